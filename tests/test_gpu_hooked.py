"""GPU: INTEGRATION.md taken literally.  oracle/_ref/liboatk_ref_hooked.so is the reference with two function-pointer hooks compiled into
scg_syncmer_consensus and calc_syncmer_overlap at the places INTEGRATION.md 3b / 3b' name (`make -C oracle ref_hooked`: syncasm.c goes through
sed into gcc, nothing else changes).  With the hooks pointing at liboatk_host's adaptors, every consensus sum and every distance table of all
four scg_consensus calls comes from the MI355X -- together with the scan, the count, the error correction, the assembly graph and every read
alignment -- and both GFA files still equal a run of the untouched reference byte for byte."""
import ctypes as C
import filecmp
import os

import numpy as np
import pytest

import adversarial as A
import ref_lib as R
from oatk_amd import _lib
from test_gpu_dropin import device_dbs, host_lib

HOOKED = os.path.join(R.REF_DIR, "liboatk_ref_hooked.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (R.available() and os.path.exists(HOOKED)), reason="oracle/_ref hooked build missing")]


@pytest.mark.parametrize("K,S,cov,err,n_reads,scale", [(1001, 31, 8, 0.0008, 260, False), (301, 21, 6, 0.001, 260, False), (1001, 31, 30, 0.0, 20000, True)])
def test_gfa_identical_with_consensus_overlaps_and_everything_else_from_the_device(hip, tmp_path, K, S, cov, err, n_reads, scale):
    L, H, LH = R.lib(), host_lib(), C.CDLL(HOOKED)
    vp = C.c_void_p
    LH.refx_syncasm_tail_graph.restype = C.c_int
    LH.refx_syncasm_tail_graph.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_char_p]
    LH.refx_hooks_install.argtypes = [vp, vp, vp, vp, vp]
    LH.refx_hooks_served.argtypes = [vp]
    LH.refx_set_aligner.argtypes = [vp]
    LH.refx_scmdb_destroy.argtypes = [vp]
    LH.refx_srdb_destroy.argtypes = [vp]
    H.oatk_read_error_correction.argtypes = [vp, vp, vp, vp, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp]
    H.oatk_make_syncmer_asmg.restype = vp
    H.oatk_make_syncmer_asmg.argtypes = [vp, vp, C.c_uint32, C.c_double, C.POINTER(C.c_int)]
    H.oatk_scg_read_alignment.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(vp)]
    H.oatk_consensus_fetch.restype = vp
    H.oatk_consensus_fetch.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
    H.oatk_consensus_destroy.argtypes = [vp]
    H.oatk_overlap_fetch.restype = vp
    H.oatk_overlap_fetch.argtypes = [vp, C.POINTER(C.c_int)]
    H.oatk_overlap_destroy.argtypes = [vp]
    if scale:
        from oatk_amd.synth import ReadSet
        reads = ReadSet(1_000_000, n_reads, 15000).as_list(0, n_reads)
    else:
        reads = A.hifi_like(n_reads, 50000, 9000 if K > 500 else 5000, seed=K + 33, err=err)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    out_ref, out_dev = str(tmp_path / "ref"), str(tmp_path / "dev")
    assert L.refx_syncasm(R._files_arg([fa]), 1, K, S, cov, 0.35, 1, 3, 8, out_ref.encode()) == 0          # the untouched reference
    db, scm = device_dbs(hip, reads, K, S)
    stats = np.zeros(12, np.uint64)
    assert H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, cov, 10 * cov, cov, 0.35, stats.ctypes.data) == 0
    rc = C.c_int(0)
    cs = H.oatk_consensus_fetch(hip.h, cov, K, C.byref(rc))                       # run-length totals of every syncmer the graph can contain
    assert rc.value == 0 and cs
    ov = H.oatk_overlap_fetch(hip.h, C.byref(rc))                                 # distance tables of every adjacent pair
    assert rc.value == 0 and ov
    asmg = H.oatk_make_syncmer_asmg(hip.h, scm, cov, 0.35, C.byref(rc))
    assert rc.value == 0 and asmg
    problems = []

    def aligner(db_, v, g, n_threads, for_unzip):
        nsk = C.c_uint64(0)
        r = H.oatk_scg_read_alignment(hip.h, db_, v, g, for_unzip, C.byref(nsk), None)
        if r != 0 or nsk.value:
            problems.append((r, nsk.value))

    cb = C.CFUNCTYPE(None, vp, vp, vp, C.c_int, C.c_int)(aligner)
    LH.refx_set_aligner(cb)
    LH.refx_hooks_install(scm, C.cast(H.oatk_scg_syncmer_consensus, vp), cs, C.cast(H.oatk_overlap_lookup, vp), ov)
    try:
        assert LH.refx_syncasm_tail_graph(db, scm, asmg, K, 100000, 10000, cov, 0.35, 0.3, 3, 8, out_dev.encode()) == 0
        served = np.zeros(4, np.uint64)
        LH.refx_hooks_served(served.ctypes.data)
    finally:
        LH.refx_hooks_install(None, None, None, None, None)
        LH.refx_set_aligner(None)
    assert not problems
    assert served[0] > 50 and served[2] > 50, served                              # consensus strings and distance tables did come from the device
    for suffix in (".utg.gfa", ".utg.final.gfa"):
        assert os.path.getsize(out_ref + suffix) > 100
        assert filecmp.cmp(out_ref + suffix, out_dev + suffix, shallow=False), suffix
    H.oatk_overlap_destroy(ov)
    H.oatk_consensus_destroy(cs)
    LH.refx_scmdb_destroy(scm)
    LH.refx_srdb_destroy(db)
