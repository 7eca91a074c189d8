"""GPU drop-in test: the device path fills the reference's own structs (sr_db_t / syncmer_db_t, through
liboatk_host.so), the COMPILED REFERENCE runs the rest of syncasm() on them (graph, error correction, cleaning,
unzipping, GFA output -- oracle/ref_shim.c::refx_syncasm_tail), and the GFA bytes must equal a pure reference run.
Needs the GPU and oracle/_ref (built in the authoring container; travels to the GPU box)."""
import ctypes as C
import filecmp
import os

import numpy as np
import pytest

import adversarial as A
import ref_lib as R
from oatk_amd import _lib, pack_reads

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")]


def host_lib():
    H = C.CDLL(_lib.HOST_LIB_PATH)
    vp = C.c_void_p
    H.oatk_sr_db_new.restype = vp
    H.oatk_sr_db_new.argtypes = [C.c_int, C.c_int]
    H.oatk_sr_read_packed.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp]
    H.oatk_collect_syncmer_from_reads.restype = vp
    H.oatk_collect_syncmer_from_reads.argtypes = [vp, vp, C.POINTER(C.c_int)]
    return H


def device_dbs(hip, reads, K, S):
    H = host_lib()
    seq, off, lens = pack_reads(reads)
    db = H.oatk_sr_db_new(K, S)
    rc = H.oatk_sr_read_packed(hip.h, db, seq.ctypes.data, off.ctypes.data, lens.ctypes.data, len(reads), seq.size, None)
    assert rc == 0, hip.L.oatk_hip_last_error(hip.h)
    rcc = C.c_int(0)
    scm = H.oatk_collect_syncmer_from_reads(hip.h, db, C.byref(rcc))
    assert rcc.value == 0 and scm
    return db, scm


@pytest.mark.parametrize("K,S,cov,do_ec,do_unzip", [(1001, 31, 8, 1, 3), (1001, 31, 8, 0, 0), (301, 21, 6, 1, 3)])
def test_gfa_identical_to_reference(hip, tmp_path, K, S, cov, do_ec, do_unzip):
    L = R.lib()
    L.refx_syncasm_tail.restype = C.c_int
    L.refx_syncasm_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                    C.c_int, C.c_char_p]
    reads = A.hifi_like(260, 50000, 9000 if K > 500 else 5000, seed=K)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    out_ref, out_dev = str(tmp_path / "ref"), str(tmp_path / "dev")
    # pure reference
    rc = L.refx_syncasm(R._files_arg([fa]), 1, K, S, cov, 0.35, do_ec, do_unzip, 4, out_ref.encode())
    assert rc == 0
    # device scan + count, reference tail
    db, scm = device_dbs(hip, reads, K, S)
    rc = L.refx_syncasm_tail(db, scm, K, 100000, 10000, cov, 0.35, 0.3, do_ec, do_unzip, 4, out_dev.encode())
    assert rc == 0
    for suffix in (".utg.gfa", ".utg.final.gfa"):
        assert os.path.getsize(out_ref + suffix) > 100
        assert filecmp.cmp(out_ref + suffix, out_dev + suffix, shallow=False), suffix
    # the reference's own destructors free what liboatk_host malloc'ed
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)


@pytest.mark.parametrize("K,S,cov,err,host_graph", [(1001, 31, 8, 0.0008, True), (301, 21, 6, 0.001, True), (1001, 31, 8, 0.0008, False),
                                                    (301, 21, 6, 0.001, False)])
def test_gfa_identical_with_all_three_device_functions(hip, tmp_path, K, S, cov, err, host_graph):
    """sr_read, collect_syncmer_from_reads AND read_error_correction on the device (through liboatk_host.so); the
    reference only builds the EC graph in between (host_graph) -- or not even that -- and runs the rest of syncasm() afterwards"""
    L, H = R.lib(), host_lib()
    L.refx_syncasm_tail.restype = C.c_int
    L.refx_syncasm_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                    C.c_int, C.c_char_p]
    H.oatk_read_error_correction.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_double, C.c_void_p]
    reads = A.hifi_like(260, 50000, 9000 if K > 500 else 5000, seed=K + 5, err=err)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    out_ref, out_dev = str(tmp_path / "ref"), str(tmp_path / "dev")
    assert L.refx_syncasm(R._files_arg([fa]), 1, K, S, cov, 0.35, 1, 3, 4, out_ref.encode()) == 0
    db, scm = device_dbs(hip, reads, K, S)
    stats = np.zeros(12, np.uint64)
    if host_graph:
        g = L.refx_make_graph(db, scm, 0, 0.0)                               # run_syncasm.c:109
        L.refx_consensus(db, g, 1, 1)                                        # run_syncasm.c:117
        asmg = C.cast(g, C.POINTER(C.c_void_p))[1]                           # scg_t.utg_asmg (syncasm.h:51-54)
        rc = H.oatk_read_error_correction(hip.h, db, scm, asmg, 0.02, cov, 10 * cov, cov, 0.35, stats.ctypes.data)
        L.refx_scg_destroy(g)                                                # run_syncasm.c:132
    else:
        rc = H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, cov, 10 * cov, cov, 0.35, stats.ctypes.data)
    assert rc == 0, hip.L.oatk_hip_last_error(hip.h)
    assert int(stats[2] + stats[7]) > 0                                      # blocks were corrected
    assert L.refx_syncasm_tail(db, scm, K, 100000, 10000, cov, 0.35, 0.3, 0, 3, 4, out_dev.encode()) == 0
    for suffix in (".utg.gfa", ".utg.final.gfa"):
        assert os.path.getsize(out_ref + suffix) > 100
        assert filecmp.cmp(out_ref + suffix, out_dev + suffix, shallow=False), suffix
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)


@pytest.mark.parametrize("K,S,cov,err,do_ec,do_unzip", [(1001, 31, 8, 0.0008, 1, 3), (301, 21, 6, 0.001, 1, 3), (1001, 31, 8, 0.0008, 0, 0),
                                                       (301, 21, 6, 0.001, 0, 3)])
def test_gfa_identical_with_the_assembly_graph_from_the_device(hip, tmp_path, K, S, cov, err, do_ec, do_unzip):
    """sr_read, collect_syncmer_from_reads, the whole error-correction round AND make_syncmer_graph(sr_db, scm_db, c, a) (run_syncasm.c:138)
    on the device: the reference receives its asmg_t ready-made (oatk_make_syncmer_asmg) and starts at the unitigging (:160)"""
    L, H = R.lib(), host_lib()
    L.refx_syncasm_tail_graph.restype = C.c_int
    L.refx_syncasm_tail_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                          C.c_int, C.c_char_p]
    H.oatk_read_error_correction.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_double, C.c_void_p]
    H.oatk_make_syncmer_asmg.restype = C.c_void_p
    H.oatk_make_syncmer_asmg.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.POINTER(C.c_int)]
    reads = A.hifi_like(260, 50000, 9000 if K > 500 else (5000 if K > 200 else 2000), seed=K + 9, err=err)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    out_ref, out_dev = str(tmp_path / "ref"), str(tmp_path / "dev")
    assert L.refx_syncasm(R._files_arg([fa]), 1, K, S, cov, 0.35, do_ec, do_unzip, 4, out_ref.encode()) == 0
    db, scm = device_dbs(hip, reads, K, S)
    if do_ec:
        stats = np.zeros(12, np.uint64)
        assert H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, cov, 10 * cov, cov, 0.35, stats.ctypes.data) == 0
    rc = C.c_int(0)
    asmg = H.oatk_make_syncmer_asmg(hip.h, scm, cov, 0.35, C.byref(rc))
    assert rc.value == 0 and asmg, hip.L.oatk_hip_last_error(hip.h)
    assert L.refx_syncasm_tail_graph(db, scm, asmg, K, 100000, 10000, cov, 0.35, 0.3, do_unzip, 4, out_dev.encode()) == 0   # frees asmg
    for suffix in (".utg.gfa", ".utg.final.gfa"):
        assert os.path.getsize(out_ref + suffix) > 100
        assert filecmp.cmp(out_ref + suffix, out_dev + suffix, shallow=False), suffix
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)


@pytest.mark.parametrize("K,S,cov,err,n_reads,scale", [(1001, 31, 8, 0.0008, 260, False), (301, 21, 6, 0.001, 260, False), (1001, 31, 30, 0.0, 20000, True)])
def test_gfa_identical_with_everything_but_the_graph_surgery_on_the_device(hip, tmp_path, K, S, cov, err, n_reads, scale):
    """sr_read, collect_syncmer_from_reads, the error-correction round, make_syncmer_graph AND every scg_read_alignment call of the unzip
    rounds and the final coverage pass (run_syncasm.c:219-303) on the device; the reference does the unitigging, cleaning, multiplexing and
    the GFA output in between, and both GFA files still equal a pure reference run byte for byte"""
    L, H = R.lib(), host_lib()
    vp = C.c_void_p
    L.refx_syncasm_tail_graph.restype = C.c_int
    L.refx_syncasm_tail_graph.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_char_p]
    H.oatk_read_error_correction.argtypes = [vp, vp, vp, vp, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp]
    H.oatk_make_syncmer_asmg.restype = vp
    H.oatk_make_syncmer_asmg.argtypes = [vp, vp, C.c_uint32, C.c_double, C.POINTER(C.c_int)]
    H.oatk_scg_read_alignment.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(vp)]
    if scale:
        from oatk_amd.synth import ReadSet
        reads = ReadSet(1_000_000, n_reads, 15000).as_list(0, n_reads)
    else:
        reads = A.hifi_like(n_reads, 50000, 9000 if K > 500 else 5000, seed=K + 21, err=err)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    out_ref, out_dev = str(tmp_path / "ref"), str(tmp_path / "dev")
    assert L.refx_syncasm(R._files_arg([fa]), 1, K, S, cov, 0.35, 1, 3, 8, out_ref.encode()) == 0
    db, scm = device_dbs(hip, reads, K, S)
    stats = np.zeros(12, np.uint64)
    assert H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, cov, 10 * cov, cov, 0.35, stats.ctypes.data) == 0
    rc = C.c_int(0)
    asmg = H.oatk_make_syncmer_asmg(hip.h, scm, cov, 0.35, C.byref(rc))
    assert rc.value == 0 and asmg
    calls, problems = [], []

    def aligner(db_, v, g, n_threads, for_unzip):          # same signature as scg_read_alignment (alignment.c:596)
        nsk = C.c_uint64(0)
        r = H.oatk_scg_read_alignment(hip.h, db_, v, g, for_unzip, C.byref(nsk), None)
        calls.append(for_unzip)
        if r != 0 or nsk.value:
            problems.append((r, nsk.value))

    cb = C.CFUNCTYPE(None, vp, vp, vp, C.c_int, C.c_int)(aligner)
    L.refx_set_aligner.argtypes = [vp]
    L.refx_set_aligner(cb)
    try:
        assert L.refx_syncasm_tail_graph(db, scm, asmg, K, 100000, 10000, cov, 0.35, 0.3, 3, 8, out_dev.encode()) == 0
    finally:
        L.refx_set_aligner(None)
    assert not problems and len(calls) >= 4 and 1 in calls and 0 in calls
    for suffix in (".utg.gfa", ".utg.final.gfa"):
        assert os.path.getsize(out_ref + suffix) > 100
        assert filecmp.cmp(out_ref + suffix, out_dev + suffix, shallow=False), suffix
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)


def test_structs_equal_reference_structs(hip):
    """member-by-member: reference flatteners applied to OUR sr_db / scm_db vs the reference's own"""
    K, S = 1001, 31
    reads = A.reads(K, S, seed=77, scale=0.5) + A.hifi_like(50, 40000, 9000, seed=12)
    n_nn = np.array([sum(1 for c in r if c not in b"ACGTacgtUu\x00\x01\x02\x03") for r in reads], np.uint32)
    db, scm = device_dbs(hip, reads, K, S)
    mine_db, mine_scm = object.__new__(R.SrDb), object.__new__(R.ScmDb)
    mine_db._h, mine_db.K, mine_db.S = db, K, S
    mine_scm._h = scm
    got, got_c = mine_db.flatten(n_nn=n_nn), mine_scm.flatten()
    ref_db = R.SrDb.from_reads(reads, K, S, threads=2)
    ref_scm = R.ScmDb(ref_db)
    want, want_c = ref_db.flatten(n_nn=n_nn), ref_scm.flatten()
    for f in want:
        assert np.array_equal(got[f], want[f]), f
    for f in want_c:
        assert np.array_equal(np.asarray(got_c[f]), np.asarray(want_c[f])), f
    mine_scm.close(), mine_db.close(), ref_scm.close(), ref_db.close()


@pytest.mark.parametrize("fmt,gz,one_file", [("fa", False, False), ("fq", False, False), ("fa", True, False), ("fa", False, True), ("fq", False, True)])
def test_sr_read_files_fills_the_reference_structs(hip, tmp_path, fmt, gz, one_file):
    """oatk_sr_read_files: files -> device reader -> device scan -> sr_db, member by member equal to the reference's sr_read of the same files
    (two files, names with comments, a wrapped FASTA, optional gzip)"""
    import gzip
    L, H = R.lib(), host_lib()
    H.oatk_sr_read_files.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_int]
    L.refx_srdb_name.restype = C.c_char_p
    L.refx_srdb_name.argtypes = [C.c_void_p, C.c_uint64]
    K, S = 301, 21
    reads = A.hifi_like(90, 30000, 4000, seed=5) + A.reads(K, S, seed=4, scale=0.2)[:20]
    paths = []
    # one file: mapped and handed over as it lies (it ends in a newline); the second variant of it does not and takes the copying path
    for part, chunk in enumerate((reads,) if one_file else (reads[:60], reads[60:])):
        p = str(tmp_path / ("part%d.%s%s" % (part, fmt, ".gz" if gz else "")))
        op = gzip.open if gz else open
        with op(p, "wb") as f:
            for i, r in enumerate(chunk):
                name = b"read_%d_%d some comment %d" % (part, i, i * 7)
                if fmt == "fa":
                    f.write(b">" + name + b"\n")
                    for o in range(0, len(r), 70 if part else 10 ** 9):
                        f.write(r[o:o + (70 if part else 10 ** 9)] + b"\n")
                else:
                    f.write(b"@" + name + b"\n" + r + b"\n+\n" + b"I" * len(r) + b"\n")
        paths.append(p)
    if one_file and fmt == "fq":
        raw = open(paths[0], "rb").read()
        open(paths[0], "wb").write(raw[:-1])           # no newline behind the last quality line: not mappable as it lies
    n_nn = np.array([sum(1 for c in r if c not in b"ACGTacgtUu\x00\x01\x02\x03") for r in reads], np.uint32)
    db = H.oatk_sr_db_new(K, S)
    files = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    assert H.oatk_sr_read_files(hip.h, db, files, len(paths)) == 0, hip.L.oatk_hip_last_error(hip.h)
    mine = object.__new__(R.SrDb)
    mine._h, mine.K, mine.S = db, K, S
    ref = R.SrDb(paths, K, S, 2)
    got, want = mine.flatten(n_nn=n_nn), ref.flatten(n_nn=n_nn)
    for f in want:
        assert np.array_equal(got[f], want[f]), f
    for i in (0, 1, 59, 60, len(reads) - 1):
        assert L.refx_srdb_name(db, i) == L.refx_srdb_name(ref.handle, i) and L.refx_srdb_name(db, i).startswith(b"read_")
    mine.close(), ref.close()


class _KString(C.Structure):
    _fields_ = [("l", C.c_size_t), ("m", C.c_size_t), ("s", C.c_void_p)]


@pytest.mark.parametrize("K,S,cov,with_ec", [(101, 11, 4, False), (101, 11, 4, True), (301, 21, 4, True)])
def test_syncmer_consensus_served_from_the_device(hip, K, S, cov, with_ec):
    """scg_syncmer_consensus (syncasm.c:888) through liboatk_host.so: run-length totals from the MI355X, string assembly on the host,
    against the compiled reference's function on the very same structs -- both strands, several `beg`, hoco and base space"""
    import cons_util as CU
    L, H = R.lib(), host_lib()
    H.oatk_read_error_correction.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_double, C.c_void_p]
    H.oatk_consensus_fetch.restype = C.c_void_p
    H.oatk_consensus_fetch.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
    H.oatk_consensus_destroy.argtypes = [C.c_void_p]
    H.oatk_scg_syncmer_consensus.restype = C.c_int64
    H.oatk_scg_syncmer_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int64, C.POINTER(_KString), C.c_int]
    L.refx_syncmer_consensus.restype = C.c_int64
    L.refx_syncmer_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int64, C.c_int, C.c_char_p, C.c_int64]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    reads = CU.long_run_reads(K + 3, K)
    db, scm = device_dbs(hip, reads, K, S)
    if with_ec:
        st = np.zeros(12, np.uint64)
        assert H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, cov, 10 * cov, cov, 0.35, st.ctypes.data) == 0
        assert int(st[2] + st[7]) > 0
    rc = C.c_int(0)
    cs = H.oatk_consensus_fetch(hip.h, cov, K, C.byref(rc))
    assert rc.value == 0 and cs
    rscm = object.__new__(R.ScmDb)
    rscm._h = scm
    sc = rscm.flatten()
    ids = np.nonzero((sc["cov"] >= cov) & (sc["del"] == 0))[0]
    assert len(ids) > 20
    buf = C.create_string_buffer(1 << 20)
    served = 0
    for i in ids.tolist():
        for rev in (0, 1):
            for beg in (0, 9, K - 1, -4):
                for hoco in (0, 1):
                    ks = _KString(0, 0, None)
                    n = H.oatk_scg_syncmer_consensus(cs, db, i, rev, beg, C.byref(ks), hoco)
                    got = C.string_at(ks.s, ks.l) if ks.l else b""
                    libc.free(ks.s)
                    nr = L.refx_syncmer_consensus(db, scm, i, rev, beg, hoco, buf, len(buf))
                    assert n == nr and got == buf.raw[:nr], (i, rev, beg, hoco)
                    served += 1
    assert served == len(ids) * 16
    # a syncmer below the threshold is not prepared: the caller keeps its own routine for it
    low = np.nonzero(sc["cov"] < cov)[0]
    if len(low):
        ks = _KString(0, 0, None)
        assert H.oatk_scg_syncmer_consensus(cs, db, int(low[0]), 0, 0, C.byref(ks), 0) == -1
    H.oatk_consensus_destroy(cs)
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)


@pytest.mark.parametrize("K,S,cov", [(101, 11, 4), (301, 21, 5)])
def test_sr_db_stat_from_the_device(hip, K, S, cov):
    """sr_db_stat (syncmer.c:867) through liboatk_host.so at its two call sites -- after sr_read (k-mers are still hashes,
    run_syncasm.c:88) and after read_error_correction (:131) -- against the compiled reference on the very same structs"""
    L, H = R.lib(), host_lib()
    H.oatk_sr_db_stat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    H.oatk_read_error_correction.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_double, C.c_void_p]
    H.oatk_sr_read_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    H.oatk_sr_db_new.restype = C.c_void_p
    H.oatk_collect_syncmer_from_reads.restype = C.c_void_p
    H.oatk_collect_syncmer_from_reads.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]

    class StatT(C.Structure):             # sr_stat_t, syncmer.h:72-77
        _fields_ = [("syncmer_n", C.c_uint64), ("per_read", C.c_double), ("avg_dist", C.c_double), ("smer_avg", C.c_double), ("kmer_avg", C.c_double),
                    ("smer_unique", C.c_int), ("smer_singleton", C.c_int), ("smer_hom", C.c_int), ("smer_het", C.c_int),
                    ("kmer_unique", C.c_int), ("kmer_singleton", C.c_int), ("kmer_hom", C.c_int), ("kmer_het", C.c_int)]

    class SrDbT(C.Structure):
        _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p), ("k", C.c_int), ("s", C.c_int), ("stats", C.POINTER(StatT))]

    def snapshot(db):
        st = C.cast(db, C.POINTER(SrDbT)).contents.stats.contents
        return [getattr(st, f[0]) for f in StatT._fields_]

    def check(db):
        assert H.oatk_sr_db_stat(hip.h, db, None, 0) == 0
        mine = snapshot(db)
        i8, d5 = np.zeros(8, np.int32), np.zeros(5, np.float64)
        L.refx_srdb_stat(db, i8.ctypes.data, d5.ctypes.data)             # overwrites db->stats with the reference's own
        ref = snapshot(db)
        assert mine == ref, (mine, ref)                                   # doubles bit for bit: sums of integers below 2^53
        return mine

    reads = A.hifi_like(400, 20000, 2500 if K < 300 else 5000, seed=K, err=0.002)
    seq, off, lens = pack_reads(reads)
    db = H.oatk_sr_db_new(K, S)
    assert H.oatk_sr_read_packed(hip.h, db, seq.ctypes.data, off.ctypes.data, lens.ctypes.data, len(reads), seq.size, None) == 0
    a = check(db)                                                         # after sr_read
    assert a[0] > 1000 and a[11] > 0                                      # a k-mer coverage peak was found
    rc = C.c_int(0)
    scm = H.oatk_collect_syncmer_from_reads(hip.h, db, C.byref(rc))
    assert rc.value == 0 and scm
    b = check(db)                                                         # after the count: same numbers, k-mers are ids now
    assert a == b
    st = np.zeros(12, np.uint64)
    assert H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, cov, 10 * cov, cov, 0.35, st.ctypes.data) == 0
    c = check(db)                                                         # after the correction (run_syncasm.c:131)
    assert c != b and int(st[2] + st[7]) > 0
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)


def test_gfa_identical_at_scale(hip, tmp_path):
    """20 k synthetic HiFi reads x 15 kb (0.3 Gbases, ~300x of a 1 Mb genome, the bench's generator), k = 1001, -c 30: scan, count,
    EC graph and error correction on the MI355X (no host graph), the compiled reference's syncasm() for the rest -- both GFA files
    byte-identical to a pure reference run.  Every solver tier, 80 k error blocks, long-run escapes and coverage in the hundreds."""
    from oatk_amd.synth import ReadSet
    L, H = R.lib(), host_lib()
    L.refx_syncasm_tail.restype = C.c_int
    L.refx_syncasm_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                    C.c_int, C.c_char_p]
    H.oatk_read_error_correction.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_double, C.c_void_p]
    K, S, cov, n = 1001, 31, 30, 20000
    rs = ReadSet(1_000_000, n, 15000)
    reads = rs.as_list(0, n)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    out_ref, out_dev = str(tmp_path / "ref"), str(tmp_path / "dev")
    assert L.refx_syncasm(R._files_arg([fa]), 1, K, S, cov, 0.35, 1, 3, 16, out_ref.encode()) == 0
    db, scm = device_dbs(hip, reads, K, S)
    stats = np.zeros(12, np.uint64)
    assert H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, cov, 10 * cov, cov, 0.35, stats.ctypes.data) == 0
    assert int(stats[0] + stats[5]) > 50000 and int(stats[11]) > 0          # blocks past the first solver tier too
    assert L.refx_syncasm_tail(db, scm, K, 100000, 10000, cov, 0.35, 0.3, 0, 3, 16, out_dev.encode()) == 0
    for suffix in (".utg.gfa", ".utg.final.gfa"):
        assert os.path.getsize(out_ref + suffix) > 100000
        assert filecmp.cmp(out_ref + suffix, out_dev + suffix, shallow=False), suffix
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)
