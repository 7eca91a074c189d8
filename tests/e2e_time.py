#!/usr/bin/env python3
"""Wall clock of a whole assembly, reads in a FASTA file -> both GFA files: the compiled reference alone, and the reference with every row of
SURVEY 8 on the MI355X (scan, count, error correction, assembly graph, read alignments, consensus sums and distance tables through the hooked
build) and only its graph surgery and printing on the host.  Needs oracle/_ref (built where /root/reference exists).  Development aid.
usage: python tests/e2e_time.py [n_reads] [threads]"""
import ctypes as C
import filecmp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # tests/ may use the compiled reference; tools/ may not
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_lib as R  # noqa: E402
from oatk_amd import HipSyncasm, _lib  # noqa: E402
from oatk_amd.synth import ReadSet  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K, S, cov = 1001, 31, 30
vp = C.c_void_p
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "reads.fa")
rs = ReadSet(1_000_000, n, 15000)
seq, off, lens = rs.slice(0, n)
with open(fa, "wb") as f:
    for i in range(n):
        f.write(b">r%d\n" % i + seq[int(off[i]):int(off[i]) + int(lens[i])].tobytes() + b"\n")
bases = int(lens.sum())
print("%d reads, %.2f Gbases, -t %d" % (n, bases / 1e9, T), flush=True)

L = R.lib()
t0 = time.perf_counter()
assert L.refx_syncasm(R._files_arg([fa]), 1, K, S, cov, 0.35, 1, 3, T, os.path.join(tmp, "ref").encode()) == 0
t_ref = time.perf_counter() - t0
print("reference alone            %8.2f s" % t_ref, flush=True)

H = C.CDLL(_lib.HOST_LIB_PATH)
LH = C.CDLL(os.path.join(R.REF_DIR, "liboatk_ref_hooked.so"))
H.oatk_sr_db_new.restype = vp
H.oatk_sr_db_new.argtypes = [C.c_int, C.c_int]
H.oatk_sr_read_packed.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp]
H.oatk_collect_syncmer_from_reads.restype = vp
H.oatk_collect_syncmer_from_reads.argtypes = [vp, vp, C.POINTER(C.c_int)]
H.oatk_read_error_correction.argtypes = [vp, vp, vp, vp, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp]
H.oatk_make_syncmer_asmg.restype = vp
H.oatk_make_syncmer_asmg.argtypes = [vp, vp, C.c_uint32, C.c_double, C.POINTER(C.c_int)]
H.oatk_scg_read_alignment.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(vp)]
H.oatk_consensus_fetch.restype = vp
H.oatk_consensus_fetch.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
H.oatk_overlap_fetch.restype = vp
H.oatk_overlap_fetch.argtypes = [vp, C.POINTER(C.c_int)]
H.oatk_ingest_files.argtypes = [vp, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_uint64)]
LH.refx_syncasm_tail_graph.restype = C.c_int
LH.refx_syncasm_tail_graph.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_char_p]
LH.refx_hooks_install.argtypes = [vp] * 5
LH.refx_set_aligner.argtypes = [vp]

hip = HipSyncasm(0)
hip.scan_host(seq[:4096 * 64], off[:64], lens[:64], K, S)      # context warm-up (first kernel launch, allocations of a tiny batch)
t0 = time.perf_counter()
marks = []
db = H.oatk_sr_db_new(K, S)
H.oatk_sr_read_files.argtypes = [vp, vp, C.POINTER(C.c_char_p), C.c_int]
files = (C.c_char_p * 1)(fa.encode())
assert H.oatk_sr_read_files(hip.h, db, files, 1) == 0                # the same FASTA file the reference read
marks.append(("file -> device reader -> scan -> sr_db structs on the host", time.perf_counter()))
rc = C.c_int(0)
scm = H.oatk_collect_syncmer_from_reads(hip.h, db, C.byref(rc))
marks.append(("count + syncmer_db structs", time.perf_counter()))
st = np.zeros(12, np.uint64)
assert H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, cov, 10 * cov, cov, 0.35, st.ctypes.data) == 0
marks.append(("EC graph + correction + write-back", time.perf_counter()))
cs = H.oatk_consensus_fetch(hip.h, cov, K, C.byref(rc))
ov = H.oatk_overlap_fetch(hip.h, C.byref(rc))
asmg = H.oatk_make_syncmer_asmg(hip.h, scm, cov, 0.35, C.byref(rc))
marks.append(("consensus sums, distance tables, assembly graph", time.perf_counter()))
t_aln = [0.0]


def aligner(db_, v, g, n_threads, for_unzip):
    t = time.perf_counter()
    nsk = C.c_uint64(0)
    assert H.oatk_scg_read_alignment(hip.h, db_, v, g, for_unzip, C.byref(nsk), None) == 0 and nsk.value == 0
    t_aln[0] += time.perf_counter() - t


cb = C.CFUNCTYPE(None, vp, vp, vp, C.c_int, C.c_int)(aligner)
LH.refx_set_aligner(cb)
LH.refx_hooks_install(scm, C.cast(H.oatk_scg_syncmer_consensus, vp), cs, C.cast(H.oatk_overlap_lookup, vp), ov)
assert LH.refx_syncasm_tail_graph(db, scm, asmg, K, 100000, 10000, cov, 0.35, 0.3, 3, T, os.path.join(tmp, "dev").encode()) == 0
marks.append(("the reference's tail (unitigging, cleaning, unzip rounds, GFA), alignments on the device", time.perf_counter()))
t_dev = time.perf_counter() - t0
prev = t0
for name, t in marks:
    print("  %-88s %8.3f s" % (name, t - prev))
    prev = t
print("  of the tail: %.3f s inside the %s" % (t_aln[0], "device alignment calls (graph flattening and scg_ra_v rebuild included)"))
print("with the device             %8.2f s   (%.1fx)" % (t_dev, t_ref / t_dev))
for sfx in (".utg.gfa", ".utg.final.gfa"):
    assert filecmp.cmp(os.path.join(tmp, "ref" + sfx), os.path.join(tmp, "dev" + sfx), shallow=False), sfx
print("both GFA files identical")
