#!/usr/bin/env python3
"""CPU trace of the error-block search on the config-1 surrogate (ec_trace.c): the compiled reference scans, counts and builds the EC graph,
the oracle marks, the traced search runs every block.   python tests/trace/ec_trace.py [n_reads] [min_tried] [out]"""
import ctypes as C, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ec_util as E, ref_lib as R
from oatk_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
min_tried = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/ec_trace_%d.txt" % n
K, S = 1001, 31
which = os.environ.get("ECT_SET", "config1s")                 # config1s (the surrogate) or a key of synth.CONFIGS (plain reads of one genome)
cfg = dict(synth.CONFIG1S if which == "config1s" else synth.CONFIGS[which]); c = cfg["min_k_cov"]
t0 = time.time()
rs = synth.MixReadSet(**cfg) if which == "config1s" else synth.ReadSet(**cfg)
seq, off, lens = rs.slice(0, n)
fa = "/tmp/ec_trace_%d.fa" % n
synth.write_fasta(fa, seq, off, lens, mode=synth.FA_PLAIN)
print("reads written %.1f s" % (time.time() - t0), flush=True)
db = R.SrDb([fa], K, S, 8); scm = R.ScmDb(db)
sr0, sc0 = db.flatten(), scm.flatten()
g, Gd = E.ref_graph(db, scm)
print("reference scan + count + graph %.1f s: %d vertices, %d arcs" % (time.time() - t0, Gd["n_vtx"], Gd["n_arc"]), flush=True)
scm_del = sc0["del"].copy()
E.oracle_find_error_syncmers(Gd, sc0["cov"], scm_del, c, 10 * c, c, 0.35)
so = os.path.join(HERE, "libec_trace.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "ec_trace.c")):
    os.system("gcc -O2 -Wall -fPIC -shared -o %s %s -lm" % (so, os.path.join(HERE, "ec_trace.c")))
L = C.CDLL(so)
L.ect_trace.restype = C.c_uint64
L.ect_trace.argtypes = [C.POINTER(E.GraphT), C.c_void_p, C.c_int, C.c_double, C.c_uint64] + [C.c_void_p] * 6 + [C.c_uint64, C.c_char_p]
gs = E._graph_struct(Gd)
boff = np.zeros(n + 1, np.uint64); boff[1:] = np.cumsum((sr0["hoco_l"].astype(np.uint64) + 3) // 4)
arrs = [np.ascontiguousarray(sr0[k]) for k in ("hoco_l", "hoco_s")] + [boff] + [np.ascontiguousarray(sr0[k]) for k in ("n_scm", "k_mer", "m_pos")]
t1 = time.time()
nb = L.ect_trace(C.byref(gs), scm_del.ctypes.data, K, 0.02, n, *[a.ctypes.data for a in arrs], min_tried, out.encode())
print("traced %d blocks in %.1f s -> %s" % (nb, time.time() - t1, out))
