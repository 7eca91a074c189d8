#!/usr/bin/env python3
"""Where an arc's time goes in the second stage of the error-block solver (ec_fused.hpp), by class of workgroups: a development build of the library with -DECF_PROF2
(tools/prof_arcs.sh builds it into tools/experiments/liboatk_hip_prof2.so) accumulates thread 0's cycles by phase of the per-arc loop.  GPU only.
    OATK_HIP_LIB=tools/experiments/liboatk_hip_prof2.so python tools/prof_arcs.py [reads]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oatk_amd import HipSyncasm, synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
cfg = dict(synth.CONFIG1S); rs = synth.MixReadSet(**cfg); c = cfg["min_k_cov"]
seq, off, lens = rs.slice(0, n)
hip = HipSyncasm(0); hip.set_timing(True)
hip.scan_host(seq, off, lens, 1001, 31); hip.count()
lib = _lib.load()
buf = (C.c_ulonglong * 140)()
lib.oatk_hip_debug_ecf_prof2.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
lib.oatk_hip_debug_ecf_prof2.restype = C.c_int
for rep in range(2):
    hip.ec_graph(light_c=c); hip.sync()
    lib.oatk_hip_debug_ecf_prof2(hip.h, buf)        # clear
    t0 = time.perf_counter(); hip.ec(0.02, c, 0.35); hip.sync(); dt = time.perf_counter() - t0
    lib.oatk_hip_debug_ecf_prof2(hip.h, buf)
p = np.array(list(buf), dtype=np.float64).reshape(5, 28)
print("config1s %d reads: ec %.1f ms (solve %.1f)" % (n, dt * 1e3, hip.timing()["ec_solve"]))
names = ["top: barrier + restore + arc", "append + barrier", "table test", "alignment", "outcome + frame push", "table building"]
for row, nw in enumerate((2, 4, 8, 16)):
    r = p[row]
    if r[14] == 0: continue
    arcs = r[0]
    print("%2d waves: %d blocks, %d arcs (%.0f %% from a frame, %.0f %% without alignment, %.0f %% known dead, %.0f %% aligned), %d tables, %d wavefront steps (%.1f per aligned arc)" % (
        nw, r[14], arcs, 100 * r[1] / arcs, 100 * r[2] / arcs, 100 * r[3] / arcs, 100 * r[5] / arcs, r[4], r[15], r[15] / max(r[5], 1)))
    print("     arcs of >= 32 bases: %d (%.1f %% of the arcs), %d of them with a lagging wavefront, %d known dead" % (r[6], 100 * r[6] / arcs, r[7], r[3]))
    print("     aligned arcs / their steps: short alive %d / %d, short dead by score %d / %d, long alive %d / %d, long dead by score %d / %d" % (r[16], r[20], r[17], r[21], r[18], r[22], r[19], r[23]))
    print("     long arcs not asked: lagging wavefront %d, string > 1024 %d, target's end within reach of the rows before %d, no table slot left %d; asked %d" % (r[7], r[24], r[25], r[26], r[27]))
    tot = r[8:14].sum()
    for i, nm in enumerate(names):
        print("     %-30s %6.1f %% of the cycles, %8.0f cycles per arc" % (nm, 100 * r[8 + i] / tot, r[8 + i] / arcs))
    print("     cycles per arc in all %.0f; alignment cycles per wavefront step %.0f; table test cycles per test %.0f" % (tot / arcs, r[11] / max(r[15], 1), r[10] / max(r[3], 1)))
