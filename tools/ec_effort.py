#!/usr/bin/env python3
"""What the error-block search costs, block by block (OATK_BUF_EC_BLOCK_WORK / _OUT): the distribution of DFS steps over the blocks of a read set,
and the heaviest blocks with their shape.   python tools/ec_effort.py [config1s|config2|config3] [n_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oatk_amd import HipSyncasm, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "config1s"
K, S = 1001, 31
if wl == "config1s":
    cfg = dict(synth.CONFIG1S); rs = synth.MixReadSet(**cfg)
else:
    cfg = dict(synth.CONFIGS[wl]); rs = synth.ReadSet(**cfg)
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["n_reads"]
c = cfg["min_k_cov"]
seq, off, lens = rs.slice(0, n)
hip = HipSyncasm(0); hip.set_timing(True)
hip.scan_host(seq, off, lens, K, S); hip.count(); hip.ec_graph(light_c=c)
if os.environ.get("EC_EFFORT_WARM"):           # (a first call sizes the solver's slabs: hipMalloc of a gigabyte or two)
    hip.ec(0.02, c, 0.35); hip.sync(); hip.ec_graph(light_c=c)
t0 = time.perf_counter(); st = hip.ec(0.02, c, 0.35); hip.sync(); dt = time.perf_counter() - t0
tm = hip.timing()
w = hip.fetch("EC_BLOCK_WORK").reshape(-1, 12); o = hip.fetch("EC_BLOCK_OUT").reshape(-1, 12)
tried, npath, status, l, r = o[:, 6].astype(np.int64), o[:, 7].astype(np.int64), o[:, 0], w[:, 6].astype(np.int64), w[:, 7]
end_none = (w[:, 2] == 0xFFFFFFFF) & (w[:, 3] == 0xFFFFFFFF)
print("%s, %d reads: ec %.1f ms (mark %.1f solve %.1f refresh %.1f); %d blocks, %d DFS steps in all" % (wl, n, dt * 1e3, tm["ec_mark"], tm["ec_solve"], tm["ec_refresh"], len(o), tried.sum()))
for lo, hi in ((0, 1), (1, 4), (4, 16), (16, 64), (64, 256), (256, 1024), (1024, 4096), (4096, 16384), (16384, 1 << 40)):
    m = (tried >= lo) & (tried < hi)
    if m.any():
        print("  steps [%6d, %6s): %8d blocks, %12d steps (%.1f %%), mean length %6.0f, dead ends %d" % (lo, hi if hi < 1 << 40 else "inf", m.sum(), tried[m].sum(), 100.0 * tried[m].sum() / max(tried.sum(), 1), l[m].mean(), npath[m].sum()))
wfs, wfd = o[:, 8].astype(np.int64), o[:, 9].astype(np.int64) * 64
print("wavefront steps %d, diagonal extensions %d in all" % (wfs.sum(), wfd.sum()))
tk, tier, steals = o[:, 10].astype(np.int64), o[:, 11] & 0xFF, (o[:, 11] >> 8).astype(np.int64)      # (the tree solver leaves the number of sub-tasks thieves ran above its tier)
print("sub-tasks run by thieves (tree solver): %d in %d blocks" % (steals.sum(), (steals > 0).sum()))
print("time on the waves: %.1f ms in all, longest block %.2f ms; by kernel variant: %s" % (tk.sum() * 1e-5, tk.max() * 1e-5, {int(t): "%d blocks %.1f ms" % ((tier == t).sum(), tk[tier == t].sum() * 1e-5) for t in np.unique(tier)}))
for name, key in (("DFS steps", tried), ("diagonal extensions", wfd), ("time", tk)):
    print("heaviest by %s:" % name)
    for i in np.argsort(-key)[:10]:
        print("  block %8d: read %7d len %6d %s arcs %7d dead ends %6d wavefront steps %8d diagonals %10d status %d path %d  %.2f ms (variant %d, %d sub-tasks): %.0f ns per step" % (
              i, w[i, 4], l[i], "leading" if r[i] else ("trailing" if end_none[i] else "middle"), tried[i], npath[i], wfs[i], wfd[i], status[i], o[i, 1], tk[i] * 1e-5, tier[i], steals[i], tk[i] * 10.0 / max(wfs[i], 1)))
# the second stage's blocks by the waves their band needs (ec_fused.hpp: a wave owns 56 slots, 2 bw + 3 slots in all): what narrower workgroups would hold
bw = np.maximum(6, np.ceil(l * 0.02)).astype(np.int64)
need = (2 * bw + 3 + 55) // 56
m2 = tier >= 32
print("second-stage blocks by waves needed (2 bw + 3 slots, 56 owned per wave):")
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 99)):
    m = m2 & (need >= lo) & (need <= hi)
    if m.any():
        print("  %2d-%2d waves: %6d blocks, time %9.1f ms (longest %7.1f), arcs %10d, wavefront steps %10d, mean diagonals per step %6.1f" % (
              lo, hi, m.sum(), tk[m].sum() * 1e-5, tk[m].max() * 1e-5, tried[m].sum(), wfs[m].sum(), wfd[m].sum() / max(wfs[m].sum(), 1)))
