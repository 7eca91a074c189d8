#!/usr/bin/env python3
"""What of a step does not shrink with the reads (the strong-scaling model's non-scaling compute term, 2.9 ms, was never itemised): config 3's step at
2 M reads and at one eighth of them, phase by phase (the phase timers synchronise; the step without them beside it).   python tools/fixed_term.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oatk_amd import HipSyncasm
from oatk_amd.synth import CONFIGS, ReadSet

cfg = dict(CONFIGS["config3"]); c = cfg["min_k_cov"]
rs = ReadSet(**cfg)
res = {}
for n in (cfg["n_reads"], cfg["n_reads"] // 8):
    seq, off, lens = rs.slice(0, n)
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(seq).to(dev), torch.from_numpy(off.view(np.int64)).to(dev), torch.from_numpy(lens.view(np.int32)).to(dev)]
    hip = HipSyncasm(0)
    def step():
        hip.scan_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, int(seq.size), 1001, 31); hip.count(); hip.ec_graph(light_c=c); hip.ec(0.02, c, 0.35)
    step(); step(); hip.sync()
    t0 = time.perf_counter()
    for _ in range(5): step()
    hip.sync(); wall = (time.perf_counter() - t0) / 5 * 1e3
    hip.set_timing(True); step(); tm = dict(hip.timing()); hip.set_timing(False)
    res[n] = (wall, tm)
    hip.close(); del d
(w1, t1), (w8, t8) = res[cfg["n_reads"]], res[cfg["n_reads"] // 8]
print("step: %.2f ms at %d reads, %.2f ms at an eighth (x 8 = %.2f): %.2f ms of the eighth do not scale" % (w1, cfg["n_reads"], w8, 8 * w8, w8 - w1 / 8))
print("%-14s %9s %9s %12s" % ("phase", "2 M", "an eighth", "not scaling"))
for k in t1:
    if t1[k] > 0.0005 or t8.get(k, 0) > 0.0005:
        print("%-14s %9.3f %9.3f %12.3f" % (k, t1[k], t8.get(k, 0.0), t8.get(k, 0.0) - t1[k] / 8))
print("%-14s %9.3f %9.3f %12.3f   (the timers' sum against the step: %.2f / %.2f ms outside any timer)" % ("sum", sum(t1.values()), sum(t8.values()), sum(t8.values()) - sum(t1.values()) / 8, w1 - sum(t1.values()), w8 - sum(t8.values())))
