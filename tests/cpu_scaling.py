#!/usr/bin/env python3
"""Reference (oracle/_ref) scan + count [+ EC round] on the host cores at several thread counts (development aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oatk_amd.synth import CONFIGS, ReadSet
cfg = dict(CONFIGS["config2"])
rs = ReadSet(**cfg)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
for t in [int(x) for x in (sys.argv[2:] or ["8", "32", "64", "128"])]:
    r = bench.cpu_baseline(rs, 0, n, 1001, 31, t, 30)
    print("threads %3d: scan+count %.3f Gbases/s   with syncerr %.3f Gbases/s" % (t, r["value"], r["with_syncerr"]["value"]), flush=True)
