#!/usr/bin/env python3
"""Quick per-phase timing of scan + count on synthetic reads (development aid; bench.py is the contract)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oatk_amd import HipSyncasm  # noqa: E402
from oatk_amd.synth import ReadSet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=50000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--mean-len", type=int, default=15000)
a = ap.parse_args()
rs = ReadSet(genome_len=1_000_000, n_reads=a.reads, mean_len=a.mean_len)
seq, off, lens = rs.slice(0, a.reads)
hip = HipSyncasm(0)
hip.set_timing(True)
acc = {}
for it in range(a.steps + 1):
    try:
        hip.scan_host(seq, off, lens, 1001, 31)
        hip.count()
    except Exception as ex:        # experimental builds with phases compiled out produce garbage the count refuses; the timers still hold
        if it == 1:
            print("(%s)" % str(ex)[:80])
    if it:
        for k, v in hip.timing().items():
            acc[k] = acc.get(k, 0) + v / a.steps
bases = int(lens.sum())
print("hoco positions per dispatch %d" % int(hip.fetch("HOCO_L").astype(np.uint64).sum()))
print("reads %d bases %.3f G  " % (a.reads, bases / 1e9) + "  ".join("%s %.3f" % kv for kv in acc.items()))
tot = sum(v for k, v in acc.items() if k not in ("scan_post",))
print("device ms/step %.3f -> %.1f Gbases/s ; hpc %.1f GB/s" % (tot, bases / tot / 1e6, (bases * 1.9375) / acc["hpc"] / 1e6))
