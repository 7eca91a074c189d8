#!/usr/bin/env python3
"""(needs tools/experiments/liboatk_hip_experiments.so: bash tools/experiments/build.sh, then OATK_HIP_LIB=... python tools/edbench.py)
SURVEY 7-5 "benchmark both": the wavefront (Landau-Vishkin) edit distance the correction uses against Myers' bit-vector algorithm, on pairs shaped like
error blocks -- a target of ~2000 hoco bases, a query that is the target with a few differences plus the overhang of the last appended k-mer, band 2 % --
and on the other extreme, unrelated strings.  GPU only.  Output goes to profiles/ by hand."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oatk_amd import HipSyncasm

rng = np.random.default_rng(5)
def dna(n): return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tolist())
def mutate(t, k):
    q = bytearray(t)
    for _ in range(k):
        p, kind = int(rng.integers(0, len(q))), int(rng.integers(0, 3))
        if kind == 0: q[p] = b"ACGT"[int(rng.integers(0, 4))]
        elif kind == 1: q.insert(p, b"ACGT"[int(rng.integers(0, 4))])
        else: del q[p]
    return bytes(q)
hip = HipSyncasm(0)
for name, n, tl, edits, over, band in (("error blocks: 2000-base target, 0-4 differences, 300-base overhang, band 40", 32768, 2000, 4, 300, 40),
                                       ("short blocks: 300-base target, 0-2 differences, 100-base overhang, band 6", 65536, 300, 2, 100, 6),
                                       ("no band, 1 % differences: 1000-base target", 16384, 1000, 10, 0, -1)):
    pairs = []
    for _ in range(n // 64):
        t = dna(tl)
        for _ in range(64):
            pairs.append((t, mutate(t, int(rng.integers(0, edits + 1))) + dna(over), band))
    res = {}
    for myers in (False, True):
        best = None
        for rep in range(3):
            out, ms = hip.ed_ab(pairs, myers)
            best = ms if best is None or ms < best else best
        res[myers] = (out, best)
    same = res[False][0] == res[True][0]
    print("%-95s %6d pairs: wavefront %8.3f ms (%6.1f ns/pair, one wave per pair)   Myers %8.3f ms (%7.1f ns/pair, one lane per pair)   ratio %.1fx   same results: %s"
          % (name, n, res[False][1], res[False][1] * 1e6 / n, res[True][1], res[True][1] * 1e6 / n, res[True][1] / res[False][1], same), flush=True)
