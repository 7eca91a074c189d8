#!/usr/bin/env python3
"""BASELINE.json configs[2] (2 M reads x ~15 kb = 30 Gbases) through every device row once, with wall-clock per call (development aid;
bench.py is the contract).  usage: python tools/config3_run.py [workload]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oatk_amd import HipSyncasm  # noqa: E402
from oatk_amd.synth import CONFIGS, ReadSet  # noqa: E402

cfg = dict(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "config3"])
K, S, c = cfg.get("k", 1001), cfg.get("s", 31), cfg.get("min_k_cov", 30)
n = cfg["n_reads"]
t0 = time.perf_counter()
rs = ReadSet(**cfg)
seq, off, lens = rs.slice(0, n)
print("generated %d reads, %.2f Gbases in %.1f s" % (n, lens.sum() / 1e9, time.perf_counter() - t0), flush=True)
hip = HipSyncasm(0)


def timed(name, fn):
    hip.sync()
    t = time.perf_counter()
    r = fn()
    hip.sync()
    print("%-18s %9.2f ms  %s" % (name, (time.perf_counter() - t) * 1e3, r if r is not None else ""), flush=True)
    return r


timed("scan (H2D incl.)", lambda: hip.scan_host(seq, off, lens, K, S))
timed("count", hip.count)
timed("stat", lambda: hip.stat_raw()["kmer_unique"])
timed("ec_graph", hip.ec_graph)
st = timed("ec", lambda: hip.ec(0.02, c, 0.35))
timed("stat after ec", lambda: hip.stat_raw()["kmer_unique"])
timed("consensus", lambda: hip.consensus(c))
timed("overlap_hist", hip.overlap_hist)
nv, na = timed("asm_graph", lambda: hip.asm_graph(c, 0.35))
ag = hip.fetch_asm_graph()
ns = len(ag["scm_del"])
su_off = np.zeros(ns + 1, np.uint64)
su_off[1:] = np.cumsum(ag["scm_del"] == 0)
graph = {"n_scm": ns, "su_off": su_off, "su_uid": np.arange(nv, dtype=np.uint64) << np.uint64(1), "su_pos": np.zeros(nv, np.uint32),
         "utg_n": np.ones(nv, np.uint32), "idx_p": ag["idx_p"], "idx_n": ag["idx_n"].astype(np.uint64), "arc_w": ag["arc_w"],
         "arc_ln": np.zeros(na, np.uint64), "arc_del": np.zeros(na, np.uint8)}
timed("read_alignment", lambda: hip.read_alignment(graph)[:2])
print("info", hip.info())
