#!/usr/bin/env python3
"""Times the device EC round (graph + correction) after scan + count on a synthetic read set.  GPU only."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oatk_amd import HipSyncasm
from oatk_amd.synth import ReadSet

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=20000)
ap.add_argument("--genome", type=int, default=1_000_000)
ap.add_argument("--len", type=int, default=15000)
ap.add_argument("--err-ppm", type=int, default=500)
ap.add_argument("-c", type=int, default=30)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--tiers", default="", help="semicolon-separated cap_t0,cap_t1 pairs to try (oatk_hip_debug_ec_tiers), e.g. '2048,16384;3072,16384'")
a = ap.parse_args()
rs = ReadSet(a.genome, a.reads, a.len, err_ppm=a.err_ppm)
seq, off, lens = rs.slice(0, a.reads)
bases = int(lens.sum())
dev = torch.device("cuda", 0)
d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off.view(np.int64)).to(dev); d_len = torch.from_numpy(lens.view(np.int32)).to(dev)
hip = HipSyncasm(0)
def T(f):
    hip.sync(); t = time.perf_counter(); r = f(); hip.sync(); return (time.perf_counter() - t) * 1e3, r
hip.set_timing(True)
for tier in [t for t in a.tiers.split(";") if t]:
    t0_, t1_ = [int(x) for x in tier.split(",")]
    hip._check(hip.L.oatk_hip_debug_ec_tiers(hip.h, t0_, t1_), "tiers")
    hip.scan_device(d_seq.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), a.reads, int(seq.size), 1001, 31)
    hip.count(); hip.ec_graph()
    best = None
    for _ in range(3):
        t_ec, st = T(lambda: hip.ec(0.02, a.c, 0.35))
        tm = hip.timing()
        if best is None or t_ec < best[0]:
            best = (t_ec, tm["ec_mark"], tm["ec_solve"], tm["ec_refresh"], int(st[11]))
    print("tiers %-12s ec %.2f ms (mark %.2f solve %.2f refresh %.2f) past first tier %d" % ((tier,) + best), flush=True)
hip._check(hip.L.oatk_hip_debug_ec_tiers(hip.h, 0, 0), "tiers")
for rep in range(a.reps):
    t_scan, _ = T(lambda: hip.scan_device(d_seq.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), a.reads, int(seq.size), 1001, 31))
    t_cnt, _ = T(hip.count)
    t_g, _ = T(hip.ec_graph)
    t_ec, st = T(lambda: hip.ec(0.02, a.c, 0.35))
    t_cons, _ = T(lambda: hip.consensus(a.c))
    n_sel = len(hip.fetch("CONS_SEL"))
    info = hip.info()
    print("rep %d: %.2f Gbases  scan %.2f ms  count %.2f ms  graph %.2f ms  ec %.2f ms  consensus %.2f ms (%d syncmers)  -> %.2f Gbases/s  | occ %d scm %d  stats %s" % (
        rep, bases / 1e9, t_scan, t_cnt, t_g, t_ec, t_cons, n_sel, bases / (t_scan + t_cnt + t_g + t_ec) / 1e6, info["n_occ"], info["n_scm"], st.tolist()), flush=True)
