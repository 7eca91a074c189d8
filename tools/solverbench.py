#!/usr/bin/env python3
"""A/B of the error-block solver on one resident batch (GPU only): the EC round is repeated under different environment switches of
oatk_hip_ec_correct (OATK_DEBUG_EC_QUAD, OATK_DEBUG_EC_WAVES, ...), printing the phase timers each time.
    python tools/solverbench.py --workload config3 --set OATK_DEBUG_EC_QUAD=0 --set OATK_DEBUG_EC_QUAD=1 --set "OATK_DEBUG_EC_QUAD=1 OATK_DEBUG_EC_WAVES=6" """
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oatk_amd import HipSyncasm
from oatk_amd.synth import CONFIGS, CONFIG1S, MixReadSet, ReadSet

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config3")
ap.add_argument("--reads", type=int, default=0)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
cfg = dict(CONFIG1S if a.workload == "config1s" else CONFIGS[a.workload])          # (config1s: the config-1 surrogate, oatk_amd/synth.py)
if a.reads:
    cfg["n_reads"] = a.reads
c = int(cfg.get("min_k_cov", 30))
rs = MixReadSet(**cfg) if a.workload == "config1s" else ReadSet(**cfg)
seq, off, lens = rs.slice(0, cfg["n_reads"])
dev = torch.device("cuda", 0)
d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off.view(np.int64)).to(dev); d_len = torch.from_numpy(lens.view(np.int32)).to(dev)
hip = HipSyncasm(0)
hip.set_timing(True)
hip.scan_device(d_seq.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), cfg["n_reads"], int(seq.size), 1001, 31)
hip.count()
ref = None
for setting in (a.set or [""]):
    keys = []
    for kv in setting.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
        keys.append(k)
    best = None
    for _ in range(a.reps):
        hip.ec_graph(light_c=c)
        hip.sync(); t0 = time.perf_counter()
        st = hip.ec(0.02, c, 0.35)
        hip.sync(); dt = (time.perf_counter() - t0) * 1e3
        tm = hip.timing()
        if best is None or tm["ec_solve"] < best[1]["ec_solve"]:
            best = (dt, tm, st)
    import zlib
    crc = zlib.crc32(hip.fetch("EC_KMER").view(np.uint8))
    if ref is None:
        ref = crc
    print("%-60s ec %.2f ms: mark %.2f solve %.2f refresh %.2f | blocks %d past first tier %d | chains crc %08x %s" % (
        setting or "(default)", best[0], best[1]["ec_mark"], best[1]["ec_solve"], best[1]["ec_refresh"], int(best[2][0] + best[2][5] + best[2][10]), int(best[2][11]),
        crc, "same" if crc == ref else "DIFFERENT"), flush=True)
    for k in keys:
        del os.environ[k]
