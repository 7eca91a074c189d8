#!/usr/bin/env python3
"""How much does the graph the correction runs against BRANCH?  (development aid)  Out-degree of every oriented vertex of the EC graph of a workload.
    python tools/graph_branching.py [config3|config2|config1s] [reads]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oatk_amd import HipSyncasm
from oatk_amd.synth import CONFIGS, CONFIG1S, MixReadSet, ReadSet

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
cfg = dict(CONFIG1S if wl == "config1s" else CONFIGS[wl])
if len(sys.argv) > 2:
    cfg["n_reads"] = int(sys.argv[2])
c = int(cfg.get("min_k_cov", 30))
rs = MixReadSet(**cfg) if wl == "config1s" else ReadSet(**cfg)
seq, off, lens = rs.slice(0, cfg["n_reads"])
dev = torch.device("cuda", 0)
d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off.view(np.int64)).to(dev); d_len = torch.from_numpy(lens.view(np.int32)).to(dev)
hip = HipSyncasm(0)
hip.scan_device(d_seq.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), cfg["n_reads"], int(seq.size), 1001, 31)
hip.count()
hip.ec_graph(light_c=c)
n = hip.fetch("EG_IDX_N")
cov = hip.fetch("EG_ARC_COV")
print("%s, %d reads, -c %d: %d oriented vertices, %d arcs" % (wl, cfg["n_reads"], c, n.size, cov.size))
h = np.bincount(np.minimum(n, 8))
print("out-degree histogram (8 = 8 or more):", h.tolist())
print("vertices with more than one arc out: %d (%.4f %%)" % (int((n > 1).sum()), 100.0 * float((n > 1).sum()) / max(1, n.size)))
