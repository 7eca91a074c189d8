#!/usr/bin/env python3
"""Throughput of the device record scan (oatk_hip_ingest) on synthetic FASTA text: kernels alone (text resident) and with the PCIe copy."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oatk_amd import HipSyncasm
from oatk_amd.synth import ReadSet

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=100000)
ap.add_argument("--wrap", type=int, default=0)
a = ap.parse_args()
rs = ReadSet(1_000_000, a.reads, 15000)
seq, off, lens = rs.slice(0, a.reads)
t0 = time.perf_counter()
parts = []
for i in range(a.reads):
    r = seq[int(off[i]):int(off[i]) + int(lens[i])].tobytes()
    parts.append(b">r%d\n" % i)
    if a.wrap:
        parts.append(b"\n".join(r[j:j + a.wrap] for j in range(0, len(r), a.wrap)) + b"\n")
    else:
        parts.append(r + b"\n")
text = np.frombuffer(b"".join(parts), dtype=np.uint8)
print("text %.2f GB built in %.1f s" % (text.size / 1e9, time.perf_counter() - t0), flush=True)
dev = torch.device("cuda", 0)
hip = HipSyncasm(0)
pinned = torch.from_numpy(text.copy()).pin_memory()
for rep in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    d = pinned.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    t_copy = time.perf_counter() - t
    t = time.perf_counter()
    n, used = hip.ingest_device(d.data_ptr(), int(d.numel()), 1, True)
    hip.sync()
    t_ing = time.perf_counter() - t
    t = time.perf_counter()
    hip.scan_ingested(1001, 31)
    hip.sync()
    t_scan = time.perf_counter() - t
    bases = int(lens.sum())
    print("rep %d: %d reads  H2D %.1f ms (%.1f GB/s)  ingest %.2f ms (%.1f GB/s of text)  scan %.2f ms  -> file-to-syncmers %.1f Gbases/s (with PCIe), %.1f (text resident)"
          % (rep, n, t_copy * 1e3, text.size / t_copy / 1e9, t_ing * 1e3, text.size / t_ing / 1e9, t_scan * 1e3,
             bases / (t_copy + t_ing + t_scan) / 1e9, bases / (t_ing + t_scan) / 1e9), flush=True)
