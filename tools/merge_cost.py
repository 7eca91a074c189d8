#!/usr/bin/env python3
"""Compute cost of the count-table merge (oatk_amd/multi.py) at the scale of N ranks x config-2 tables, on one GPU: the collectives are
replaced by local concatenation, so this is what every rank computes between them (development aid for the N-GPU scaling)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oatk_amd import multi

dev = torch.device("cuda", 0)
n_local = 1_336_000


class FakeDist:
    """all ranks hold a table of the same size; rank 0's view"""
    def __init__(self, world, tabs): self.world, self.tabs, self.k = world, tabs, 0
    def get_world_size(self, group=None): return self.world
    def all_gather(self, out, t, group=None):
        if t.numel() == 1:
            for o in out: o.copy_(t)
        else:
            src = self.tabs[self.k % 2]; self.k += 1
            for o, s in zip(out, src): o[: s.numel()] = s
    def all_reduce(self, t, group=None): t.mul_(1)


for world in (2, 4, 8):
    g = torch.Generator(device=dev); g.manual_seed(world)
    hs = [torch.sort(torch.randint(-2**62, 2**62, (n_local,), device=dev, dtype=torch.int64, generator=g))[0] for _ in range(world)]
    common = hs[0][:2000].clone()
    hs = [torch.sort(torch.cat([t[2000:], common]))[0] for t in hs]
    ss = [(t ^ 0x5555) & 0x3FFFFFFFFFFFFFFF for t in hs]
    cov = torch.ones(n_local, dtype=torch.int32, device=dev)
    fd = FakeDist(world, [[t ^ multi._BIAS for t in hs], ss])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        fd.k = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        G, S, C, l2g = multi.merge_syncmer_tables(hs[0] ^ multi._BIAS ^ multi._BIAS, ss[0], cov, fd)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("N=%d: merged table %d rows, local compute %.2f ms; all_gather %.0f MB, all_reduce %.0f MB per rank"
          % (world, G.numel(), best * 1e3, world * n_local * 16 / 1e6, G.numel() * 4 / 1e6), flush=True)
