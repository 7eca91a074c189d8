"""Multi-GPU merge of per-GPU syncmer tables: one process per GPU, RCCL over xGMI.

Reads shard by record -- rank r owns a contiguous range of read ids -- so the scan and the local count need no
communication.  The one exchange step is the merge of the per-GPU syncmer tables: syncmer IDs in the reference are
ranks in the sorted order of k-mer hashes (syncmer.c:1419-1438), so every rank needs the same global key array.

    all_gather(sorted unique hashes)  ->  identical merged key array G on every rank (global id = rank in G)
    local coverage scattered into a dense vector over G  ->  all_reduce(sum)        (the count-table all-reduce)

The collectives go through torch.distributed ("nccl" = RCCL on the GPUs, "gloo" in the CPU tests); the arithmetic in
between is tensor plumbing (sort / unique / searchsorted) on the device the tables live on.
"""
import numpy as np
import torch

_BIAS = -(1 << 63)          # xor with the sign bit: unsigned order of uint64 == signed order of int64


def _to_ordered_i64(u64_tensor_as_i64):
    return u64_tensor_as_i64 ^ _BIAS


def merge_syncmer_tables(h, s, cov, dist=None, group=None):
    """h, s: int64 tensors holding the raw uint64 bit patterns of the local table (h ascending as unsigned);
    cov: integer tensor.  Returns (G_h, G_s, G_cov, local_to_global) with G_* identical on every rank."""
    dev = h.device
    hk = _to_ordered_i64(h)
    n_local = torch.tensor([h.numel()], dtype=torch.int64, device=dev)
    world = dist.get_world_size(group) if dist is not None else 1
    if world > 1:
        sizes = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes, n_local, group=group)
        sizes = [int(x.item()) for x in sizes]
        nmax = max(max(sizes), 1)

        def gather(t, fill):
            pad = torch.full((nmax,), fill, dtype=t.dtype, device=dev)
            pad[: t.numel()] = t
            out = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(out, pad, group=group)
            return torch.cat([o[:n] for o, n in zip(out, sizes)])

        all_h = gather(hk, torch.iinfo(torch.int64).max)
        all_s = gather(s, 0)
    else:
        all_h, all_s = hk, s
    order = torch.argsort(all_h, stable=True)
    sh, ss = all_h[order], all_s[order]
    G, inverse = torch.unique_consecutive(sh, return_inverse=True)
    # the same k-mer must carry the same s-mer everywhere (syncmer.c:1370-1376); equal hash with different s-mers
    # across GPUs is either that fatal condition or a true 64-bit hash collision -- both need the sequences
    smin = torch.full((G.numel(),), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev).scatter_reduce(0, inverse, ss, "amin")
    smax = torch.full((G.numel(),), torch.iinfo(torch.int64).min, dtype=torch.int64, device=dev).scatter_reduce(0, inverse, ss, "amax")
    if bool((smin != smax).any()):
        raise RuntimeError("equal k-mer hash with different s-mers across shards: sequence-level merge required")
    l2g = torch.searchsorted(G, hk)
    dense = torch.zeros(G.numel(), dtype=torch.int64, device=dev)
    dense.index_add_(0, l2g, cov.to(torch.int64))
    if world > 1:
        dist.all_reduce(dense, group=group)         # the count-table all-reduce over xGMI
    return G ^ _BIAS, smin, dense, l2g


class _DevView:
    """zero-copy view of a device buffer owned by the HIP context (numba-style array interface)"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class CountMerger:
    """merge the resident per-GPU syncmer table of a HipSyncasm context across the ranks of a process group"""

    def __init__(self, hip, dist, device):
        self.hip, self.dist, self.device = hip, dist, device
        self.result = None

    def _tensor(self, name, typestr, itemsize):
        ptr, nbytes = self.hip.buffer(name)
        n = nbytes // itemsize
        if n == 0:
            return torch.zeros(0, dtype={"<i8": torch.int64, "<i4": torch.int32}[typestr], device=self.device)
        return torch.as_tensor(_DevView(ptr, n, typestr), device=self.device)

    def merge(self):
        self.hip.sync()
        h = self._tensor("SCM_H", "<i8", 8)
        s = self._tensor("SCM_S", "<i8", 8)
        cov = self._tensor("SCM_COV", "<i4", 4)
        self.result = merge_syncmer_tables(h, s, cov, self.dist)
        torch.cuda.synchronize(self.device)
        return self.result


def merge_numpy(h_u64, s_u64, cov, dist=None):
    """convenience for the CPU tests: numpy uint64 in, numpy out"""
    h = torch.from_numpy(h_u64.view(np.int64).copy())
    s = torch.from_numpy(s_u64.view(np.int64).copy())
    c = torch.from_numpy(cov.astype(np.int64))
    G, S, C, l2g = merge_syncmer_tables(h, s, c, dist)
    return G.numpy().view(np.uint64), S.numpy().view(np.uint64), C.numpy(), l2g.numpy()
