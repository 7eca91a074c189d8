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
    # a local table that split a 64-bit hash collision holds one hash twice: ranking by hash would fold two different k-mers into one row
    if hk.numel() > 1 and bool((hk[1:] == hk[:-1]).any()):
        raise RuntimeError("this shard's table holds one k-mer hash more than once (a split hash collision): sequence-level merge required")
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
    new = torch.ones(sh.numel(), dtype=torch.bool, device=dev)
    if sh.numel() > 1:
        new[1:] = sh[1:] != sh[:-1]
        # the same k-mer must carry the same s-mer everywhere (syncmer.c:1370-1376); equal hash with different s-mers
        # across GPUs is either that fatal condition or a true 64-bit hash collision -- both need the sequences
        if bool(((~new[1:]) & (ss[1:] != ss[:-1])).any()):
            raise RuntimeError("equal k-mer hash with different s-mers across shards: sequence-level merge required")
    G, S = sh[new], ss[new]
    l2g = torch.searchsorted(G, hk)
    # a local table holds every hash once, so its rows land on distinct global rows: a plain scatter, then the count-table
    # all-reduce over xGMI (32-bit counts: half the bytes of the index type)
    dense = torch.zeros(G.numel(), dtype=torch.int32, device=dev)
    dense[l2g] = cov.to(torch.int32)
    if world > 1:
        dist.all_reduce(dense, group=group)
    return G ^ _BIAS, S, dense.to(torch.int64), l2g


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
            return torch.zeros(0, dtype={"<i8": torch.int64, "<i4": torch.int32, "|u1": torch.uint8}[typestr], device=self.device)
        return torch.as_tensor(_DevView(ptr, n, typestr), device=self.device)

    def merge(self):
        self.hip.sync()
        h = self._tensor("SCM_H", "<i8", 8)
        s = self._tensor("SCM_S", "<i8", 8)
        cov = self._tensor("SCM_COV", "<i4", 4)
        self.result = merge_syncmer_tables(h, s, cov, self.dist)
        torch.cuda.synchronize(self.device)
        return self.result


def _staged(dist):
    """gloo moves host memory only (the CPU tests, and two shards sharing one GPU); RCCL moves device memory"""
    return dist.get_backend() == "gloo"


def gather_var(t, dist):
    """all_gather of tensors whose first dimension differs between ranks -> list of per-rank tensors (rank order)"""
    dev = t.device
    w = t.cpu() if _staged(dist) else t
    n = torch.tensor([w.shape[0]], dtype=torch.int64, device=w.device)
    sizes = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    pad = torch.zeros((max(max(sizes), 1),) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    pad[: w.shape[0]] = w
    out = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(out, pad)
    return [o[:k].to(dev) for o, k in zip(out, sizes)]


def all_reduce(t, dist, op=None):
    op = op if op is not None else dist.ReduceOp.SUM
    if _staged(dist):
        c = t.cpu()
        dist.all_reduce(c, op=op)
        return c.to(t.device)
    dist.all_reduce(t, op=op)
    return t


class ShardedEc:
    """The error-correction round of syncasm (run_syncasm.c:107-134) with reads sharded by record over GPUs: rank r owns a
    contiguous range of read ids and a HipSyncasm context with its scan + count resident.  Every rank ends with the corrected
    chains of ITS reads in global syncmer ids and with the refreshed global syncmer table -- bit-identical to one GPU holding
    all reads (tests/test_gpu_sharded_ec.py).  Exchange steps (include/oatk_hip_ec.h lists the device calls between them):

      1. count tables:  all_gather(sorted hashes) + all_reduce(coverage)                   (CountMerger)
      2. adjacent pairs: all_gather((key, distance) lists) in rank order -- the graph is a property of all reads, every rank
         builds the same one (12 bytes per syncmer occurrence; ~0.5 GB at 2 M reads, against 30 GB of reads)
      3. k-mers of live vertices a rank never saw: owner = lowest rank that has one; all_gather of the few that are needed
      4. refreshed table: all_reduce(coverage, forward-strand counts), block statistics
    """

    def __init__(self, hip, dist, device):
        self.hip, self.dist, self.device = hip, dist, device
        self.merger = CountMerger(hip, dist, device)
        self.n_imported = 0

    def _view(self, name, typestr, itemsize):
        return self.merger._tensor(name, typestr, itemsize)

    def _ready(self):
        """tensors made by torch (its current stream) are about to be read by the HIP context (its own, non-blocking stream)"""
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def run(self, max_edist, c, a):
        hip, dist, dev = self.hip, self.dist, self.device
        rank = dist.get_rank()
        G, S, cov, l2g = self.merger.merge()
        n_global = int(G.numel())
        l2g32, cov32, s64 = l2g.to(torch.int32).contiguous(), cov.to(torch.int32).contiguous(), S.contiguous()
        self._ready()
        hip.ec_set_global(n_global, l2g32.data_ptr(), cov32.data_ptr(), s64.data_ptr())
        # the graph of all reads, from everybody's adjacent pairs in read order
        kp, dp, n = hip.ec_pairs()
        if n:
            keys = torch.as_tensor(_DevView(kp, n, "<i8"), device=dev)
            dd = torch.as_tensor(_DevView(dp, n, "<i4"), device=dev)
        else:
            keys, dd = torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev)
        keys_all = torch.cat(gather_var(keys, dist)).contiguous()
        dist_all = torch.cat(gather_var(dd, dist)).contiguous()
        self._ready()
        hip.ec_graph_from_pairs(keys_all.data_ptr(), dist_all.data_ptr(), int(keys_all.numel()))
        hip.ec_mark(c, a)
        # live vertices without a k-mer on this rank
        err_del = self._view("EC_ERR_DEL", "|u1", 1)
        src = self._view("EC_VTX_SRC", "<i8", 8)
        need = torch.nonzero((err_del == 0) & (src == -1)).flatten()
        if hip.info()["n_occ"] == 0:            # a rank without syncmers corrects nothing and needs no k-mers
            need = need[:0]
        union = torch.unique(torch.cat(gather_var(need, dist)))
        self.n_imported = 0
        if union.numel():
            big = dist.get_world_size()
            owner = torch.where(src[union] != -1, torch.full_like(union, rank), torch.full_like(union, big))
            owner = all_reduce(owner, dist, dist.ReduceOp.MIN)
            if bool((owner == big).any()):
                raise RuntimeError("a live syncmer occurs on no shard")
            mine = union[owner == rank].to(torch.int32).contiguous()
            stride = ((hip.info()["k"] + 3) // 4 + 8 + 15) // 16 * 16
            out = torch.zeros((mine.numel(), stride), dtype=torch.uint8, device=dev)
            rev = torch.zeros(mine.numel(), dtype=torch.uint8, device=dev)
            self._ready()
            hip.ec_export_kmers(mine.data_ptr(), int(mine.numel()), out.data_ptr(), stride, rev.data_ptr())
            ids_all = torch.cat(gather_var(mine, dist))
            rev_all = torch.cat(gather_var(rev, dist))
            km_all = torch.cat(gather_var(out, dist))
            lack = src[ids_all.long()] == -1
            if hip.info()["n_occ"] == 0:
                lack = torch.zeros_like(lack)
            ids_i, rev_i, km_i = ids_all[lack].contiguous(), rev_all[lack].contiguous(), km_all[lack].contiguous()
            self._ready()
            hip.ec_import_kmers(ids_i.data_ptr(), rev_i.data_ptr(), km_i.data_ptr(), int(ids_i.numel()), stride)
            self.n_imported = int(ids_i.numel())
        st = hip.ec_correct(max_edist)
        # the refreshed table of all reads
        cov_g = all_reduce(self._view("EC_SCM_COV", "<i4", 4).to(torch.int64), dist)
        fwd_g = all_reduce(self._view("EC_SCM_FWD", "<i4", 4).to(torch.int64), dist)
        st_g = all_reduce(torch.from_numpy(st.astype(np.int64)).to(dev), dist)
        self.last = {"n_global": n_global, "hash": G, "cov": cov_g, "del": (fwd_g == 0).to(torch.uint8), "stats": st_g.cpu().numpy(),
                     "local_stats": st}
        return self.last

    def consensus(self, min_cov):
        """scg_syncmer_consensus' rounded mean run lengths (syncasm.c:949-1001) for every live syncmer with GLOBAL coverage >= min_cov,
        after run(): every rank adds up its own occurrences (oatk_hip_consensus_ids), totals and counts are all-reduced.  Returns ids,
        rl [n, k], m_seq and, per id, the lowest rank that holds an uncorrected occurrence (its CONS_FIRST names the read whose bases
        the reference prints; -1 where there is none)."""
        hip, dist, dev = self.hip, self.dist, self.device
        cov, dele = self.last["cov"], self.last["del"]
        ids = torch.nonzero((dele == 0) & (cov >= max(int(min_cov), 1))).flatten().to(torch.int32).contiguous()
        n, K = int(ids.numel()), hip.info()["k"]
        self._ready()
        hip.consensus_ids(ids.data_ptr(), n)
        if n:
            tot = self._view("CONS_TOT", "<i8", 8).reshape(n, K).clone()
            m = self._view("CONS_MSEQ", "<i4", 4).to(torch.int64)
            first = self._view("CONS_FIRST", "<i8", 8).clone()
        else:
            tot, m, first = torch.zeros((0, K), dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int64, device=dev)
        tot, m = all_reduce(tot, dist), all_reduce(m, dist)
        big = dist.get_world_size()
        owner = all_reduce(torch.where(first != -1, torch.full_like(first, dist.get_rank()), torch.full_like(first, big)), dist, dist.ReduceOp.MIN)
        owner = torch.where(owner == big, torch.full_like(owner, -1), owner)
        rl = torch.floor(tot.to(torch.float64) / m.clamp(min=1).to(torch.float64).unsqueeze(1) + 0.5).to(torch.int64)     # lround of a quotient >= 0
        rl = torch.where(m.unsqueeze(1) > 0, rl, torch.zeros_like(rl))
        return {"ids": ids, "rl": rl, "m_seq": m, "owner": owner, "first_local": first}


    def asm_graph(self, min_k_cov, min_a_cov_f):
        """make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) (run_syncasm.c:138) of ALL reads after run(): the canonical keys of
        every rank's corrected chains are all-gathered (8 bytes per syncmer occurrence; their order does not matter to a counter), the
        coverage and deletion marks are the all-reduced ones of run(); every rank ends with the same resident graph (AG_* buffers, global
        ids).  Returns (n_vtx, n_arc)."""
        hip, dist, dev = self.hip, self.dist, self.device
        kp, n = hip.asm_pairs()
        keys = torch.as_tensor(_DevView(kp, n, "<i8"), device=dev) if n else torch.zeros(0, dtype=torch.int64, device=dev)
        keys_all = torch.cat(gather_var(keys, dist)).contiguous()
        cov32, del8 = self.last["cov"].to(torch.int32).contiguous(), self.last["del"].contiguous()
        self._ready()
        return hip.asm_graph_from_pairs(keys_all.data_ptr(), int(keys_all.numel()), self.last["n_global"], cov32.data_ptr(), del8.data_ptr(),
                                        int(min_k_cov), float(min_a_cov_f))


    def overlap_hist(self):
        """calc_syncmer_overlap's tables (include/oatk_hip_cons.h) for ALL reads after run(): the (key, distance) lists of the ranks' corrected
        chains all-gathered in rank order -- the order one context holding all reads walks them in -- and the same tables built on every rank
        (OVL_* buffers, global ids).  Returns (n_pairs, n_entries)."""
        hip, dist, dev = self.hip, self.dist, self.device
        kp, dp, n = hip.overlap_pairs()
        if n:
            keys = torch.as_tensor(_DevView(kp, n, "<i8"), device=dev)
            dd = torch.as_tensor(_DevView(dp, n, "<i4"), device=dev)
        else:
            keys, dd = torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev)
        keys_all = torch.cat(gather_var(keys, dist)).contiguous()
        dist_all = torch.cat(gather_var(dd, dist)).contiguous()
        self._ready()
        return hip.overlap_hist_from_pairs(keys_all.data_ptr(), dist_all.data_ptr(), int(keys_all.numel()))

    def stat_raw(self):
        """sr_db_stat's tabulation (include/oatk_hip_stat.h) over ALL reads after run(): the s-mer codes and k-mer keys (global ids) of every rank's
        chain entries all-gathered (16 bytes per entry; the statistics run twice per assembly, not per batch), the additive figures all-reduced,
        the same result on every rank."""
        hip, dist, dev = self.hip, self.dist, self.device
        sp, kp, n, add4 = hip.stat_keys()
        if n:
            sm = torch.as_tensor(_DevView(sp, n, "<i8"), device=dev)
            kk = torch.as_tensor(_DevView(kp, n, "<i8"), device=dev)
        else:
            sm = kk = torch.zeros(0, dtype=torch.int64, device=dev)
        sm_all = torch.cat(gather_var(sm, dist)).contiguous()
        kk_all = torch.cat(gather_var(kk, dist)).contiguous()
        tot = all_reduce(torch.from_numpy(add4).to(dev), dist).cpu().numpy()
        self._ready()
        return hip.stat_from_keys(sm_all.data_ptr(), kk_all.data_ptr(), int(sm_all.numel()), tot)

    def read_alignment(self, graph, old_ra=None):
        """scg_read_alignment (alignment.c:596) with sharded reads: no exchange at all -- the graph (dict shaped like oatk_ra_graph_t, global
        syncmer ids, e.g. built from asm_graph()'s result) is the same on every rank and a read aligns on its own; the alignments of all
        reads are the ranks' results in rank order.  Returns this rank's (n_aln, n_frg, stats3)."""
        return self.hip.read_alignment(graph, old_ra)


def merge_numpy(h_u64, s_u64, cov, dist=None):
    """convenience for the CPU tests: numpy uint64 in, numpy out"""
    h = torch.from_numpy(h_u64.view(np.int64).copy())
    s = torch.from_numpy(s_u64.view(np.int64).copy())
    c = torch.from_numpy(cov.astype(np.int64))
    G, S, C, l2g = merge_syncmer_tables(h, s, c, dist)
    return G.numpy().view(np.uint64), S.numpy().view(np.uint64), C.numpy(), l2g.numpy()
