"""GPU: sharded reads up to the graph hand-off (include/oatk_hip_multi.h, second half) equal one handle holding all the reads.

After oatk_hip_merge_counts / oatk_hip_ec_sharded the ranks' results are partitioned; what syncasm() consumes next are properties of all reads.
Checked here against ONE handle on the same reads, array for array:
  oatk_hip_gather_table          h, s, cov, del and the per-syncmer occurrence lists in (sid, idx) order (syncmer.c:1353-1360; after the correction
                                 update_syncmer_db's, syncerr.c:796-805), and every read's k_mer as the count leaves it (global id << 1, syncmer.c:1378)
  oatk_hip_asm_graph_sharded     the graph of run_syncasm.c:138 with asmg_finalize's order, flags and link ids
  oatk_hip_consensus_sharded     selection, rounded mean run lengths, counts and first occurrences (syncasm.c:888-1003)
  oatk_hip_overlap_hist_sharded  the distance tables of calc_syncmer_overlap (syncasm.c:477-582) with their first-appearance order and tail flags
  oatk_hip_stat_sharded          sr_db_stat's tabulation (syncmer.c:867-1028) right after the scan and after the correction
The ranks are threads of this process, each with its own handle on the test box's one GPU, over the in-process communicator group."""
import threading

import numpy as np
import pytest

import adversarial as A
import test_gpu_ec as E
from oatk_amd import HipSyncasm, _lib, pack_reads

pytestmark = pytest.mark.gpu

CASES = [
    (301, 21, 6, lambda: A.hifi_like(300, 30000, 4000, seed=307, err=0.004), (0.0, 0.5, 1.0)),
    (1001, 31, 6, lambda: E.sample_reads(E.genome_with_repeats(5, 50000), 260, 9000, 0.001, 6), (0.0, 0.35, 1.0)),
    (101, 11, 5, lambda: E.sample_reads(E.genome_with_repeats(9, 9000, unit=600, copies=3), 300, 1500, 0.004, 10), (0.0, 0.4, 0.4, 1.0)),      # one shard is empty
    (301, 21, 6, lambda: A.hifi_like(300, 30000, 4000, seed=311, err=0.003), (0.0, 0.2, 0.5, 0.7, 1.0)),
    (101, 11, 4, lambda: E.diploid_reads(101, 6000, 150, 500, 1200, 0.006), (0.0, 0.3, 1.0)),      # bubbles: ties in the distance tables, ambiguous blocks
]

AG = ["AG_SCM_DEL", "AG_VTX_SCM", "AG_VTX_COV", "AG_IDX_N", "AG_ARC_V", "AG_ARC_W", "AG_ARC_COV", "AG_ARC_COMP", "AG_ARC_LINK"]
OVL = ["OVL_KEY", "OVL_OFF", "OVL_DIST", "OVL_CNT", "OVL_TAIL"]
CONS = ["CONS_SEL", "CONS_SLOT", "CONS_RL", "CONS_MSEQ", "CONS_FIRST"]


def stat_equal(a, b):
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], (k, a[k], b[k])


def single(hip, reads, K, S, c, a):
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    out = {"stat0": hip.stat_raw()}
    hip.count()
    out["cnt"] = hip.fetch_count()
    out["kid"] = hip.fetch("POS_KID")
    hip.ec_graph(light_c=c)
    hip.ec(0.02, c, a)
    out["ec"] = {k: hip.fetch(k) for k in ["EC_SCM_COV", "EC_SCM_DEL", "EC_SCM_OCC_OFF", "EC_SCM_OCC"]}
    out["stat1"] = hip.stat_raw()
    out["dims"] = hip.asm_graph(c, a)
    out["ag"] = {k: hip.fetch(k) for k in AG}
    ipn, ipp = out["ag"]["AG_IDX_N"], hip.fetch("AG_IDX_P")
    out["ag"]["AG_IDX_P"] = np.where(ipn > 0, ipp, 0)
    hip.consensus(c)
    out["cons"] = {k: hip.fetch(k) for k in CONS}
    hip.overlap_hist()
    out["ovl"] = {k: hip.fetch(k) for k in OVL}
    return out


def run_ranks(world, grp, reads, bounds, K, S, c, a, root=0, make_comm=None, devices=None):
    """make_comm(rank): another communicator factory than the in-process group (tests/mock_rccl_run.py: the RCCL branch over a mock library)"""
    L = _lib.load()
    out, errs = [None] * world, []

    def work(rank):
        try:
            h = HipSyncasm(devices[rank] if devices else 0)      # (real RCCL: a device per rank, tests/test_gpu_real_rccl.py)
            comm = make_comm(rank) if make_comm else L.oatk_comm_group_rank(grp, rank)
            lo, hi = bounds[rank], bounds[rank + 1]
            seq, off, lens = pack_reads(reads[lo:hi])
            h.scan_host(seq, off, lens, K, S, sid0=lo)
            res = {"stat0": h.stat_sharded(comm)}
            h.count()
            h.merge_counts(comm)
            h.gather_table(comm, root)
            res["gkid"] = h.fetch("MG_POS_GKID")
            if rank == root:
                res["tab0"] = {k: h.fetch(k) for k in ["MG_G_H", "MG_G_S", "MG_G_COV", "MG_G_DEL", "MG_G_OCC_OFF", "MG_G_OCC"]}
            h.ec_sharded(comm, 0.02, c, a)
            h.gather_table(comm, root)
            if rank == root:
                res["tab1"] = {k: h.fetch(k) for k in ["MG_G_H", "MG_G_S", "MG_G_COV", "MG_G_DEL", "MG_G_OCC_OFF", "MG_G_OCC"]}
            res["stat1"] = h.stat_sharded(comm)
            res["dims"] = h.asm_graph_sharded(comm, c, a)
            res["ag"] = {k: h.fetch(k) for k in AG}
            res["ag"]["AG_IDX_P"] = np.where(res["ag"]["AG_IDX_N"] > 0, h.fetch("AG_IDX_P"), 0)
            h.consensus_sharded(comm, c)
            res["cons"] = {k: h.fetch(k) for k in CONS}
            res["ovl_dims"] = h.overlap_hist_sharded(comm, c)
            res["ovl"] = {k: h.fetch(k) for k in OVL}
            h.overlap_hist_sharded(comm, 0)
            res["ovl_all"] = {k: h.fetch(k) for k in OVL}
            out[rank] = res
            L.oatk_comm_destroy(comm)
            h.close()
        except Exception as ex:                          # noqa: BLE001
            errs.append((rank, ex))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a rank hangs in a collective"
    assert not errs, errs
    return out


def ovl_subset(ref, keep_vtx):
    """the tables of `ref` restricted to pairs both of whose members are in keep_vtx (bool per syncmer id)"""
    key, off = ref["OVL_KEY"], ref["OVL_OFF"].astype(np.int64)
    a, b = (key >> np.uint64(33)).astype(np.int64), ((key & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.int64)
    sel = keep_vtx[a] & keep_vtx[b]
    n = (off[1:] - off[:-1])
    idx = np.concatenate([np.arange(off[i], off[i + 1]) for i in np.nonzero(sel)[0]]) if sel.any() else np.zeros(0, np.int64)
    new_off = np.concatenate([[0], np.cumsum(n[sel])]).astype(np.uint64)
    return {"OVL_KEY": key[sel], "OVL_OFF": new_off, "OVL_DIST": ref["OVL_DIST"][idx], "OVL_CNT": ref["OVL_CNT"][idx], "OVL_TAIL": ref["OVL_TAIL"][sel]}


def check(out, ref, c, root):
    """every array of the sharded run against one handle's"""
    cnt = ref["cnt"]
    # statistics right after the scan and after the correction: the same on every rank, and one handle's
    for r in out:
        stat_equal(ref["stat0"], r["stat0"])
        stat_equal(ref["stat1"], r["stat1"])
    # the counted table on root, the reads' ids on their ranks
    t0 = out[root]["tab0"]
    for key, want in (("MG_G_H", cnt["h"]), ("MG_G_S", cnt["s"]), ("MG_G_COV", cnt["cov"]), ("MG_G_OCC_OFF", cnt["occ_off"]), ("MG_G_OCC", cnt["occ"])):
        assert np.array_equal(t0[key], want), key
    assert not t0["MG_G_DEL"].any()
    assert np.array_equal(np.concatenate([r["gkid"] for r in out]), ref["kid"])
    # the refreshed table on root
    t1 = out[root]["tab1"]
    for key, want in (("MG_G_H", cnt["h"]), ("MG_G_S", cnt["s"]), ("MG_G_COV", ref["ec"]["EC_SCM_COV"]), ("MG_G_DEL", ref["ec"]["EC_SCM_DEL"]),
                      ("MG_G_OCC_OFF", ref["ec"]["EC_SCM_OCC_OFF"]), ("MG_G_OCC", ref["ec"]["EC_SCM_OCC"])):
        assert np.array_equal(t1[key], want), key
    # graph, consensus, tables: identical on every rank and equal to one handle's
    keep = (ref["ec"]["EC_SCM_DEL"] == 0) & (ref["ec"]["EC_SCM_COV"] >= c)
    ovl_kept = ovl_subset(ref["ovl"], keep)
    assert ref["dims"][0] > 0
    for r in out:
        assert r["dims"] == ref["dims"]
        for k in AG + ["AG_IDX_P"]:
            assert np.array_equal(r["ag"][k], ref["ag"][k]), k
        for k in CONS:
            assert np.array_equal(r["cons"][k], ref["cons"][k]), k
        for k in OVL:
            assert np.array_equal(r["ovl"][k], ovl_kept[k]), k
            assert np.array_equal(r["ovl_all"][k], ref["ovl"][k]), k
        assert r["ovl_dims"] == (len(ovl_kept["OVL_KEY"]), len(ovl_kept["OVL_DIST"]))


@pytest.mark.parametrize("case", range(len(CASES)))
def test_sharded_tail_equals_one_handle(hip, case):
    K, S, c, mk, frac = CASES[case]
    a = 0.35
    reads = mk()
    bounds = [int(round(f * len(reads))) for f in frac]
    world = len(bounds) - 1
    root = world - 1 if case == 3 else 0
    L = _lib.load()
    grp = L.oatk_comm_group_create(world)
    assert grp
    try:
        out = run_ranks(world, grp, reads, bounds, K, S, c, a, root)
    finally:
        L.oatk_comm_group_destroy(grp)
    ref = single(hip, reads, K, S, c, a)
    check(out, ref, c, root)


def test_stat_without_singletons_replays_the_count_table(hip):
    """every read twice: no s-mer or k-mer occurs exactly once, and the reference then reports a stale variable that depends on the layout of its
    count table (kh_ctab_stat, syncmer.c:637-643) -- the owners' counts are gathered in key order and the table is replayed"""
    K, S = 101, 11
    base = A.hifi_like(40, 6000, 1500, seed=9, err=0.0)
    reads = base + base
    bounds = [0, 25, 60, len(reads)]
    L = _lib.load()
    grp = L.oatk_comm_group_create(3)
    out, errs = [None] * 3, []

    def work(rank):
        try:
            h = HipSyncasm(0)
            comm = L.oatk_comm_group_rank(grp, rank)
            seq, off, lens = pack_reads(reads[bounds[rank]:bounds[rank + 1]])
            h.scan_host(seq, off, lens, K, S, sid0=bounds[rank])
            out[rank] = h.stat_sharded(comm)
            L.oatk_comm_destroy(comm)
            h.close()
        except Exception as ex:                          # noqa: BLE001
            errs.append((rank, ex))

    th = [threading.Thread(target=work, args=(r,)) for r in range(3)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    L.oatk_comm_group_destroy(grp)
    assert not any(t.is_alive() for t in th) and not errs, errs
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    ref = hip.stat_raw()
    assert ref["smer_cnt"][1] == 0 and ref["kmer_cnt"][1] == 0
    for r in out:
        stat_equal(ref, r)


def test_a_failing_rank_releases_the_others(hip):
    """a rank that fails between two collectives poisons the group: the other ranks come back with an error instead of waiting for ever"""
    K, S = 101, 11
    reads = A.hifi_like(60, 6000, 1500, seed=3, err=0.002)
    L = _lib.load()
    grp = L.oatk_comm_group_create(2)
    res = [None, None]

    def work(rank):
        h = HipSyncasm(0)
        comm = L.oatk_comm_group_rank(grp, rank)
        seq, off, lens = pack_reads(reads[rank * 30:(rank + 1) * 30])
        h.scan_host(seq, off, lens, K, S, sid0=rank * 30)
        if rank == 0:
            h.count()                                    # rank 1 never counted: its merge fails on the spot (call order)
        res[rank] = L.oatk_hip_merge_counts(h.h, comm, None)
        L.oatk_comm_destroy(comm)
        h.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=60) for t in th]
    alive = any(t.is_alive() for t in th)
    L.oatk_comm_group_destroy(grp) if not alive else None
    assert not alive, "a rank hangs although its peer failed"
    assert res[0] != 0 and res[1] != 0
