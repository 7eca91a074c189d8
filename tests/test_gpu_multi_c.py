"""GPU: reads sharded by record through the C collectives (include/oatk_hip_multi.h: oatk_hip_merge_counts, oatk_hip_ec_sharded) equal one handle
holding all the reads -- merged table (partitioned by hash range: the ranks' ranges in rank order ARE the table of one handle), global ids and
coverage of every shard's own syncmers, chains in global ids, refreshed coverage and deletion flags, block statistics, imported k-mers.

The test box has one GPU and RCCL refuses two ranks on one device, so the N-rank cases run the ranks as threads of this process, each with its
own handle, over the in-process communicator group (same code path above the three primitives; device-to-device copies instead of xGMI); the RCCL
backend itself -- librccl loaded at run time, communicator from a unique id, ncclAllGather / grouped ncclBroadcast / ncclAllReduce on the handle's
stream -- runs with a world of one."""
import ctypes as C
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
    # a shard of a handful of reads sees few of the good syncmers: their k-mers must be imported
    (101, 11, 5, lambda: E.sample_reads(E.genome_with_repeats(9, 9000, unit=600, copies=3), 300, 1500, 0.004, 10), (0.0, 0.02, 1.0)),
    # three shards, one of them empty; four shards
    (101, 11, 5, lambda: E.sample_reads(E.genome_with_repeats(9, 9000, unit=600, copies=3), 300, 1500, 0.004, 10), (0.0, 0.4, 0.4, 1.0)),
    (301, 21, 6, lambda: A.hifi_like(300, 30000, 4000, seed=311, err=0.003), (0.0, 0.2, 0.5, 0.7, 1.0)),
]


def single(hip, reads, K, S, c):
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    cnt = hip.fetch_count()
    hip.ec_graph()
    st = hip.ec(0.02, c, 0.35)
    want = {k: hip.fetch(k) for k in ["EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER", "EC_SCM_COV", "EC_SCM_DEL"]}
    return cnt, st, want


def run_ranks(world, make_comm, reads, bounds, K, S, c, mask=None, devices=None):
    out, errs = [None] * world, []

    def work(rank):
        try:
            h = HipSyncasm(devices[rank] if devices else 0)      # (real RCCL: a device per rank, tests/test_gpu_real_rccl.py)
            if mask is not None:
                h.debug_hash_mask(mask)
            comm = make_comm(rank)
            lo, hi = bounds[rank], bounds[rank + 1]
            seq, off, lens = pack_reads(reads[lo:hi])
            h.scan_host(seq, off, lens, K, S, sid0=lo)
            h.count()
            ng = h.merge_counts(comm)
            merged = {k: h.fetch(k) for k in ("MG_H", "MG_S", "MG_COV", "MG_L2G", "MG_LCOV")}
            merged["range"] = h.multi_range()
            local_h = h.fetch("SCM_H")
            st, n_imp = h.ec_sharded(comm, 0.02, c, 0.35)
            res = {k: h.fetch(k) for k in ("EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER", "MG_EC_COV", "MG_EC_DEL")}
            out[rank] = (ng, merged, local_h, st, n_imp, res)
            _lib.load().oatk_comm_destroy(comm)
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


@pytest.mark.parametrize("graph", ["light", "full"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_sharded_through_the_c_collectives_equals_one_handle(hip, case, graph, monkeypatch):
    """graph = light: candidates' pairs travel as weighted segments plus one flag per oriented candidate (the default when err_arc_c >= err_mer_c);
    full: every pair of every shard travels (any thresholds)"""
    if graph == "full":
        monkeypatch.setenv("OATK_DEBUG_FULL_GRAPH", "1")
    K, S, c, mk, frac = CASES[case]
    reads = mk()
    bounds = [int(round(f * len(reads))) for f in frac]
    world = len(bounds) - 1
    L = _lib.load()
    grp = L.oatk_comm_group_create(world)
    assert grp
    try:
        out = run_ranks(world, lambda r: L.oatk_comm_group_rank(grp, r), reads, bounds, K, S, c)
    finally:
        L.oatk_comm_group_destroy(grp)
    cnt, st, want = single(hip, reads, K, S, c)
    order = np.argsort(cnt["h"], kind="stable")         # the merged table is in hash order; one handle numbers syncmers the same way
    assert np.array_equal(order, np.arange(len(order)))
    first = 0
    for rank, (ng, mg, local_h, st_r, n_imp, res) in enumerate(out):
        assert ng == cnt["n_scm"]
        f, no, g = mg["range"]
        assert f == first and g == ng and no == len(mg["MG_H"])           # the ranges tile the table in rank order
        first += no
        if no:                                                                # rank r owns the hashes with floor(h * world / 2^64) == r
            assert int(mg["MG_H"][0]) * world >> 64 == rank and int(mg["MG_H"][-1]) * world >> 64 == rank
        l2g = mg["MG_L2G"].astype(np.int64)
        assert np.array_equal(cnt["h"][l2g], local_h) and np.array_equal(cnt["cov"][l2g], mg["MG_LCOV"])
        assert st_r[:11].tolist() == st[:11].tolist()
    assert first == cnt["n_scm"]
    for key, ref in (("MG_H", cnt["h"]), ("MG_S", cnt["s"]), ("MG_COV", cnt["cov"])):
        assert np.array_equal(np.concatenate([o[1][key] for o in out]), ref), key
    assert np.array_equal(np.concatenate([o[5]["MG_EC_COV"] for o in out]), want["EC_SCM_COV"])
    assert np.array_equal(np.concatenate([o[5]["MG_EC_DEL"] for o in out]), want["EC_SCM_DEL"])
    assert np.array_equal(np.concatenate([o[5]["EC_N_SCM"] for o in out]), want["EC_N_SCM"])
    for key in ("EC_KMER", "EC_MPOS", "EC_SMER"):
        assert np.array_equal(np.concatenate([o[5][key] for o in out]), want[key]), key
    assert int(st[0] + st[5] + st[10]) > 0
    if case == 2:
        assert sum(o[4] for o in out) > 0                 # k-mers did travel


@pytest.mark.parametrize("mask", [0xFF, 0x3FF])
def test_sharded_with_forced_hash_collisions(hip, mask):
    """hashes ANDed down to a few bits (the hook of tests/test_gpu_scan.py::test_forced_hash_collisions): unrelated k-mers share a hash inside a
    shard and across shards.  The owners compare the k-mers and cluster them in first-seen order, so ids, coverage and everything downstream
    still equal one handle's -- which splits the same collisions on its own (process_kmer_cluster, syncmer.c:1293-1335)."""
    K, S, c = 101, 11, 4
    reads = A.hifi_like(120, 5000, 1500, seed=5, err=0.004)
    bounds = [0, 35, 80, len(reads)]
    L = _lib.load()
    grp = L.oatk_comm_group_create(3)
    try:
        out = run_ranks(3, lambda r: L.oatk_comm_group_rank(grp, r), reads, bounds, K, S, c, mask=mask)
    finally:
        L.oatk_comm_group_destroy(grp)
    hip.debug_hash_mask(mask)
    try:
        cnt, st, want = single(hip, reads, K, S, c)
        assert hip.info()["collisions"] == 1
    finally:
        hip.debug_hash_mask(0xFFFFFFFFFFFFFFFF)
    assert len(np.unique(cnt["h"])) < cnt["n_scm"]                       # several syncmers per "hash"
    for key, ref in (("MG_H", cnt["h"]), ("MG_S", cnt["s"]), ("MG_COV", cnt["cov"])):
        assert np.array_equal(np.concatenate([o[1][key] for o in out]), ref), key
    for rank, (ng, mg, local_h, st_r, n_imp, res) in enumerate(out):
        assert ng == cnt["n_scm"] and st_r[:11].tolist() == st[:11].tolist()
        l2g = mg["MG_L2G"].astype(np.int64)
        assert np.array_equal(cnt["h"][l2g], local_h) and np.array_equal(cnt["cov"][l2g], mg["MG_LCOV"]) and len(np.unique(l2g)) == len(l2g)
    for key in ("EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER"):
        assert np.array_equal(np.concatenate([o[5][key] for o in out]), want[key]), key
    assert np.array_equal(np.concatenate([o[5]["MG_EC_COV"] for o in out]), want["EC_SCM_COV"])
    assert np.array_equal(np.concatenate([o[5]["MG_EC_DEL"] for o in out]), want["EC_SCM_DEL"])


def test_sharded_with_no_candidates_at_all(hip):
    """a coverage threshold nothing reaches: no candidate is announced, no segment travels, every syncmer is deleted, no read changes"""
    K, S, c = 301, 21, 5000
    reads = A.hifi_like(200, 30000, 4000, seed=331, err=0.003)
    bounds = [0, 70, 70, len(reads)]
    L = _lib.load()
    grp = L.oatk_comm_group_create(3)
    try:
        out = run_ranks(3, lambda r: L.oatk_comm_group_rank(grp, r), reads, bounds, K, S, c)
    finally:
        L.oatk_comm_group_destroy(grp)
    cnt, st, want = single(hip, reads, K, S, c)
    assert int(st[:11].sum()) == 0
    for key in ("EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER"):
        assert np.array_equal(np.concatenate([o[5][key] for o in out]), want[key]), key
    assert np.array_equal(np.concatenate([o[5]["MG_EC_COV"] for o in out]), want["EC_SCM_COV"])
    assert np.array_equal(np.concatenate([o[5]["MG_EC_DEL"] for o in out]), want["EC_SCM_DEL"])
    assert all(o[3][:11].tolist() == st[:11].tolist() for o in out)


def test_rccl_backend_world_of_one(hip):
    """the RCCL code path itself: unique id, ncclCommInitRank, all three primitives on the handle's stream"""
    K, S, c = 301, 21, 6
    reads = A.hifi_like(300, 30000, 4000, seed=313, err=0.004)
    L = _lib.load()
    uid = (C.c_uint8 * 128)()
    assert L.oatk_comm_unique_id(uid) == 0, "librccl could not be loaded"
    comm = L.oatk_comm_create(uid, 0, 1, 0)
    assert comm and L.oatk_comm_backend(comm) == b"rccl" and L.oatk_comm_size(comm) == 1 and L.oatk_comm_rank(comm) == 0
    out = run_ranks(1, lambda r: comm, reads, [0, len(reads)], K, S, c)       # (run_ranks destroys the communicator)
    cnt, st, want = single(hip, reads, K, S, c)
    ng, mg, local_h, st_r, n_imp, res = out[0]
    assert ng == cnt["n_scm"] and np.array_equal(mg["MG_H"], cnt["h"]) and np.array_equal(mg["MG_COV"], cnt["cov"]) and mg["range"] == (0, ng, ng)
    assert np.array_equal(mg["MG_L2G"], np.arange(ng, dtype=np.uint32)) and n_imp == 0
    for key in ("EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER"):
        assert np.array_equal(res[key], want[key]), key
    assert np.array_equal(res["MG_EC_COV"], want["EC_SCM_COV"]) and np.array_equal(res["MG_EC_DEL"], want["EC_SCM_DEL"])
    assert st_r[:11].tolist() == st[:11].tolist()


@pytest.mark.parametrize("which,world", [("config2", 4), ("config3", 2), ("config3", 8)])
def test_full_size_over_several_ranks(hip, which, world):
    """BASELINE.json configs[1] (200 k reads x ~15 kb, k = 1001) sharded over four ranks, configs[2] (2 M reads, the headline workload) over two, and
    over EIGHT -- which is configs[3], "2 M reads read-sharded across 8 MI355X", with the eight ranks on the one GPU of the test box -- through the
    C collectives equal one handle: the ranks' ranges of the merged table, every corrected chain, the refreshed table, the statistics"""
    import zlib
    from oatk_amd.synth import CONFIGS, ReadSet
    cfg = dict(CONFIGS[which])
    rs = ReadSet(**cfg)
    n, c = cfg["n_reads"], cfg["min_k_cov"]
    bounds = [n * r // world for r in range(world + 1)]
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).view(np.uint8))      # noqa: E731
    out, errs = [None] * world, []
    L = _lib.load()
    grp = L.oatk_comm_group_create(world)
    parts = [rs.slice(bounds[r], bounds[r + 1] - bounds[r]) for r in range(world)]       # (first, count)

    def work(rank):
        try:
            h = HipSyncasm(0)
            comm = L.oatk_comm_group_rank(grp, rank)
            seq, off, lens = parts[rank]
            h.scan_host(seq, off, lens, 1001, 31, sid0=bounds[rank])
            parts[rank] = None
            h.count()
            ng = h.merge_counts(comm)
            tab = {k: h.fetch(k) for k in ("MG_H", "MG_S", "MG_COV")}
            st, n_imp = h.ec_sharded(comm, 0.02, c, 0.35)
            res = {k: h.fetch(k) for k in ("EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER", "MG_EC_COV", "MG_EC_DEL")}
            out[rank] = (ng, tab, st, res)
            L.oatk_comm_destroy(comm)
            h.close()
        except Exception as ex:                          # noqa: BLE001
            errs.append((rank, ex))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    L.oatk_comm_group_destroy(grp)
    assert not any(t.is_alive() for t in th) and not errs, errs
    del parts
    seq, off, lens = rs.slice(0, n)
    hip.scan_host(seq, off, lens, 1001, 31)
    hip.count()
    cnt = hip.fetch_count()
    hip.ec_graph(light_c=c)
    st = hip.ec(0.02, c, 0.35)
    assert [o[0] for o in out] == [cnt["n_scm"]] * world
    assert [o[2][:11].tolist() for o in out] == [st[:11].tolist()] * world
    for key, ref in (("MG_H", cnt["h"]), ("MG_S", cnt["s"]), ("MG_COV", cnt["cov"])):
        assert crc(np.concatenate([o[1][key] for o in out])) == crc(ref), key
    for key, name in (("EC_N_SCM", "EC_N_SCM"), ("EC_KMER", "EC_KMER"), ("EC_MPOS", "EC_MPOS"), ("EC_SMER", "EC_SMER"), ("MG_EC_COV", "EC_SCM_COV"), ("MG_EC_DEL", "EC_SCM_DEL")):
        assert crc(np.concatenate([o[3][key] for o in out])) == crc(hip.fetch(name)), key
    assert int(st[0] + st[5] + st[10]) > 3.5 * n


def test_a_new_batch_is_merged_again(hip):
    """a merged table belongs to the batch it was merged for: after another scan + count on the same handles, oatk_hip_ec_sharded merges again"""
    K, S, c = 301, 21, 6
    first = A.hifi_like(200, 30000, 4000, seed=337, err=0.004)
    second = A.hifi_like(260, 30000, 4000, seed=347, err=0.003)
    world = 2
    L = _lib.load()
    grp = L.oatk_comm_group_create(world)
    out, errs = [None] * world, []

    def work(rank):
        try:
            h = HipSyncasm(0)
            comm = L.oatk_comm_group_rank(grp, rank)
            for reads in (first, second):
                lo, hi = len(reads) * rank // world, len(reads) * (rank + 1) // world
                seq, off, lens = pack_reads(reads[lo:hi])
                h.scan_host(seq, off, lens, K, S, sid0=lo)
                h.count()
                st, _ = h.ec_sharded(comm, 0.02, c, 0.35)            # (no explicit merge_counts)
                out[rank] = (h.multi_range(), st, {k: h.fetch(k) for k in ("MG_H", "EC_KMER", "MG_EC_COV")})
            L.oatk_comm_destroy(comm)
            h.close()
        except Exception as ex:                          # noqa: BLE001
            errs.append((rank, ex))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    L.oatk_comm_group_destroy(grp)
    assert not any(t.is_alive() for t in th) and not errs, errs
    cnt, st, want = single(hip, second, K, S, c)
    assert out[0][0][2] == cnt["n_scm"] and out[0][1][:11].tolist() == st[:11].tolist()
    assert np.array_equal(np.concatenate([o[2]["MG_H"] for o in out]), cnt["h"])
    assert np.array_equal(np.concatenate([o[2]["EC_KMER"] for o in out]), want["EC_KMER"])
    assert np.array_equal(np.concatenate([o[2]["MG_EC_COV"] for o in out]), want["EC_SCM_COV"])


def test_rccl_branch_with_several_ranks_over_a_mock_library(tmp_path):
    """RCCL refuses two ranks on one device, so on this box the RCCL branch of the collectives (grouped ncclSend / ncclRecv with the library's offsets,
    grouped broadcasts, all-gather, all-reduce) only ever runs with a world of one.  tests/c/mock_rccl.cpp implements those entry points for ranks
    that are threads of one process; a child process loads it through OATK_RCCL_LIB and runs 2 - 4 ranks through oatk_comm_create, an empty shard and
    forced hash collisions included, against one handle (tests/mock_rccl_run.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = str(tmp_path / "libmock_rccl.so")
    subprocess.run(["hipcc", "-shared", "-fPIC", "-O2", "-o", lib, os.path.join(here, "c", "mock_rccl.cpp")], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    env = dict(os.environ, OATK_RCCL_LIB=lib)
    p = subprocess.run([sys.executable, os.path.join(here, "mock_rccl_run.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0 and p.stdout.strip().endswith(b"ok 8"), (p.returncode, p.stdout[-300:], p.stderr.decode(errors="replace")[-1500:])


def test_a_rank_that_fails_mid_way_releases_its_peers(tmp_path):
    """include/oatk_hip_multi.h promises that a rank-local failure between two collectives does not leave the peers waiting.  Over the mock library (ranks are
    threads) rank 1 fails on request in the middle of the table merge; its comm_finish aborts the communicator, which releases the others with an error
    (tests/mock_rccl_abort.py).  Against real RCCL a peer's abort is not visible to the others at once: there oatk's wait polls ncclCommGetAsyncError and gives
    up after OATK_COMM_TIMEOUT_S (api_multi.inc: comm_wait) -- that form needs two GPUs (test_real_rccl_*)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = str(tmp_path / "libmock_rccl.so")
    subprocess.run(["hipcc", "-shared", "-fPIC", "-O2", "-o", lib, os.path.join(here, "c", "mock_rccl.cpp")], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    env = dict(os.environ, OATK_RCCL_LIB=lib, OATK_DEBUG_FAIL_RANK="1,5")
    p = subprocess.run([sys.executable, os.path.join(here, "mock_rccl_abort.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().startswith(b"ok"), (p.returncode, p.stdout[-300:], p.stderr.decode(errors="replace")[-1500:])
