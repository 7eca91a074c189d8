"""GPU: the EC round with reads SHARDED by record (oatk_amd/multi.py: ShardedEc) equals the EC round of one context holding all
reads -- chains (in global ids), refreshed coverage / deletion flags, block statistics.  Two processes share the one GPU of the
test box and talk over gloo (RCCL refuses two ranks on one device); on a node the same code runs one rank per GPU over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import adversarial as A
import test_gpu_ec as E
from oatk_amd import pack_reads

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, reads, bounds, K, S, c, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oatk_amd import HipSyncasm
    from oatk_amd.multi import ShardedEc
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    hip = HipSyncasm(0)
    lo, hi = bounds[rank], bounds[rank + 1]
    seq, off, lens = pack_reads(reads[lo:hi])
    hip.scan_host(seq, off, lens, K, S, sid0=lo)
    hip.count()
    sh = ShardedEc(hip, dist, dev)
    res = sh.run(0.02, c, 0.35)
    cons = sh.consensus(c)
    nv, na = sh.asm_graph(c, 0.35)
    ag = hip.fetch_asm_graph()
    assert nv == len(ag["vtx_scm"]) and na == len(ag["arc_v"])
    np.savez(os.path.join(outdir, "ag%d.npz" % rank), **ag)
    stt = sh.stat_raw()
    np.savez(os.path.join(outdir, "st%d.npz" % rank), **{k: np.asarray(v) for k, v in stt.items()})
    sh.overlap_hist()
    np.savez(os.path.join(outdir, "ov%d.npz" % rank), **{k: hip.fetch("OVL_" + k) for k in OVL_NAMES})
    sh.read_alignment(vertex_graph(ag, nv, na))
    np.savez(os.path.join(outdir, "ra%d.npz" % rank), **{k: hip.fetch("RA_" + k) for k in RA_NAMES})
    np.savez(os.path.join(outdir, "r%d.npz" % rank), n_scm=hip.fetch("EC_N_SCM"), k_mer=hip.fetch("EC_KMER"), m_pos=hip.fetch("EC_MPOS"),
             s_mer=hip.fetch("EC_SMER"), occ=hip.fetch("EC_SCM_OCC"), occ_off=hip.fetch("EC_SCM_OCC_OFF"), cov=res["cov"].cpu().numpy(),
             dele=res["del"].cpu().numpy(), stats=res["stats"], imported=sh.n_imported,
             c_ids=cons["ids"].cpu().numpy(), c_rl=cons["rl"].cpu().numpy(), c_m=cons["m_seq"].cpu().numpy(), c_owner=cons["owner"].cpu().numpy(),
             c_first=cons["first_local"].cpu().numpy())
    hip.close()
    dist.barrier()
    dist.destroy_process_group()


OVL_NAMES = ["KEY", "OFF", "DIST", "CNT", "TAIL"]
RA_NAMES = ["ALN_SID", "ALN_OFF", "ALN_S", "FRG_UID", "FRG_UBEG", "FRG_UEND", "FRG_SBEG", "FRG_SEND"]


def vertex_graph(ag, nv, na):
    """the assembly graph (one syncmer per vertex) as scg_read_alignment reads it"""
    ns = len(ag["scm_del"])
    su_off = np.zeros(ns + 1, np.uint64)
    su_off[1:] = np.cumsum(ag["scm_del"] == 0)
    return {"n_scm": ns, "su_off": su_off, "su_uid": np.arange(nv, dtype=np.uint64) << np.uint64(1), "su_pos": np.zeros(nv, np.uint32),
            "utg_n": np.ones(nv, np.uint32), "idx_p": ag["idx_p"], "idx_n": ag["idx_n"].astype(np.uint64), "arc_w": ag["arc_w"],
            "arc_ln": np.zeros(na, np.uint64), "arc_del": np.zeros(na, np.uint8)}


CASES = [
    # K, S, c, reads, shard boundaries as fractions
    (301, 21, 6, lambda: A.hifi_like(300, 30000, 4000, seed=307, err=0.004), (0.0, 0.5, 1.0)),
    (1001, 31, 6, lambda: E.sample_reads(E.genome_with_repeats(5, 50000), 260, 9000, 0.001, 6), (0.0, 0.35, 1.0)),
    # a shard of a handful of reads sees few of the good syncmers: their k-mers must be imported
    (101, 11, 5, lambda: E.sample_reads(E.genome_with_repeats(9, 9000, unit=600, copies=3), 300, 1500, 0.004, 10), (0.0, 0.02, 1.0)),
    # three shards, one of them empty
    (101, 11, 5, lambda: E.sample_reads(E.genome_with_repeats(9, 9000, unit=600, copies=3), 300, 1500, 0.004, 10), (0.0, 0.4, 0.4, 1.0)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_sharded_ec_equals_single_context(hip, tmp_path, case):
    K, S, c, mk, frac = CASES[case]
    reads = mk()
    bounds = [int(round(f * len(reads))) for f in frac]
    world = len(bounds) - 1
    mp.spawn(_worker, args=(world, _free_port(), reads, bounds, K, S, c, str(tmp_path)), nprocs=world, join=True)
    # one context, all reads
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    hip.ec_graph()
    st = hip.ec(0.02, c, 0.35)
    want = {k: hip.fetch(k) for k in ["EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER", "EC_SCM_COV", "EC_SCM_DEL", "EC_SCM_OCC", "EC_SCM_OCC_OFF"]}
    z = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate([x["n_scm"] for x in z]), want["EC_N_SCM"])
    for name, key in (("k_mer", "EC_KMER"), ("m_pos", "EC_MPOS"), ("s_mer", "EC_SMER")):
        assert np.array_equal(np.concatenate([x[name] for x in z]), want[key]), name
    for x in z:
        assert np.array_equal(x["cov"], want["EC_SCM_COV"].astype(np.int64))
        assert np.array_equal(x["dele"], want["EC_SCM_DEL"])
        assert np.array_equal(x["stats"][:11], st[:11].astype(np.int64))
    # occurrence lists: a syncmer's global list is the concatenation of the shards' lists in shard order
    n = len(want["EC_SCM_COV"])
    parts = [[x["occ"][int(x["occ_off"][i]):int(x["occ_off"][i + 1])] for x in z] for i in range(0, n, max(1, n // 400))]
    got = np.concatenate([np.concatenate(p) for p in parts])
    ref = np.concatenate([want["EC_SCM_OCC"][int(want["EC_SCM_OCC_OFF"][i]):int(want["EC_SCM_OCC_OFF"][i + 1])] for i in range(0, n, max(1, n // 400))])
    assert np.array_equal(got, ref)
    if case == 2:
        assert sum(int(x["imported"]) for x in z) > 0
    # base-space consensus: totals summed over the shards round to the single-context values; the first uncorrected occurrence
    # is the first one of the lowest rank that has any
    hip.consensus(c)
    sel, rl, ms, fo = hip.fetch("CONS_SEL"), hip.fetch("CONS_RL").reshape(-1, K), hip.fetch("CONS_MSEQ"), hip.fetch("CONS_FIRST")
    for x in z:
        assert np.array_equal(x["c_ids"].astype(np.uint32), sel) and np.array_equal(x["c_rl"], rl.astype(np.int64))
        assert np.array_equal(x["c_m"], ms.astype(np.int64))
    owner = z[0]["c_owner"]
    first = np.array([z[int(o)]["c_first"][i] if o >= 0 else -1 for i, o in enumerate(owner)], dtype=np.int64)
    assert np.array_equal(first.view(np.uint64), fo) and len(sel) > 0
    # the assembly graph of all reads (run_syncasm.c:138): every rank holds the graph the single context builds
    nv, na = hip.asm_graph(c, 0.35)
    want_g = hip.fetch_asm_graph()
    assert nv > 0 and na > 0
    for r in range(world):
        got_g = np.load(os.path.join(str(tmp_path), "ag%d.npz" % r))
        for k, v in want_g.items():
            if k == "idx_p":
                has = want_g["idx_n"] > 0
                assert np.array_equal(got_g[k][has], v[has]), k
            else:
                assert np.array_equal(got_g[k], v), (r, k)
    # sr_db_stat's tabulation over all reads (run_syncasm.c:131)
    want_st = hip.stat_raw()
    for r in range(world):
        zs = np.load(os.path.join(str(tmp_path), "st%d.npz" % r))
        for k, v in want_st.items():
            assert np.array_equal(zs[k], np.asarray(v)), (r, k)
    assert want_st["kmer_unique"] > 10 and want_st["n_dist"] > 0
    # pair-distance tables of all reads: every rank holds the tables the single context builds
    hip.overlap_hist()
    for r in range(world):
        zo = np.load(os.path.join(str(tmp_path), "ov%d.npz" % r))
        for k in OVL_NAMES:
            assert np.array_equal(zo[k], hip.fetch("OVL_" + k)), (r, k)
    assert len(hip.fetch("OVL_KEY")) > 10
    # read alignment: every shard aligns its own reads against the same graph; the results concatenate
    hip.read_alignment(vertex_graph(want_g, nv, na))
    want_ra = {k: hip.fetch("RA_" + k) for k in RA_NAMES}
    zr = [np.load(os.path.join(str(tmp_path), "ra%d.npz" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate([x["ALN_SID"] + np.uint32(bounds[r]) for r, x in enumerate(zr)]), want_ra["ALN_SID"])
    assert np.array_equal(np.concatenate([x["ALN_S"] for x in zr]), want_ra["ALN_S"]) and len(want_ra["ALN_SID"]) > 0
    for k in RA_NAMES[3:]:
        assert np.array_equal(np.concatenate([x[k] for x in zr]), want_ra[k]), k
    assert np.array_equal(np.concatenate([np.diff(x["ALN_OFF"]) for x in zr]), np.diff(want_ra["ALN_OFF"]))
