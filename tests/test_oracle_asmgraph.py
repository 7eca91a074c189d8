"""CPU: the oracle's assembly graph (oracle/asmgraph.c) against golden vectors produced by the compiled reference's
make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) (syncasm.c:203-299, run_syncasm.c:138), and side by side with it."""
import numpy as np
import pytest

import asm_util as AU
import golden_util as G
import ref_lib as R

EC_CASES = ["ec_diploid_k101", "ec_repeats_k301", "ec_hifi_k1001"]


@pytest.mark.parametrize("stage", ["raw", "ec"])
@pytest.mark.parametrize("case", EC_CASES)
def test_oracle_asmgraph_matches_reference_golden(case, stage):
    e, g = G.load(case), G.load("asmgraph_" + case)
    pre = "in_" if stage == "raw" else "out_"
    og = AU.oracle_asmgraph(e[pre + "n_scm"], e[pre + "k_mer"], e[pre + "m_pos"], e[pre + "scm_cov"], e[pre + "scm_del"], int(g["c"]), float(g["a"]))
    assert not og["multi_arc"] and len(og["arc_v"]) > 0 and 0 < len(og["vtx_scm"]) < len(e["in_scm_s"])
    AU.assert_asm_equal(og, g, stage + "_")
    # the two filters do something: fewer arcs than the unfiltered graph over the same vertices
    og0 = AU.oracle_asmgraph(e[pre + "n_scm"], e[pre + "k_mer"], e[pre + "m_pos"], e[pre + "scm_cov"], e[pre + "scm_del"], int(g["c"]), 0.0)
    assert len(og0["arc_v"]) >= len(og["arc_v"])


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("K,S,err,seed,c,a", [(101, 11, 0.01, 5, 3, 0.35), (301, 21, 0.002, 6, 0, 0.0), (301, 21, 0.004, 8, 2, 0.9),
                                               (1001, 31, 0.0008, 7, 5, 0.35), (101, 11, 0.01, 9, 1000000, 0.35)])
def test_oracle_asmgraph_side_by_side(K, S, err, seed, c, a):
    import adversarial as A
    reads = A.hifi_like(220, 12 * K, 5 * K, seed=seed, err=err)
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    scm = R.ScmDb(db)
    sr0, sc0 = db.flatten(), scm.flatten()
    og = AU.oracle_asmgraph(sr0["n_scm"], sr0["k_mer"], sr0["m_pos"], sc0["cov"], sc0["del"], c, a)
    want = AU.reference_asmgraph(db.handle, scm.handle, c, a)
    want["scm_del"] = scm.flatten()["del"]
    AU.assert_asm_equal(og, want)
    if c == 1000000:
        assert len(og["vtx_scm"]) == 0 and len(og["arc_v"]) == 0 and og["scm_del"].all()
    scm.close()
    db.close()


def synthetic_pairs(rng, ns, n_pairs, multi):
    """two-syncmer "reads": pairs between near and far syncmers on either strand, some their own complement (a+ -> a-), optionally the
    a -> a corner on both strands that makes the reference emit duplicate arcs"""
    a, b = rng.integers(0, ns, n_pairs), rng.integers(0, ns, n_pairs)
    near = rng.random(n_pairs) < 0.7
    b[near] = (a[near] + rng.integers(1, 4, near.sum())) % ns
    sa, sb = rng.integers(0, 2, n_pairs), rng.integers(0, 2, n_pairs)
    selfc = rng.random(n_pairs) < 0.03
    b[selfc], sb[selfc] = a[selfc], 1 - sa[selfc]
    if multi:
        a[:4], b[:4], sa[:4], sb[:4] = [7, 7, 9, 9], [7, 7, 9, 9], [0, 1, 0, 1], [0, 1, 0, 1]
    same = (a == b) & (sa == sb) & (np.arange(n_pairs) >= (4 if multi else 0))
    b[same] = (b[same] + 1) % ns
    k_mer = np.stack([a, b], 1).reshape(-1).astype(np.uint64) << np.uint64(1)
    m_pos = (np.stack([sa, sb], 1).reshape(-1) | (np.arange(2 * n_pairs) << 1)).astype(np.uint32)
    return np.full(n_pairs, 2, np.uint32), k_mer, m_pos


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(5))
def test_oracle_asmgraph_on_synthetic_chains_side_by_side(seed):
    """chains no genome would produce, in hand-made databases in front of the compiled reference's make_syncmer_graph: self-complementary arcs
    and their flag (asmg_arc_fix_symm), random coverage and deletion marks, filters 0 .. 2"""
    import ctypes as C
    import ec_util as E
    L = R.lib()
    L.refx_fake_srdb.restype = C.c_void_p
    L.refx_fake_srdb.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.refx_fake_scmdb.restype = C.c_void_p
    L.refx_fake_scmdb.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p]
    L.refx_fake_scmdb_del.argtypes = [C.c_void_p, C.c_void_p]
    L.refx_fake_dbs_free.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(100 + seed)
    ns = 300
    n_scm, k_mer, m_pos = synthetic_pairs(rng, ns, 6000, False)
    cov = rng.integers(0, 60, ns).astype(np.uint32)
    dele = (rng.random(ns) < 0.05).astype(np.uint8)
    self_arcs = 0
    for c, f in ((0, 0.0), (5, 0.35), (20, 0.9), (1, 2.0), (59, 0.0)):
        db = L.refx_fake_srdb(len(n_scm), n_scm.ctypes.data, k_mer.ctypes.data, m_pos.ctypes.data)
        scm = L.refx_fake_scmdb(ns, cov.ctypes.data, dele.ctypes.data)
        want = AU.reference_asmgraph(db, scm, c, f)
        d = np.zeros(ns, np.uint8)
        L.refx_fake_scmdb_del(scm, d.ctypes.data)
        want["scm_del"] = d
        og = AU.oracle_asmgraph(n_scm, k_mer, m_pos, cov, dele, c, f)
        assert not og["multi_arc"]
        AU.assert_asm_equal(og, want)
        self_arcs += int(((og["arc_w"] ^ np.uint64(1)) == og["arc_v"]).sum())
        L.refx_fake_dbs_free(db, scm)
    assert self_arcs > 0
    # the duplicate-arc corner is flagged (the reference's order of the two copies is an accident of its hash table)
    n2, k2, m2 = synthetic_pairs(np.random.default_rng(seed), ns, 200, True)
    assert AU.oracle_asmgraph(n2, k2, m2, np.full(ns, 10, np.uint32), np.zeros(ns, np.uint8), 0, 0.0)["multi_arc"]
