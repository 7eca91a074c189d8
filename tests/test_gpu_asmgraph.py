"""GPU parity of the assembly graph built on the device (oatk_hip_asm_graph, include/oatk_hip_graph.h) against the COMPILED
REFERENCE's make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) (syncasm.c:203-299, run_syncasm.c:138) on the same
databases -- before and after the error correction -- against the committed golden vectors, and against the oracle at a
size the reference's hash-table arc counter is slow at.  Bit-exact: vertices, arcs, flags, link ids, index, deletion marks."""
import numpy as np
import pytest

import adversarial as A
import asm_util as AU
import ec_util as E
import golden_util as G
import ref_lib as R
import test_gpu_ec as T
from oatk_amd import pack_reads
from test_gpu_dropin import device_dbs

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")


def scan_count(hip, reads, K, S):
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    hip.count()


GOLDEN = {
    "ec_diploid_k101": (101, 11, lambda: T.diploid_reads(101, 6000, 150, 500, 1200, 0.006)),
    "ec_repeats_k301": (301, 21, lambda: T.sample_reads(T.genome_with_repeats(7, 25000, unit=1500, copies=4), 320, 4000, 0.003, 8)),
    "ec_hifi_k1001": (1001, 31, lambda: A.hifi_like(120, 30000, 9000, seed=1009, err=0.0008)),
}


@pytest.mark.parametrize("stage", ["raw", "ec"])
@pytest.mark.parametrize("case", sorted(GOLDEN))
def test_device_asmgraph_matches_golden(hip, case, stage):
    K, S, mk = GOLDEN[case]
    g = G.load("asmgraph_" + case)
    c, a = int(g["c"]), float(g["a"])
    scan_count(hip, mk(), K, S)
    if stage == "ec":
        hip.ec_graph()
        hip.ec(0.02, c, a)
        assert np.array_equal(hip.fetch("EC_KMER"), G.load(case)["out_k_mer"])
    nv, na = hip.asm_graph(c, a)
    D = hip.fetch_asm_graph()
    assert nv == len(D["vtx_scm"]) > 0 and na == len(D["arc_v"]) > 0
    AU.assert_asm_equal(D, g, stage + "_")
    nv2, na2 = hip.asm_graph(c, a)                          # repeatable on the same resident state
    assert (nv2, na2) == (nv, na)
    AU.assert_asm_equal(hip.fetch_asm_graph(), g, stage + "_")


@needs_ref
@pytest.mark.parametrize("stage", ["raw", "ec"])
@pytest.mark.parametrize("case,a", [(0, 0.35), (1, 0.0), (2, 0.35), (3, 0.9), (4, 0.35), (5, 0.2), (6, 0.5)])
def test_device_asmgraph_matches_reference(hip, case, a, stage):
    K, S, c, mk = T.CASES[case]
    db, scm = device_dbs(hip, mk(), K, S)                   # reference-layout structs built from the device scan + count
    L = R.lib()
    if stage == "ec":
        g = L.refx_make_graph(db, scm, 0, 0.0)              # run_syncasm.c:109-124 on the reference side ...
        L.refx_consensus(db, g, 1, 1)
        E.reference_ec(T._H(db), T._H(scm), g, 0.02, c, a, threads=3)
        L.refx_scg_destroy(g)
        hip.ec_graph()                                      # ... and on the device
        hip.ec(0.02, c, a)
    want = AU.reference_asmgraph(db, scm, c, a)             # run_syncasm.c:138
    rscm = object.__new__(R.ScmDb)
    rscm._h = scm
    want["scm_del"] = rscm.flatten()["del"]
    nv, na = hip.asm_graph(c, a)
    D = hip.fetch_asm_graph()
    assert nv == len(want["vtx_scm"]) and na == len(want["arc_v"]) and nv > 0
    AU.assert_asm_equal(D, want)
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)


def test_device_asmgraph_edges(hip):
    K, S = 101, 11
    reads = A.hifi_like(150, 6000, 1500, seed=77, err=0.003)
    scan_count(hip, reads, K, S)
    n_scm = hip.info()["n_scm"]
    # nothing survives the coverage filter: no vertices, no arcs, every syncmer marked (syncasm.c:228)
    assert hip.asm_graph(1 << 20, 0.35) == (0, 0)
    D = hip.fetch_asm_graph()
    assert len(D["scm_del"]) == n_scm and D["scm_del"].all() and len(D["idx_n"]) == 0 and len(D["arc_link"]) == 0
    # no filter at all: one vertex per syncmer, the arcs of the EC graph
    nv, na = hip.asm_graph(0, 0.0)
    assert nv == n_scm
    hip.ec_graph()
    assert na == hip.buffer("EG_ARC_V")[1] // 8
    D = hip.fetch_asm_graph()
    assert np.array_equal(D["arc_v"], hip.fetch("EG_ARC_V")) and np.array_equal(D["arc_w"], hip.fetch("EG_ARC_W"))
    assert np.array_equal(D["arc_cov"], hip.fetch("EG_ARC_COV")) and np.array_equal(D["arc_comp"], hip.fetch("EG_ARC_COMP"))
    # link ids: shared by an arc and its complement, dense, opened in arc order
    link = D["arc_link"]
    pos = {(int(v), int(w)): i for i, (v, w) in enumerate(zip(D["arc_v"], D["arc_w"]))}
    comp = np.array([pos[(int(w) ^ 1, int(v) ^ 1)] for v, w in zip(D["arc_v"], D["arc_w"])])
    assert np.array_equal(link, link[comp]) and link.max() + 1 == len(np.unique(link))
    first = np.minimum(np.arange(na), comp)
    assert np.array_equal(link, np.unique(first, return_inverse=True)[1])
    # reads without syncmers, and no reads
    scan_count(hip, [b"ACGT" * 10, b"A" * 300], K, S)
    assert hip.asm_graph(0, 0.0) == (0, 0)
    with pytest.raises(RuntimeError):
        hip.asm_graph(0, -1.0)


def test_device_asmgraph_matches_oracle_at_scale(hip):
    """20 k reads: device graph after the error correction against oracle/asmgraph.c fed with the device's own corrected chains"""
    from oatk_amd.synth import ReadSet
    K, S, c, a, n = 1001, 31, 30, 0.35, 20000
    scan_count(hip, ReadSet(1_000_000, n, 15000).as_list(0, n), K, S)
    hip.ec_graph()
    hip.ec(0.02, c, a)
    nv, na = hip.asm_graph(c, a)
    D = hip.fetch_asm_graph()
    og = AU.oracle_asmgraph(hip.fetch("EC_N_SCM"), hip.fetch("EC_KMER"), hip.fetch("EC_MPOS"), hip.fetch("EC_SCM_COV"), hip.fetch("EC_SCM_DEL"), c, a)
    assert not og["multi_arc"] and nv > 300 and na > nv
    AU.assert_asm_equal(D, og)


@pytest.mark.parametrize("seed", range(5))
def test_device_asmgraph_on_injected_pairs_matches_oracle(hip, seed):
    """corners that reads from a genome rarely produce: arcs that are their own complement (v -> v^1), pairs on both strands, random coverage
    and deletion marks, filters from 0 to 2 -- two-syncmer "reads" for the oracle, their canonical keys through oatk_hip_asm_graph_from_pairs
    for the device; and the duplicate-arc corner (a -> a on both strands) refused with OATK_E_SPLIT exactly when the oracle flags it"""
    import ctypes as C
    from oatk_amd import OatkHipError
    scan_count(hip, A.hifi_like(20, 3000, 1000, seed=1), 101, 11)      # the result buffers are handed out on a context that holds a batch
    rng = np.random.default_rng(100 + seed)
    ns = 300
    n_pairs = 6000
    a, b = rng.integers(0, ns, n_pairs), rng.integers(0, ns, n_pairs)
    near = rng.random(n_pairs) < 0.7                          # most pairs between neighbours, so that keys repeat and coverages add up
    b[near] = (a[near] + rng.integers(1, 4, near.sum())) % ns
    sa, sb = rng.integers(0, 2, n_pairs), rng.integers(0, 2, n_pairs)
    selfc = rng.random(n_pairs) < 0.03                        # a+ -> a-: its own complement
    b[selfc], sb[selfc] = a[selfc], 1 - sa[selfc]
    if seed == 4:                                             # a+ -> a+ and a- -> a-: duplicate (v, w) after the complements are added
        a[:4], b[:4], sa[:4], sb[:4] = [7, 7, 9, 9], [7, 7, 9, 9], [0, 1, 0, 1], [0, 1, 0, 1]
    same = (a == b) & (sa == sb) & (np.arange(n_pairs) >= (4 if seed == 4 else 0))
    b[same] = (b[same] + 1) % ns
    k_mer = np.stack([a, b], 1).reshape(-1).astype(np.uint64) << np.uint64(1)
    m_pos = (np.stack([sa, sb], 1).reshape(-1) | (np.arange(2 * n_pairs) << 1)).astype(np.uint32)
    n_scm = np.full(n_pairs, 2, np.uint32)
    cov = rng.integers(0, 60, ns).astype(np.uint32)
    dele = (rng.random(ns) < 0.05).astype(np.uint8)
    v0, v1 = (a << 1 | sa).astype(np.uint64), (b << 1 | sb).astype(np.uint64)
    keys = np.where(v0 <= v1, v0 << np.uint64(32) | v1, (v1 ^ np.uint64(1)) << np.uint64(32) | (v0 ^ np.uint64(1)))
    keys = np.concatenate([keys, np.full(50, 0xFFFFFFFFFFFFFFFF, np.uint64)])       # fillers, as slot 0 of every read produces them
    rng.shuffle(keys)
    rt = C.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    dev = []
    for arr in (keys, cov, dele):
        p = C.c_void_p()
        assert rt.hipMalloc(C.byref(p), arr.nbytes) == 0 and rt.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        dev.append(p)
    try:
        for c, f in ((0, 0.0), (5, 0.35), (20, 0.9), (1, 2.0), (59, 0.0)):
            og = AU.oracle_asmgraph(n_scm, k_mer, m_pos, cov, dele, c, f)
            if og["multi_arc"]:
                with pytest.raises(OatkHipError, match="duplicate arcs"):
                    hip.asm_graph_from_pairs(dev[0].value, len(keys), ns, dev[1].value, dev[2].value, c, f)
                continue
            nv, na = hip.asm_graph_from_pairs(dev[0].value, len(keys), ns, dev[1].value, dev[2].value, c, f)
            AU.assert_asm_equal(hip.fetch_asm_graph(), og)
            assert nv == len(og["vtx_scm"]) and na == len(og["arc_v"])
        if seed == 4:
            assert AU.oracle_asmgraph(n_scm, k_mer, m_pos, cov, dele, 0, 0.0)["multi_arc"] or cov[7] == 0 or cov[9] == 0 or dele[7] or dele[9]
    finally:
        for p in dev:
            rt.hipFree(p)
