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
