"""CPU: the oracle (oracle/*.c) against the golden vectors produced by the compiled reference."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O


@pytest.mark.parametrize("case", G.SCAN_CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_scan_matches_reference(case, mode):
    g = G.load(case)
    K, S = int(g["K"]), int(g["S"])
    got = O.scan(G.reads_of(g), K, S, mode=mode)
    for f in G.SCAN_FIELDS:
        assert got[f].shape == g[f].shape and np.array_equal(got[f], g[f]), (case, mode, f)
    assert np.array_equal(got["n_nn"], g["n_nn"])


@pytest.mark.parametrize("case", G.SCAN_CASES)
def test_count_matches_reference(case):
    g = G.load(case)
    K, S = int(g["K"]), int(g["S"])
    _, c = O.scan_and_count(G.reads_of(g), K, S, mode=1)
    assert c["err"] == 0
    assert c["n_scm"] == len(g["scm_h"])
    assert np.array_equal(c["h"], g["scm_h"])
    assert np.array_equal(c["s"], g["scm_s"])
    assert np.array_equal(c["cov"], g["scm_cov"])
    assert np.array_equal(c["occ"], g["scm_occ"])
    assert np.array_equal(c["k_id"], g["k_id"])          # sr->k_mer rewritten to id << 1 (syncmer.c:1378)


def test_levdist_known_answer():
    """the reference's built-in LEVDIST_TEST_NAIVE pair (levdist.c:445-446): ED=8 t_EN=59 q_EN=56"""
    g = G.load("levdist")
    assert O.wf_ed(g["pairs_t"][0], g["pairs_q"][0], -1) == (8, 59, 56)
    assert O.ed_bruteforce(g["pairs_t"][0], g["pairs_q"][0]) == (8, 59, 56)


def test_levdist_pairs():
    g = G.load("levdist")
    for ts, qs, bw, want in zip(g["pairs_t"], g["pairs_q"], g["pairs_bw"], g["pairs_out"]):
        got = O.wf_ed(ts, qs, int(bw))
        assert got == tuple(int(v) for v in want), (ts, qs, bw)
        if int(bw) < 0 or want[0] <= bw:
            # closed form: min over last row/column of the DP, ties -> smallest diagonal
            assert O.ed_bruteforce(ts, qs) == tuple(int(v) for v in want)


def test_levdist_resumable_traces():
    g = G.load("levdist")
    for ts, qs, bw, steps in zip(g["tr_t"], g["tr_q"], g["tr_bw"], g["tr_steps"]):
        w = O.Wavefront(ts, int(bw))
        for ql, sc, te, qe in steps:
            assert w.step(qs[:int(ql)]) == (int(sc), int(te), int(qe))
        w.close()


def test_hash64_is_a_bijection_sample():
    mask = (1 << 62) - 1
    L = O.lib()
    xs = np.random.default_rng(1).integers(0, 1 << 62, size=2000, dtype=np.uint64)
    hs = {int(L.orc_hash64(int(x), mask)) for x in xs}
    assert len(hs) == len(set(int(x) for x in xs))
    assert all(h <= mask for h in hs)


def test_empty_and_short_inputs():
    out = O.scan([b"", b"A", b"ACGT" * 10, b"N" * 50], 1001, 31, mode=0)
    assert out["hoco_l"].tolist() == [0, 1, 40, 50]
    assert out["n_scm"].sum() == 0
    _, c = O.scan_and_count([b"", b"ACGT"], 1001, 31)
    assert c["n_scm"] == 0
