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


def _matrix_steps(ts, qs, bw, qls):
    """what wf_ed_core, resumed from one query length to the next, returns -- said by the edit-distance MATRIX alone: the score is the least value on
    the boundary (the query's last row, the target's last column) but not below the call before; the end is the boundary cell of the lowest diagonal within that score; beyond
    the band the call ends with score bw + 1 and no end"""
    tl, T = len(ts), np.frombuffer(ts, np.uint8)
    prev = np.arange(tl + 1, dtype=np.int64)               # the row before the first: D(-1, t) = t + 1, at index t + 1
    idx = np.arange(tl + 1)
    lastcol, out, score, done = [], [], 0, 0
    for ql in qls:
        for q in range(done, ql):
            m = np.minimum(prev[:-1] + (T != qs[q]), prev[1:] + 1)
            prev = np.minimum.accumulate(np.concatenate(([q + 1], m)) - idx) + idx      # ... and from the cell to the left: a min-plus scan
            lastcol.append(int(prev[tl]))
        done = ql
        cells = [(int(prev[t + 1]), ql - 1 - t, t, ql - 1) for t in range(tl)] + [(lastcol[q], q - (tl - 1), tl - 1, q) for q in range(ql)]
        sc = max(min(c[0] for c in cells), score)
        if bw >= 0 and sc > bw:
            score = bw + 1
            out.append((bw + 1, 0, 0))
            continue
        _, t, q = min((c[1], c[2], c[3]) for c in cells if c[0] <= sc)
        score = sc
        out.append((sc, t + 1, q + 1))
    return out


def test_levdist_resumable_traces_are_the_matrix():
    """the reference's own resumed calls (tests/golden/levdist.npz: 120 traces, 515 calls written by the compiled reference) equal the closed form above: the wavefront's
    lowest-diagonal-first end, its skipped diagonals and its one-sided pruning (levdist.c:99-113,156-224) add nothing to the matrix -- which is what lets a search keep the
    matrix's last row instead of a wavefront (tests/trace/ec_trace.c checks the same on 7.6 M arcs of the config-1 surrogate)"""
    g = G.load("levdist")
    n = 0
    for ts, qs, bw, steps in zip(g["tr_t"], g["tr_q"], g["tr_bw"], g["tr_steps"]):
        got = _matrix_steps(ts, qs, int(bw), [int(s[0]) for s in steps])
        for p, s in zip(got, steps):
            assert p == (int(s[1]), int(s[2]), int(s[3])), (ts, qs, bw, s, p)
            n += 1
    assert n > 500


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
