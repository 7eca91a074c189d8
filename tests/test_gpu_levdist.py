"""GPU: the device edit distance ON ITS OWN (oatk_hip_debug_wf_ed over ec_wave.hpp's ecw_step -- the routine every error block is solved with)
against the reference's wf_ed / wf_ed_core (levdist.c:265-345): all of tests/golden/levdist.npz -- 601 pairs incl. the reference's only built-in
known answer (levdist.c:445-446: ED=8 t_EN=59 q_EN=56) and the band cut-offs, 120 resumable traces -- then fresh random jobs against the CPU
oracle (oracle/levdist.c, itself pinned to the same goldens and to the compiled reference), long strings and many-diagonal wavefronts included.
"north_star: edit distances match exactly"."""
import os

import numpy as np
import pytest

import adversarial as A
import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu


def two_bit(q, ts):
    """the device alphabet is sr_t.hoco_s's 2-bit one (the correction never sees anything else: syncmer.c:268-283 stores N as A).  An N of the
    known-answer query is replaced by a base for which the char-level routine -- pinned on the ORIGINAL pair -- still returns the same answer."""
    want = O.wf_ed(ts, q, -1)
    q = bytearray(q)
    for i, ch in enumerate(q):
        if ch in b"Nn":
            for sub in b"ACGT":
                q[i] = sub
                if O.wf_ed(ts, bytes(q), -1) == want:
                    break
            else:
                raise AssertionError("no substitution keeps the answer")
    return bytes(q)


def test_known_answer_of_the_reference(hip):
    g = G.load("levdist")
    ts, qs = g["pairs_t"][0], g["pairs_q"][0]
    assert tuple(int(v) for v in g["pairs_out"][0]) == (8, 59, 56) == O.wf_ed(ts, qs, -1)
    q2 = two_bit(qs, ts)
    assert O.wf_ed(ts, q2, -1) == (8, 59, 56)
    assert hip.wf_ed([(ts, q2, -1, [len(q2)])]) == [[(8, 59, 56)]]


def fits(job, wg):
    """does the job's wavefront fit the registers of the workgroup solver's variant (include/oatk_hip_ec.h: oatk_hip_debug_wf_ed_wg)?"""
    ts, qs, bw, steps = job
    if wg == 32:
        return (2 * bw + 5 if bw >= 0 else len(ts) + len(qs) + 5) <= 512
    if wg == 33:
        return bw >= 0 and 2 * bw + 5 <= 640
    return wg == 0 or (2 * bw + 3 if bw >= 0 else len(ts) + len(qs) + 3) <= (896 if wg == 16 else 256 * wg)


WG = pytest.mark.parametrize("wg", [0, 1, 2, 6, 16])       # 0: ecw_step (a wave per block); 1, 2, 6: ech_step, a workgroup per block with 1, 2, 6 diagonals per lane; 16: ecf_align (sixteen waves, four steps per barrier).  (The tree solver's step and the alignment by matrix rows live in tools/experiments/ since round 6.)


@WG
def test_golden_pairs(hip, wg):
    g = G.load("levdist")
    jobs, want = [], []
    for ts, qs, bw, out in list(zip(g["pairs_t"], g["pairs_q"], g["pairs_bw"], g["pairs_out"]))[1:]:
        jobs.append((ts, qs, int(bw), [len(qs)]))
        want.append([tuple(int(v) for v in out)])
    assert len(jobs) == 600 and sum(1 for j, w in zip(jobs, want) if j[2] >= 0 and w[0][0] > j[2]) > 20      # band cut-offs are among them
    keep = [i for i, j in enumerate(jobs) if fits(j, wg)]
    assert len(keep) > 300
    jobs, want = [jobs[i] for i in keep], [want[i] for i in keep]
    got = hip.wf_ed(jobs, wg)
    for j, (a, b) in enumerate(zip(got, want)):
        assert a == b, (j, jobs[j], a, b)


@WG
def test_golden_resumable_traces(hip, wg):
    g = G.load("levdist")
    jobs, want = [], []
    for ts, qs, bw, steps in zip(g["tr_t"], g["tr_q"], g["tr_bw"], g["tr_steps"]):
        jobs.append((ts, qs, int(bw), [int(s[0]) for s in steps]))
        want.append([(int(s[1]), int(s[2]), int(s[3])) for s in steps])
    assert len(jobs) == 120 and sum(len(w) for w in want) > 300
    keep = [i for i, j in enumerate(jobs) if fits(j, wg)]
    assert len(keep) > 60
    jobs, want = [jobs[i] for i in keep], [want[i] for i in keep]
    got = hip.wf_ed(jobs, wg)
    for j, (a, b) in enumerate(zip(got, want)):
        assert a == b, (j, a, b)


def mutate(rng, ts, n_edits, alpha=b"ACGT"):
    q = bytearray(ts)
    for _ in range(n_edits):
        if not q:
            break
        p, kind = int(rng.integers(0, len(q))), int(rng.integers(0, 3))
        if kind == 0:
            q[p] = alpha[int(rng.integers(0, len(alpha)))]
        elif kind == 1:
            q.insert(p, alpha[int(rng.integers(0, len(alpha)))])
        else:
            del q[p]
    return bytes(q) or b"A"


@WG
def test_fresh_jobs_against_the_oracle(hip, wg):
    rng = np.random.default_rng(265)
    jobs, want = [], []
    for it in range(400):
        alpha = [b"ACGT", b"AC", b"A", b"ACGT"][it % 4]
        tl = int(rng.integers(1, 3000 if it % 10 == 0 else 300))
        ts = A.rand_dna(rng, tl, alpha)
        q = mutate(rng, ts, int(rng.integers(0, 1 + tl // 12)), alpha)
        if it % 3 == 0:
            q = q + A.rand_dna(rng, int(rng.integers(1, 80)), alpha)                       # the query runs past the target
        if it % 7 == 0:
            q = A.rand_dna(rng, int(rng.integers(1, 120)), alpha)                          # unrelated: every diagonal is alive, up to 200 of them
        bw = [-1, 2, 6, 12, 40, 100][it % 6]
        if wg and it % 5 == 0:
            bw = [126, 254, 300, 760, 446][it % 5 if wg == 16 else it % 4]                                              # the widest band each variant takes: 2 bw + 3 <= 256 R
            tl = int(rng.integers(2000, 6000))
            ts = A.rand_dna(rng, tl, alpha)
            q = mutate(rng, ts, int(rng.integers(0, 2 * bw)), alpha) if it % 10 else A.rand_dna(rng, tl, alpha)   # (unrelated: the climb to the band's edge on wavefronts as wide as the band)
        if not fits((ts, q, bw, None), wg):
            continue
        steps, ql = [], 0
        w = O.Wavefront(ts, bw)
        res = []
        while ql < len(q):
            ql = min(len(q), ql + int(rng.integers(1, max(2, len(q) // 2))))
            r = w.step(q[:ql])
            steps.append(ql), res.append(r)
            if bw >= 0 and r[0] > bw:
                break
        w.close()
        jobs.append((ts, q, bw, steps)), want.append(res)
    assert len(jobs) > 150
    got = hip.wf_ed(jobs, wg)
    n_wide = 0
    for j, (a, b) in enumerate(zip(got, want)):
        assert a == b, (j, jobs[j][2], jobs[j][3], a, b)
        n_wide += b[-1][0] > 32
    assert n_wide > 10                                                                      # wavefronts wider than one lane group per diagonal


EXPERIMENTS_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "experiments", "liboatk_hip_experiments.so")


@pytest.mark.skipif(not os.path.exists(EXPERIMENTS_LIB), reason="tools/experiments/build.sh not run: Myers' kernel is not part of liboatk_hip.so (round 4)")
def test_myers_bit_vector_variant_gives_the_same_distances():
    """(runs in a process of its own over tools/experiments/liboatk_hip_experiments.so: the product library no longer carries the experiment)"""
    import subprocess
    import sys
    env = dict(os.environ, OATK_HIP_LIB=EXPERIMENTS_LIB, OATK_TEST_MYERS_CHILD="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__ + "::test_myers_child"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]


@pytest.mark.skipif(not os.environ.get("OATK_TEST_MYERS_CHILD"), reason="the child of test_myers_bit_vector_variant_gives_the_same_distances")
def test_myers_child(hip):
    """north_star names a bit-parallel (Myers) kernel, SURVEY 7-5 asks to benchmark both: the bit-vector algorithm with one lane per pair
    (ec_wave.hpp: myers_ed_kernel) returns wf_ed's (score, t_end, q_end) on every golden pair and on fresh ones, band cut-offs included -- the
    closed form the wavefront equals (minimum over last row and last column, smallest diagonal first).  tools/edbench.py times the two."""
    g = G.load("levdist")
    pairs, want = [], []
    for ts, qs, bw, out in list(zip(g["pairs_t"], g["pairs_q"], g["pairs_bw"], g["pairs_out"]))[1:]:
        pairs.append((ts, qs, int(bw))), want.append(tuple(int(v) for v in out))
    rng = np.random.default_rng(77)
    for it in range(300):
        tl = int(rng.integers(1, 2500 if it % 10 == 0 else 260))
        ts = A.rand_dna(rng, tl)
        q = mutate(rng, ts, int(rng.integers(0, 1 + tl // 15)))
        if it % 3 == 0:
            q = q + A.rand_dna(rng, int(rng.integers(1, 80)))
        bw = [-1, 3, 8, 40][it % 4]
        w = O.Wavefront(ts, bw)
        want.append(w.step(q))
        w.close()
        pairs.append((ts, q, bw))
    for myers in (False, True):
        got, ms = hip.ed_ab(pairs, myers)
        for j, (a, b) in enumerate(zip(got, want)):
            assert a == b, (myers, j, pairs[j][2], len(pairs[j][0]), len(pairs[j][1]), a, b)
        assert ms > 0


def test_bad_arguments_are_refused(hip):
    from oatk_amd import OatkHipError
    with pytest.raises(OatkHipError):
        hip.wf_ed([(b"ACGT", b"ACGT", 2, [3, 2])])                                          # query lengths must ascend
    with pytest.raises(ValueError):
        hip.wf_ed([(b"ACGN", b"ACGT", 2, [4])])


@pytest.mark.parametrize("seg", [None, (2, 6), (4, 40), (8, 25), (16, 120)])
def test_tables_of_a_long_arc(hip, seg, monkeypatch):
    """the two tables by which a long arc can be known to die without a step (oatk_hip_debug_tables, ec_tables.hpp: ecb_table) against their plain recurrences
    (tests/c/prof_bitpar_test.c holds the same on the CPU): table 0 [u] = least cost of the whole string inside target[u ..], table 1 [u] = of a prefix against target[u ..] to its end"""
    rng = np.random.default_rng(77)

    def plain(ts, ex):
        tl, m = len(ts), len(ex)
        T, big = np.frombuffer(ts, np.uint8), 1 << 20
        f = np.full(tl + 1, 0, np.int64)                       # F[m][u] = 0
        g = (tl - np.arange(tl + 1)).astype(np.int64)          # G[m][u] = tl - u
        for i in range(m - 1, -1, -1):
            neq = (T != ex[i]).astype(np.int64)
            for tab, last in ((f, m - i), (g, 0)):
                prev = tab.copy()
                cand = np.minimum(np.concatenate((prev[1:] + neq, [big])), prev + 1)
                cand[tl] = last
                # tab[u] = min(cand[u], tab[u + 1] + 1): a min-plus scan from the right
                idx = np.arange(tl + 1)
                tab[:] = (np.minimum.accumulate((cand + idx)[::-1])[::-1]) - idx
        return f.astype(np.int32), g.astype(np.int32)

    jobs = []
    for it in range(40):
        tl = int(rng.integers(1, 700))
        ts = A.rand_dna(rng, tl, [b"ACGT", b"AC"][it % 2])
        m = int(rng.integers(1, 1001 if it % 3 else 60))
        if it % 2 and m < tl:
            at = int(rng.integers(0, tl - m + 1))
            ex = mutate(rng, ts[at:at + m], int(rng.integers(0, 1 + m // 10)))[:1024]
        else:
            ex = A.rand_dna(rng, m, b"ACGT")
        jobs.append((ts, ex))
    # seg = (waves, cut): as the solver's second stage builds them (round 6, ecb_tables_wg) -- table 0 in stretches by the waves, exact where it is <= cut and beyond cut
    # elsewhere; table 1 only where it can be <= cut (near the target's end), not written (-1) below
    if seg:
        monkeypatch.setenv("OATK_DEBUG_TABLES_SEG", "%d:%d" % seg)
        jobs += [(A.rand_dna(rng, 5000 + 1700 * i, b"ACGT"), A.rand_dna(rng, 970 - 200 * i, b"ACGT")) for i in range(3)]       # stretches with a run-up shorter than the target
        tgt = A.rand_dna(rng, 6000, b"ACGT")
        jobs += [(tgt, mutate(rng, tgt[at:at + 900], 12)[:1024]) for at in (0, 2500, 5100)]                                  # ... and a string that does fit somewhere
    got = hip.tables(jobs)
    for (ts, ex), (t0, t1) in zip(jobs, got):
        w0, w1 = plain(ts, ex)
        if not seg:
            assert np.array_equal(t0, w0), (len(ts), len(ex))
            assert np.array_equal(t1, w1), (len(ts), len(ex))
            continue
        cut = seg[1]
        assert np.array_equal(np.minimum(t0, cut + 1), np.minimum(w0, cut + 1)), (len(ts), len(ex), seg)
        lo = max(0, len(ts) - (len(ex) + cut + 1))
        assert np.array_equal(t1[lo:], w1[lo:]) and np.all(t1[:lo] == -1) and np.all(w1[:lo] > cut), (len(ts), len(ex), seg)
