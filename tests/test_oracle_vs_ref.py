"""CPU, authoring container only: oracle side by side with the compiled reference (oracle/_ref) on fresh inputs.
Skipped where the compiled reference is absent."""
import numpy as np
import pytest

import adversarial as A
import oracle_lib as O
import ref_lib as R

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("K,S", [(1001, 31), (101, 11), (40, 8)])
def test_scan_and_count_side_by_side(K, S):
    reads = A.reads(K, S, seed=123, scale=0.4) + A.hifi_like(40, 20000 if K > 500 else 5000, 6000 if K > 500 else 1200, seed=K)
    o0 = O.scan(reads, K, S, 0)
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    ref = db.flatten(n_nn=o0["n_nn"])
    for mode in (0, 1):
        o = O.scan(reads, K, S, mode)
        for f in ["hoco_l", "n_scm", "hoco_s", "ho_rl", "ho_l_rl", "n_nucl", "m_pos", "s_mer", "k_mer"]:
            assert np.array_equal(ref[f], o[f]), (mode, f)
    sc = R.ScmDb(db)
    rf, after = sc.flatten(), db.flatten()
    _, c = O.scan_and_count(reads, K, S, 1)
    for a, b in [("h", "h"), ("s", "s"), ("cov", "cov"), ("occ", "occ")]:
        assert np.array_equal(rf[a], c[b]), a
    assert np.array_equal(after["k_mer"], c["k_id"])
    sc.close()
    db.close()


def test_wavefront_random_side_by_side():
    rng = np.random.default_rng(77)
    for it in range(300):
        ts = A.rand_dna(rng, int(rng.integers(1, 120)))
        q = A.rand_dna(rng, int(rng.integers(1, 120))) if it % 3 == 0 else ts[:int(rng.integers(1, len(ts) + 1))] + A.rand_dna(rng, 3)
        bw = int(rng.integers(1, 10))
        w = R.Wavefront(ts, bw)
        r = w.step(q)
        w.close()
        assert O.wf_ed(ts, q, bw) == r
