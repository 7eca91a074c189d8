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


def test_config1s_shaped_reads_side_by_side(tmp_path):
    """a small read set of the config-1 surrogate's shape (oatk_amd.synth.CONFIG1S: organelles over a low-coverage background, low-complexity arrays,
    N runs, short and lower-case reads) through oracle and compiled reference; the reference reads it from the three gzip forms host/fasta_out.c writes"""
    from oatk_amd import synth
    K, S = 1001, 31
    cfg = dict(synth.CONFIG1S)
    cfg.update(n_reads=600, nuclear_len=3_000_000, n_ppm=30_000, lower_ppm=20_000, short_ppm=30_000)
    rs = synth.MixReadSet(**cfg)
    seq, off, lens = rs.slice(0, 600)
    reads = [seq[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)]
    assert sum(b"N" in r or b"n" in r for r in reads) >= 5 and sum(len(r) < K for r in reads) >= 5 and sum(r.islower() for r in reads) >= 3
    o = O.scan(reads, K, S, 1)
    assert o["ho_l_rl"].size > 0 and o["n_nucl"].size > 0
    flat = None
    for name, mode in (("one", synth.FA_GZ), ("bgzf", synth.FA_BGZF), ("members", synth.FA_GZ_MEMBERS)):
        path = str(tmp_path / (name + ".fa.gz"))
        synth.write_fasta(path, seq, off, lens, mode=mode, member_bytes=1_000_000, threads=4)
        db = R.SrDb([path], K, S, threads=2)
        ref = db.flatten(n_nn=o["n_nn"])
        for f in ["hoco_l", "n_scm", "hoco_s", "ho_rl", "ho_l_rl", "n_nucl", "m_pos", "s_mer", "k_mer"]:
            assert np.array_equal(ref[f], o[f]), (name, f)
        if flat is None:
            sc = R.ScmDb(db)
            rf, after = sc.flatten(), db.flatten()
            _, c = O.scan_and_count(reads, K, S, 1)
            for a in ("h", "s", "cov", "occ"):
                assert np.array_equal(rf[a], c[a]), a
            assert np.array_equal(after["k_mer"], c["k_id"])
            sc.close()
            flat = True
        db.close()
