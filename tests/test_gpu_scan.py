"""GPU parity: HIP scan + count (through the C ABI) against the CPU oracle on the same inputs. Bit-exact."""
import numpy as np
import pytest

import adversarial as A
import oracle_lib as O
from oatk_amd import pack_reads

pytestmark = pytest.mark.gpu

SCAN_FIELDS = ["hoco_l", "n_scm", "n_nn", "n_lrl", "hoco_s", "ho_rl", "ho_l_rl", "n_nucl", "m_pos", "s_mer", "k_mer"]
KS = [(1001, 31), (101, 11), (61, 15), (33, 31), (25, 5), (64, 16), (40, 8), (2500, 31)]


def run_hip(hip, reads, K, S):
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    return hip.fetch_scan(off), off


def compare_scan(got, want):
    for f in SCAN_FIELDS:
        g, w = got[f], want[f]
        assert g.shape == w.shape, (f, g.shape, w.shape)
        if not np.array_equal(g, w):
            bad = np.nonzero(g != w)[0]
            raise AssertionError("%s differs at %d positions, first %s: got %s want %s" % (f, len(bad), bad[:5], g[bad[:5]], w[bad[:5]]))


@pytest.mark.parametrize("K,S", KS)
def test_scan_adversarial(hip, K, S):
    reads = A.reads(K, S)
    got, _ = run_hip(hip, reads, K, S)
    want = O.scan(reads, K, S, mode=0)
    compare_scan(got, want)


@pytest.mark.parametrize("K,S", [(1001, 31), (101, 11)])
def test_scan_byte_soup(hip, K, S):
    """kernel A classifies clean 16-byte vectors without its table; every byte that could be mistaken for a base sends its vector
    (and the one after it) to the table path -- at vector, wave and tile boundaries"""
    reads = A.byte_soup(K)
    got, _ = run_hip(hip, reads, K, S)
    compare_scan(got, O.scan(reads, K, S, mode=0))


@pytest.mark.parametrize("K,S", [(1001, 31), (101, 11), (25, 5)])
def test_count_adversarial(hip, K, S):
    reads = A.reads(K, S) + A.hifi_like(60, 6000 if K < 1001 else 30000, 1500 if K < 1001 else 8000)
    got, _ = run_hip(hip, reads, K, S)
    hip.count()
    c = hip.fetch_count()
    want_scan, want = O.scan_and_count(reads, K, S, mode=0)
    compare_scan(got, want_scan)
    assert c["n_scm"] == want["n_scm"]
    for f in ["h", "s", "cov", "occ_off", "occ", "k_id"]:
        assert np.array_equal(c[f], want[f]), f


def test_scan_hifi_like_k1001(hip):
    reads = A.hifi_like(300, 200000, 15000, seed=3)
    got, _ = run_hip(hip, reads, 1001, 31)
    hip.count()
    c = hip.fetch_count()
    want_scan, want = O.scan_and_count(reads, 1001, 31, mode=0)
    compare_scan(got, want_scan)
    for f in ["h", "s", "cov", "occ_off", "occ", "k_id"]:
        assert np.array_equal(c[f], want[f]), f
    assert int(got["n_scm"].sum()) > 0


@pytest.mark.parametrize("mask,collide,full", [(0xFF, 1, False), (0xFF, 1, True),
                                               (0xFFC0000000FFFFFF, 0, False), (0xFFC0000000FFFFFF, 0, True), (0xFF00000000000FFF, None, False)])
def test_forced_hash_collisions(hip, mask, collide, full, monkeypatch):
    """AND the hashes down to a few bits so unrelated k-mers share a 'hash': exercises the sequence comparison
    and first-seen split of process_kmer_cluster (syncmer.c:1293-1335).  (Batches this small are sorted on all 64 bits whatever the switch says; the sort on
    the top 40 bits with its repair pass is compared with it at 4.6 M records in tests/test_gpu_fullsize.py, with the same kinds of masks.)"""
    K, S = 101, 11
    reads = A.hifi_like(200 if mask == 0xFF else 80, 5000, 1500, seed=5)       # (0xFF: one run of ~6600 records, beyond OATK_SORT_REPAIR_MAX)
    seq, off, lens = pack_reads(reads)
    if full:
        monkeypatch.setenv("OATK_DEBUG_FULL_SORT", "1")
    hip.debug_hash_mask(mask)
    try:
        hip.scan_host(seq, off, lens, K, S)
        hip.count()
        c = hip.fetch_count()
        if collide is not None:
            assert hip.info()["collisions"] == collide
    finally:
        hip.debug_hash_mask(0xFFFFFFFFFFFFFFFF)
    # oracle with the same masked hashes
    oseq, ooff = O.pack_reads(reads)
    L = O.lib()
    out, p = O.scan_raw(oseq, ooff, K, S, 0)
    n = int(p.contents.tot_scm)
    for i in range(n):
        p.contents.k_mer[i] &= mask
    cp = L.orc_count(p, K)
    cc = cp.contents
    want = {"n_scm": int(cc.n_scm), "h": O._arr(cc.h, cc.n_scm, np.uint64), "s": O._arr(cc.s, cc.n_scm, np.uint64),
            "cov": O._arr(cc.cov, cc.n_scm, np.uint32), "occ": O._arr(cc.occ, cc.tot_occ, np.uint64),
            "k_id": O._arr(cc.k_id, cc.tot_occ, np.uint64), "err": int(cc.err)}
    L.orc_count_free(cp)
    L.orc_scan_free(p)
    assert c["n_scm"] == want["n_scm"]
    for f in ["h", "s", "cov", "occ", "k_id"]:
        assert np.array_equal(c[f], want[f]), f


import golden_util as G  # noqa: E402


@pytest.mark.parametrize("case", G.SCAN_CASES)
def test_against_reference_golden_vectors(hip, case):
    """HIP path vs the outputs of the compiled reference committed under tests/golden (no oracle in between)."""
    g = G.load(case)
    K, S = int(g["K"]), int(g["S"])
    reads = G.reads_of(g)
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    got = hip.fetch_scan(off)
    for f in G.SCAN_FIELDS:
        assert got[f].shape == g[f].shape and np.array_equal(got[f], g[f]), (case, f)
    hip.count()
    c = hip.fetch_count()
    assert np.array_equal(c["h"], g["scm_h"]) and np.array_equal(c["s"], g["scm_s"])
    assert np.array_equal(c["cov"], g["scm_cov"]) and np.array_equal(c["occ"], g["scm_occ"])
    assert np.array_equal(c["k_id"], g["k_id"])


def test_general_kernel_still_matches_at_k1001(hip):
    """K=1001 normally takes the fast syncmer kernel; force the general one and check it agrees with the oracle too."""
    reads = A.reads(1001, 31, seed=31, scale=0.6) + A.hifi_like(40, 40000, 9000, seed=8)
    want = O.scan(reads, 1001, 31, mode=0)
    hip.debug_force_general(True)
    try:
        got, _ = run_hip(hip, reads, 1001, 31)
    finally:
        hip.debug_force_general(False)
    compare_scan(got, want)
    got2, _ = run_hip(hip, reads, 1001, 31)
    compare_scan(got2, want)


@pytest.mark.parametrize("cap", [1, 3, 16])
def test_fast_kernel_record_list_overflows(hip, cap):
    """the fast kernel collects a read's syncmers in LDS and writes the records when the read is done; with room for only
    `cap` of them the list is written out in the middle of the read, and tiles with more syncmers than that go straight to
    records -- the same records either way"""
    reads = A.reads(1001, 31, seed=31, scale=0.6) + A.hifi_like(40, 60000, 20000, seed=8) + [A.hifi_like(1, 400000, 300000, seed=5)[0]]
    want = O.scan(reads, 1001, 31, mode=0)
    hip.debug_list_cap(cap)
    try:
        got, _ = run_hip(hip, reads, 1001, 31)
    finally:
        hip.debug_list_cap(0)
    compare_scan(got, want)
    with pytest.raises(RuntimeError):
        hip.debug_list_cap(129)


@pytest.mark.parametrize("K,S", [(550, 31), (1500, 21), (1976, 31), (1977, 31), (1054, 31), (1055, 31), (551, 31), (552, 31)])
def test_fast_kernel_k_range(hip, K, S):
    reads = A.hifi_like(30, 30000, 9000, seed=K) + A.reads(K, S, seed=3, scale=0.3)[:30]
    got, _ = run_hip(hip, reads, K, S)
    compare_scan(got, O.scan(reads, K, S, mode=1))


@pytest.mark.parametrize("K,S", [(1001, 31), (561, 31), (1060, 31), (1054, 31), (551, 31)])
def test_long_low_complexity_reads(hip, K, S):
    """many tiles of tandem repeats inside long reads: equal s-mer hashes everywhere, so the top-word filter of the fast kernel
    ties constantly and the exact 64-bit rule (chunk minima included) decides -- at the edges of the fast kernel's K - S range too"""
    rng = np.random.default_rng(K)
    w = K - S
    reads = []
    for unit_len in (2, 3, 7, S, S + 1, 64, w - 1, w, w + 1, K, 2048, 4096, 5000):
        unit = A.rand_nohp(rng, unit_len)
        rep = (unit * (26000 // unit_len + 2))[:26000]
        reads.append(A.rand_nohp(rng, 2500) + rep + A.rand_nohp(rng, 1800))
        reads.append(rep[:9000] + A.rand_dna(rng, 3000) + rep[:7000])
    reads.append(A.rand_dna(rng, 40000))
    got, _ = run_hip(hip, reads, K, S)
    want = O.scan(reads, K, S, mode=0)
    compare_scan(got, want)
    assert int(got["n_scm"].sum()) > 1000


@pytest.mark.parametrize("K", [1001, 991, 561])
def test_top_words_tie_between_different_smers(hip, K):
    """r03h: the fast kernel's ring holds top words only and re-hashes a position from the packed bases when top words tie.  Exact tandem repeats
    tie on the whole hash; here DIFFERENT s-mers share a top word (built by inverting the mixing function), at every distance around the window
    length, in both orders, with exact duplicates between them and on either strand (goldens of the compiled reference: topties_k*)"""
    reads = A.top_word_tie_reads(K)
    got, _ = run_hip(hip, reads, K, 31)
    compare_scan(got, O.scan(reads, K, 31, mode=0))


@pytest.mark.parametrize("K,S", [(1000, 30), (800, 20), (904, 27), (1024, 24), (601, 15), (1060, 31), (700, 17)])
def test_fast_kernel_other_s_and_every_window_alignment(hip, K, S):
    """r03j: the fast kernel's decision reads the ring at offsets fixed by (K - S) mod 8 -- one instantiation per alignment, for S = 31 and for any
    other S (even S: s-mers that are their own reverse complement; S <= 16: hashes without a top word, where every comparison goes to the 64-bit rule)"""
    rng = np.random.default_rng(K * 31 + S)
    w = K - S
    reads = A.hifi_like(12, 30000, 9000, seed=K + S) + A.reads(K, S, seed=5, scale=0.3)[:20]
    for unit_len in (2, 7, S, w - 1, w + 1):
        unit = A.rand_nohp(rng, unit_len)
        reads.append(A.rand_nohp(rng, 1500) + (unit * (9000 // unit_len + 2))[:9000] + A.rand_nohp(rng, 1200))
    pal = A.rand_nohp(rng, S // 2)
    reads.append(A.rand_nohp(rng, 3000) + pal + A.revcomp(pal) + A.rand_nohp(rng, 3000))       # an even-length s-mer equal to its reverse complement
    got, _ = run_hip(hip, reads, K, S)
    compare_scan(got, O.scan(reads, K, S, mode=1))


@pytest.mark.parametrize("K", [1001, 991, 1060])
def test_fast_kernel_four_wave_form(hip, K, monkeypatch):
    """r03p: the fast kernel runs as two waves per workgroup on a 2048-slot ring wherever K - S <= 1023 and as four waves on 4096 slots beyond
    (K = 1060 here); OATK_DEBUG_SYNCMER_NT256 takes the four-wave form for every K, on tandem repeats, top-word ties and ordinary reads"""
    monkeypatch.setenv("OATK_DEBUG_SYNCMER_NT256", "1")
    rng = np.random.default_rng(K)
    reads = A.hifi_like(10, 30000, 9000, seed=K) + A.top_word_tie_reads(K)[:40]
    for unit_len in (2, 7, 64, K - 32):
        unit = A.rand_nohp(rng, unit_len)
        reads.append(A.rand_nohp(rng, 2500) + (unit * (12000 // unit_len + 2))[:12000] + A.rand_nohp(rng, 1800))
    got, _ = run_hip(hip, reads, K, 31)
    compare_scan(got, O.scan(reads, K, 31, mode=0))
