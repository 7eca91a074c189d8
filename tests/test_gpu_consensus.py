"""GPU: the device syncmer consensus (oatk_hip_consensus) against the oracle's totals (oracle/consensus.c, pinned to the compiled
reference by tests/test_oracle_consensus.py) for EVERY selected syncmer, and the strings built from the device arrays against the
compiled reference's scg_syncmer_consensus -- as scanned and after error correction."""
import numpy as np
import pytest

import cons_util as CU
import oracle_lib as O
import ref_lib as R
from oatk_amd import pack_reads

pytestmark = pytest.mark.gpu


def device_consensus(hip, min_cov):
    hip.consensus(min_cov)
    return {k: hip.fetch("CONS_" + k) for k in ("SEL", "SLOT", "RL", "MSEQ", "FIRST")}


def compare_with_oracle(hip, reads, K, S, after_ec, min_cov, c_ec=4):
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    sr = hip.fetch_scan(off)
    if after_ec:
        hip.ec_graph()
        st = hip.ec(0.02, c_ec, 0.35)
        assert int(st[2] + st[7]) > 0
        sr = dict(sr)
        sr["n_scm"], sr["k_mer"], sr["m_pos"] = hip.fetch("EC_N_SCM"), hip.fetch("EC_KMER"), hip.fetch("EC_MPOS")
        cov, occ, occ_off, dele = hip.fetch("EC_SCM_COV"), hip.fetch("EC_SCM_OCC"), hip.fetch("EC_SCM_OCC_OFF"), hip.fetch("EC_SCM_DEL")
        assert int((sr["k_mer"] & np.uint64(1)).sum()) > 0
    else:
        c = hip.fetch_count()
        sr = dict(sr)
        sr["k_mer"] = c["k_id"]
        cov, occ, occ_off, dele = c["cov"], c["occ"], c["occ_off"], np.zeros(len(c["cov"]), np.uint8)
    D = device_consensus(hip, min_cov)
    want_sel = np.nonzero((dele == 0) & (cov >= max(min_cov, 1)))[0].astype(np.uint32)
    assert np.array_equal(D["SEL"], want_sel) and len(want_sel) > 0
    slot = np.full(len(cov), 0xFFFFFFFF, np.uint32)
    slot[want_sel] = np.arange(len(want_sel), dtype=np.uint32)
    assert np.array_equal(D["SLOT"], slot)
    view, keep = CU.make_view(sr)
    rl = D["RL"].reshape(len(want_sel), K)
    n_long = 0
    for s_i, i in enumerate(want_sel.tolist()):
        tot, m, first = CU.oracle_rl(view, occ[int(occ_off[i]):int(occ_off[i + 1])], K)
        assert m == int(D["MSEQ"][s_i]) and first == int(D["FIRST"][s_i]), i
        if m:
            want = np.floor(tot.astype(np.float64) / m + 0.5).astype(np.uint32)        # lround of a non-negative quotient
            assert np.array_equal(rl[s_i], want), i
            n_long += int(tot.max() // m >= 255)
    return D, view, keep, n_long


@pytest.mark.parametrize("K,S,after_ec", [(101, 11, False), (101, 11, True), (301, 21, True), (1001, 31, False)])
def test_device_consensus_matches_oracle(hip, K, S, after_ec):
    reads = CU.long_run_reads(K + 1, K)
    D, view, keep, n_long = compare_with_oracle(hip, reads, K, S, after_ec, 2)
    assert n_long > 0                                              # run lengths behind the 255 escape took part


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
def test_strings_from_device_arrays_equal_the_reference(hip):
    """the string scg_syncmer_consensus appends, rebuilt from CONS_RL / CONS_MSEQ / CONS_FIRST, for both strands and several `beg`"""
    K, S = 101, 11
    reads = CU.long_run_reads(7, K)
    D, view, keep, _ = compare_with_oracle(hip, reads, K, S, False, 1)
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    scm = R.ScmDb(db)
    rl = D["RL"].reshape(len(D["SEL"]), K)
    rng = np.random.default_rng(3)
    for s_i in rng.choice(len(D["SEL"]), min(80, len(D["SEL"])), replace=False).tolist():
        i, m = int(D["SEL"][s_i]), int(D["MSEQ"][s_i])
        tot = rl[s_i].astype(np.uint64) * np.uint64(m)             # lround(tot / m) of these totals is CONS_RL again
        for rev in (0, 1):
            for beg in (0, 5, K - 1, -2):
                assert CU.oracle_string(view, tot, m, int(D["FIRST"][s_i]), K, rev, beg, 0) == CU.reference_string(db, scm, i, rev, beg, 0), (i, rev, beg)
    scm.close()
    db.close()
