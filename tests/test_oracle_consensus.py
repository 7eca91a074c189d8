"""CPU: oracle/consensus.c against the compiled reference's scg_syncmer_consensus (syncasm.c:888-1003), side by side: before and
after error correction (corrected entries are skipped), both strands, positive / zero / negative `beg`, hoco and base space,
homopolymers beyond the 255 escape."""
import numpy as np
import pytest

import adversarial as A
import cons_util as CU
import ec_util as E
import ref_lib as R

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")


def check_all(db, scm, K, rng, n_pick=60):
    sr, sc = db.flatten(), scm.flatten()
    view, keep = CU.make_view(sr)
    occ_off = np.concatenate([[0], np.cumsum(sc["cov"].astype(np.uint64))]).astype(np.int64)
    ids = np.nonzero(sc["cov"] > 0)[0]
    pick = ids if len(ids) <= n_pick else rng.choice(ids, n_pick, replace=False)
    n_checked = n_rl = 0
    for i in pick.tolist():
        tot, m, first = CU.oracle_rl(view, sc["occ"][occ_off[i]:occ_off[i + 1]], K)
        n_rl += int(tot.sum() > 0)
        for rev in (0, 1):
            for beg in (0, 1, 17, K // 2, K - 1, -3):
                for hoco in (0, 1):
                    want = CU.reference_string(db, scm, i, rev, beg, hoco)
                    got = CU.oracle_string(view, tot, m, first, K, rev, beg, hoco)
                    assert got == want, (i, rev, beg, hoco)
                    n_checked += 1
    return n_checked, n_rl


@pytest.mark.parametrize("K,S", [(101, 11), (301, 21)])
def test_consensus_oracle_matches_reference(K, S):
    rng = np.random.default_rng(K)
    reads = CU.long_run_reads(K, K)
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    scm = R.ScmDb(db)
    assert int((db.flatten()["ho_rl"] == 255).sum()) > 0          # long runs are in play
    n, n_rl = check_all(db, scm, K, rng)
    assert n > 500 and n_rl > 10
    # after error correction some entries are 'corrected' and must be skipped
    g, _ = E.ref_graph(db, scm)
    E.reference_ec(db, scm, g, 0.02, 4, 0.35)
    assert int((db.flatten()["k_mer"] & 1).sum()) > 0
    n2, _ = check_all(db, scm, K, rng)
    assert n2 > 500
    R.lib().refx_scg_destroy(g)
    scm.close()
    db.close()
