"""CPU + compiled reference: DESIGN.md 10's claim about duplicate (v, w) arcs.  They need a syncmer adjacent to ITSELF on one strand (keys (2v, 2v) and
(2v + 1, 2v + 1), syncasm.c:256-257), i.e. two consecutive occurrences of one k-mer on a read -- which a perfect tandem repeat longer than K + period offers
in abundance, and which the closed-syncmer rule never selects: between a Close and its copy one period on there is always the Open of the same minimum
(and the other way round).  The reference's own graph of such reads, for several K and repeat units shorter and longer than the window, has thousands of
repeat syncmers and no self arc."""
import numpy as np
import pytest

import adversarial as A
import ec_util as E
import ref_lib as R

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("K,S,unit,mean_len", [(101, 11, 47, 2500), (301, 21, 47, 6000), (301, 21, 150, 6000), (301, 21, 286, 6000), (301, 21, 400, 6000),
                                               (1001, 31, 47, 12000), (1001, 31, 700, 12000), (1001, 31, 1300, 14000)])
def test_perfect_tandem_repeats_never_make_a_syncmer_adjacent_to_itself(K, S, unit, mean_len):
    reads = A.tandem_repeat_reads(K, unit_len=unit, mean_len=mean_len, n_reads=160)
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    scm = R.ScmDb(db)
    f = db.flatten()
    # the repeat did produce syncmers: some k-mer occurs several times on ONE read
    off = np.concatenate([[0], np.cumsum(f["n_scm"].astype(np.int64))])
    ids = f["k_mer"] >> np.uint64(1)
    most = max(int(np.bincount(ids[off[i]:off[i + 1]].astype(np.int64)).max()) for i in range(len(reads)) if off[i + 1] > off[i])
    assert most >= 3
    same_strand_neighbours = 0
    for i in range(len(reads)):
        a, m = ids[off[i]:off[i + 1]], f["m_pos"][off[i]:off[i + 1]] & 1
        same_strand_neighbours += int(((a[1:] == a[:-1]) & (m[1:] == m[:-1])).sum())
    assert same_strand_neighbours == 0
    g = R.lib().refx_make_graph(db.handle, scm.handle, 0, 0.0)
    G = E.flatten_graph(g)
    na = G["n_arc"]
    v, w = G["arc_v"][:na], G["arc_w"][:na]
    key = (v << np.uint64(32)) | w
    assert len(np.unique(key)) == na and not np.any(v == w)
    R.lib().refx_scg_destroy(g)
    scm.close()
    db.close()
