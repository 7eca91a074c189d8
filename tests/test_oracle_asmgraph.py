"""CPU: the oracle's assembly graph (oracle/asmgraph.c) against golden vectors produced by the compiled reference's
make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) (syncasm.c:203-299, run_syncasm.c:138), and side by side with it."""
import numpy as np
import pytest

import asm_util as AU
import golden_util as G
import ref_lib as R

EC_CASES = ["ec_diploid_k101", "ec_repeats_k301", "ec_hifi_k1001"]


@pytest.mark.parametrize("stage", ["raw", "ec"])
@pytest.mark.parametrize("case", EC_CASES)
def test_oracle_asmgraph_matches_reference_golden(case, stage):
    e, g = G.load(case), G.load("asmgraph_" + case)
    pre = "in_" if stage == "raw" else "out_"
    og = AU.oracle_asmgraph(e[pre + "n_scm"], e[pre + "k_mer"], e[pre + "m_pos"], e[pre + "scm_cov"], e[pre + "scm_del"], int(g["c"]), float(g["a"]))
    assert not og["multi_arc"] and len(og["arc_v"]) > 0 and 0 < len(og["vtx_scm"]) < len(e["in_scm_s"])
    AU.assert_asm_equal(og, g, stage + "_")
    # the two filters do something: fewer arcs than the unfiltered graph over the same vertices
    og0 = AU.oracle_asmgraph(e[pre + "n_scm"], e[pre + "k_mer"], e[pre + "m_pos"], e[pre + "scm_cov"], e[pre + "scm_del"], int(g["c"]), 0.0)
    assert len(og0["arc_v"]) >= len(og["arc_v"])


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("K,S,err,seed,c,a", [(101, 11, 0.01, 5, 3, 0.35), (301, 21, 0.002, 6, 0, 0.0), (301, 21, 0.004, 8, 2, 0.9),
                                               (1001, 31, 0.0008, 7, 5, 0.35), (101, 11, 0.01, 9, 1000000, 0.35)])
def test_oracle_asmgraph_side_by_side(K, S, err, seed, c, a):
    import adversarial as A
    reads = A.hifi_like(220, 12 * K, 5 * K, seed=seed, err=err)
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    scm = R.ScmDb(db)
    sr0, sc0 = db.flatten(), scm.flatten()
    og = AU.oracle_asmgraph(sr0["n_scm"], sr0["k_mer"], sr0["m_pos"], sc0["cov"], sc0["del"], c, a)
    want = AU.reference_asmgraph(db.handle, scm.handle, c, a)
    want["scm_del"] = scm.flatten()["del"]
    AU.assert_asm_equal(og, want)
    if c == 1000000:
        assert len(og["vtx_scm"]) == 0 and len(og["arc_v"]) == 0 and og["scm_del"].all()
    scm.close()
    db.close()
