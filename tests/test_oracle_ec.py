"""CPU: the oracle's error correction (oracle/ec.c) against golden vectors produced by the compiled reference's
read_error_correction (syncerr.c:819), and side by side with the compiled reference where it is available."""
import numpy as np
import pytest

import ec_util as E
import golden_util as G
import ref_lib as R

EC_CASES = ["ec_diploid_k101", "ec_repeats_k301", "ec_hifi_k1001"]


def run_oracle(g):
    K, c, a = int(g["K"]), int(g["c"]), float(g["a"])
    Gd = {"n_vtx": len(g["g_vtx_len"]), "n_arc": len(g["g_arc_w"]) if g["g_idx_n"].sum() else 0,
          "vtx_len": g["g_vtx_len"].copy(), "vtx_del": g["g_vtx_del"].copy(), "vtx_seq_off": g["g_vtx_seq_off"].copy(), "seq": g["g_seq"].copy(),
          "arc_w": g["g_arc_w"].copy(), "arc_ls": g["g_arc_ls"].copy(), "arc_cov": g["g_arc_cov"].copy(), "arc_del": g["g_arc_del"].copy(),
          "idx_p": g["g_idx_p"].copy(), "idx_n": g["g_idx_n"].copy()}
    scm_del = g["in_scm_del"].copy()
    E.oracle_find_error_syncmers(Gd, g["in_scm_cov"], scm_del, c, 10 * c, c, a)
    sr = {"hoco_l": g["in_hoco_l"], "hoco_s": g["in_hoco_s"], "n_scm": g["in_n_scm"], "k_mer": g["in_k_mer"], "m_pos": g["in_m_pos"], "s_mer": g["in_s_mer"]}
    oc = E.oracle_ec_reads(Gd, scm_del, g["in_scm_s"], K, float(g["max_edist"]), sr)
    od = E.oracle_update_db(oc["n_scm"], oc["k_mer"], oc["m_pos"], len(g["in_scm_s"]))
    return oc, od


@pytest.mark.parametrize("case", EC_CASES)
def test_oracle_ec_matches_reference_golden(case):
    g = G.load(case)
    oc, od = run_oracle(g)
    for f in ["n_scm", "k_mer", "m_pos", "s_mer"]:
        assert np.array_equal(oc[f], g["out_" + f]), f
    assert np.array_equal(od["cov"], g["out_scm_cov"])
    assert np.array_equal(od["del"], g["out_scm_del"])
    assert np.array_equal(od["occ"], g["out_scm_occ"])
    st, want = oc["stats"], g["out_summary"]
    assert [st[0] + st[5] + st[10], st[1] + st[6], st[2] + st[7], st[3] + st[8], st[4] + st[9]] == want.tolist()
    assert (oc["k_mer"] & 1).sum() > 0          # something was actually corrected


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_ec_side_by_side():
    import adversarial as A
    K, S, c = 301, 21, 5
    reads = A.hifi_like(250, 25000, 4000, seed=99, err=0.003)
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    scm = R.ScmDb(db)
    sr0, sc0 = db.flatten(), scm.flatten()
    g, Gd = E.ref_graph(db, scm)
    scm_del = sc0["del"].copy()
    E.oracle_find_error_syncmers(Gd, sc0["cov"], scm_del, c, 10 * c, c, 0.35)
    oc = E.oracle_ec_reads(Gd, scm_del, sc0["s"], K, 0.02, sr0)
    od = E.oracle_update_db(oc["n_scm"], oc["k_mer"], oc["m_pos"], sc0["n_scm"])
    E.reference_ec(db, scm, g, 0.02, c, 0.35)
    sr1, sc1 = db.flatten(), scm.flatten()
    for f in ["n_scm", "k_mer", "m_pos", "s_mer"]:
        assert np.array_equal(oc[f], sr1[f]), f
    assert np.array_equal(od["cov"], sc1["cov"]) and np.array_equal(od["del"], sc1["del"]) and np.array_equal(od["occ"], sc1["occ"])
    R.lib().refx_scg_destroy(g)
    scm.close()
    db.close()


def assert_ecgraph(og, want_v, want_w, want_ls, want_cov, want_idx_p, want_idx_n):
    na = og["n_arc"]
    assert not og["multi_arc"] and na == int(want_idx_n.sum()) and na > 0
    assert np.array_equal(og["arc_v"], want_v[:na]) and np.array_equal(og["arc_w"], want_w[:na])
    assert np.array_equal(og["arc_cov"], want_cov[:na])
    assert np.array_equal(og["arc_ls"], want_ls[:na])
    assert np.array_equal(og["idx_n"], want_idx_n)
    has = want_idx_n > 0
    assert np.array_equal(og["idx_p"][has], want_idx_p[has])


@pytest.mark.parametrize("case", EC_CASES)
def test_oracle_ecgraph_matches_reference_golden(case):
    """oracle/ecgraph.c against the graph the compiled reference built for the golden EC cases (make_syncmer_graph +
    scg_consensus), including the khashl-order tie rule of the arc overlaps"""
    g = G.load(case)
    off, occ = E.occ_lists(g["in_n_scm"], g["in_k_mer"], g["in_m_pos"], len(g["in_scm_s"]))
    og = E.oracle_ecgraph(g["in_n_scm"], g["in_k_mer"], g["in_m_pos"], off, occ, int(g["K"]))
    assert_ecgraph(og, g["g_arc_v"], g["g_arc_w"], g["g_arc_ls"], g["g_arc_cov"], g["g_idx_p"], g["g_idx_n"])


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("K,S,err,seed", [(101, 11, 0.01, 5), (301, 21, 0.002, 6), (1001, 31, 0.0008, 7)])
def test_oracle_ecgraph_side_by_side(K, S, err, seed):
    import adversarial as A
    reads = A.hifi_like(220, 12 * K, 5 * K, seed=seed, err=err)
    db = R.SrDb.from_reads(reads, K, S, threads=2)
    scm = R.ScmDb(db)
    sr0, sc0 = db.flatten(), scm.flatten()
    g, Gd = E.ref_graph(db, scm)
    occ_off = np.zeros(sc0["n_scm"] + 1, np.uint64)
    occ_off[1:] = np.cumsum(sc0["cov"], dtype=np.uint64)
    og = E.oracle_ecgraph(sr0["n_scm"], sr0["k_mer"], sr0["m_pos"], occ_off, sc0["occ"], K)
    off2, occ2 = E.occ_lists(sr0["n_scm"], sr0["k_mer"], sr0["m_pos"], sc0["n_scm"])       # the golden test rebuilds the lists this way
    assert np.array_equal(off2, occ_off) and np.array_equal(occ2, sc0["occ"])
    assert_ecgraph(og, Gd["arc_v"], Gd["arc_w"], Gd["arc_ls"], Gd["arc_cov"], Gd["idx_p"], Gd["idx_n"])
    assert np.array_equal(og["arc_comp"], Gd["arc_comp"][:og["n_arc"]])
    R.lib().refx_scg_destroy(g)
    scm.close()
    db.close()
