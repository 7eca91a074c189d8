"""GPU + compiled reference, BASELINE.json configs[1] at FULL size: the same 200 k reads x ~15 kb through the reference (sr_read,
collect_syncmer_from_reads, make_syncmer_graph(0, 0.) + scg_consensus(hoco), read_error_correction, sr_db_stat; run_syncasm.c:81-131) and
through the device, every array compared element for element: the scan (hoco strings, run lengths, syncmer positions, s-mers, k-mer hashes),
the syncmer table (hashes, s-mers, coverage, occurrence lists, ids written back to the reads), the EC graph (arcs, coverage, hoco overlaps),
the corrected chains, the refreshed table, the block statistics and sr_stat_t.  About a minute: half of it the reference on host cores."""
import ctypes as C
import os

import numpy as np
import pytest

import ec_util as E
import ref_lib as R
from oatk_amd.synth import CONFIGS, ReadSet

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")]

K, S = 1001, 31


def test_config2_scan_count_graph_ec_equal_the_compiled_reference(hip, tmp_path):
    cfg = dict(CONFIGS["config2"])
    c = cfg["min_k_cov"]
    n = cfg["n_reads"]
    rs = ReadSet(**cfg)
    seq, off, lens = rs.slice(0, n)
    fa = str(tmp_path / "config2.fa")
    with open(fa, "wb") as f:
        for i in range(n):
            f.write(b">r%d\n" % i)
            f.write(seq[int(off[i]):int(off[i]) + int(lens[i])].tobytes())
            f.write(b"\n")
    threads = min(32, os.cpu_count() or 8)

    # ---- the reference, stage by stage ----
    db = R.SrDb([fa], K, S, threads)
    os.unlink(fa)
    assert db.n() == n
    r_scan = db.flatten()
    r_stat0 = db.stat()
    scm = R.ScmDb(db)
    r_kid = db.flatten()["k_mer"]
    r_cnt = scm.flatten()
    g, G = E.ref_graph(db, scm)
    summ = E.reference_ec(db, scm, g, 0.02, c, 0.35, threads=threads)
    r_ec, r_tab = db.flatten(), scm.flatten()
    r_stat1 = db.stat()

    # ---- the device ----
    hip.scan_host(seq, off, lens, K, S)
    d = hip.fetch_scan(off)
    for f in ("hoco_l", "n_scm", "hoco_s", "ho_rl", "ho_l_rl", "m_pos", "s_mer", "k_mer"):
        assert np.array_equal(d[f], r_scan[f]), "scan: " + f
    assert int(d["n_nn"].sum()) == 0
    st = hip.stat_raw()
    hip.count()
    dc = hip.fetch_count()
    assert dc["n_scm"] == r_cnt["n_scm"]
    for f in ("h", "s", "cov", "occ"):
        assert np.array_equal(dc[f], r_cnt[f]), "count: " + f
    assert np.array_equal(dc["k_id"], r_kid)
    hip.ec_graph()
    na = G["n_arc"]
    for name, key in (("EG_ARC_V", "arc_v"), ("EG_ARC_W", "arc_w"), ("EG_ARC_COV", "arc_cov"), ("EG_ARC_LS", "arc_ls"), ("EG_ARC_COMP", "arc_comp")):
        assert np.array_equal(hip.fetch(name).astype(np.uint64), G[key][:na].astype(np.uint64)), "EC graph: " + key
    stats = hip.ec(0.02, c, 0.35)
    for name, key in (("EC_N_SCM", "n_scm"), ("EC_KMER", "k_mer"), ("EC_MPOS", "m_pos"), ("EC_SMER", "s_mer")):
        assert np.array_equal(hip.fetch(name), r_ec[key]), "corrected chains: " + key
    assert np.array_equal(hip.fetch("EC_SCM_COV"), r_tab["cov"]) and np.array_equal(hip.fetch("EC_SCM_DEL"), r_tab["del"])
    assert np.array_equal(hip.fetch("EC_SCM_OCC"), r_tab["occ"])
    assert int(stats[0] + stats[5] + stats[10]) == summ["total"] and int(stats[2] + stats[7]) == summ["corrected"]
    assert int(stats[1] + stats[6]) == summ["uncorrected"] and int(stats[3] + stats[8]) == summ["ambiseq"] and int(stats[4] + stats[9]) == summ["ambipath"]
    assert summ["total"] > 100000
    full_stat1 = hip.stat_raw()

    # ---- the same round through the LIGHT graph -- the headline path of bench.py and the drop-in (include/oatk_hip_ec.h) -- against the reference directly ----
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    hip.ec_graph(light_c=c)
    kept = hip.fetch("EG_ARC_V").astype(np.uint64)
    rcov = r_cnt["cov"].astype(np.int64)
    both = (rcov[(G["arc_v"][:na] >> np.uint64(1)).astype(np.int64)] >= c) & (rcov[(G["arc_w"][:na] >> np.uint64(1)).astype(np.int64)] >= c)
    for name, key in (("EG_ARC_V", "arc_v"), ("EG_ARC_W", "arc_w"), ("EG_ARC_COV", "arc_cov"), ("EG_ARC_LS", "arc_ls"), ("EG_ARC_COMP", "arc_comp")):
        assert np.array_equal(hip.fetch(name).astype(np.uint64), G[key][:na][both].astype(np.uint64)), "light EC graph = the reference's arcs between candidates: " + key
    assert 0 < len(kept) < na
    stats_l = hip.ec(0.02, c, 0.35)
    for name, key in (("EC_N_SCM", "n_scm"), ("EC_KMER", "k_mer"), ("EC_MPOS", "m_pos"), ("EC_SMER", "s_mer")):
        assert np.array_equal(hip.fetch(name), r_ec[key]), "light graph, corrected chains: " + key
    assert np.array_equal(hip.fetch("EC_SCM_COV"), r_tab["cov"]) and np.array_equal(hip.fetch("EC_SCM_DEL"), r_tab["del"])
    assert np.array_equal(hip.fetch("EC_SCM_OCC"), r_tab["occ"])
    assert list(stats_l[:11]) == list(stats[:11])

    # ---- sr_db_stat before the count and after the correction (run_syncasm.c:88, :131) ----
    import test_gpu_dropin as TD
    H = TD.host_lib()
    H.oatk_stat_peaks.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for raw, (i8, d5) in ((st, r_stat0), (full_stat1, r_stat1), (hip.stat_raw(), r_stat1)):
        assert raw["n_syncmers"] == int(d5[0]) and raw["smer_unique"] == i8[0] and raw["kmer_unique"] == i8[4]
        assert raw["smer_cnt"][1] == i8[1] and raw["kmer_cnt"][1] == i8[5]
        for cnt, hom, het in ((raw["smer_cnt"], i8[2], i8[3]), (raw["kmer_cnt"], i8[6], i8[7])):
            a, b = C.c_int(), C.c_int()
            cc = np.ascontiguousarray(cnt, np.int64)
            H.oatk_stat_peaks(cc.ctypes.data, C.byref(a), C.byref(b))
            assert (a.value, b.value) == (hom, het)
        assert d5[2] == raw["sum_dist"] / raw["n_dist"]
    R.lib().refx_scg_destroy(g)
    scm.close()
    db.close()
