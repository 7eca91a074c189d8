"""GPU + compiled reference on a read set SHAPED like BASELINE.json configs[0] (`syncasm -k 1001 -c 30 -t 8 ddAraThal4_organelle.hifi.fa.gz`,
/root/reference/README.md:34,77; the file itself is not available offline): oatk_amd.synth.CONFIG1S -- two organelle genomes at thousand-fold coverage
inside a 256 Mb nuclear background at ~7x (a table dominated by syncmers below -c 30, which find_error_syncmers marks deleted, syncerr.c:679-757),
homopolymers beyond 256 (ho_l_rl, syncmer.c:301-304), telomere / microsatellite / satellite arrays (every window minimum ties), reads with runs
of N (syncmer.c:316-323), reads shorter than K, lower-case reads; written as `.fa.gz`.

1. 200 k reads, stage by stage against the compiled reference reading the .fa.gz: scan (incl. n_nucl and ho_l_rl), count, EC graph (full and light),
   marks, corrected chains, refreshed table, block statistics, sr_stat_t -- every array, element for element.
2. the `syncasm` CLI on the .fa.gz in its three gzip forms (one member, BGZF, several members): both GFA files byte-identical, no call served by
   its original body."""
import ctypes as C
import filecmp
import os

import numpy as np
import pytest

import cli_util as CU
import ec_util as E
import ref_lib as R
from oatk_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")]

K, S = 1001, 31


@pytest.fixture(scope="module")
def config1s():
    cfg = dict(synth.CONFIG1S)
    rs = synth.MixReadSet(**cfg)
    seq, off, lens = rs.slice(0, cfg["n_reads"])
    return cfg, seq, off, lens


def test_config1s_every_array_equals_the_compiled_reference(hip, tmp_path, config1s):
    cfg, seq, off, lens = config1s
    c, n = cfg["min_k_cov"], cfg["n_reads"]
    fa = str(tmp_path / "config1s.fa.gz")
    synth.write_fasta(fa, seq, off, lens, mode=synth.FA_GZ)
    threads = min(32, os.cpu_count() or 8)
    # the shape this read set is meant to have
    assert int((lens < K).sum()) > n // 200
    pos = np.flatnonzero((seq == 78) | (seq == 110))                    # the only non-bases the generator writes: N / n (the padding between reads is not looked at)
    rd = np.searchsorted(off, pos.astype(np.uint64), side="right") - 1
    keep = pos < (off[rd] + lens[rd]).astype(np.int64)
    n_nn = np.bincount(rd[keep], minlength=n).astype(np.uint32)
    assert int((n_nn > 0).sum()) > 100, "reads with N"

    # ---- the reference, stage by stage, from the gzip'ed file ----
    db = R.SrDb([fa], K, S, threads)
    assert db.n() == n
    r_scan = db.flatten(n_nn=n_nn)
    r_stat0 = db.stat()
    scm = R.ScmDb(db)
    r_kid = db.flatten()["k_mer"]
    r_cnt = scm.flatten()
    assert float((r_cnt["cov"] < c).mean()) > 0.9, "most distinct syncmers lie below the cutoff"
    g, G = E.ref_graph(db, scm)
    summ = E.reference_ec(db, scm, g, 0.02, c, 0.35, threads=threads)
    r_ec, r_tab = db.flatten(), scm.flatten()
    r_stat1 = db.stat()
    assert int(r_scan["ho_l_rl"].size) > 0, "homopolymers beyond 256"

    # ---- the device ----
    hip.scan_host(seq, off, lens, K, S)
    d = hip.fetch_scan(off)
    assert np.array_equal(d["n_nn"], n_nn)
    for f in ("hoco_l", "n_scm", "hoco_s", "ho_rl", "ho_l_rl", "n_nucl", "m_pos", "s_mer", "k_mer"):
        assert np.array_equal(d[f], r_scan[f]), "scan: " + f
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
    assert summ["total"] > 10000
    full_stat1 = hip.stat_raw()

    # ---- the LIGHT graph (the headline path of bench.py and the drop-in) ----
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    hip.ec_graph(light_c=c)
    rcov = r_cnt["cov"].astype(np.int64)
    both = (rcov[(G["arc_v"][:na] >> np.uint64(1)).astype(np.int64)] >= c) & (rcov[(G["arc_w"][:na] >> np.uint64(1)).astype(np.int64)] >= c)
    for name, key in (("EG_ARC_V", "arc_v"), ("EG_ARC_W", "arc_w"), ("EG_ARC_COV", "arc_cov"), ("EG_ARC_LS", "arc_ls"), ("EG_ARC_COMP", "arc_comp")):
        assert np.array_equal(hip.fetch(name).astype(np.uint64), G[key][:na][both].astype(np.uint64)), "light EC graph: " + key
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


SIX = ("sr_read", "sr_db_stat", "collect_syncmer_from_reads", "make_syncmer_graph", "read_error_correction", "scg_read_alignment")


@pytest.mark.skipif(not CU.available(), reason="CLI binaries not built")
def test_config1s_cli_on_the_gzipped_file_in_three_forms(tmp_path, config1s):
    """40 k reads of the set (the organelles at ~600x / ~300x): the reference binary once, the drop-in binary on the single-member, the BGZF and the
    multi-member form of the same text"""
    cfg, seq, off, lens = config1s
    n, c = 40_000, cfg["min_k_cov"]
    d = str(tmp_path)
    forms = (("one", synth.FA_GZ, 0), ("bgzf", synth.FA_BGZF, 0), ("members", synth.FA_GZ_MEMBERS, 50_000_000))
    for name, mode, mb in forms:
        synth.write_fasta(os.path.join(d, name + ".fa.gz"), seq, off[:n], lens[:n], mode=mode, member_bytes=mb)
    threads = min(32, os.cpu_count() or 8)
    t_ref, _ = CU.run_cli(CU.CLI_REF, os.path.join(d, "one.fa.gz"), os.path.join(d, "ref"), K, c, threads)
    for name, _, _ in forms:
        t_dev, err = CU.run_cli(CU.CLI_DROPIN, os.path.join(d, name + ".fa.gz"), os.path.join(d, name), K, c, threads, {"OATK_DROPIN_LOG": "1"})
        for x in (".utg.gfa", ".utg.final.gfa"):
            assert filecmp.cmp(os.path.join(d, "ref" + x), os.path.join(d, name + x), shallow=False), "%s: %s differs from the reference's" % (name, x)
        tab = CU.served_table(err)
        assert all(f in tab and tab[f][0] > 0 for f in SIX), (name, tab)
        assert {f: v[2] for f, v in tab.items() if v[2] > 0} == {}, "%s: calls served by their original bodies" % name
        print("config1s CLI, %s: reference %.1f s, drop-in %.1f s" % (name, t_ref, t_dev))
    assert os.path.getsize(os.path.join(d, "ref.utg.final.gfa")) > 100_000


def test_an_arc_that_is_its_own_complement_gets_no_overlap(hip):
    """fold-back reads (A followed by its reverse complement: a syncmer next to its own reverse complement) make arcs v -> v^1; asmg_arc_fix_symm finds such
    an arc as its own complement and flags it (graph.c:221), and scg_consensus never gives a flagged arc an overlap (syncasm.c:779): ls stays 0.  The device
    assigned K - distance until round 4 (16 of the 4.5 M arcs of the config-1 surrogate)."""
    import oracle_lib as O
    rng = np.random.default_rng(11)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    genome = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 40000)].tolist())
    reads = [genome[a:a + 12000] for a in rng.integers(0, 28000, 40)]
    for a in rng.integers(0, 30000, 12):
        half = genome[a:a + 7000]
        reads.append(half + half.translate(comp)[::-1])
    from oatk_amd import pack_reads
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    got = hip.fetch_scan(off)
    hip.count()
    cnt = hip.fetch_count()
    hip.ec_graph()
    og = E.oracle_ecgraph(got["n_scm"], cnt["k_id"], got["m_pos"], cnt["occ_off"], cnt["occ"], K)
    v, w = hip.fetch("EG_ARC_V").astype(np.uint64), hip.fetch("EG_ARC_W").astype(np.uint64)
    own = (w ^ np.uint64(1)) == v
    assert int(own.sum()) >= 5, "fold-back reads make self-complementary arcs"
    for name, key in (("EG_ARC_V", "arc_v"), ("EG_ARC_W", "arc_w"), ("EG_ARC_COV", "arc_cov"), ("EG_ARC_LS", "arc_ls"), ("EG_ARC_COMP", "arc_comp")):
        assert np.array_equal(hip.fetch(name).astype(np.uint64), og[key].astype(np.uint64)), key
    assert not hip.fetch("EG_ARC_LS")[own].any() and hip.fetch("EG_ARC_COMP")[own].all()


def many_distance_reads(n=90):
    """two unique flanks around an (AC)n array whose length differs from read to read: the array itself yields no syncmers (period 2 divides K - S: the first
    and the last s-mer of every window tie, Open and Close cancel, syncmer.c:337,393), so the last syncmer before it and the first one behind it are adjacent
    on every read -- at a different distance on each"""
    rng = np.random.default_rng(3)
    nt = np.frombuffer(b"ACGT", np.uint8)
    fa, fb = (bytes(nt[rng.integers(0, 4, 4000)].tolist()) for _ in range(2))
    return [fa + b"AC" * (700 + 3 * i) + fb for i in range(n)] + [fa + b"AC" * 800 + fb] * 7


def test_an_arc_with_more_distinct_distances_than_the_lds_tables_hold(hip):
    """> 48 (EC graph) / > 64 (pair tables) distinct distances for one arc: OATK_E_SPLIT until round 4, now the khashl replay continues in global memory"""
    import oracle_lib as O
    import test_gpu_overlap as TO
    from oatk_amd import pack_reads
    reads = many_distance_reads()
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    got = hip.fetch_scan(off)
    hip.count()
    cnt = hip.fetch_count()
    hip.ec_graph()
    og = E.oracle_ecgraph(got["n_scm"], cnt["k_id"], got["m_pos"], cnt["occ_off"], cnt["occ"], K)
    # the shape the case is meant to have: some pair of syncmers adjacent at > 64 different distances
    tabs = TO.tables_from_chains(got["n_scm"], cnt["k_id"], got["m_pos"])
    assert max(len(t[0]) for t in tabs.values()) > 64
    for name, key in (("EG_ARC_V", "arc_v"), ("EG_ARC_W", "arc_w"), ("EG_ARC_COV", "arc_cov"), ("EG_ARC_LS", "arc_ls"), ("EG_ARC_COMP", "arc_comp")):
        assert np.array_equal(hip.fetch(name).astype(np.uint64), og[key].astype(np.uint64)), key
    assert TO.check_tables(hip, got["n_scm"], cnt["k_id"], got["m_pos"]) > 5
