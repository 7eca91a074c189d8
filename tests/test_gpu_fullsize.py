"""GPU, BASELINE.json configs[1] (200 k reads x ~15 kb), configs[2] (2 M reads x ~15 kb = 30 Gbases, the configuration the metric is quoted
on) and one GPU's eighth of configs[4] (1.25 M reads x ~20 kb, -c 150) at FULL size, k = 1001, s = 31: properties that do not need an oracle run at that size -- conservation sums, orderings, run-to-run
determinism of every resident result (the solver pulls blocks from a shared queue, so scheduling differs between runs), strand symmetry --
plus a bit-exact oracle check of reads sampled from the full batch.  (tests/test_gpu_fullsize_ref.py compares config 2 with the compiled
reference element for element.)"""
import zlib

import numpy as np
import pytest

import oracle_lib as O
from oatk_amd.synth import CONFIGS, ReadSet

pytestmark = pytest.mark.gpu

K, S = 1001, 31
COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
    COMP[a] = b


@pytest.fixture(scope="module", params=["config2", "config3", "config5/8"])
def batch(request):
    """config5/8: what ONE of eight GPUs holds of BASELINE.json configs[4] (10 M reads x ~20 kb, -c 150): 1.25 M reads, 25 Gbases; reads of up to 40 kb
    carry more syncmers than a wave has lanes, so the lane-serial block walk runs beside the wave-per-read one at scale"""
    cfg = dict(CONFIGS[request.param.split("/")[0]])
    if "/" in request.param:
        cfg["n_reads"] //= int(request.param.split("/")[1])
    rs = ReadSet(**cfg)
    seq, off, lens = rs.slice(0, cfg["n_reads"])
    yield cfg, seq, off, lens
    del seq


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8))


SCAN_BUFS = ["HOCO_L", "N_SCM", "SCM_OFF", "POS_MPOS", "POS_SMER", "POS_HASH"]
COUNT_BUFS = ["POS_KID", "SCM_H", "SCM_S", "SCM_COV", "SCM_OCC_OFF", "SCM_OCC"]
EC_BUFS = ["EC_N_SCM", "EC_SCM_OFF", "EC_KMER", "EC_MPOS", "EC_SMER", "EC_SCM_COV", "EC_SCM_DEL", "EC_SCM_OCC_OFF", "EC_SCM_OCC", "EC_ERR_DEL"]
EG_BUFS = ["EG_IDX_N", "EG_ARC_V", "EG_ARC_W", "EG_ARC_LS", "EG_ARC_COV", "EG_ARC_COMP"]


def test_full_size_pipeline_properties(hip, batch):
    cfg, seq, off, lens = batch
    n = len(off)
    c = cfg["min_k_cov"]
    sums = []
    for rep in range(2):
        hip.scan_host(seq, off, lens, K, S)
        hip.count()
        hip.ec_graph()
        st = hip.ec(0.02, c, 0.35)
        sums.append({b: crc(hip.fetch(b)) for b in SCAN_BUFS + COUNT_BUFS + EC_BUFS + EG_BUFS} | {"stats": st[:11].tolist()})
    assert sums[0] == sums[1]                                     # nothing depends on scheduling

    info = hip.info()
    hoco_l, n_scm, scm_off = hip.fetch("HOCO_L"), hip.fetch("N_SCM"), hip.fetch("SCM_OFF")
    m_pos = hip.fetch("POS_MPOS")
    assert n == info["n_reads"] and int(n_scm.sum()) == info["n_occ"] == len(m_pos)
    assert np.all(hoco_l <= lens) and np.all(hoco_l > 0)
    assert np.array_equal(scm_off[1:] - scm_off[:-1], n_scm.astype(np.uint64))
    # chains are ordered by position inside a read, and every k-mer fits its read
    pos = (m_pos >> 1).astype(np.int64)
    first = np.zeros(len(pos), bool)
    first[scm_off[:-1][n_scm > 0].astype(np.int64)] = True
    assert np.all((np.diff(pos) > 0) | first[1:])
    assert np.all(pos + K <= np.repeat(hoco_l.astype(np.int64), n_scm))
    # count: coverage is conserved; ids are dense; equal hashes got equal ids; the table is sorted by first appearance
    cov, occ_off, occ, kid, h = hip.fetch("SCM_COV"), hip.fetch("SCM_OCC_OFF"), hip.fetch("SCM_OCC"), hip.fetch("POS_KID"), hip.fetch("POS_HASH")
    assert int(cov.sum()) == info["n_occ"] and len(cov) == info["n_scm"]
    assert np.array_equal(np.bincount((kid >> np.uint64(1)).astype(np.int64), minlength=len(cov)), cov)
    assert np.array_equal(hip.fetch("SCM_H")[(kid >> np.uint64(1)).astype(np.int64)], h)
    assert np.all(np.diff(occ.astype(np.int64))[np.ones(len(occ) - 1, bool) & ~np.isin(np.arange(1, len(occ)), occ_off[1:-1].astype(np.int64))] > 0)
    # EC graph: symmetric (every arc has its complement with equal coverage and overlap), CSR consistent
    av, aw, acov, als, idx_n = hip.fetch("EG_ARC_V"), hip.fetch("EG_ARC_W"), hip.fetch("EG_ARC_COV"), hip.fetch("EG_ARC_LS"), hip.fetch("EG_IDX_N")
    assert int(idx_n.sum()) == len(av) and np.all(np.diff((av << np.uint64(32) | aw).astype(np.uint64).view(np.int64)) > 0)
    akey = av << np.uint64(32) | aw
    ckey = (aw ^ np.uint64(1)) << np.uint64(32) | (av ^ np.uint64(1))
    ci = np.searchsorted(akey, ckey)
    assert np.array_equal(akey[ci], ckey) and np.array_equal(acov[ci], acov) and np.array_equal(als[ci], als)
    assert int(acov.sum()) + int(acov[(aw ^ np.uint64(1)) == av].sum()) == 2 * (info["n_occ"] - int((n_scm > 0).sum()))
    # error correction: conservation, the refreshed table is the histogram of the corrected chains, corrected entries point at live syncmers
    st = np.array(sums[0]["stats"])
    new_n, new_k, ecov, edel = hip.fetch("EC_N_SCM"), hip.fetch("EC_KMER"), hip.fetch("EC_SCM_COV"), hip.fetch("EC_SCM_DEL")
    assert int(new_n.sum()) == len(new_k) == int(ecov.sum())
    assert np.array_equal(np.bincount((new_k >> np.uint64(1)).astype(np.int64), minlength=len(ecov)), ecov)
    assert int(st[0] + st[5] + st[10]) > 0 and int(st[2] + st[7]) > 0.9 * int(st[0] + st[5])      # HiFi-like errors are correctable
    err_del = hip.fetch("EC_ERR_DEL")
    assert not np.any(err_del[(new_k[(new_k & np.uint64(1)) == 1] >> np.uint64(1)).astype(np.int64)])
    assert np.all(edel[ecov == 0] == 1)


def test_full_size_sample_against_oracle(hip, batch):
    cfg, seq, off, lens = batch
    hip.scan_host(seq, off, lens, K, S)
    n_scm, scm_off, hoco_l = hip.fetch("N_SCM"), hip.fetch("SCM_OFF"), hip.fetch("HOCO_L")
    m_pos, s_mer, k_mer = hip.fetch("POS_MPOS"), hip.fetch("POS_SMER"), hip.fetch("POS_HASH")
    pick = np.random.default_rng(7).choice(len(off), 96, replace=False)
    reads = [seq[int(off[i]):int(off[i]) + int(lens[i])].tobytes() for i in pick]
    want = O.scan(reads, K, S, mode=0)
    assert np.array_equal(hoco_l[pick], want["hoco_l"]) and np.array_equal(n_scm[pick], want["n_scm"])
    sel = np.concatenate([np.arange(int(scm_off[i]), int(scm_off[i + 1])) for i in pick])
    assert np.array_equal(m_pos[sel], want["m_pos"]) and np.array_equal(s_mer[sel], want["s_mer"]) and np.array_equal(k_mer[sel], want["k_mer"])


def test_full_size_ec_sample_against_oracle(hip, batch):
    """the correction at full size against the CPU restatement (oracle/ec.c: the block finder, the DFS over the graph, the wavefront edit distance,
    the splice) on reads sampled from the full batch.  The oracle corrects the sampled reads against the device's own graph of ALL reads -- the light
    graph's arcs with the marks of find_error_syncmers, the k-mer of every live vertex cut out of the hoco string of its first occurrence -- so what is
    compared is every decision the solver made for those reads: which blocks, which path, which ambiguity, the spliced chains."""
    import ec_util as E
    cfg, seq, off, lens = batch
    c = cfg["min_k_cov"]
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    hip.ec_graph(light_c=c)
    hip.ec(0.02, c, 0.35)
    ns = hip.info()["n_scm"]
    err_del = hip.fetch("EC_ERR_DEL")
    av, aw = hip.fetch("EG_ARC_V"), hip.fetch("EG_ARC_W")
    idx_n = hip.fetch("EG_IDX_N").astype(np.uint64)
    G = {"n_vtx": ns, "n_arc": len(av), "vtx_len": np.full(ns, K, np.uint64), "vtx_del": err_del.copy(), "vtx_seq_off": np.zeros(ns, np.uint64),
         "arc_w": aw, "arc_ls": hip.fetch("EG_ARC_LS").astype(np.uint64), "arc_cov": hip.fetch("EG_ARC_COV"),
         "arc_del": (err_del[(av >> np.uint64(1)).astype(np.int64)] | err_del[(aw >> np.uint64(1)).astype(np.int64)]).astype(np.uint8) if len(av) else np.zeros(1, np.uint8),
         "idx_p": np.where(idx_n > 0, hip.fetch("EG_IDX_P"), 0).astype(np.uint64), "idx_n": idx_n}
    # vertex strings of the live vertices: the oriented k-mer of the syncmer's first occurrence (syncasm.c:910-940), one base letter per byte
    hoco_l, n_scm, scm_off = hip.fetch("HOCO_L"), hip.fetch("N_SCM"), hip.fetch("SCM_OFF").astype(np.int64)
    m_pos, s_mer, kid = hip.fetch("POS_MPOS"), hip.fetch("POS_SMER"), hip.fetch("POS_KID")
    hoco_s = hip.fetch("HOCO_S")
    occ_off, occ = hip.fetch("SCM_OCC_OFF").astype(np.int64), hip.fetch("SCM_OCC")
    live = np.nonzero(err_del == 0)[0]
    assert 0 < len(live) < 200000
    nt = np.frombuffer(b"ACGT", np.uint8)
    seqs = np.zeros(len(live) * K, np.uint8)
    for t, v in enumerate(live):
        o = int(occ[occ_off[v]])
        rd, idx = o >> 32, (o >> 1) & 0x7FFFFFFF
        mp = int(m_pos[scm_off[rd] + idx])
        p = (mp >> 1) + np.arange(K)
        codes = (hoco_s[int(off[rd]) // 4 + (p >> 2)] >> (((p & 3) ^ 3) << 1)) & 3
        seqs[t * K:(t + 1) * K] = nt[(3 - codes[::-1]) if mp & 1 else codes]
        G["vtx_seq_off"][v] = t * K
    G["seq"] = seqs
    # the sampled reads as a little sr_db of their own (ids stay global)
    pick = np.sort(np.random.default_rng(11).choice(len(off), 400, replace=False))
    sel = np.concatenate([np.arange(scm_off[i], scm_off[i + 1]) for i in pick])
    sr = {"hoco_l": hoco_l[pick], "n_scm": n_scm[pick], "k_mer": kid[sel], "m_pos": m_pos[sel], "s_mer": s_mer[sel],
          "hoco_s": np.concatenate([hoco_s[int(off[i]) // 4:int(off[i]) // 4 + (int(hoco_l[i]) + 3) // 4] for i in pick])}
    want = E.oracle_ec_reads(G, err_del.copy(), hip.fetch("SCM_S"), K, 0.02, sr)
    new_n, new_off = hip.fetch("EC_N_SCM"), hip.fetch("EC_SCM_OFF").astype(np.int64)
    assert np.array_equal(new_n[pick], want["n_scm"])
    nsel = np.concatenate([np.arange(new_off[i], new_off[i + 1]) for i in pick])
    for name, key in (("EC_KMER", "k_mer"), ("EC_MPOS", "m_pos"), ("EC_SMER", "s_mer")):
        assert np.array_equal(hip.fetch(name)[nsel], want[key]), key
    assert int(want["stats"][2] + want["stats"][7]) > 100            # the sample did hold blocks that were corrected


def test_strand_symmetry(hip, batch):
    """a read and its reverse complement select the same k-mers: same hashes in reverse order, mirrored positions, flipped strands"""
    cfg, seq, off, lens = batch
    if cfg["n_reads"] > 200000:
        pytest.skip("per-read property: the 200 k-read batch covers it")
    m = 20000
    sub_end = int(off[m - 1]) + int((int(lens[m - 1]) + 63) // 64 * 64)
    fwd = seq[:sub_end]
    rc = np.zeros_like(fwd)
    for i in range(m):
        o, l = int(off[i]), int(lens[i])
        rc[o:o + l] = COMP[fwd[o:o + l][::-1]]
    res = []
    for s in (fwd, rc):
        hip.scan_host(s, off[:m], lens[:m], K, S)
        res.append((hip.fetch("N_SCM"), hip.fetch("SCM_OFF"), hip.fetch("HOCO_L"), hip.fetch("POS_MPOS"), hip.fetch("POS_HASH"), hip.fetch("POS_SMER")))
    (n0, o0, hl0, mp0, h0, s0), (n1, o1, hl1, mp1, h1, s1) = res
    assert np.array_equal(n0, n1) and np.array_equal(hl0, hl1) and int(n0.sum()) > 0
    # reverse every chain of the second scan
    idx = np.concatenate([np.arange(int(o1[i + 1]) - 1, int(o1[i]) - 1, -1) for i in range(m)])
    assert np.array_equal(h0, h1[idx]) and np.array_equal(s0 >> np.uint64(1), s1[idx] >> np.uint64(1))
    hl = np.repeat(hl0.astype(np.int64), n0)
    assert np.array_equal((mp0 >> 1).astype(np.int64), hl - K - (mp1[idx] >> 1).astype(np.int64))
    assert np.array_equal(mp0 & 1, (mp1[idx] & 1) ^ 1)


def test_full_size_graph_tables_and_alignment_properties(hip, batch):
    """the rows behind the error correction at full size: the assembly graph is symmetric and indexed consistently, the pair tables add up to
    the uncorrected adjacencies, every read aligns along its own corrected chain on the one-syncmer-per-vertex graph, and the two output
    paths of the aligner (pool + gather, count + second run) agree"""
    cfg, seq, off, lens = batch
    n, c = len(off), cfg["min_k_cov"]
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    hip.ec_graph()
    hip.ec(0.02, c, 0.35)
    n_scm, k_mer, m_pos = hip.fetch("EC_N_SCM"), hip.fetch("EC_KMER"), hip.fetch("EC_MPOS")
    cov, dele = hip.fetch("EC_SCM_COV"), hip.fetch("EC_SCM_DEL")

    # ---- assembly graph (make_syncmer_graph -c 30 -a 0.35 + asmg_finalize) ----
    nv, na = hip.asm_graph(c, 0.35)
    g = hip.fetch_asm_graph()
    keep = (dele == 0) & (cov >= c)
    assert nv == int(keep.sum()) and np.array_equal(g["vtx_scm"], np.nonzero(keep)[0].astype(np.uint32)) and np.array_equal(g["vtx_cov"], cov[keep])
    assert np.array_equal(g["scm_del"], (~keep).astype(np.uint8))
    v, w = g["arc_v"], g["arc_w"]
    key = v << np.uint64(32) | w
    assert na == len(v) and np.all(key[1:] > key[:-1])                                  # (v, w) order, no duplicate arcs
    ckey = (w ^ np.uint64(1)) << np.uint64(32) | (v ^ np.uint64(1))
    ci = np.searchsorted(key, ckey)
    assert np.array_equal(key[ci], ckey)                                                # every arc has its complement ...
    assert np.array_equal(g["arc_cov"][ci], g["arc_cov"]) and np.array_equal(g["arc_link"][ci], g["arc_link"])    # ... same coverage, same link
    self_c = ci == np.arange(na)
    assert np.array_equal(g["arc_comp"][ci][~self_c] ^ 1, g["arc_comp"][~self_c]) and np.all(g["arc_comp"][self_c] == 1)
    assert int(g["arc_link"].max()) + 1 == len(np.unique(g["arc_link"])) == (na + int(self_c.sum())) // 2
    idx_n = np.bincount(v.astype(np.int64), minlength=2 * nv)
    assert np.array_equal(g["idx_n"], idx_n.astype(np.uint32))
    has = idx_n > 0
    assert np.array_equal(g["idx_p"][has], (np.cumsum(idx_n) - idx_n)[has].astype(np.uint64))
    assert np.all(g["arc_cov"] >= 0.35 * np.minimum(g["vtx_cov"][(v >> np.uint64(1)).astype(np.int64)], g["vtx_cov"][(w >> np.uint64(1)).astype(np.int64)]))

    # ---- pair tables (calc_syncmer_overlap): counts add up to the adjacent pairs without a corrected member ----
    np_, ne = hip.overlap_hist()
    okey, ooff, odist, ocnt = hip.fetch("OVL_KEY"), hip.fetch("OVL_OFF"), hip.fetch("OVL_DIST"), hip.fetch("OVL_CNT")
    first = np.zeros(len(k_mer), bool)
    first[(np.cumsum(n_scm, dtype=np.int64) - n_scm)[n_scm > 0]] = True
    corr = (k_mer & np.uint64(1)).astype(bool)
    pair_ok = ~first & ~corr
    pair_ok[1:] &= ~corr[:-1]
    assert int(ocnt.sum()) == int(pair_ok.sum()) and np_ == len(okey) and np.all(okey[1:] > okey[:-1]) and int(ooff[-1]) == ne == len(odist)
    assert np.all(odist > 0) and np.all(ocnt > 0)
    d_all = (m_pos[1:] >> 1).astype(np.int64) - (m_pos[:-1] >> 1).astype(np.int64)
    assert int((odist.astype(np.int64) * ocnt).sum()) == int(d_all[pair_ok[1:]].sum())    # and so do the distances

    # ---- read alignment against the graph with one syncmer per vertex ----
    su_off = np.zeros(len(dele) + 1, np.uint64)
    su_off[1:] = np.cumsum(g["scm_del"] == 0)
    graph = {"n_scm": len(dele), "su_off": su_off, "su_uid": np.arange(nv, dtype=np.uint64) << np.uint64(1), "su_pos": np.zeros(nv, np.uint32),
             "utg_n": np.ones(nv, np.uint32), "idx_p": g["idx_p"], "idx_n": g["idx_n"].astype(np.uint64), "arc_w": w,
             "arc_ln": np.zeros(na, np.uint64), "arc_del": np.zeros(na, np.uint8)}
    res = []
    for two_pass in (0, 1):
        hip._check(hip.L.oatk_hip_debug_align_two_pass(hip.h, two_pass), "oatk_hip_debug_align_two_pass")
        n_aln, n_frg, st = hip.read_alignment(graph)
        res.append((n_aln, n_frg, st.tolist(), [crc(hip.fetch("RA_" + b)) for b in ("ALN_SID", "ALN_OFF", "ALN_S", "FRG_UID", "FRG_UBEG", "FRG_UEND", "FRG_SBEG", "FRG_SEND")]))
    hip._check(hip.L.oatk_hip_debug_align_two_pass(hip.h, 0), "oatk_hip_debug_align_two_pass")
    assert res[0] == res[1]
    n_aln, n_frg, st = res[0][:3]
    # (reads beyond the aligner's working limits are handed back for the original routine: none at 15 kb, a handful of the longest at 20 kb)
    assert st[2] <= (0 if cfg["mean_len"] <= 15000 else 1e-5 * n) and st[0] > 0.99 * n and st[1] <= st[0]
    sid, aoff, uid = hip.fetch("RA_ALN_SID"), hip.fetch("RA_ALN_OFF"), hip.fetch("RA_FRG_UID")
    sb, se = hip.fetch("RA_FRG_SBEG"), hip.fetch("RA_FRG_SEND")
    assert np.all(sid[1:] >= sid[:-1]) and int(aoff[-1]) == n_frg == len(uid)
    # on this graph a fragment is one syncmer of the read, on the vertex of that syncmer in the read's orientation
    assert np.array_equal(sb, se)
    starts = np.cumsum(n_scm, dtype=np.int64) - n_scm
    slot = np.repeat(starts[sid.astype(np.int64)], np.diff(aoff).astype(np.int64)) + sb
    vtx_of = np.full(len(dele), -1, np.int64)
    vtx_of[g["vtx_scm"].astype(np.int64)] = np.arange(nv)
    want_uid = (vtx_of[(k_mer[slot] >> np.uint64(1)).astype(np.int64)] << 1 | (m_pos[slot] & 1)).astype(np.uint64)
    assert np.array_equal(uid, want_uid)
    per_aln = np.diff(aoff).astype(np.int64)
    assert np.all(per_aln >= np.ceil(0.9 * n_scm[sid.astype(np.int64)] - 1e-9))         # min_a_frac


@pytest.mark.parametrize("mask", [0xFFFFFFFFFFFFFFFF, 0xFFFC000000FFFFFF, 0xFFFFC00000000FFF, 0x0000000000FFFFFF])
def test_sort_on_top_bits_with_repair_equals_the_sort_on_all_bits(hip, mask, monkeypatch):
    """r03o: batches of 4 M records and more are sorted on the top 40 bits of their k-mer hashes (five radix passes), and the runs in which different hashes
    share those bits are repaired (count.hpp: sort_repair_kernel).  220 k reads of config 2 (4.6 M records): untouched hashes (a handful of such runs at most), hashes masked
    to 14 + 24 bits (every run holds several hashes, interleaved in slot order), to 18 + 12 bits (mixed runs WITH true collisions inside them) and to the low 24 bits
    alone (ONE run of everything: too long for the repair, the count sorts again on all 64 bits) must
    give the table, the ids and the occurrence lists of the eight-pass sort, array for array."""
    cfg = dict(CONFIGS["config2"])
    cfg["n_reads"] = 220_000                    # (config 2 itself is 4.18 M records, just under the 2^22 from which the sort looks at the top bits only)
    rs = ReadSet(**cfg)
    seq, off, lens = rs.slice(0, cfg["n_reads"])
    res = {}
    hip.debug_hash_mask(mask)
    try:
        for mode in ("top", "full"):
            if mode == "full":
                monkeypatch.setenv("OATK_DEBUG_FULL_SORT", "1")
            hip.scan_host(seq, off, lens, K, S)
            hip.count()
            assert hip.info()["n_occ"] >= 1 << 22
            res[mode] = {b: crc(hip.fetch(b)) for b in COUNT_BUFS} | {"collisions": hip.info()["collisions"], "n_scm": hip.info()["n_scm"]}
    finally:
        hip.debug_hash_mask(0xFFFFFFFFFFFFFFFF)
    assert res["top"] == res["full"]
    if mask == 0xFFFFC00000000FFF:
        assert res["top"]["collisions"] == 1


def test_a_sort_whose_output_is_not_trusted_is_done_again_on_all_bits(hip, monkeypatch):
    """round 4: what the library sort hands back is checked to be its input, permuted and in order -- two 64-bit sums over a mix of every
    (hash, slot) pair before and after, and the order directly (count.hpp: pair_sum_kernel).  OATK_DEBUG_SORT_DISTRUST makes the check
    of the sort on the top bits fail: the count must come back with the table of the sort on all 64 bits, not with an error and not with something else."""
    cfg = dict(CONFIGS["config2"])
    cfg["n_reads"] = 220_000
    rs = ReadSet(**cfg)
    seq, off, lens = rs.slice(0, cfg["n_reads"])
    res = {}
    for mode in ("distrust", "full"):
        monkeypatch.delenv("OATK_DEBUG_SORT_DISTRUST", raising=False)
        monkeypatch.setenv("OATK_DEBUG_SORT_DISTRUST" if mode == "distrust" else "OATK_DEBUG_FULL_SORT", "1")
        hip.scan_host(seq, off, lens, K, S)
        hip.count()
        assert hip.info()["n_occ"] >= 1 << 22
        res[mode] = {b: crc(hip.fetch(b)) for b in COUNT_BUFS} | {"n_scm": hip.info()["n_scm"]}
    assert res["distrust"] == res["full"]
