"""GPU parity of the read -> unitig alignment (oatk_hip_read_alignment through liboatk_host.so's oatk_scg_read_alignment) against the
COMPILED REFERENCE's scg_read_alignment (alignment.c:596-691) on the same graph: after unitigging, and through rounds of multiplexing
with the for_unzip filter (run_syncasm.c:209-232).  Bit-exact: which reads align, every alignment's fragments, order, scores."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import adversarial as A
import ref_lib as R
import test_gpu_ec as T
from test_gpu_dropin import device_dbs, host_lib

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")


def setup_libs():
    L, H = R.lib(), host_lib()
    vp = C.c_void_p
    H.oatk_read_error_correction.argtypes = [vp, vp, vp, vp, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp]
    H.oatk_scg_read_alignment.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(vp)]
    L.refx_ra_new.restype = vp
    L.refx_ra_destroy.argtypes = [vp]
    L.refx_read_alignment.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.refx_ra_dims.argtypes = [vp, vp, vp]
    L.refx_ra_flatten.argtypes = [vp] * 9
    L.refx_process_unitigs.argtypes = [vp]
    L.refx_update_utg_cov.argtypes = [vp]
    L.refx_multiplex.argtypes = [vp, vp, C.c_uint32, C.c_double, C.c_double]
    return L, H


def flatten(L, v):
    na, nf = C.c_uint64(), C.c_uint64()
    L.refx_ra_dims(v, C.byref(na), C.byref(nf))
    na, nf = na.value, nf.value
    out = {"sid": np.zeros(na, np.uint64), "n": np.zeros(na, np.uint32), "s": np.zeros(na, np.float64), "uid": np.zeros(nf, np.uint64),
           "u_beg": np.zeros(nf, np.uint64), "u_end": np.zeros(nf, np.uint64), "s_beg": np.zeros(nf, np.uint32), "s_end": np.zeros(nf, np.uint32)}
    L.refx_ra_flatten(v, *[out[k].ctypes.data for k in ("sid", "n", "s", "uid", "u_beg", "u_end", "s_beg", "s_end")])
    return out


def both(L, H, hip, db, g, v_ref, v_dev, for_unzip):
    L.refx_read_alignment(db, v_ref, g, 3, for_unzip)
    want = flatten(L, v_ref)
    # the device routine twice on copies of the previous alignments: writing into its pool (normal) and counting first, then running again
    for two_pass in (1, 0):
        hip._check(hip.L.oatk_hip_debug_align_two_pass(hip.h, two_pass), "oatk_hip_debug_align_two_pass")
        v = v_dev if two_pass == 0 else clone(L, H, v_dev)
        nsk = C.c_uint64(0)
        rc = H.oatk_scg_read_alignment(hip.h, db, v, g, for_unzip, C.byref(nsk), None)
        assert rc == 0, hip.L.oatk_hip_last_error(hip.h)
        assert nsk.value == 0
        got = flatten(L, v)
        for k in want:
            assert len(got[k]) == len(want[k]), (k, two_pass)
            assert np.array_equal(got[k], want[k]), (k, two_pass)
        if two_pass:
            L.refx_ra_destroy(v)
    return want


def clone(L, H, v):
    """a deep copy of a scg_ra_v (the for_unzip filter reads the previous alignments, and the call replaces them)"""
    f = flatten(L, v)
    L.refx_ra_build.restype = C.c_void_p
    L.refx_ra_build.argtypes = [C.c_uint64] + [C.c_void_p] * 8
    return L.refx_ra_build(len(f["sid"]), *[f[k].ctypes.data for k in ("sid", "n", "s", "uid", "u_beg", "u_end", "s_beg", "s_end")])


CASES = [
    (1001, 31, 8, lambda: A.hifi_like(260, 50000, 9000, seed=1010, err=0.0008)),
    (301, 21, 6, lambda: A.hifi_like(300, 30000, 4000, seed=307, err=0.002)),
    (101, 11, 4, lambda: T.diploid_reads(101, 6000, 150, 500, 1200, 0.006)),
    (301, 21, 5, lambda: T.sample_reads(T.genome_with_repeats(7, 25000, unit=1500, copies=4), 320, 4000, 0.003, 8)),
    (101, 11, 5, lambda: T.sample_reads(T.genome_with_repeats(9, 9000, unit=600, copies=3), 300, 1500, 0.004, 10)),
    (1001, 31, 6, lambda: T.sample_reads(T.genome_with_repeats(5, 50000), 260, 9000, 0.001, 6)),
]


@needs_ref
@pytest.mark.parametrize("case", range(len(CASES)))
def test_read_alignment_matches_reference(hip, case):
    K, S, c, mk = CASES[case]
    L, H = setup_libs()
    db, scm = device_dbs(hip, mk(), K, S)
    st = np.zeros(12, np.uint64)
    assert H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, c, 10 * c, c, 0.35, st.ctypes.data) == 0
    g = L.refx_make_graph(db, scm, c, 0.35)                 # run_syncasm.c:138
    assert g
    v_ref, v_dev = L.refx_ra_new(), L.refx_ra_new()
    w = both(L, H, hip, db, g, v_ref, v_dev, 0)             # one syncmer per vertex: every read maps along its own chain
    assert len(w["sid"]) > 10
    L.refx_process_unitigs(g)                               # :160
    w = both(L, H, hip, db, g, v_ref, v_dev, 0)
    assert len(w["sid"]) > 10 and w["n"].max() >= 1
    # the unzip rounds: align with the previous round's filter, update coverage, multiplex (run_syncasm.c:219-232)
    max_n_scm = int(math.ceil(30000.0 / K))
    rounds = 0
    for _ in range(3):
        w = both(L, H, hip, db, g, v_ref, v_dev, 1)
        L.refx_update_utg_cov(g)
        rounds += 1
        if L.refx_multiplex(g, v_ref, max_n_scm, 10.0, 0.3) == 0:
            break
    both(L, H, hip, db, g, v_ref, v_dev, 1)
    both(L, H, hip, db, g, v_ref, v_dev, 0)
    assert rounds >= 1
    L.refx_ra_destroy(v_ref)
    L.refx_ra_destroy(v_dev)                                # the reference's destructor frees what liboatk_host allocated
    L.refx_scg_destroy(g)
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)


GOLDEN = {
    "align_repeats_k301": (301, 21, 5, lambda: T.sample_reads(T.genome_with_repeats(7, 25000, unit=1500, copies=4), 320, 4000, 0.003, 8)),
    "align_diploid_k101": (101, 11, 4, lambda: T.diploid_reads(101, 6000, 150, 500, 1200, 0.006)),
}


@pytest.mark.parametrize("two_pass", [0, 1])
@pytest.mark.parametrize("case", sorted(GOLDEN))
def test_read_alignment_matches_golden(hip, case, two_pass):
    """the committed vectors of the compiled reference (tests/golden/align_*.npz: every stage of its pipeline, multi-mapping reads
    included) through the C ABI: device scan + count + EC must reproduce the golden chains, then every stage's alignments"""
    import align_util as AU
    import golden_util as G
    from oatk_amd import pack_reads
    K, S, c, mk = GOLDEN[case]
    g = G.load(case)
    hip.scan_host(*pack_reads(mk()), K, S)
    hip.count()
    hip.ec_graph()
    hip.ec(0.02, c, 0.35)
    assert np.array_equal(hip.fetch("EC_KMER"), g["k_mer"]) and np.array_equal(hip.fetch("EC_MPOS"), g["m_pos"])
    hip._check(hip.L.oatk_hip_debug_align_two_pass(hip.h, two_pass), "oatk_hip_debug_align_two_pass")
    try:
        multi = 0
        for st in range(int(g["n_stages"])):
            pre = "s%d_" % st
            graph = {k: g[pre + k] for k, _ in AU.GRAPH_FIELDS}
            graph["n_scm"] = int(g["n_scm_table"])
            got = AU.device_align(hip, graph, g[pre + "old_ra"])
            assert got["skipped"] == 0
            AU.assert_same(got, {k: g[pre + "out_" + k] for k in AU.OUT_FIELDS}, (case, st))
            multi += got["n_mapped"] - got["n_unique"]
    finally:
        hip._check(hip.L.oatk_hip_debug_align_two_pass(hip.h, 0), "oatk_hip_debug_align_two_pass")
    assert multi > 0 or case != "align_repeats_k301"


def test_read_alignment_matches_oracle_at_scale_and_reports_reads_over_the_limits(hip):
    """20 k reads against the one-syncmer-per-vertex graph of their own corrected chains (the densest chaining case) vs oracle/align.c;
    then a graph in which one syncmer sits on 200 unitigs: reads that carry it exceed the per-read limits and are reported, not aligned"""
    import align_util as AU
    from oatk_amd import pack_reads
    from oatk_amd.synth import ReadSet
    K, S, c, n = 1001, 31, 30, 20000
    hip.scan_host(*pack_reads(ReadSet(1_000_000, n, 15000).as_list(0, n)), K, S)
    hip.count()
    hip.ec_graph()
    hip.ec(0.02, c, 0.35)
    nv, na = hip.asm_graph(c, 0.35)
    ag = hip.fetch_asm_graph()
    ns = len(ag["scm_del"])
    su_off = np.zeros(ns + 1, np.uint64)
    su_off[1:] = np.cumsum(ag["scm_del"] == 0)
    graph = {"n_scm": ns, "su_off": su_off, "su_uid": np.arange(nv, dtype=np.uint64) << np.uint64(1), "su_pos": np.zeros(nv, np.uint32),
             "utg_n": np.ones(nv, np.uint32), "idx_p": ag["idx_p"], "idx_n": ag["idx_n"].astype(np.uint64), "arc_w": ag["arc_w"],
             "arc_ln": np.zeros(na, np.uint64), "arc_del": np.zeros(na, np.uint8)}
    chains = hip.fetch("EC_N_SCM"), hip.fetch("EC_KMER"), hip.fetch("EC_MPOS")
    got = AU.device_align(hip, graph)
    want = AU.oracle_align(*chains, graph)
    AU.assert_same(got, want)
    assert got["skipped"] == 0 and len(got["sid"]) > 0.9 * n and (got["n_mapped"], got["n_unique"]) == (want["n_mapped"], want["n_unique"])
    # one popular syncmer on 200 (then 3000) extra single-syncmer unitigs: reads that carry it are over the lane-per-read limits (160 hits) and go through
    # the data-parallel routine (align_big.hpp, round 4); with that switched off they are reported, not aligned, and everybody else's alignments are the same
    pop = int(ag["vtx_scm"][0])
    carriers = np.unique(np.repeat(np.arange(n), chains[0])[(chains[1] >> np.uint64(1)) == pop])
    assert len(carriers) > 0
    for extra in (200, 3000):
        su_uid = np.concatenate([graph["su_uid"][:1], (np.arange(nv, nv + extra, dtype=np.uint64) << np.uint64(1)), graph["su_uid"][1:]])
        su_off2 = su_off.copy()
        su_off2[pop + 1:] += np.uint64(extra)
        g2 = dict(graph, su_off=su_off2, su_uid=su_uid, su_pos=np.zeros(nv + extra, np.uint32), utg_n=np.ones(nv + extra, np.uint32),
                  idx_p=np.concatenate([graph["idx_p"], np.zeros(2 * extra, np.uint64)]), idx_n=np.concatenate([graph["idx_n"], np.zeros(2 * extra, np.uint64)]))
        got2 = AU.device_align(hip, g2)
        want2 = AU.oracle_align(*chains, g2)
        assert got2["skipped"] == 0
        AU.assert_same(got2, want2, extra)
        if extra == 200:
            os.environ["OATK_DEBUG_RA_NO_BIG"] = "1"
            try:
                got3 = AU.device_align(hip, g2)
            finally:
                del os.environ["OATK_DEBUG_RA_NO_BIG"]
            assert got3["skipped"] == len(carriers)
            assert np.array_equal(np.sort(hip.fetch("RA_SKIPPED")), carriers.astype(np.uint32))
            keep = ~np.isin(want2["sid"], carriers)
            assert np.array_equal(got3["sid"], want2["sid"][keep]) and np.array_equal(got3["s"], want2["s"][keep])


@pytest.mark.parametrize("seed", range(6))
def test_read_alignment_on_random_graphs_matches_oracle(hip, seed):
    import align_util as AU
    from oatk_amd import pack_reads
    K, S, c = 101, 11, 3
    hip.scan_host(*pack_reads(A.hifi_like(400, 12000, 2500, seed=50 + seed, err=0.002)), K, S)
    hip.count()
    hip.ec_graph()
    hip.ec(0.02, c, 0.35)
    chains = hip.fetch("EC_N_SCM"), hip.fetch("EC_KMER"), hip.fetch("EC_MPOS")
    n_scm = len(hip.fetch("EC_SCM_COV"))
    rng = np.random.default_rng(seed)
    total = multi = 0
    for n_src, max_len, p_overlap, p_noise in ((25, 1, 0.0, 0.2), (25, 4, 0.3, 0.3), (60, 12, 0.5, 0.5), (150, 30, 0.2, 1.0), (12, 2, 0.5, 2.0)):
        graph = AU.random_graph(rng, n_scm, chains, n_src, max_len, p_overlap, p_noise)
        old = np.where(rng.random(len(chains[0])) < 0.8, 1, 0).astype(np.int64) | (rng.integers(0, 6, len(chains[0])).astype(np.int64) << 1)
        for o in (None, old):
            got = AU.device_align(hip, graph, o)
            want = AU.oracle_align(*chains, graph, o)
            if got["skipped"]:                               # reads over the device limits are reported, the rest must agree
                keep = ~np.isin(want["sid"], hip.fetch("RA_SKIPPED"))
                fk = np.repeat(keep, want["n"])
                want = {k: (want[k][keep] if k in ("sid", "n", "s") else want[k][fk]) for k in AU.OUT_FIELDS}
            AU.assert_same(got, want, (seed, n_src, max_len))
            total += len(got["sid"])
            multi += int((np.bincount(got["sid"].astype(np.int64)) > 1).sum()) if len(got["sid"]) else 0
    assert total > 100 and multi > 0
