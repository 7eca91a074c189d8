"""GPU: the LIGHT error-correction graph (include/oatk_hip_ec.h, oatk_hip_ec_graph_light) and the weighted-segment builder under it.

read_error_correction (syncerr.c:819) deletes every syncmer seen fewer than err_mer_c times together with its arcs; of those arcs
find_error_syncmers (:690-718) only asks whether one exists.  The light graph therefore sorts only pairs between two candidates and keeps one
flag per oriented vertex for the rest.  Here: marks, corrected reads, refreshed table and statistics equal those from the full graph (which
test_gpu_ec.py holds against the reference), the kept arcs are exactly the full graph's arcs between candidates, thresholds it cannot serve
are refused, and the run-length compressed form shards exchange (oatk_hip_ec_graph_from_segments) gives the arcs of the expanded pair list --
overlap mode with khashl's tie order included."""
import ctypes as C

import numpy as np
import pytest
import torch

import adversarial as A
import test_gpu_ec as E
from oatk_amd import pack_reads

pytestmark = pytest.mark.gpu

CASES = E.CASES + [
    # a wide band between c and 10 c, where the existence of arcs to rare syncmers decides the mark
    (101, 11, 12, lambda: E.sample_reads(E.genome_with_repeats(9, 9000, unit=600, copies=3), 700, 1500, 0.006, 21)),
    (301, 21, 3, lambda: A.hifi_like(120, 30000, 4000, seed=977, err=0.004)),
]
RES = ["EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER", "EC_SCM_COV", "EC_SCM_DEL", "EC_ERR_DEL", "EC_SCM_OCC", "EC_SCM_OCC_OFF"]


def graph_arrays(hip):
    out = {}
    for k, (which, dt) in E.EG_BUF.items():
        p, b = C.c_void_p(), C.c_uint64()
        hip._check(hip.L.oatk_hip_buffer(hip.h, which, C.byref(p), C.byref(b)), "oatk_hip_buffer(EG %s)" % k)
        out[k] = np.zeros(b.value // np.dtype(dt).itemsize, dtype=dt)
        if b.value:
            hip._check(hip.L.oatk_hip_d2h(hip.h, out[k].ctypes.data, p, b.value), "d2h")
    return out


@pytest.mark.parametrize("case", range(len(CASES)))
def test_light_graph_corrects_like_the_full_graph(hip, case):
    K, S, c, mk = CASES[case]
    reads = mk()
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    cov = hip.fetch_count()["cov"]
    hip.ec_graph()
    full = graph_arrays(hip)
    st_full = hip.ec(0.02, c, 0.35)
    want = {k: E.fetch_ec(hip, k) for k in RES}
    hip.ec_graph(light_c=c)
    light = graph_arrays(hip)
    st_light = hip.ec(0.02, c, 0.35)
    got = {k: E.fetch_ec(hip, k) for k in RES}
    for k in RES:
        assert np.array_equal(got[k], want[k]), k
    assert st_light.tolist() == st_full.tolist()
    # the kept arcs: the full graph's arcs with both ends at or above c, in the same order
    keep = (cov[(full["arc_v"] >> 1).astype(np.int64)] >= c) & (cov[(full["arc_w"] >> 1).astype(np.int64)] >= c)
    assert 0 < keep.sum() < len(keep)
    for k in ("arc_v", "arc_w", "arc_ls", "arc_cov", "arc_comp"):
        assert np.array_equal(light[k], full[k][keep]), k
    assert int(st_full[0] + st_full[5] + st_full[10]) > 0


def test_light_graph_refuses_thresholds_it_cannot_serve(hip):
    reads = A.hifi_like(200, 30000, 4000, seed=307, err=0.004)
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, 301, 21)
    hip.count()
    hip.ec_graph(light_c=6)
    L = hip.L
    assert L.oatk_hip_ec(hip.h, None, 0.02, 5, 60, 5, 0.35) == 2 and b"light" in L.oatk_hip_last_error(hip.h)      # below the graph's coverage
    hip.ec_graph(light_c=6)
    assert L.oatk_hip_ec(hip.h, None, 0.02, 6, 60, 4, 0.35) == 2                                                     # arcs rarer than syncmers
    hip.ec_graph(light_c=6)
    assert L.oatk_hip_ec(hip.h, None, 0.02, 8, 80, 9, 0.35) == 0                                                      # stricter is fine
    a = E.fetch_ec(hip, "EC_KMER")
    hip.ec_graph()
    assert L.oatk_hip_ec(hip.h, None, 0.02, 8, 80, 9, 0.35) == 0
    assert np.array_equal(a, E.fetch_ec(hip, "EC_KMER"))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_segments_give_the_arcs_of_the_expanded_pairs(hip, seed):
    """random pair lists in which an arc meets many distances in many orders, cut into random segments"""
    reads = A.hifi_like(60, 20000, 3000, seed=5, err=0.0)
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, 301, 21)
    hip.count()
    nv = int(hip.fetch_count()["n_scm"])
    rng = np.random.default_rng(seed)
    n_keys = 400
    v0 = rng.integers(0, 2 * nv - 1, size=n_keys)
    v1 = np.array([rng.integers(a, 2 * nv) for a in v0])          # canonical: v0 <= v1
    ok = (v1 ^ 1) != v0                                              # (a key that is its own complement is fine too, keep some)
    keys_u = (v0.astype(np.uint64) << np.uint64(32) | v1.astype(np.uint64))
    c0, c1 = (v1 ^ 1).astype(np.uint64), (v0 ^ 1).astype(np.uint64)         # the complementary arc; were it a key too, the graph would hold the arc twice
    comp = c0 << np.uint64(32) | c1
    keys_u = np.unique(np.where((c0 <= c1) & (comp < keys_u), comp, keys_u))
    seg_k, seg_d, seg_w = [], [], []
    for k in keys_u:
        kind = int(rng.integers(0, 5))
        n_dist = [1, 2, 3, 7, 40][kind]
        pool = rng.choice(np.arange(40, 400), size=n_dist, replace=False)
        n_seg = int(rng.integers(1, [4, 6, 12, 60, 200][kind]))
        for _ in range(n_seg):
            seg_k.append(k)
            seg_d.append(int(rng.choice(pool)))
            seg_w.append(int(rng.choice([1, 1, 1, 2, 3, 9, 50])))
    # shuffle whole segments between keys (the builder sorts by key, stably), keep the order inside a key
    order = np.argsort(rng.integers(0, 8, size=len(seg_k)), kind="stable")
    seg_k, seg_d, seg_w = np.array(seg_k, dtype=np.uint64)[order], np.array(seg_d, dtype=np.uint64)[order], np.array(seg_w, dtype=np.uint64)[order]
    pk, pd = np.repeat(seg_k, seg_w.astype(np.int64)), np.repeat(seg_d, seg_w.astype(np.int64)).astype(np.uint32)
    dev = torch.device("cuda", 0)
    t_pk, t_pd = torch.from_numpy(pk.view(np.int64)).to(dev), torch.from_numpy(pd.view(np.int32)).to(dev)
    hip.ec_graph_from_pairs(t_pk.data_ptr(), t_pd.data_ptr(), len(pk))
    want = graph_arrays(hip)
    val = seg_d | seg_w << np.uint64(32)
    t_sk, t_sv = torch.from_numpy(seg_k.view(np.int64)).to(dev), torch.from_numpy(val.view(np.int64)).to(dev)
    hip._check(hip.L.oatk_hip_ec_graph_from_segments(hip.h, t_sk.data_ptr(), t_sv.data_ptr(), len(seg_k)), "oatk_hip_ec_graph_from_segments")
    got = graph_arrays(hip)
    assert len(want["arc_v"]) >= len(keys_u) and ok.any()
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert len(np.unique(want["arc_ls"])) > 20


def test_light_graph_with_no_candidate_at_all(hip):
    """a threshold nothing reaches: no pair is kept, no arc exists, every syncmer is deleted, no read changes -- as with the full graph"""
    reads = A.hifi_like(150, 30000, 4000, seed=353, err=0.003)
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, 301, 21)
    hip.count()
    hip.ec_graph()
    st_full = hip.ec(0.02, 5000, 0.35)
    want = {k: E.fetch_ec(hip, k) for k in RES}
    hip.ec_graph(light_c=5000)
    assert len(graph_arrays(hip)["arc_v"]) == 0
    st_light = hip.ec(0.02, 5000, 0.35)
    for k in RES:
        assert np.array_equal(E.fetch_ec(hip, k), want[k]), k
    assert st_light.tolist() == st_full.tolist() and int(st_full[:11].sum()) == 0
