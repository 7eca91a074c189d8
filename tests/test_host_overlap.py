"""CPU: the host side of calc_syncmer_overlap (liboatk_host.so: the khashl<int,int> replica and the lookup of host/ovl_host.c) against the
arc overlaps of the compiled reference's EC graphs (tests/golden/ec_*.npz: make_syncmer_graph + scg_consensus(hoco), khashl tie order and
all).  The pair tables are built here from the golden chains the way the device builds them (first-appearance order, counts, trailing-repeat
flag); no GPU involved."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G
from oatk_amd import _lib
from test_gpu_overlap import tables_from_chains

EC_CASES = ["ec_diploid_k101", "ec_repeats_k301", "ec_hifi_k1001"]


class OverlapT(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_entries", C.c_uint64), ("key", C.c_void_p), ("off", C.c_void_p), ("dist", C.c_void_p),
                ("cnt", C.c_void_p), ("tail", C.c_void_p)]


@pytest.mark.parametrize("case", EC_CASES)
def test_host_replay_reproduces_the_references_arc_overlaps(case):
    g = G.load(case)
    K = int(g["K"])
    tabs = tables_from_chains(g["in_n_scm"], g["in_k_mer"], g["in_m_pos"])
    keys = np.array(sorted(tabs), np.uint64)
    off = np.zeros(len(keys) + 1, np.uint64)
    dist, cnt, tail = [], [], np.zeros(len(keys), np.uint8)
    for i, k in enumerate(keys.tolist()):
        order, counts, rep = tabs[k]
        dist += order
        cnt += [counts[d] for d in order]
        off[i + 1] = len(dist)
        tail[i] = rep
    dist, cnt = np.array(dist, np.int32), np.array(cnt, np.uint32)
    ov = OverlapT(len(keys), len(dist), keys.ctypes.data, off.ctypes.data, dist.ctypes.data, cnt.ctypes.data, tail.ctypes.data)
    H = C.CDLL(_lib.HOST_LIB_PATH)
    H.oatk_calc_syncmer_overlap.argtypes = [C.POINTER(OverlapT), C.c_uint64, C.c_uint64, C.c_void_p]
    H.oatk_ovl_table_new.restype = C.c_void_p
    H.oatk_ovl_table_destroy.argtypes = [C.c_void_p]
    na = int(g["g_idx_n"].sum())
    av, aw, als = g["g_arc_v"][:na], g["g_arc_w"][:na], g["g_arc_ls"][:na]
    ties = 0
    for v, w, ls in zip(av.tolist(), aw.tolist(), als.tolist()):
        movl = H.oatk_calc_syncmer_overlap(C.byref(ov), v, w, None)          # a fresh table, as scg_consensus uses for arcs (syncasm.c:803)
        want = K if movl < 0 else (K - movl if movl < K else 0)              # scg_syncmer_consensus(beg = movl) capped by the vertex length
        assert want == ls, (v, w, movl, ls)
        key = v << 32 | w if v <= w else (w ^ 1) << 32 | (v ^ 1)
        ties += len(tabs[key][1]) > 1
    assert na > 100 and ties == 0       # two syncmers adjacent on reads overlap, so their distance is the same on every read outside tandem repeats:
    #                                     ties and table growth are exercised on random walks below, against the reference's own khashl.h
    # the same answers from a table that is kept across calls only where its size happens not to matter: just exercise the path
    h = H.oatk_ovl_table_new()
    for v, w in list(zip(av.tolist(), aw.tolist()))[:50]:
        H.oatk_calc_syncmer_overlap(C.byref(ov), v, w, h)
    H.oatk_ovl_table_destroy(h)
    assert H.oatk_calc_syncmer_overlap(C.byref(ov), 1 << 30, (1 << 30) + 6, None) == 0


def reduce_walk(seq):
    """what the device hands over for one pair: distinct distances in first-appearance order, counts, 'the last call was a repeat'"""
    order, counts = [], {}
    for d in seq:
        if d in counts:
            counts[d] += 1
        else:
            counts[d] = 1
            order.append(d)
    return order, [counts[d] for d in order], bool(seq) and counts[seq[-1]] > 1


def test_host_replay_equals_the_references_khashl_on_random_walks():
    """the claim behind the pair tables: first-appearance order + counts + one flag reproduce the table the reference's walk leaves behind, so
    the most frequent distance comes out with the same tie-break -- checked against the reference's own khashl.h (instantiated in the shim
    as syncasm.c:63 does) on random walks full of ties and table growth, with a fresh table and with one table kept across many pairs"""
    import ref_lib as R
    if not R.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    L, H = R.lib(), C.CDLL(_lib.HOST_LIB_PATH)
    L.refx_kh_new.restype = C.c_void_p
    L.refx_kh_free.argtypes = [C.c_void_p]
    L.refx_kh_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    H.oatk_calc_syncmer_overlap.argtypes = [C.POINTER(OverlapT), C.c_uint64, C.c_uint64, C.c_void_p]
    H.oatk_ovl_table_new.restype = C.c_void_p
    H.oatk_ovl_table_destroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(11)
    walks = []
    for _ in range(3000):
        n_distinct = int(rng.integers(1, 40))
        vals = rng.integers(1, 3000, n_distinct)
        n = int(rng.integers(1, 4 * n_distinct + 2))
        seq = vals[rng.integers(0, n_distinct, n)].astype(np.int32)
        if rng.random() < 0.5:                                                 # force ties for the maximum
            seq = np.concatenate([seq, seq]).astype(np.int32)
            rng.shuffle(seq)
        walks.append(seq)
    walks += [np.array([5], np.int32), np.array([7, 7, 7], np.int32), np.array([1, 2, 3, 3, 2, 1], np.int32), np.arange(1, 200, dtype=np.int32)]
    # one pair per walk: keys (2 i, 2 i + 2^20) are canonical as they stand
    keys = np.array([(2 * i) << 32 | (2 * i + (1 << 20)) for i in range(len(walks))], np.uint64)
    red = [reduce_walk(w.tolist()) for w in walks]
    off = np.zeros(len(walks) + 1, np.uint64)
    off[1:] = np.cumsum([len(r[0]) for r in red])
    dist = np.array([d for r in red for d in r[0]], np.int32)
    cnt = np.array([c for r in red for c in r[1]], np.uint32)
    tail = np.array([r[2] for r in red], np.uint8)
    ov = OverlapT(len(keys), len(dist), keys.ctypes.data, off.ctypes.data, dist.ctypes.data, cnt.ctypes.data, tail.ctypes.data)
    hr, hh = L.refx_kh_new(), H.oatk_ovl_table_new()
    differ = 0
    for i, w in enumerate(walks):
        v, x = 2 * i, 2 * i + (1 << 20)
        fresh_ref = L.refx_kh_mode(None, w.ctypes.data, len(w))
        assert H.oatk_calc_syncmer_overlap(C.byref(ov), v, x, None) == fresh_ref, i
        kept_ref = L.refx_kh_mode(hr, w.ctypes.data, len(w))               # the table keeps the size its history gave it
        assert H.oatk_calc_syncmer_overlap(C.byref(ov), v, x, hh) == kept_ref, i
        differ += kept_ref != fresh_ref
    assert differ > 0                                                          # the history did change answers: the kept-table path is not vacuous
    L.refx_kh_free(hr)
    H.oatk_ovl_table_destroy(hh)
