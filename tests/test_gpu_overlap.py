"""GPU parity of the pair-distance tables (oatk_hip_overlap_hist) and of what is served from them through liboatk_host.so:
calc_syncmer_overlap's most frequent distance with the reference's tie-break, and scg_unitig_consensus (syncasm.c:1004-1046) -- the
strings of real unitigs of the compiled reference's graph, hoco and base space, both orientations."""
import ctypes as C

import numpy as np
import pytest

import adversarial as A
import ref_lib as R
import test_gpu_ec as T
from oatk_amd import pack_reads
from test_gpu_dropin import _KString, device_dbs, host_lib

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")


def tables_from_chains(n_scm, k_mer, m_pos):
    """calc_syncmer_overlap's tabulation restated over the chains: canonical pair -> distinct distances in (read, slot) order, counts,
    and whether the last addition was a repeat; pairs with a corrected member (low bit of k_mer) are left out (syncasm.c:499, :511)"""
    out = {}
    o = 0
    for n in n_scm.tolist():
        for j in range(1, n):
            a, b = int(k_mer[o + j - 1]), int(k_mer[o + j])
            if (a | b) & 1:
                continue
            v0 = (a >> 1) << 1 | (int(m_pos[o + j - 1]) & 1)
            v1 = (b >> 1) << 1 | (int(m_pos[o + j]) & 1)
            key = v0 << 32 | v1 if v0 <= v1 else (v1 ^ 1) << 32 | (v0 ^ 1)
            d = (int(m_pos[o + j]) >> 1) - (int(m_pos[o + j - 1]) >> 1)
            t = out.setdefault(key, [[], {}, False])
            if d in t[1]:
                t[1][d] += 1
                t[2] = True
            else:
                t[0].append(d)
                t[1][d] = 1
                t[2] = False
        o += n
    return out


def check_tables(hip, n_scm, k_mer, m_pos):
    np_, ne = hip.overlap_hist()
    key, off, dist, cnt, tail = (hip.fetch("OVL_" + x) for x in ("KEY", "OFF", "DIST", "CNT", "TAIL"))
    want = tables_from_chains(n_scm, k_mer, m_pos)
    assert np_ == len(want) == len(key) and ne == len(dist) and np.all(key[1:] > key[:-1])
    for i, k in enumerate(key.tolist()):
        order, counts, rep = want[k]
        lo, hi = int(off[i]), int(off[i + 1])
        assert dist[lo:hi].tolist() == order, hex(k)
        assert cnt[lo:hi].tolist() == [counts[d] for d in order]
        assert bool(tail[i]) == rep
    return np_


@pytest.mark.parametrize("K,S,c,err,with_ec", [(101, 11, 4, 0.006, False), (101, 11, 4, 0.006, True), (301, 21, 5, 0.003, True), (1001, 31, 6, 0.0008, True)])
def test_pair_tables_equal_the_walk_over_the_chains(hip, K, S, c, err, with_ec):
    reads = A.hifi_like(260, 30 * K, 6 * K, seed=K + 1, err=err)
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    if with_ec:
        hip.ec_graph()
        hip.ec(0.02, c, 0.35)
        chains = hip.fetch("EC_N_SCM"), hip.fetch("EC_KMER"), hip.fetch("EC_MPOS")
        assert (chains[1] & 1).sum() > 0
    else:
        chains = hip.fetch("N_SCM"), hip.fetch("POS_KID"), hip.fetch("POS_MPOS")
    assert check_tables(hip, *chains) > 50
    # no reads with two syncmers: no pairs
    hip.scan_host(*pack_reads([b"ACGT" * 10]), K, S)
    hip.count()
    assert hip.overlap_hist() == (0, 0)


@needs_ref
@pytest.mark.parametrize("K,S,c,mk", [
    (101, 11, 4, lambda: T.diploid_reads(101, 6000, 150, 500, 1200, 0.006)),
    (301, 21, 5, lambda: T.sample_reads(T.genome_with_repeats(7, 25000, unit=1500, copies=4), 320, 4000, 0.003, 8)),
    (1001, 31, 6, lambda: A.hifi_like(200, 40000, 9000, seed=1009, err=0.0005)),
    (101, 11, 5, lambda: T.sample_reads(T.genome_with_repeats(9, 9000, unit=600, copies=3), 300, 1500, 0.004, 10)),
])
def test_unitig_consensus_served_from_the_device(hip, K, S, c, mk):
    """the reference builds its graph and unitigs on structs filled by the device; every unitig's sequence from oatk_scg_unitig_consensus
    (pair tables + run-length totals from the MI355X, table replay and string assembly on the host) equals scg_unitig_consensus"""
    L, H = R.lib(), host_lib()
    H.oatk_read_error_correction.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_double, C.c_void_p]
    H.oatk_consensus_fetch.restype = C.c_void_p
    H.oatk_consensus_fetch.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
    H.oatk_consensus_destroy.argtypes = [C.c_void_p]
    H.oatk_overlap_fetch.restype = C.c_void_p
    H.oatk_overlap_fetch.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    H.oatk_overlap_destroy.argtypes = [C.c_void_p]
    H.oatk_scg_unitig_consensus.restype = C.c_int64
    H.oatk_scg_unitig_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(_KString), C.c_int]
    H.oatk_calc_syncmer_overlap.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.refx_utg_lists.restype = C.c_uint64
    L.refx_utg_lists.argtypes = [C.c_void_p] * 4
    L.refx_unitig_consensus.restype = C.c_int64
    L.refx_unitig_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_char_p, C.c_int64]
    L.refx_process_unitigs.argtypes = [C.c_void_p]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    db, scm = device_dbs(hip, mk(), K, S)
    st = np.zeros(12, np.uint64)
    assert H.oatk_read_error_correction(hip.h, db, scm, None, 0.02, c, 10 * c, c, 0.35, st.ctypes.data) == 0
    rc = C.c_int(0)
    cs = H.oatk_consensus_fetch(hip.h, c, K, C.byref(rc))
    assert rc.value == 0 and cs
    ov = H.oatk_overlap_fetch(hip.h, C.byref(rc))
    assert rc.value == 0 and ov
    g = L.refx_make_graph(db, scm, c, 0.35)                 # run_syncasm.c:138
    L.refx_process_unitigs(g)                               # :160
    import ec_util as E
    nv = E.flatten_graph(g)["n_vtx"]
    n = np.zeros(nv, np.uint64)
    de = np.zeros(nv, np.uint8)
    tot = L.refx_utg_lists(g, n.ctypes.data, de.ctypes.data, None)
    a = np.zeros(max(tot, 1), np.uint64)
    L.refx_utg_lists(g, None, None, a.ctypes.data)
    starts = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    buf = C.create_string_buffer(1 << 22)
    checked = longest = 0
    for u in range(nv):
        if de[u]:
            continue
        v = np.ascontiguousarray(a[starts[u]:starts[u + 1]])
        for lst in (v, np.ascontiguousarray(v[::-1] ^ np.uint64(1))):          # the unitig and its reverse complement
            for hoco in (0, 1):
                ks = _KString(0, 0, None)
                l = H.oatk_scg_unitig_consensus(cs, ov, db, lst.ctypes.data, len(lst), C.byref(ks), hoco)
                got = C.string_at(ks.s, ks.l) if ks.l else b""
                libc.free(ks.s)
                lr = L.refx_unitig_consensus(db, g, lst.ctypes.data, len(lst), hoco, buf, len(buf))
                assert l == lr and got == buf.raw[:lr], (u, hoco, len(lst))
                checked += 1
        longest = max(longest, len(v))
    assert checked >= 4 and longest >= 3
    # a pair that is adjacent on no read: empty table, distance 0 (syncasm.c:555-571 with nothing counted)
    assert H.oatk_calc_syncmer_overlap(ov, (1 << 30) + 2, (1 << 30) + 8, None) == 0
    H.oatk_overlap_destroy(ov)
    H.oatk_consensus_destroy(cs)
    L.refx_scg_destroy(g)
    L.refx_scmdb_destroy(scm)
    L.refx_srdb_destroy(db)


def test_tables_and_arc_overlaps_on_injected_walks(hip):
    """Two syncmers adjacent on reads overlap, so real data gives a pair one distance (outside tandem repeats) and the multi-distance paths of
    ovh_kernel and egr_mode_kernel would go untested.  The sharded entry points take (key, distance) lists from outside: random walks full of
    ties and table growth, interleaved across pairs -- tables against the walks themselves, arc overlaps against the reference's own
    khashl.h fed the same walks (oracle/ref_shim.c: refx_kh_mode)."""
    from test_host_overlap import reduce_walk
    K, S = 101, 11
    hip.scan_host(*pack_reads(A.hifi_like(200, 8000, 2000, seed=9, err=0.001)), K, S)
    hip.count()
    n_scm = hip.info()["n_scm"]
    rng = np.random.default_rng(3)
    walks, keys = [], []
    for i in range(min(n_scm // 2 - 1, 1500)):
        nd = int(rng.integers(1, 30)) if i % 97 else int(rng.integers(1, 8))
        vals = rng.integers(1, K - 1, nd)
        seq = vals[rng.integers(0, nd, int(rng.integers(1, 5 * nd + 2)))]
        if rng.random() < 0.5:
            seq = np.concatenate([seq, seq])
            rng.shuffle(seq)
        if i % 97 == 0:
            seq = np.concatenate([seq, rng.choice(np.arange(1, K - 1), 40, replace=False)[rng.integers(0, 40, 300)]])   # long runs: the wave-wide path of egr_mode_kernel; <= 40 + 30 > 48 distinct values would be refused
        walks.append(seq.astype(np.uint32))
        v, w = 2 * (2 * i) + int(rng.integers(0, 2)), 2 * (2 * i + 1) + int(rng.integers(0, 2))
        keys.append(v << 32 | w)                                            # v < w: canonical as it stands
    # interleave the pairs' entries the way reads would, keeping every pair's own order
    owner = np.repeat(np.arange(len(walks)), [len(w) for w in walks])
    rng.shuffle(owner)
    cursor = np.zeros(len(walks), np.int64)
    k_all, d_all = np.zeros(len(owner), np.uint64), np.zeros(len(owner), np.uint32)
    for t, o in enumerate(owner.tolist()):
        k_all[t], d_all[t] = keys[o], walks[o][cursor[o]]
        cursor[o] += 1
    # device copies through the HIP runtime the library itself runs on (torch's own copy of it does not initialise once that one is up)
    rt = C.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]

    class _Dev:
        def __init__(self, a):
            self.p = C.c_void_p()
            assert rt.hipMalloc(C.byref(self.p), a.nbytes) == 0 and rt.hipMemcpy(self.p, a.ctypes.data, a.nbytes, 1) == 0

        def data_ptr(self):
            return self.p.value

    dk, dd = _Dev(k_all), _Dev(d_all)
    # --- tables
    np_, ne = hip.overlap_hist_from_pairs(dk.data_ptr(), dd.data_ptr(), len(owner))
    okey, ooff, odist, ocnt, otail = (hip.fetch("OVL_" + x) for x in ("KEY", "OFF", "DIST", "CNT", "TAIL"))
    order = np.argsort(np.array(keys, np.uint64))
    assert np_ == len(walks) and np.array_equal(okey, np.array(keys, np.uint64)[order])
    many = 0
    for j, i in enumerate(order.tolist()):
        want = reduce_walk(walks[i].tolist())
        lo, hi = int(ooff[j]), int(ooff[j + 1])
        assert odist[lo:hi].tolist() == want[0] and ocnt[lo:hi].tolist() == want[1] and bool(otail[j]) == want[2], i
        many += len(want[0]) > 1
    assert many > 100 and ne == len(odist)
    # --- arc overlaps of the EC graph built from the same lists
    if not R.available():
        rt.hipFree(dk.p), rt.hipFree(dd.p)
        return
    L = R.lib()
    L.refx_kh_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    hip.ec_graph_from_pairs(dk.data_ptr(), dd.data_ptr(), len(owner))
    av, aw, als, comp = hip.fetch("EG_ARC_V"), hip.fetch("EG_ARC_W"), hip.fetch("EG_ARC_LS"), hip.fetch("EG_ARC_COMP")
    by_key = {k: i for i, k in enumerate(keys)}
    checked = tied = 0
    for v, w, ls, cp in zip(av.tolist(), aw.tolist(), als.tolist(), comp.tolist()):
        key = v << 32 | w if v <= w else (w ^ 1) << 32 | (v ^ 1)
        seq = walks[by_key[key]].astype(np.int32)
        movl = L.refx_kh_mode(None, seq.ctypes.data, len(seq))
        assert ls == (K - movl if movl < K else 0), (v, w, movl, ls)
        c = sorted(np.unique(seq, return_counts=True)[1].tolist(), reverse=True)
        tied += len(c) > 1 and c[0] == c[1]
        checked += 1
    assert checked == 2 * len(walks) and tied > 20            # the bucket-order tie-break decided many of these
    rt.hipFree(dk.p), rt.hipFree(dd.p)
