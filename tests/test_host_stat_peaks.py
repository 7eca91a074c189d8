"""CPU: the host peak finder of sr_db_stat (liboatk_host.so: oatk_stat_peaks) against the compiled reference's ha_analyze_count
(syncmer.c:768-864, static -- reached through sr_db_stat on a hand-made read database whose k-mer multiplicities realise a chosen histogram)."""
import ctypes as C

import numpy as np
import pytest

import ref_lib
from oatk_amd import _lib

pytestmark = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref not built (needs /root/reference)")


def reference_peaks(hist):
    """hist[c] distinct k-mers occur c times -> (peak_hom, peak_het) of the reference's sr_db_stat"""
    L = ref_lib.lib()
    L.refx_fake_srdb.restype = C.c_void_p
    L.refx_fake_srdb.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.refx_fake_srdb_smer.argtypes = [C.c_void_p, C.c_void_p]
    L.refx_fake_dbs_free.argtypes = [C.c_void_p, C.c_void_p]
    ids = np.repeat(np.arange(int(hist.sum()), dtype=np.uint64), np.repeat(np.arange(len(hist)), hist))
    n_scm = np.array([len(ids)], np.uint32)
    k_mer = np.ascontiguousarray(ids << np.uint64(1))
    m_pos = np.zeros(len(ids), np.uint32)
    db = L.refx_fake_srdb(1, n_scm.ctypes.data, k_mer.ctypes.data, m_pos.ctypes.data)
    L.refx_fake_srdb_smer(db, ids.ctypes.data)
    o8, od = np.zeros(8, np.int32), np.zeros(5, np.float64)
    L.refx_srdb_stat(db, o8.ctypes.data, od.ctypes.data)
    L.refx_fake_dbs_free(db, None)
    assert (o8[2], o8[3]) == (o8[6], o8[7])           # s-mers were given the k-mers' multiplicities
    return int(o8[6]), int(o8[7])


def host_peaks(hist):
    H = C.CDLL(_lib.HOST_LIB_PATH)
    H.oatk_stat_peaks.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    cnt = np.zeros(1001, np.int64)
    cnt[:len(hist)] = hist
    hom, het = C.c_int(), C.c_int()
    H.oatk_stat_peaks(cnt.ctypes.data, C.byref(hom), C.byref(het))
    return hom.value, het.value


def bumps(rng, n_bins):
    """error tail + one to three bumps of random place, width and height, sometimes flat-topped or touching the last bins"""
    h = np.zeros(n_bins, np.int64)
    x = np.arange(n_bins)
    if rng.random() < 0.8:
        h += (rng.integers(5, 400) * np.exp(-x / rng.uniform(0.5, 6))).astype(np.int64)
    for _ in range(rng.integers(0, 4)):
        mu, sd, top = rng.uniform(3, n_bins + 5), rng.uniform(0.7, n_bins / 6), rng.integers(2, 120)
        h += (top * np.exp(-0.5 * ((x - mu) / sd) ** 2)).astype(np.int64)
    if rng.random() < 0.3:
        h += rng.integers(0, 3, n_bins)
    h[0] = 0
    return h


def test_peak_finder_equals_the_reference():
    rng = np.random.default_rng(768)
    n_checked, outcomes = 0, set()
    for t in range(400):
        n_bins = int(rng.choice([8, 12, 30, 60, 120]))
        h = bumps(rng, n_bins) if t % 4 else rng.integers(0, 6, n_bins)
        h[0] = 0
        if h.sum() == 0:
            continue
        got, want = host_peaks(h), reference_peaks(h)
        assert got == want, (h.tolist(), got, want)
        n_checked += 1
        outcomes.add("low" if want[0] < 0 else "single" if want[1] < 0 else "right" if h[want[1]] > h[want[0]] else "left")
    assert n_checked > 300 and outcomes == {"low", "single", "left", "right"}          # low coverage, no shoulder, left shoulder, right shoulder all occurred


def test_peak_finder_edges():
    for h in ([0, 0, 0, 0, 0, 9, 8, 7, 6, 5, 4],            # only ever falls: low coverage
              [0, 50, 20, 10, 5, 2, 1, 3],                  # rises in the very last bin
              [0, 0, 9, 1, 1, 1, 1, 4, 4, 4, 1, 8, 8, 1],   # flat tops, ties between shoulders
              [0, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 4],      # main peak is the last bin: nothing to its right
              [0, 1, 1, 1, 1, 1, 2, 40, 2, 1, 38, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 39]):  # right shoulder beyond 2.5 x
        h = np.array(h, np.int64)
        assert host_peaks(h) == reference_peaks(h), h.tolist()
