"""GPU parity of the device error correction (oatk_hip_ec) against the COMPILED REFERENCE's read_error_correction
(syncerr.c:819) on the same databases and the same EC graph.  Bit-exact: corrected chains, refreshed syncmer table,
error-syncmer marks and the block statistics."""
import ctypes as C
import os

import numpy as np
import pytest

import adversarial as A
import ec_util as E
import ref_lib as R
from test_gpu_dropin import device_dbs

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")]


class EcGraphT(C.Structure):
    _fields_ = [("n_vtx", C.c_uint64), ("n_arc", C.c_uint64), ("idx_p", C.c_void_p), ("idx_n", C.c_void_p), ("arc_v", C.c_void_p),
                ("arc_w", C.c_void_p), ("arc_ls", C.c_void_p), ("arc_cov", C.c_void_p), ("arc_del", C.c_void_p)]


EC_BUF = {"EC_N_SCM": (100, np.uint32), "EC_SCM_OFF": (101, np.uint64), "EC_KMER": (102, np.uint64), "EC_MPOS": (103, np.uint32),
          "EC_SMER": (104, np.uint64), "EC_SCM_COV": (105, np.uint32), "EC_SCM_DEL": (106, np.uint8), "EC_SCM_OCC_OFF": (107, np.uint64),
          "EC_SCM_OCC": (108, np.uint64), "EC_ERR_DEL": (109, np.uint8)}


def fetch_ec(hip, name):
    which, dt = EC_BUF[name]
    p, b = C.c_void_p(), C.c_uint64()
    hip._check(hip.L.oatk_hip_buffer(hip.h, which, C.byref(p), C.byref(b)), "oatk_hip_buffer(%s)" % name)
    out = np.zeros(b.value // np.dtype(dt).itemsize, dtype=dt)
    if b.value:
        hip._check(hip.L.oatk_hip_d2h(hip.h, out.ctypes.data, p, b.value), "d2h")
    return out


EG_BUF = {"idx_p": (120, np.uint64), "idx_n": (121, np.uint32), "arc_v": (122, np.uint64), "arc_w": (123, np.uint64),
          "arc_ls": (124, np.uint32), "arc_cov": (125, np.uint32), "arc_comp": (126, np.uint8)}


def device_graph(hip):
    """oatk_hip_ec_graph on the resident scan + count; returns the resident graph as numpy arrays"""
    L = hip.L
    L.oatk_hip_ec_graph.argtypes = [C.c_void_p]
    hip._check(L.oatk_hip_ec_graph(hip.h), "oatk_hip_ec_graph")
    out = {}
    for k, (which, dt) in EG_BUF.items():
        p, b = C.c_void_p(), C.c_uint64()
        hip._check(L.oatk_hip_buffer(hip.h, which, C.byref(p), C.byref(b)), "oatk_hip_buffer(EG %s)" % k)
        out[k] = np.zeros(b.value // np.dtype(dt).itemsize, dtype=dt)
        if b.value:
            hip._check(L.oatk_hip_d2h(hip.h, out[k].ctypes.data, p, b.value), "d2h")
    return out


def assert_graph_equal(D, G):
    na = G["n_arc"]
    assert len(D["arc_v"]) == na and len(D["idx_n"]) == 2 * G["n_vtx"]
    for k in ("arc_v", "arc_w", "arc_cov", "arc_comp", "arc_ls"):
        assert np.array_equal(D[k].astype(np.uint64), G[k][:na].astype(np.uint64)), k
    assert np.array_equal(D["idx_n"].astype(np.uint64), G["idx_n"])
    has = G["idx_n"] > 0
    assert np.array_equal(D["idx_p"][has], G["idx_p"][has])


def device_ec(hip, G, max_edist, c, a):
    """G = flattened host graph, or None to correct against the graph oatk_hip_ec_graph left resident"""
    L = hip.L
    L.oatk_hip_ec.argtypes = [C.c_void_p, C.POINTER(EcGraphT), C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double]
    L.oatk_hip_ec_stats.argtypes = [C.c_void_p, C.c_void_p]
    if G is None:
        hip._check(L.oatk_hip_ec(hip.h, None, max_edist, c, 10 * c, c, a), "oatk_hip_ec")
    else:
        g = EcGraphT(G["n_vtx"], G["n_arc"], G["idx_p"].ctypes.data, G["idx_n"].ctypes.data, G["arc_v"].ctypes.data, G["arc_w"].ctypes.data,
                     G["arc_ls"].ctypes.data, G["arc_cov"].ctypes.data, G["arc_del"].ctypes.data)
        hip._check(L.oatk_hip_ec(hip.h, C.byref(g), max_edist, c, 10 * c, c, a), "oatk_hip_ec")
    st = np.zeros(12, np.uint64)
    hip._check(L.oatk_hip_ec_stats(hip.h, st.ctypes.data), "oatk_hip_ec_stats")
    return st


class _H:                       # gives the ref_lib helpers something with a .handle
    def __init__(self, h):
        self.handle = h


def genome_with_repeats(seed, n, unit=3000, copies=3):
    """random genome with a few near-identical repeat copies: the DFS then meets ambiguous paths"""
    rng = np.random.default_rng(seed)
    g = bytearray(A.rand_dna(rng, n))
    rep = bytearray(A.rand_dna(rng, unit))
    for c in range(copies):
        at = (c + 1) * n // (copies + 1)
        r2 = bytearray(rep)
        for _ in range(c):       # copy c differs from the original in c positions
            p = int(rng.integers(0, unit))
            r2[p] = b"ACGT"[(b"ACGT".index(bytes([r2[p]])) + 1) & 3]
        g[at:at + unit] = r2
    return bytes(g)


def sample_reads(genome, n_reads, mean_len, err, seed):
    rng = np.random.default_rng(seed)
    gg = genome + genome
    out = []
    for _ in range(n_reads):
        ln = int(np.clip(rng.normal(mean_len, 0.1 * mean_len), 500, min(len(genome), 2 * mean_len)))
        st = int(rng.integers(0, len(genome)))
        r = bytearray(gg[st:st + ln])
        for p in sorted(rng.integers(0, ln, size=rng.binomial(ln, err)).tolist(), reverse=True):
            kind = int(rng.integers(0, 3))
            if kind == 0:
                r[p] = b"ACGT"[(b"ACGT".index(bytes([r[p]])) + 1 + int(rng.integers(0, 3))) & 3]
            elif kind == 1:
                r.insert(p, b"ACGT"[int(rng.integers(0, 4))])
            else:
                del r[p]
        r = bytes(r)
        out.append(A.revcomp(r) if rng.integers(0, 2) else r)
    return out


def diploid_reads(K, n, snp_every, n_reads, mean_len, err):
    """two haplotypes with a SNP every `snp_every` bases: bubbles in the graph, so some blocks end ambiguous (EC_AMBISEQ)"""
    rng = np.random.default_rng(K)
    h1 = bytearray(A.rand_dna(rng, n))
    h2 = bytearray(h1)
    for p in range(snp_every // 2, n, snp_every):
        h2[p] = b"ACGT"[(b"ACGT".index(bytes([h2[p]])) + 1 + int(rng.integers(0, 3))) & 3]
    return sample_reads(bytes(h1), n_reads // 2, mean_len, err, 1) + sample_reads(bytes(h2), n_reads // 2, mean_len, err, 2)


CASES = [
    # K, S, min cov, reads
    (101, 11, 4, lambda: diploid_reads(101, 6000, 150, 500, 1200, 0.006)),
    (101, 11, 3, lambda: diploid_reads(101, 5000, 90, 600, 1000, 0.01)),
    (1001, 31, 8, lambda: A.hifi_like(200, 40000, 9000, seed=1009, err=0.0005)),
    (301, 21, 6, lambda: A.hifi_like(300, 30000, 4000, seed=307, err=0.004)),
    (1001, 31, 6, lambda: sample_reads(genome_with_repeats(5, 50000), 260, 9000, 0.001, 6)),
    (301, 21, 5, lambda: sample_reads(genome_with_repeats(7, 25000, unit=1500, copies=4), 320, 4000, 0.003, 8)),
    (101, 11, 5, lambda: sample_reads(genome_with_repeats(9, 9000, unit=600, copies=3), 300, 1500, 0.004, 10)),
    # chains longer than a wave (about a hundred syncmers per read), mixed with short ones: the lane-serial walk beside the wave-per-read one
    (101, 11, 5, lambda: A.hifi_like(150, 20000, 6000, seed=463, err=0.003) + A.hifi_like(150, 20000, 1500, seed=467, err=0.003)),
]


@pytest.mark.parametrize("graph", ["host", "device", "device-tiers", "device-tiers-serial", "device-heavy", "device-heavy-mix", "device-heavy-spill", "device-fused", "device-fused-spill", "device-fused-nocert", "device-fused-nw2", "device-fused-nw4", "device-fused-nw8", "device-fused-nw16"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_device_ec_matches_reference(hip, case, graph, monkeypatch):
    # the table test for long arcs (ec_fused.hpp CERT) is on by default since round 6; device-fused-nocert: without.  device-fused-nw4 / 8 / 16: the second stage's narrowest
    # class of workgroups has that many waves (by default the classes are one wave -- ec_heavy.hpp without a budget -- and 4, 8 and 16 waves of ec_fused.hpp by the block's band, and these cases' blocks all fit one wave or four)
    monkeypatch.setenv("OATK_DEBUG_EC_CERT", "0" if graph == "device-fused-nocert" else "1")
    monkeypatch.setenv("OATK_DEBUG_EC_FUSED_MIN_NW", graph[len("device-fused-nw"):] if graph.startswith("device-fused-nw") else "0")
    K, S, c, mk = CASES[case]
    reads = mk()
    # tiny first tier: most blocks run in the classes behind it -- routed there by length and run beside the first tier, or left over by it.
    #   device-tiers[-serial]: the tiers of round 4 (one wave per block, larger LDS carve-ups, HBM slabs: OATK_DEBUG_EC_HEAVY=0)
    #   device-heavy:          everything longer than 48 bases in the first class of the workgroup solver (ec_heavy.hpp, one diagonal per lane)
    #   device-heavy-mix:      48 < l <= 160 in the first class, <= 400 in the second (two diagonals per lane), the rest in the third (six)
    #   device-heavy-spill:    as device-heavy with an LDS frame arena of 64 bytes: every DFS frame goes to the HBM slab
    t0, t1 = (48, 160) if graph.startswith("device-tiers") or graph == "device-heavy-mix" else ((48, 0) if graph.startswith("device-heavy") or graph.startswith("device-fused") else (0, 0))
    monkeypatch.setenv("OATK_DEBUG_EC_FUSED", "0" if graph in ("device-heavy", "device-heavy-spill") else "1")
    # device-fused: every block that takes more than one wavefront step goes past its budget and starts again in ec_fused.hpp (several steps per barrier); device-heavy-mix:
    # more than eight; by default three thousand
    monkeypatch.setenv("OATK_DEBUG_EC_STEP_BUDGET", "1" if graph.startswith("device-fused") else ("8" if graph == "device-heavy-mix" else "0"))
    monkeypatch.setenv("OATK_DEBUG_EC_SERIAL_TIERS", "1" if graph == "device-tiers-serial" else "0")
    # which solver: "device" leaves the choice to the library (by the graph: round 4's tiers when the live graph does not branch anywhere, else the classes with budgets and the
    # second stage -- api_ec.inc); the other variants name theirs
    if graph == "device":
        monkeypatch.delenv("OATK_DEBUG_EC_HEAVY", raising=False)
    else:
        monkeypatch.setenv("OATK_DEBUG_EC_HEAVY", "0" if graph.startswith("device-tiers") else "1")
    monkeypatch.setenv("OATK_DEBUG_EC_HEAVY_CAP2", "400" if graph == "device-heavy-mix" else "0")
    monkeypatch.setenv("OATK_DEBUG_EC_HEAVY_FL", "64" if graph in ("device-heavy-spill", "device-fused-spill") else "0")
    hip._check(hip.L.oatk_hip_debug_ec_tiers(hip.h, t0, t1), "oatk_hip_debug_ec_tiers")
    db, scm = device_dbs(hip, reads, K, S)                  # reference-layout structs built from the device scan + count
    L = R.lib()
    g = L.refx_make_graph(db, scm, 0, 0.0)                  # run_syncasm.c:109
    L.refx_consensus(db, g, 1, 1)                           # run_syncasm.c:117
    G = E.flatten_graph(g)
    if graph != "host":                                     # graph built on the device too: nothing of the EC round on the host
        assert_graph_equal(device_graph(hip), G)
        st = device_ec(hip, None, 0.02, c, 0.35)
        st2 = device_ec(hip, None, 0.02, c, 0.35)           # the resident graph survives a correction
        assert np.array_equal(st, st2)
    else:
        st = device_ec(hip, G, 0.02, c, 0.35)
    got = {k: fetch_ec(hip, k) for k in EC_BUF}
    # reference on the very same structs
    summary = E.reference_ec(_H(db), _H(scm), g, 0.02, c, 0.35, threads=3)
    rdb, rscm = object.__new__(R.SrDb), object.__new__(R.ScmDb)
    rdb._h, rdb.K, rdb.S, rscm._h = db, K, S, scm
    sr1, sc1 = rdb.flatten(), rscm.flatten()
    assert np.array_equal(got["EC_N_SCM"], sr1["n_scm"])
    assert np.array_equal(got["EC_KMER"], sr1["k_mer"])
    assert np.array_equal(got["EC_MPOS"], sr1["m_pos"])
    assert np.array_equal(got["EC_SMER"], sr1["s_mer"])
    assert np.array_equal(got["EC_SCM_COV"], sc1["cov"])
    assert np.array_equal(got["EC_SCM_DEL"], sc1["del"])
    assert np.array_equal(got["EC_SCM_OCC"], sc1["occ"])
    total = int(st[0] + st[5] + st[10])
    assert total == summary["total"] and total > 0
    assert int(st[2] + st[7]) == summary["corrected"] and int(st[1] + st[6]) == summary["uncorrected"]
    assert int(st[3] + st[8]) == summary["ambiseq"] and int(st[4] + st[9]) == summary["ambipath"]   # the reference prints stats[3]+[8] under "ambiguous seqs"
    if graph.startswith("device-tiers") or graph.startswith("device-heavy") or graph.startswith("device-fused"):
        assert int(st[11]) > 0                               # blocks did fall through
        hip._check(hip.L.oatk_hip_debug_ec_tiers(hip.h, 0, 0), "oatk_hip_debug_ec_tiers")
    L.refx_scg_destroy(g)
    rscm.close(), rdb.close()
