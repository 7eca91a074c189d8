"""Helpers for the base-space consensus tests: flat read views, oracle (oracle/consensus.c) and compiled-reference calls."""
import ctypes as C

import numpy as np

import oracle_lib as O


class ReadsView(C.Structure):
    _fields_ = [("sid0", C.c_uint64), ("scm_off", C.c_void_p), ("k_mer", C.c_void_p), ("m_pos", C.c_void_p), ("hs_off", C.c_void_p),
                ("hoco_s", C.c_void_p), ("rl_off", C.c_void_p), ("ho_rl", C.c_void_p), ("lrl_off", C.c_void_p), ("ho_l_rl", C.c_void_p)]


def make_view(sr, sid0=0):
    """sr: flat image of the reads (hoco_l, hoco_s, ho_rl, ho_l_rl, n_scm, k_mer, m_pos) -> (ReadsView, keep-alive dict)"""
    hl = sr["hoco_l"].astype(np.uint64)
    n = len(hl)
    k = {}
    k["scm_off"] = np.concatenate([[0], np.cumsum(sr["n_scm"].astype(np.uint64))]).astype(np.uint64)
    k["hs_off"] = np.concatenate([[0], np.cumsum((hl + 3) // 4)])[:n].astype(np.uint64)
    k["rl_off"] = np.concatenate([[0], np.cumsum(hl)]).astype(np.uint64)
    is255 = (sr["ho_rl"] == 255).astype(np.uint64)
    c255 = np.concatenate([[0], np.cumsum(is255)]).astype(np.uint64)
    k["lrl_off"] = c255[k["rl_off"][:n].astype(np.int64)].astype(np.uint64)
    k["rl_off"] = k["rl_off"][:n].copy()
    for f in ("k_mer", "m_pos", "hoco_s", "ho_rl"):
        k[f] = np.ascontiguousarray(sr[f])
    k["ho_l_rl"] = np.ascontiguousarray(sr["ho_l_rl"] if len(sr["ho_l_rl"]) else np.zeros(1, np.uint32))
    v = ReadsView(sid0, *[k[f].ctypes.data for f in ("scm_off", "k_mer", "m_pos", "hs_off", "hoco_s", "rl_off", "ho_rl", "lrl_off", "ho_l_rl")])
    return v, k


def oracle_rl(view, occ, K):
    L = O.lib()
    L.orc_consensus_rl.argtypes = [C.POINTER(ReadsView), C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    tot = np.zeros(K, np.uint64)
    m, first = C.c_uint32(), C.c_uint64()
    occ = np.ascontiguousarray(occ, np.uint64)
    L.orc_consensus_rl(C.byref(view), len(occ), occ.ctypes.data, K, tot.ctypes.data, C.byref(m), C.byref(first))
    return tot, m.value, first.value


def oracle_string(view, tot, m_seq, first, K, rev, beg, hoco):
    L = O.lib()
    L.orc_consensus_string.restype = C.c_int64
    L.orc_consensus_string.argtypes = [C.POINTER(ReadsView), C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_char_p]
    cap = (abs(beg) + K + 8) + int(tot.sum() // max(m_seq, 1)) + 2 * K + 64
    buf = C.create_string_buffer(cap)
    n = L.orc_consensus_string(C.byref(view), tot.ctypes.data, m_seq, first, K, rev, beg, hoco, buf)
    return n, buf.raw[:n]


def reference_string(db, scm, sid, rev, beg, hoco, cap=1 << 20):
    import ref_lib as R
    L = R.lib()
    L.refx_syncmer_consensus.restype = C.c_int64
    L.refx_syncmer_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int64, C.c_int, C.c_char_p, C.c_int64]
    buf = C.create_string_buffer(cap)
    n = L.refx_syncmer_consensus(db.handle, scm.handle, sid, rev, beg, hoco, buf, cap)
    return n, buf.raw[:n]


def long_run_reads(seed, K):
    """HiFi-like reads over a genome that holds homopolymers of 300 and 700 bases (run lengths beyond the 255 escape of ho_rl)
    and plenty of short ones; reads disagree on run lengths, so the consensus has something to average"""
    import adversarial as A
    rng = np.random.default_rng(seed)
    g = bytearray(A.rand_dna(rng, 40 * K))
    for at, ln, ch in ((9 * K, 300, b"A"), (21 * K + 11, 700, b"C"), (30 * K, 260, b"T")):
        g[at:at + ln] = ch * ln
    g = bytes(g)
    reads = []
    for _ in range(200):
        st = int(rng.integers(0, len(g) - 12 * K))
        ln = int(rng.integers(8 * K, 12 * K))
        r = bytearray(g[st:st + ln])
        # homopolymer length errors, the dominant HiFi error: lengthen or shorten some runs
        for p in sorted(rng.integers(1, len(r) - 1, size=30).tolist(), reverse=True):
            if rng.integers(0, 2):
                r.insert(p, r[p])
            elif r[p] == r[p - 1]:
                del r[p]
        r = bytes(r)
        reads.append(A.revcomp(r) if rng.integers(0, 2) else r)
    return reads
