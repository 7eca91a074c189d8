"""Helpers for the read-alignment parity tests: the reference's scg_t flattened the way scg_ra_analysis_thread reads it, the oracle's
alignment (oracle/align.c), the old_ra filter of alignment.c:610-634 restated, the device result as arrays."""
import ctypes as C
import math
import sys

import numpy as np

import oracle_lib as O
import ref_lib as R

GRAPH_FIELDS = (("su_off", np.uint64), ("su_uid", np.uint64), ("su_pos", np.uint32), ("utg_n", np.uint32), ("idx_p", np.uint64), ("idx_n", np.uint64),
                ("arc_w", np.uint64), ("arc_ln", np.uint64), ("arc_del", np.uint8))
OUT_FIELDS = ("sid", "n", "s", "uid", "u_beg", "u_end", "s_beg", "s_end")


class RaGraphT(C.Structure):
    _fields_ = [("n_scm", C.c_uint64), ("n_utg", C.c_uint64), ("n_arc", C.c_uint64)] + [(k, C.c_void_p) for k, _ in GRAPH_FIELDS]


class RaOutT(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_aln", "n_frg", "m_aln", "m_frg", "n_mapped", "n_unique")] + [
        ("sid", C.POINTER(C.c_uint64)), ("n", C.POINTER(C.c_uint32)), ("s", C.POINTER(C.c_double)), ("uid", C.POINTER(C.c_uint64)),
        ("u_beg", C.POINTER(C.c_uint64)), ("u_end", C.POINTER(C.c_uint64)), ("s_beg", C.POINTER(C.c_uint32)), ("s_end", C.POINTER(C.c_uint32))]


def ref_ra_graph(g):
    """scg_t of the compiled reference -> dict of arrays shaped like oatk_ra_graph_t"""
    L = R.lib()
    L.refx_ra_graph_dims.argtypes = [C.c_void_p] * 5
    L.refx_ra_graph_flatten.argtypes = [C.c_void_p] * 10
    d = [C.c_uint64() for _ in range(4)]
    L.refx_ra_graph_dims(g, *[C.byref(x) for x in d])
    ns, nsu, nu, na = (x.value for x in d)
    G = {"n_scm": ns, "su_off": np.zeros(ns + 1, np.uint64), "su_uid": np.zeros(max(nsu, 1), np.uint64), "su_pos": np.zeros(max(nsu, 1), np.uint32),
         "utg_n": np.zeros(max(nu, 1), np.uint32), "idx_p": np.zeros(max(2 * nu, 1), np.uint64), "idx_n": np.zeros(max(2 * nu, 1), np.uint64),
         "arc_w": np.zeros(max(na, 1), np.uint64), "arc_ln": np.zeros(max(na, 1), np.uint64), "arc_del": np.zeros(max(na, 1), np.uint8)}
    L.refx_ra_graph_flatten(g, *[G[k].ctypes.data for k, _ in GRAPH_FIELDS])
    G["su_uid"], G["su_pos"] = G["su_uid"][:nsu], G["su_pos"][:nsu]
    G["utg_n"], G["idx_p"], G["idx_n"] = G["utg_n"][:nu], G["idx_p"][:2 * nu], G["idx_n"][:2 * nu]
    G["arc_w"], G["arc_ln"], G["arc_del"] = G["arc_w"][:na], G["arc_ln"][:na], G["arc_del"][:na]
    return G


def old_ra_filter(prev, n_reads, for_unzip):
    """alignment.c:610-634: which reads the next call aligns and the score they must reach; `prev` = flattened previous alignments"""
    old = np.zeros(n_reads, np.int64)
    if for_unzip and len(prev["sid"]):
        for sid, n, s in zip(prev["sid"].tolist(), prev["n"].tolist(), prev["s"].tolist()):
            if n > 2 and (old[sid] & 1) == 0:
                frac, whole = math.modf(s)
                if frac < sys.float_info.epsilon:
                    whole -= 1
                old[sid] = int(whole) << 1 | 1
    else:
        old[:] = 1
    return old


def oracle_align(n_scm, k_mer, m_pos, G, old_ra=None):
    L = O.lib()
    L.orc_read_alignment.restype = C.POINTER(RaOutT)
    L.orc_read_alignment.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RaGraphT), C.c_void_p]
    L.orc_ra_out_free.argtypes = [C.POINTER(RaOutT)]
    keep = [np.ascontiguousarray(G[k], dtype=dt) for k, dt in GRAPH_FIELDS]
    gs = RaGraphT(int(G["n_scm"]), len(G["utg_n"]), len(G["arc_w"]), *[a.ctypes.data for a in keep])
    a = [np.ascontiguousarray(n_scm, dtype=np.uint32), np.ascontiguousarray(k_mer, dtype=np.uint64), np.ascontiguousarray(m_pos, dtype=np.uint32)]
    o = None if old_ra is None else np.ascontiguousarray(old_ra, dtype=np.int64)
    p = L.orc_read_alignment(len(a[0]), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, C.byref(gs), None if o is None else o.ctypes.data)
    r = p.contents
    na, nf = r.n_aln, r.n_frg
    out = {"sid": O._arr(r.sid, na, np.uint64), "n": O._arr(r.n, na, np.uint32), "s": O._arr(r.s, na, np.float64), "uid": O._arr(r.uid, nf, np.uint64),
           "u_beg": O._arr(r.u_beg, nf, np.uint64), "u_end": O._arr(r.u_end, nf, np.uint64), "s_beg": O._arr(r.s_beg, nf, np.uint32),
           "s_end": O._arr(r.s_end, nf, np.uint32), "n_mapped": r.n_mapped, "n_unique": r.n_unique}
    L.orc_ra_out_free(p)
    return out


def device_align(hip, G, old_ra=None):
    """oatk_hip_read_alignment on the resident chains; result shaped like the flattened scg_ra_v"""
    na, nf, st = hip.read_alignment(G, old_ra)
    off = hip.fetch("RA_ALN_OFF")
    out = {"sid": hip.fetch("RA_ALN_SID").astype(np.uint64), "n": np.diff(off).astype(np.uint32), "s": hip.fetch("RA_ALN_S"),
           "uid": hip.fetch("RA_FRG_UID"), "u_beg": hip.fetch("RA_FRG_UBEG").astype(np.uint64), "u_end": hip.fetch("RA_FRG_UEND").astype(np.uint64),
           "s_beg": hip.fetch("RA_FRG_SBEG"), "s_end": hip.fetch("RA_FRG_SEND"), "n_mapped": int(st[0]), "n_unique": int(st[1]), "skipped": int(st[2])}
    assert len(out["sid"]) == na and len(out["uid"]) == nf
    return out


def assert_same(got, want, what=""):
    for k in OUT_FIELDS:
        assert len(got[k]) == len(want[k]), (what, k, len(got[k]), len(want[k]))
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), (what, k)


def random_graph(rng, n_scm, chains, n_src, max_len, p_overlap, p_noise):
    """unitigs cut out of the chains of n_src random reads (pieces of 1..max_len syncmers, consecutive pieces joined by an arc, with probability
    p_overlap sharing one syncmer across an arc of ln = 1), so other reads of the same region hit several unitigs at once; plus random arcs
    and a few deleted ones.  Nothing an assembler would build, but every path of the aligner gets walked: ties, many predecessors, several
    best chains, chains that miss the 90 % mark"""
    n_chain, k_mer, m_pos = chains
    starts = np.cumsum(n_chain, dtype=np.int64) - n_chain
    hits = [[] for _ in range(n_scm)]
    utg_n, arcs = [], {}
    for r in rng.choice(len(n_chain), size=min(n_src, len(n_chain)), replace=False).tolist():
        n = int(n_chain[r])
        ids = (k_mer[starts[r]:starts[r] + n] >> np.uint64(1)).astype(np.int64)
        rev = (m_pos[starts[r]:starts[r] + n] & 1).astype(np.int64)
        flip = int(rng.integers(0, 2))                   # lay the read down in either orientation
        if flip:
            ids, rev = ids[::-1], 1 - rev[::-1]
        j, prev_u = 0, None
        while j < n:
            ln = int(rng.integers(1, max_len + 1))
            ovl = 1 if (prev_u is not None and j > 0 and rng.random() < p_overlap) else 0
            seg = list(range(j - ovl, min(j - ovl + ln + ovl, n)))
            u = len(utg_n)
            for q, t in enumerate(seg):
                hits[int(ids[t])].append((u << 1 | int(rev[t]), q))
            utg_n.append(len(seg))
            if prev_u is not None:
                arcs[(prev_u << 1, u << 1)] = ovl
                arcs[(u << 1 | 1, prev_u << 1 | 1)] = ovl
            prev_u, j = u, seg[-1] + 1
    n_utg = len(utg_n)
    for _ in range(int(p_noise * n_utg)):
        v, w = int(rng.integers(0, 2 * n_utg)), int(rng.integers(0, 2 * n_utg))
        if (v, w) not in arcs:
            arcs[(v, w)] = arcs[(w ^ 1, v ^ 1)] = int(rng.integers(0, 3))
    keys = sorted(arcs)
    su_off = np.zeros(n_scm + 1, np.uint64)
    su_uid, su_pos = [], []
    for sid in range(n_scm):
        hs = sorted(hits[sid], key=lambda t: (t[0] & 1, t[0] >> 1, t[1]))            # scg_scm_utg_index orders by (rev, utg, pos)
        su_uid += [h[0] for h in hs]
        su_pos += [h[1] for h in hs]
        su_off[sid + 1] = len(su_uid)
    idx_p, idx_n = np.zeros(2 * n_utg, np.uint64), np.zeros(2 * n_utg, np.uint64)
    for i, (v, w) in enumerate(keys):
        if idx_n[v] == 0:
            idx_p[v] = i
        idx_n[v] += 1
    return {"n_scm": n_scm, "su_off": su_off, "su_uid": np.array(su_uid, np.uint64), "su_pos": np.array(su_pos, np.uint32), "utg_n": np.array(utg_n, np.uint32),
            "idx_p": idx_p, "idx_n": idx_n, "arc_v": np.array([v for v, _ in keys], np.uint64), "arc_w": np.array([w for _, w in keys], np.uint64), "arc_ln": np.array([arcs[k] for k in keys], np.uint64),
            "arc_del": (rng.random(len(keys)) < 0.03).astype(np.uint8)}
