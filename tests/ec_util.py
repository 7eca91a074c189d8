"""Helpers for the error-correction parity tests: flatten the reference's EC graph, run the oracle's EC (oracle/ec.c)."""
import ctypes as C
import os
import re
import tempfile

import numpy as np

import oracle_lib as O
import ref_lib as R


class GraphT(C.Structure):
    _fields_ = [("n_vtx", C.c_uint64), ("n_arc", C.c_uint64),
                ("vtx_len", C.c_void_p), ("vtx_del", C.c_void_p), ("vtx_seq_off", C.c_void_p), ("seq", C.c_void_p),
                ("arc_w", C.c_void_p), ("arc_ls", C.c_void_p), ("arc_cov", C.c_void_p), ("arc_del", C.c_void_p),
                ("idx_p", C.c_void_p), ("idx_n", C.c_void_p)]


class EcOutT(C.Structure):
    _fields_ = [("tot", C.c_uint64), ("updated_reads", C.c_uint64), ("n_scm", C.POINTER(C.c_uint32)),
                ("k_mer", C.POINTER(C.c_uint64)), ("m_pos", C.POINTER(C.c_uint32)), ("s_mer", C.POINTER(C.c_uint64)),
                ("stats", C.c_long * 11)]


def ref_graph(db, scm, min_k_cov=0, min_a_cov_f=0.0):
    """reference make_syncmer_graph + hoco consensus (run_syncasm.c:109,117), flattened to numpy; returns (handle, dict)"""
    L = R.lib()
    g = L.refx_make_graph(db.handle, scm.handle, min_k_cov, min_a_cov_f)
    L.refx_consensus(db.handle, g, 1, 1)
    return g, flatten_graph(g)


def flatten_graph(g):
    L = R.lib()
    nv, na, sb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    L.refx_graph_dims(g, C.byref(nv), C.byref(na), C.byref(sb))
    nv, na, sb = nv.value, na.value, sb.value
    G = {
        "vtx_len": np.zeros(nv, np.uint64), "vtx_del": np.zeros(nv, np.uint8), "vtx_cov": np.zeros(nv, np.uint32),
        "vtx_seq_off": np.zeros(nv, np.uint64), "seq": np.zeros(max(sb, 1), np.uint8),
        "arc_v": np.zeros(max(na, 1), np.uint64), "arc_w": np.zeros(max(na, 1), np.uint64), "arc_ls": np.zeros(max(na, 1), np.uint64),
        "arc_cov": np.zeros(max(na, 1), np.uint32), "arc_del": np.zeros(max(na, 1), np.uint8), "arc_comp": np.zeros(max(na, 1), np.uint8),
        "idx_p": np.zeros(2 * nv, np.uint64), "idx_n": np.zeros(2 * nv, np.uint64),
    }
    order = ["vtx_len", "vtx_del", "vtx_cov", "vtx_seq_off", "seq", "arc_v", "arc_w", "arc_ls", "arc_cov", "arc_del", "arc_comp", "idx_p", "idx_n"]
    L.refx_graph_flatten(g, *[G[k].ctypes.data for k in order])
    G["n_vtx"], G["n_arc"] = nv, na
    return G


def _graph_struct(G):
    gs = GraphT(G["n_vtx"], G["n_arc"], G["vtx_len"].ctypes.data, G["vtx_del"].ctypes.data, G["vtx_seq_off"].ctypes.data,
                G["seq"].ctypes.data, G["arc_w"].ctypes.data, G["arc_ls"].ctypes.data, G["arc_cov"].ctypes.data, G["arc_del"].ctypes.data,
                G["idx_p"].ctypes.data, G["idx_n"].ctypes.data)
    return gs


def oracle_find_error_syncmers(G, scm_cov, scm_del, err_mer_c, max_err_c, err_arc_c, max_arc_f):
    """mutates G['vtx_del'], G['arc_del'] and scm_del like find_error_syncmers(..., del_err=1)"""
    L = O.lib()
    L.orc_find_error_syncmers.restype = C.c_int64
    L.orc_find_error_syncmers.argtypes = [C.POINTER(GraphT), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double]
    gs = _graph_struct(G)
    return L.orc_find_error_syncmers(C.byref(gs), scm_cov.ctypes.data, scm_del.ctypes.data, err_mer_c, max_err_c, err_arc_c, max_arc_f)


def oracle_ec_reads(G, scm_del, scm_s, K, max_edist, sr):
    """sr: flat image of sr_db (hoco_l, hoco_s, n_scm, k_mer (ids), m_pos, s_mer); returns dict with the corrected chains"""
    L = O.lib()
    L.orc_ec_reads.argtypes = [C.POINTER(GraphT), C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_uint64] + [C.c_void_p] * 7 + [C.POINTER(EcOutT)]
    L.orc_ec_out_free.argtypes = [C.POINTER(EcOutT)]
    gs = _graph_struct(G)
    n = len(sr["hoco_l"])
    boff = np.zeros(n + 1, np.uint64)
    boff[1:] = np.cumsum((sr["hoco_l"].astype(np.uint64) + 3) // 4)
    out = EcOutT()
    arrs = [np.ascontiguousarray(sr[k]) for k in ("hoco_l", "hoco_s")] + [boff] + [np.ascontiguousarray(sr[k]) for k in ("n_scm", "k_mer", "m_pos", "s_mer")]
    L.orc_ec_reads(C.byref(gs), scm_del.ctypes.data, scm_s.ctypes.data, K, max_edist, n, *[a.ctypes.data for a in arrs], C.byref(out))
    res = {"n_scm": O._arr(out.n_scm, n, np.uint32), "k_mer": O._arr(out.k_mer, out.tot, np.uint64),
           "m_pos": O._arr(out.m_pos, out.tot, np.uint32), "s_mer": O._arr(out.s_mer, out.tot, np.uint64),
           "stats": np.array(list(out.stats), np.int64)}
    L.orc_ec_out_free(C.byref(out))
    return res


def oracle_update_db(n_scm, k_mer, m_pos, n_syncmers):
    L = O.lib()
    L.orc_update_db.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    cov = np.zeros(n_syncmers, np.uint32)
    de = np.zeros(n_syncmers, np.uint8)
    occ_off = np.zeros(n_syncmers + 1, np.uint64)
    occ = np.zeros(max(len(k_mer), 1), np.uint64)
    L.orc_update_db(len(n_scm), n_scm.ctypes.data, k_mer.ctypes.data, m_pos.ctypes.data, n_syncmers, cov.ctypes.data, de.ctypes.data,
                    occ_off.ctypes.data, occ.ctypes.data)
    return {"cov": cov, "del": de, "occ_off": occ_off, "occ": occ[:len(k_mer)]}


def reference_ec(db, scm, g, max_edist, c, a, threads=2):
    """read_error_correction of the compiled reference with syncasm's arguments (run_syncasm.c:124); returns its stderr summary"""
    L = R.lib()
    fd, path = tempfile.mkstemp()
    os.close(fd)
    saved = os.dup(2)
    f = os.open(path, os.O_WRONLY | os.O_TRUNC)
    os.dup2(f, 2)
    try:
        L.refx_ec(db.handle, g, max_edist, c, c * 10, c, a, threads)
    finally:
        os.dup2(saved, 2)
        os.close(f)
        os.close(saved)
    txt = open(path).read()
    os.unlink(path)
    def grab(label):
        m = re.search(re.escape(label) + r"\s*:\s*(\d+)", txt)
        return int(m.group(1)) if m else None
    return {"total": grab("total number of error blocks"), "uncorrected": grab("- uncorrected"), "corrected": grab("- corrected"),
            "ambiseq": grab("- ambiguous seqs"), "ambipath": grab("- ambiguous path"), "text": txt}


class CountViewT(C.Structure):
    _fields_ = [("n_scm", C.c_uint64), ("occ_off", C.c_void_p), ("occ", C.c_void_p)]


class EcGraphOutT(C.Structure):
    _fields_ = [("n_vtx", C.c_uint64), ("n_arc", C.c_uint64), ("arc_v", C.POINTER(C.c_uint64)), ("arc_w", C.POINTER(C.c_uint64)),
                ("arc_ls", C.POINTER(C.c_uint64)), ("arc_cov", C.POINTER(C.c_uint32)), ("arc_comp", C.POINTER(C.c_uint8)),
                ("idx_p", C.POINTER(C.c_uint64)), ("idx_n", C.POINTER(C.c_uint64)), ("multi_arc", C.c_int)]


def occ_lists(n_scm, k_mer, m_pos, n_syncmers):
    """syncmer occurrence lists of a fresh count: sid << 32 | idx << 1 | rev in (sid, idx) order (syncmer.c:560-575)"""
    sid = np.repeat(np.arange(len(n_scm), dtype=np.uint64), n_scm)
    start = np.repeat(np.cumsum(n_scm, dtype=np.uint64) - n_scm, n_scm)
    idx = np.arange(len(k_mer), dtype=np.uint64) - start
    ids = k_mer >> np.uint64(1)
    occ = sid << np.uint64(32) | idx << np.uint64(1) | (m_pos.astype(np.uint64) & np.uint64(1))
    order = np.argsort(ids, kind="stable")
    off = np.zeros(n_syncmers + 1, np.uint64)
    off[1:] = np.cumsum(np.bincount(ids.astype(np.int64), minlength=n_syncmers))
    return off, np.ascontiguousarray(occ[order])


def oracle_ecgraph(n_scm, k_mer, m_pos, occ_off, occ, K):
    """oracle/ecgraph.c: make_syncmer_graph(sr_db, scm_db, 0, 0.) + arc overlaps, as a dict shaped like flatten_graph's"""
    L = O.lib()
    L.orc_ecgraph_build.restype = C.POINTER(EcGraphOutT)
    L.orc_ecgraph_build.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(CountViewT), C.c_int]
    L.orc_ecgraph_free.argtypes = [C.POINTER(EcGraphOutT)]
    arrs = [np.ascontiguousarray(a) for a in (n_scm, k_mer, m_pos, occ_off, occ)]
    cv = CountViewT(len(occ_off) - 1, arrs[3].ctypes.data, arrs[4].ctypes.data)
    gp = L.orc_ecgraph_build(len(n_scm), arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, C.byref(cv), K)
    g = gp.contents
    nv, na = g.n_vtx, g.n_arc
    out = {"n_vtx": nv, "n_arc": na, "multi_arc": g.multi_arc,
           "arc_v": O._arr(g.arc_v, na, np.uint64), "arc_w": O._arr(g.arc_w, na, np.uint64), "arc_ls": O._arr(g.arc_ls, na, np.uint64),
           "arc_cov": O._arr(g.arc_cov, na, np.uint32), "arc_comp": O._arr(g.arc_comp, na, np.uint8),
           "idx_p": O._arr(g.idx_p, 2 * nv, np.uint64), "idx_n": O._arr(g.idx_n, 2 * nv, np.uint64)}
    L.orc_ecgraph_free(gp)
    return out
