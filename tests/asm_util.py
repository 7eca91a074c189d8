"""Helpers for the assembly-graph parity tests: the oracle's graph (oracle/asmgraph.c) and the compiled reference's, as dicts of arrays."""
import ctypes as C

import numpy as np

import oracle_lib as O
import ref_lib as R

FIELDS = ["vtx_scm", "vtx_cov", "arc_v", "arc_w", "arc_cov", "arc_comp", "arc_link", "idx_p", "idx_n", "scm_del"]


class AsmGraphOutT(C.Structure):
    _fields_ = [("n_vtx", C.c_uint64), ("n_arc", C.c_uint64), ("vtx_scm", C.POINTER(C.c_uint32)), ("vtx_cov", C.POINTER(C.c_uint32)),
                ("arc_v", C.POINTER(C.c_uint64)), ("arc_w", C.POINTER(C.c_uint64)), ("arc_link", C.POINTER(C.c_uint64)),
                ("arc_cov", C.POINTER(C.c_uint32)), ("arc_comp", C.POINTER(C.c_uint8)), ("idx_p", C.POINTER(C.c_uint64)),
                ("idx_n", C.POINTER(C.c_uint64)), ("multi_arc", C.c_int)]


def oracle_asmgraph(n_scm, k_mer, m_pos, scm_cov, scm_del, min_k_cov, min_a_cov_f):
    """oracle/asmgraph.c: make_syncmer_graph(sr_db, scm_db, min_k_cov, min_a_cov_f) + asmg_finalize(g, 1)"""
    L = O.lib()
    L.orc_asmgraph_build.restype = C.POINTER(AsmGraphOutT)
    L.orc_asmgraph_build.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_double]
    L.orc_asmgraph_free.argtypes = [C.POINTER(AsmGraphOutT)]
    n_scm, k_mer, m_pos, scm_cov = (np.ascontiguousarray(a) for a in (n_scm, k_mer, m_pos, scm_cov))
    de = np.ascontiguousarray(scm_del).astype(np.uint8).copy()
    gp = L.orc_asmgraph_build(len(n_scm), n_scm.ctypes.data, k_mer.ctypes.data, m_pos.ctypes.data, len(scm_cov), scm_cov.ctypes.data, de.ctypes.data,
                              min_k_cov, min_a_cov_f)
    g = gp.contents
    nv, na = g.n_vtx, g.n_arc
    out = {"multi_arc": g.multi_arc, "scm_del": de,
           "vtx_scm": O._arr(g.vtx_scm, nv, np.uint32), "vtx_cov": O._arr(g.vtx_cov, nv, np.uint32),
           "arc_v": O._arr(g.arc_v, na, np.uint64), "arc_w": O._arr(g.arc_w, na, np.uint64), "arc_link": O._arr(g.arc_link, na, np.uint64),
           "arc_cov": O._arr(g.arc_cov, na, np.uint32), "arc_comp": O._arr(g.arc_comp, na, np.uint8),
           "idx_p": O._arr(g.idx_p, 2 * nv, np.uint64), "idx_n": O._arr(g.idx_n, 2 * nv, np.uint64)}
    L.orc_asmgraph_free(gp)
    return out


def reference_asmgraph(db_handle, scm_handle, min_k_cov, min_a_cov_f):
    """the compiled reference's make_syncmer_graph on its own (or layout-compatible) databases; None when it returns no graph"""
    import ec_util as E
    L = R.lib()
    g = L.refx_make_graph(db_handle, scm_handle, min_k_cov, min_a_cov_f)
    if not g:
        return None
    Gd = E.flatten_graph(g)
    nv, na = Gd["n_vtx"], Gd["n_arc"]
    vn, va0, aln, alink = (np.zeros(max(n, 1), np.uint64) for n in (nv, nv, na, na))
    L.refx_graph_flatten2(g, vn.ctypes.data, va0.ctypes.data, aln.ctypes.data, alink.ctypes.data)
    assert (vn[:nv] == 1).all() and (aln[:na] == 0).all() and (Gd["arc_ls"][:na] == 0).all() and (Gd["arc_del"][:na] == 0).all()
    assert (Gd["vtx_del"] == 0).all()
    out = {"vtx_scm": (va0[:nv] >> np.uint64(1)).astype(np.uint32), "vtx_cov": Gd["vtx_cov"], "arc_v": Gd["arc_v"][:na], "arc_w": Gd["arc_w"][:na],
           "arc_cov": Gd["arc_cov"][:na], "arc_comp": Gd["arc_comp"][:na], "arc_link": alink[:na], "idx_p": Gd["idx_p"], "idx_n": Gd["idx_n"]}
    L.refx_scg_destroy(g)
    return out


def assert_asm_equal(got, want, prefix=""):
    """got: dict from the oracle or the device (FIELDS); want: dict with the same fields under `prefix`"""
    w = {k: want[prefix + k] for k in FIELDS if (prefix + k) in want}
    for k in ("vtx_scm", "vtx_cov", "arc_v", "arc_w", "arc_cov", "arc_comp", "arc_link"):
        assert len(got[k]) == len(w[k]), k
        assert np.array_equal(np.asarray(got[k]).astype(np.uint64), np.asarray(w[k]).astype(np.uint64)), k
    idx_n = np.asarray(w["idx_n"]).astype(np.uint64)
    assert np.array_equal(np.asarray(got["idx_n"]).astype(np.uint64), idx_n)
    has = idx_n > 0
    assert np.array_equal(np.asarray(got["idx_p"])[has], np.asarray(w["idx_p"])[has])
    if "scm_del" in w:
        assert np.array_equal(np.asarray(got["scm_del"]).astype(np.uint8), np.asarray(w["scm_del"]).astype(np.uint8))
