"""ctypes window onto oracle/_ref/liboatk_ref.so -- the reference compiled from its own sources.

Test infrastructure only.  The library exists only after `make -C oracle ref` (which needs
/root/reference, i.e. this container); it then travels to the GPU box as a built artefact.
"""
import ctypes as C
import os
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
LIB_PATH = os.path.join(REF_DIR, "liboatk_ref.so")
BIN_PATH = os.path.join(REF_DIR, "syncasm")

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.refx_scan.restype = vp
        L.refx_scan.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int]
        L.refx_srdb_destroy.argtypes = [vp]
        L.refx_srdb_n.restype = C.c_uint64
        L.refx_srdb_n.argtypes = [vp]
        L.refx_srdb_lengths.argtypes = [vp, vp, vp, vp]
        L.refx_srdb_flatten.argtypes = [vp] * 8
        L.refx_srdb_nnucl.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
        L.refx_srdb_stat.argtypes = [vp, vp, vp]
        L.refx_collect.restype = vp
        L.refx_collect.argtypes = [vp]
        L.refx_scmdb_destroy.argtypes = [vp]
        L.refx_scmdb_n.restype = C.c_uint64
        L.refx_scmdb_n.argtypes = [vp]
        L.refx_scmdb_total_cov.restype = C.c_uint64
        L.refx_scmdb_total_cov.argtypes = [vp]
        L.refx_scmdb_flatten.argtypes = [vp] * 6
        L.refx_make_graph.restype = vp
        L.refx_make_graph.argtypes = [vp, vp, C.c_uint32, C.c_double]
        L.refx_consensus.argtypes = [vp, vp, C.c_int, C.c_int]
        L.refx_scg_destroy.argtypes = [vp]
        L.refx_find_error_syncmers.restype = C.c_int64
        L.refx_find_error_syncmers.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int]
        L.refx_graph_dims.argtypes = [vp, vp, vp, vp]
        L.refx_graph_flatten.argtypes = [vp] * 14
        L.refx_graph_flatten2.argtypes = [vp] * 5
        L.refx_ec.argtypes = [vp, vp, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int]
        L.refx_wf_ed.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, vp]
        L.refx_wf_new.restype = vp
        L.refx_wf_new.argtypes = [C.c_int32, C.c_char_p, C.c_int32]
        L.refx_wf_step.argtypes = [vp, C.c_int32, C.c_char_p, vp]
        L.refx_wf_free.argtypes = [vp]
        L.refx_syncasm.restype = C.c_int
        L.refx_syncasm.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_char_p]
        _lib = L
    return _lib


def write_fasta(reads, path, width=0):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">r%d\n" % i)
            if width <= 0:
                f.write(r + b"\n")
            else:
                for j in range(0, max(len(r), 1), width):
                    f.write(r[j:j + width] + b"\n")


def _files_arg(paths):
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    return arr


class SrDb:
    """reference sr_db_t built by sr_read (syncmer.c:487) from FASTA files"""

    def __init__(self, paths, K, S, threads=1, m_data=0):
        self.K, self.S = K, S
        if m_data:
            L = lib()
            L.refx_scan_cap.restype = C.c_void_p
            L.refx_scan_cap.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t]
            self._h = L.refx_scan_cap(_files_arg(paths), len(paths), K, S, threads, m_data)
        else:
            self._h = lib().refx_scan(_files_arg(paths), len(paths), K, S, threads)
        if not self._h:
            raise RuntimeError("reference failed to open input")

    @classmethod
    def from_reads(cls, reads, K, S, threads=1):
        fd, path = tempfile.mkstemp(suffix=".fa")
        os.close(fd)
        try:
            write_fasta(reads, path)
            return cls([path], K, S, threads)
        finally:
            os.unlink(path)

    @property
    def handle(self):
        return self._h

    def n(self):
        return int(lib().refx_srdb_n(self._h))

    def flatten(self, n_nn=None):
        L = lib()
        n = self.n()
        hoco_l = np.zeros(n, np.uint32)
        n_scm = np.zeros(n, np.uint32)
        sid = np.zeros(n, np.uint64)
        L.refx_srdb_lengths(self._h, hoco_l.ctypes.data, n_scm.ctypes.data, sid.ctypes.data)
        tot_h = int(hoco_l.astype(np.uint64).sum())
        tot_b = int(((hoco_l.astype(np.uint64) + 3) // 4).sum())
        tot_s = int(n_scm.astype(np.uint64).sum())
        hoco_s = np.zeros(max(tot_b, 1), np.uint8)
        ho_rl = np.zeros(max(tot_h, 1), np.uint8)
        lrl = np.zeros(max(tot_h, 1), np.uint32)
        nl = C.c_uint64(0)
        m_pos = np.zeros(max(tot_s, 1), np.uint32)
        s_mer = np.zeros(max(tot_s, 1), np.uint64)
        k_mer = np.zeros(max(tot_s, 1), np.uint64)
        L.refx_srdb_flatten(self._h, hoco_s.ctypes.data, ho_rl.ctypes.data, lrl.ctypes.data, C.addressof(nl),
                            m_pos.ctypes.data, s_mer.ctypes.data, k_mer.ctypes.data)
        out = {"hoco_l": hoco_l, "n_scm": n_scm, "sid": sid, "hoco_s": hoco_s[:tot_b], "ho_rl": ho_rl[:tot_h],
               "ho_l_rl": lrl[:nl.value].copy(), "m_pos": m_pos[:tot_s], "s_mer": s_mer[:tot_s], "k_mer": k_mer[:tot_s]}
        if n_nn is not None:
            parts = []
            for i in range(n):
                a = np.zeros(max(int(n_nn[i]), 1), np.uint32)
                L.refx_srdb_nnucl(self._h, i, int(n_nn[i]), a.ctypes.data)
                parts.append(a[:int(n_nn[i])])
            out["n_nucl"] = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
        return out

    def stat(self):
        i8 = np.zeros(8, np.int32)
        d5 = np.zeros(5, np.float64)
        lib().refx_srdb_stat(self._h, i8.ctypes.data, d5.ctypes.data)
        return i8, d5

    def close(self):
        if self._h:
            lib().refx_srdb_destroy(self._h)
            self._h = None


class ScmDb:
    """reference syncmer_db_t from collect_syncmer_from_reads (syncmer.c:1397); rewrites sr_db k_mer to ids"""

    def __init__(self, srdb: SrDb):
        self._h = lib().refx_collect(srdb.handle)

    @property
    def handle(self):
        return self._h

    def n(self):
        return int(lib().refx_scmdb_n(self._h)) if self._h else 0

    def flatten(self):
        L = lib()
        n = self.n()
        if n == 0:
            return {"n_scm": 0, "h": np.zeros(0, np.uint64), "s": np.zeros(0, np.uint64), "cov": np.zeros(0, np.uint32),
                    "del": np.zeros(0, np.uint8), "occ": np.zeros(0, np.uint64)}
        tot = int(L.refx_scmdb_total_cov(self._h))
        h = np.zeros(n, np.uint64)
        s = np.zeros(n, np.uint64)
        cov = np.zeros(n, np.uint32)
        de = np.zeros(n, np.uint8)
        occ = np.zeros(max(tot, 1), np.uint64)
        L.refx_scmdb_flatten(self._h, h.ctypes.data, s.ctypes.data, cov.ctypes.data, de.ctypes.data, occ.ctypes.data)
        return {"n_scm": n, "h": h, "s": s, "cov": cov, "del": de, "occ": occ[:tot]}

    def close(self):
        if self._h:
            lib().refx_scmdb_destroy(self._h)
            self._h = None


def wf_ed(ts: bytes, qs: bytes, bw: int, is_ext=1):
    out = (C.c_int32 * 3)()
    lib().refx_wf_ed(len(ts), ts, len(qs), qs, is_ext, bw, out)
    return tuple(out)


class Wavefront:
    def __init__(self, ts: bytes, bw: int):
        self._h = lib().refx_wf_new(len(ts), ts, bw)
        self._keep = None

    def step(self, qs: bytes):
        out = (C.c_int32 * 3)()
        self._keep = C.create_string_buffer(qs, len(qs) + 8)
        lib().refx_wf_step(self._h, len(qs), self._keep, out)
        return tuple(out)

    def close(self):
        if self._h:
            lib().refx_wf_free(self._h)
            self._h = None
