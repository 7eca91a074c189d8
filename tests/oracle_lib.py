"""ctypes window onto oracle/_build/liboatk_oracle.so (the CPU restatement, test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboatk_oracle.so")


class ScanT(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64),
        ("tot_hoco", C.c_uint64), ("tot_bytes", C.c_uint64), ("tot_scm", C.c_uint64),
        ("tot_lrl", C.c_uint64), ("tot_nn", C.c_uint64),
        ("hoco_l", C.POINTER(C.c_uint32)), ("n_scm", C.POINTER(C.c_uint32)),
        ("n_lrl", C.POINTER(C.c_uint32)), ("n_nn", C.POINTER(C.c_uint32)),
        ("hoco_s", C.POINTER(C.c_uint8)), ("ho_rl", C.POINTER(C.c_uint8)),
        ("ho_l_rl", C.POINTER(C.c_uint32)), ("n_nucl", C.POINTER(C.c_uint32)),
        ("m_pos", C.POINTER(C.c_uint32)), ("s_mer", C.POINTER(C.c_uint64)), ("k_mer", C.POINTER(C.c_uint64)),
    ]


class CountT(C.Structure):
    _fields_ = [
        ("n_scm", C.c_uint64), ("tot_occ", C.c_uint64),
        ("h", C.POINTER(C.c_uint64)), ("s", C.POINTER(C.c_uint64)), ("cov", C.POINTER(C.c_uint32)),
        ("occ_off", C.POINTER(C.c_uint64)), ("occ", C.POINTER(C.c_uint64)), ("k_id", C.POINTER(C.c_uint64)),
        ("err", C.c_int),
    ]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_scan_batch.restype = C.POINTER(ScanT)
        L.orc_scan_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int]
        L.orc_scan_free.argtypes = [C.POINTER(ScanT)]
        L.orc_count.restype = C.POINTER(CountT)
        L.orc_count.argtypes = [C.POINTER(ScanT), C.c_int]
        L.orc_count_free.argtypes = [C.POINTER(CountT)]
        L.orc_kmer_hash.restype = C.c_uint64
        L.orc_kmer_hash.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_hash64.restype = C.c_uint64
        L.orc_hash64.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_murmur64a.restype = C.c_uint64
        L.orc_murmur64a.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.orc_wf_new.restype = C.c_void_p
        L.orc_wf_new.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
        L.orc_wf_step.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p]
        L.orc_wf_free.argtypes = [C.c_void_p]
        L.orc_wf_ed.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p]
        L.orc_ed_bruteforce.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_void_p]
        _lib = L
    return _lib


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


def pack_reads(reads):
    """list of bytes -> (uint8 concatenation, uint64 offsets[n+1])"""
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    seq = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    if seq.size == 0:
        seq = np.zeros(1, np.uint8)
    return seq, off


SCAN_FIELDS = ["hoco_l", "n_scm", "n_lrl", "n_nn", "hoco_s", "ho_rl", "ho_l_rl", "n_nucl", "m_pos", "s_mer", "k_mer"]


def scan_raw(seq, off, K, S, mode=0):
    L = lib()
    n = len(off) - 1
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    p = L.orc_scan_batch(seq.ctypes.data, off.ctypes.data, n, K, S, mode)
    r = p.contents
    out = {
        "hoco_l": _arr(r.hoco_l, n, np.uint32), "n_scm": _arr(r.n_scm, n, np.uint32),
        "n_lrl": _arr(r.n_lrl, n, np.uint32), "n_nn": _arr(r.n_nn, n, np.uint32),
        "hoco_s": _arr(r.hoco_s, r.tot_bytes, np.uint8), "ho_rl": _arr(r.ho_rl, r.tot_hoco, np.uint8),
        "ho_l_rl": _arr(r.ho_l_rl, r.tot_lrl, np.uint32), "n_nucl": _arr(r.n_nucl, r.tot_nn, np.uint32),
        "m_pos": _arr(r.m_pos, r.tot_scm, np.uint32), "s_mer": _arr(r.s_mer, r.tot_scm, np.uint64),
        "k_mer": _arr(r.k_mer, r.tot_scm, np.uint64),
    }
    return out, p


def scan(reads, K, S, mode=0):
    seq, off = pack_reads(reads)
    out, p = scan_raw(seq, off, K, S, mode)
    lib().orc_scan_free(p)
    return out


def scan_and_count(reads, K, S, mode=0):
    seq, off = pack_reads(reads)
    return scan_and_count_raw(seq, off, K, S, mode)


def scan_and_count_raw(seq, off, K, S, mode=0):
    L = lib()
    out, p = scan_raw(seq, off, K, S, mode)
    cp = L.orc_count(p, K)
    c = cp.contents
    cnt = {
        "n_scm": int(c.n_scm), "err": int(c.err),
        "h": _arr(c.h, c.n_scm, np.uint64), "s": _arr(c.s, c.n_scm, np.uint64), "cov": _arr(c.cov, c.n_scm, np.uint32),
        "occ_off": _arr(c.occ_off, c.n_scm + 1, np.uint64) if c.n_scm else np.zeros(1, np.uint64),
        "occ": _arr(c.occ, c.tot_occ if c.n_scm else 0, np.uint64),
        "k_id": _arr(c.k_id, c.tot_occ if c.n_scm else 0, np.uint64),
    }
    L.orc_count_free(cp)
    L.orc_scan_free(p)
    return out, cnt


def wf_ed(ts: bytes, qs: bytes, bw: int):
    out = (C.c_int32 * 3)()
    lib().orc_wf_ed(len(ts), ts, len(qs), qs, bw, out)
    return tuple(out)


def ed_bruteforce(ts: bytes, qs: bytes):
    out = (C.c_int32 * 3)()
    lib().orc_ed_bruteforce(len(ts), ts, len(qs), qs, out)
    return tuple(out)


class Wavefront:
    """resumable wavefront: step(qs) with qs extending the previous query"""

    def __init__(self, ts: bytes, bw: int):
        self._h = lib().orc_wf_new(ts, len(ts), bw)

    def step(self, qs: bytes):
        out = (C.c_int32 * 3)()
        lib().orc_wf_step(self._h, qs, len(qs), out)
        return tuple(out)

    def close(self):
        if self._h:
            lib().orc_wf_free(self._h)
            self._h = None

    def __del__(self):
        self.close()
