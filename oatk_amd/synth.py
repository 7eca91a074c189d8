"""Synthetic HiFi read sets through liboatk_host.so's counter-based generator."""
import ctypes as C
import os

import numpy as np

from . import _lib

_host = None


class SynthT(C.Structure):
    _fields_ = [("genome_len", C.c_uint64), ("n_reads", C.c_uint64), ("genome_seed", C.c_uint64),
                ("reads_seed", C.c_uint64), ("mean_len", C.c_uint64), ("err_ppm", C.c_uint64)]


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(_lib.HOST_LIB_PATH):
            raise _lib.OatkHipError("%s is missing: build it with __graft_entry__.build()" % _lib.HOST_LIB_PATH)
        L = C.CDLL(_lib.HOST_LIB_PATH)
        L.oatk_synth_genome.argtypes = [C.POINTER(SynthT), C.c_void_p]
        L.oatk_synth_lengths.argtypes = [C.POINTER(SynthT), C.c_uint64, C.c_uint64, C.c_void_p]
        L.oatk_synth_reads.argtypes = [C.POINTER(SynthT), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        _host = L
    return _host


# BASELINE.json configs
CONFIGS = {
    "config2": dict(genome_len=1_000_000, n_reads=200_000, mean_len=15_000, min_k_cov=30),
    "config3": dict(genome_len=5_000_000, n_reads=2_000_000, mean_len=15_000, min_k_cov=30),
    "config5": dict(genome_len=5_000_000, n_reads=10_000_000, mean_len=20_000, min_k_cov=150),
}


class ReadSet:
    """genome + parameters; `slice(first, count)` materialises reads [first, first+count) as a packed read stream"""

    def __init__(self, genome_len, n_reads, mean_len, genome_seed=1001, reads_seed=31, err_ppm=500, **_):
        self.p = SynthT(genome_len, n_reads, genome_seed, reads_seed, mean_len, err_ppm)
        self.n_reads = n_reads
        self.genome = np.zeros(genome_len, dtype=np.uint8)
        host_lib().oatk_synth_genome(C.byref(self.p), self.genome.ctypes.data)

    def lengths(self, first, count):
        lens = np.zeros(count, dtype=np.uint32)
        host_lib().oatk_synth_lengths(C.byref(self.p), first, count, lens.ctypes.data)
        return lens

    def layout(self, first, count):
        lens = self.lengths(first, count)
        padded = (lens.astype(np.uint64) + (_lib.READ_ALIGN - 1)) // _lib.READ_ALIGN * _lib.READ_ALIGN
        off = np.zeros(count, dtype=np.uint64)
        if count > 1:
            off[1:] = np.cumsum(padded[:-1], dtype=np.uint64)
        total = int(padded.sum())
        return lens, off, total

    def slice(self, first, count, out=None, threads=None):
        """-> (seq uint8[total], off uint64[count], len uint32[count]); `out` may be a preallocated (pinned) buffer"""
        lens, off, total = self.layout(first, count)
        if out is None:
            out = np.empty(max(total, _lib.READ_ALIGN), dtype=np.uint8)   # padding between reads is never interpreted
        assert out.size >= total
        if threads is None:
            threads = min(64, os.cpu_count() or 1)
        host_lib().oatk_synth_reads(C.byref(self.p), self.genome.ctypes.data, first, count, off.ctypes.data, out.ctypes.data, threads)
        return out, off, lens

    def as_list(self, first, count):
        seq, off, lens = self.slice(first, count)
        return [seq[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)]
