"""Synthetic HiFi read sets through liboatk_host.so's counter-based generator."""
import ctypes as C
import os

import numpy as np

from . import _lib

_host = None


class SynthT(C.Structure):
    _fields_ = [("genome_len", C.c_uint64), ("n_reads", C.c_uint64), ("genome_seed", C.c_uint64),
                ("reads_seed", C.c_uint64), ("mean_len", C.c_uint64), ("err_ppm", C.c_uint64)]


MAX_COMP = 8


class SynthMixT(C.Structure):
    _fields_ = [("n_comp", C.c_uint64), ("genome", C.c_void_p * MAX_COMP), ("genome_len", C.c_uint64 * MAX_COMP), ("cum_ppm", C.c_uint64 * MAX_COMP),
                ("reads_seed", C.c_uint64), ("mean_len", C.c_uint64), ("err_ppm", C.c_uint64), ("short_ppm", C.c_uint64), ("short_max", C.c_uint64),
                ("n_ppm", C.c_uint64), ("lower_ppm", C.c_uint64)]


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(_lib.HOST_LIB_PATH):
            raise _lib.OatkHipError("%s is missing: build it with __graft_entry__.build()" % _lib.HOST_LIB_PATH)
        L = C.CDLL(_lib.HOST_LIB_PATH)
        L.oatk_synth_genome.argtypes = [C.POINTER(SynthT), C.c_void_p]
        L.oatk_synth_lengths.argtypes = [C.POINTER(SynthT), C.c_uint64, C.c_uint64, C.c_void_p]
        L.oatk_synth_reads.argtypes = [C.POINTER(SynthT), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        L.oatk_synth_mix_lengths.argtypes = [C.POINTER(SynthMixT), C.c_uint64, C.c_uint64, C.c_void_p]
        L.oatk_synth_mix_reads.argtypes = [C.POINTER(SynthMixT), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        L.oatk_write_fasta.restype = C.c_int
        L.oatk_write_fasta.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int]
        _host = L
    return _host


# BASELINE.json configs
CONFIGS = {
    "config2": dict(genome_len=1_000_000, n_reads=200_000, mean_len=15_000, min_k_cov=30),
    "config3": dict(genome_len=5_000_000, n_reads=2_000_000, mean_len=15_000, min_k_cov=30),
    "config5": dict(genome_len=5_000_000, n_reads=10_000_000, mean_len=20_000, min_k_cov=150),
}

# A stand-in for BASELINE.json configs[0] (ddAraThal4_organelle.hifi.fa.gz is not available offline): the SHAPE of an organelle HiFi data set.
# A 154 kb "plastid" (large inverted repeat) and a 368 kb "mitochondrion" (direct repeats) at >= 1000x inside a 256 Mb "nuclear" background at
# ~7x, so that > 90 % of the distinct syncmers sit below -c 30 (find_error_syncmers marks them deleted, syncerr.c:679-757); homopolymers beyond
# 256 (ho_l_rl, syncmer.c:301-304), telomere / microsatellite / satellite arrays of 2-12 kb (s-mer ties, first == last), organelle pieces inside
# the nuclear genome; 1 % of the reads shorter than K, 0.3 % with runs of N (syncmer.c:316-323), 0.2 % in lower case.  Written as .fa.gz.
CONFIG1S = dict(n_reads=200_000, mean_len=15_000, min_k_cov=30, nuclear_len=256_000_000, plastid_len=154_000, mito_len=368_000,
                plastid_ppm=170_000, mito_ppm=210_000, short_ppm=10_000, short_max=1000, n_ppm=3000, lower_ppm=2000)


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


class _SM64:
    """splitmix64, the generator of host/synth.c: the genomes' features are placed from it so they do not depend on numpy's generators"""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n


_CODE = {65: 0, 67: 1, 71: 2, 84: 3}


def _codes(b):
    return np.array([_CODE[c] for c in b], dtype=np.uint8)


def _low_complexity(g, rng, n_homopolymer, n_tandem):
    """homopolymer runs of 20-600 bases and tandem arrays of 2-12 kb written over g at random places"""
    units = [_codes(b"TTAGGG"), _codes(b"AC"), _codes(b"GA"), _codes(b"AAT"), _codes(b"ACGTTGCAAGT"), None, None]
    G = len(g)
    for _ in range(n_homopolymer):
        l = 20 + rng.below(581)
        a = rng.below(G - l)
        g[a:a + l] = rng.below(4)
    for _ in range(n_tandem):
        u = units[rng.below(len(units))]
        if u is None:                              # a satellite: a random unit of 37 or 171 bases
            ul = 37 if rng.below(2) else 171
            u = np.array([rng.below(4) for _ in range(ul)], dtype=np.uint8)
        l = 2000 + rng.below(10001)
        a = rng.below(G - l)
        g[a:a + l] = np.resize(u, l)


class MixReadSet:
    """reads from several genomes at once (host/synth.c: oatk_synth_mix_t); the interface of ReadSet"""

    def __init__(self, n_reads, mean_len, nuclear_len, plastid_len, mito_len, plastid_ppm, mito_ppm, short_ppm, short_max, n_ppm, lower_ppm,
                 genome_seed=1001, reads_seed=31, err_ppm=500, **_):
        H = host_lib()
        self.n_reads = n_reads
        self.genomes = []
        for i, gl in enumerate((plastid_len, mito_len, nuclear_len)):
            g = np.zeros(gl, dtype=np.uint8)
            q = SynthT(gl, 0, genome_seed + 7919 * (i + 1), 0, mean_len, 0)
            H.oatk_synth_genome(C.byref(q), g.ctypes.data)
            self.genomes.append(g)
        pl, mt, nu = self.genomes
        rng = _SM64(genome_seed * 31 + 5)
        # plastid: LSC | IRa | SSC | IRb, IRb = reverse complement of IRa (a sixth of the genome each)
        ir = plastid_len // 6
        a0 = plastid_len - 2 * ir - plastid_len // 9
        pl[plastid_len - ir:] = 3 - pl[a0:a0 + ir][::-1]
        # mitochondrion: two direct repeats of 6 kb and one of 1.5 kb
        for l in (6000, 6000, 1500):
            a, b = rng.below(mito_len // 2 - l), mito_len // 2 + rng.below(mito_len // 2 - l)
            mt[b:b + l] = mt[a:a + l]
        _low_complexity(pl, rng, 3, 0)
        _low_complexity(mt, rng, 4, 2)
        _low_complexity(nu, rng, max(4, nuclear_len // 200_000), max(2, nuclear_len // 500_000))
        # organelle pieces inside the nuclear genome (NUPTs / NUMTs), 2 % diverged
        for _ in range(max(2, nuclear_len // 8_000_000)):
            src = pl if rng.below(2) else mt
            l = 1500 + rng.below(9000)
            a, b = rng.below(len(src) - l), rng.below(nuclear_len - l)
            piece = src[a:a + l].copy()
            for _k in range(l // 50):
                piece[rng.below(l)] = rng.below(4)
            nu[b:b + l] = piece
        self.p = SynthMixT()
        self.p.n_comp = 3
        for i, g in enumerate(self.genomes):
            self.p.genome[i] = g.ctypes.data
            self.p.genome_len[i] = len(g)
        self.p.cum_ppm[0], self.p.cum_ppm[1], self.p.cum_ppm[2] = plastid_ppm, plastid_ppm + mito_ppm, 1_000_000
        self.p.reads_seed, self.p.mean_len, self.p.err_ppm = reads_seed, mean_len, err_ppm
        self.p.short_ppm, self.p.short_max, self.p.n_ppm, self.p.lower_ppm = short_ppm, short_max, n_ppm, lower_ppm

    def lengths(self, first, count):
        lens = np.zeros(count, dtype=np.uint32)
        host_lib().oatk_synth_mix_lengths(C.byref(self.p), first, count, lens.ctypes.data)
        return lens

    layout = ReadSet.layout

    def slice(self, first, count, out=None, threads=None):
        lens, off, total = self.layout(first, count)
        if out is None:
            out = np.empty(max(total, _lib.READ_ALIGN), dtype=np.uint8)
        assert out.size >= total
        if threads is None:
            threads = min(64, os.cpu_count() or 1)
        host_lib().oatk_synth_mix_reads(C.byref(self.p), first, count, off.ctypes.data, out.ctypes.data, threads)
        return out, off, lens

    as_list = ReadSet.as_list


FA_PLAIN, FA_GZ, FA_BGZF, FA_GZ_MEMBERS = 0, 1, 2, 3


def write_fasta(path, seq, off, lens, first_id=0, mode=FA_PLAIN, level=1, member_bytes=0, threads=None):
    """the packed read stream as a FASTA file `>r<i>`: plain, one gzip member, BGZF, or several gzip members (host/fasta_out.c)"""
    if threads is None:
        threads = min(64, os.cpu_count() or 1)
    off = np.ascontiguousarray(off, np.uint64)
    lens = np.ascontiguousarray(lens, np.uint32)
    rc = host_lib().oatk_write_fasta(path.encode(), seq.ctypes.data, off.ctypes.data, lens.ctypes.data, len(lens), first_id, mode, level, member_bytes, threads)
    if rc:
        raise OSError("writing %s failed" % path)
