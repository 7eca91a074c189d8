"""Deterministic adversarial / edge-case reads for parity tests (never timed).

Covers the cases SURVEY.md 8c lists: N next to syncmer boundaries, homopolymers > 256, di-/tri-nucleotide
and longer tandem repeats (s-mer ties, first == last s-mer), reads shorter than K, lower case and U,
empty reads, reverse-complement pairs, palindromes.
"""
import numpy as np

_COMP = bytes.maketrans(b"ACGTacgtUuNn", b"TGCAtgcaAaNn")


def revcomp(s: bytes) -> bytes:
    return s.translate(_COMP)[::-1]


def rand_dna(rng, n, alphabet=b"ACGT"):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return a[rng.integers(0, len(a), size=n)].tobytes()


def rand_nohp(rng, n):
    """random DNA with no two equal neighbours (HPC leaves it unchanged)"""
    out = bytearray(n)
    prev = 255
    for i in range(n):
        c = int(rng.integers(0, 4))
        if c == prev:
            c = (c + 1 + int(rng.integers(0, 3))) & 3
        out[i] = b"ACGT"[c]
        prev = c
    return bytes(out)


def reads(K, S, seed=7, scale=1.0):
    rng = np.random.default_rng(seed)
    L = int(max(4 * K, 600) * scale)
    out = []
    out.append(rand_dna(rng, L))
    base = rand_dna(rng, 3 * L)
    out.append(base)
    out.append(revcomp(base))                                  # strand symmetry
    out.append(base.lower())                                   # lower case
    out.append(base.replace(b"T", b"U")[: 2 * L])              # U == T
    # N placed around k-mer boundaries
    for gap in (0, 1, S - 1, S, K - 1, K, K + 1, 2 * K):
        r = bytearray(rand_dna(rng, 3 * K + 50))
        p = K + (gap % (K + 7))
        r[p] = ord("N")
        if gap & 1:
            r[min(len(r) - 1, p + K + 1)] = ord("n")
        out.append(bytes(r))
    r = bytearray(rand_nohp(rng, 3 * K + 10))                  # N exactly after a full k-mer, HPC-neutral
    r[K] = ord("N")
    out.append(bytes(r))
    r = bytearray(rand_nohp(rng, 3 * K + 10))
    r[2 * K + 3] = ord("R")
    r[2 * K + 4] = ord("Y")
    out.append(bytes(r))
    out.append(b"N" * 17 + rand_dna(rng, 2 * K + 30) + b"NNN")  # N runs at both ends
    out.append(b"N" * (K + 5))
    # homopolymers
    out.append(b"A" * 400 + rand_dna(rng, 2 * K) + b"C" * 256 + b"G" * 255 + b"T" * 257 + rand_dna(rng, K))
    out.append(b"A" * (3 * K))
    out.append(b"a" * 300 + b"A" * 300 + rand_nohp(rng, 2 * K + 5))
    # tandem repeats (ties between s-mers; first == last s-mer)
    for unit in (b"AC", b"ACG", b"ACGT", b"AACCGT", rand_nohp(rng, 8), rand_nohp(rng, 13), rand_nohp(rng, S),
                 rand_nohp(rng, S + 1), rand_nohp(rng, max(2, K - S)), rand_nohp(rng, K - S + 1), rand_nohp(rng, K), rand_nohp(rng, 140)):
        reps = (3 * K + 200) // len(unit) + 2
        out.append((unit * reps)[: 3 * K + 200])
        out.append(rand_nohp(rng, 57) + (unit * reps)[: 2 * K + 77] + rand_nohp(rng, K + 3))
    # palindromic stretch (even S can make an s-mer its own reverse complement)
    half = rand_nohp(rng, K)
    out.append(half + revcomp(half))
    out.append(b"ACGT" * 5 + rand_dna(rng, 10) + b"AATT" * (K // 2))
    # lengths around the thresholds
    for n in (0, 1, 3, 4, 5, S - 1, S, S + 1, K - 1, K, K + 1, K + 2, K + S):
        out.append(rand_nohp(rng, n))
    out.append(rand_dna(rng, K + 1))
    # control bytes 1..3 are legal codes in the reference's table (syncmer.c:47-48)
    out.append(rand_nohp(rng, K + 40) + bytes([1, 2, 3, 2, 1]) + rand_nohp(rng, K + 40))
    # long mixed read
    out.append(rand_dna(rng, 5 * L, b"ACGTACGTACGTN"))
    out.append(rand_dna(rng, 6 * L))
    return out


def byte_soup(K, seed=5):
    """every byte value next to every other: bytes that share their low bits or their case-folded form with a base ('!' 0x21,
    0xC1, 'I', 'Q', 'W', 'E', 0x01 .. 0x03 which ARE bases in the reference's table), at every offset inside a 16-byte vector and
    across vector, wave (1 KiB) and tile (4 KiB) boundaries; long stretches of clean bases in between so that k-mers exist"""
    rng = np.random.default_rng(seed)
    out = [bytes(rng.integers(0, 256, 3 * K + 100).astype(np.uint8).tolist())]
    alias = bytes([0x21, 0xC1, 0x49, 0x51, 0x57, 0x45, 0x01, 0x02, 0x03, 0x00, 0x63, 0x67, 0x74, 0x75, 0x55, 0x81, 0xE1, 0xD4, 0x14, 0x07])
    for step in (1, 7, 16, 17, 63, 64, 255, 1023, 1024, 1025, 4095, 4096, 4097):
        r = bytearray(rand_dna(rng, 9000 + 2 * K, b"ACGTacgtU"))
        for j, p in enumerate(range(step, len(r), max(step, 13) * 3 + 1)):
            r[p] = alias[j % len(alias)]
        out.append(bytes(r))
    for pos in (0, 15, 16, 1023, 1024, 4095, 4096, 4097, 8191, 8192):        # one odd byte, clean everywhere else
        for c in (0x4E, 0x01, 0xC1, 0x21):
            r = bytearray(rand_dna(rng, 8300 + K))
            r[pos] = c
            out.append(bytes(r))
    out.append(rand_dna(rng, 4096) + b"A" * 5000 + rand_dna(rng, K + 77))      # a run across a whole tile of clean lanes
    out.append(rand_dna(rng, 1000) + b"C" * 300 + b"N" + b"C" * 300 + rand_dna(rng, K + 5))
    return out


def hifi_like(n_reads, genome_len, mean_len, seed=11, err=0.0005):
    """small numpy model of HiFi sampling used by the CPU-side tests (the product generator is oatk_amd.synth)"""
    rng = np.random.default_rng(seed)
    genome = rand_dna(rng, genome_len)
    gg = genome + genome
    out = []
    for _ in range(n_reads):
        ln = int(np.clip(rng.normal(mean_len, 0.1 * mean_len), 200, min(genome_len, 2 * mean_len)))
        st = int(rng.integers(0, genome_len))
        r = bytearray(gg[st:st + ln])
        ne = rng.binomial(ln, err)
        for p in sorted(rng.integers(0, ln, size=ne).tolist(), reverse=True):
            kind = int(rng.integers(0, 3))
            if kind == 0:
                r[p] = b"ACGT"[(b"ACGT".index(bytes([r[p]])) + 1 + int(rng.integers(0, 3))) & 3]
            elif kind == 1:
                r.insert(p, b"ACGT"[int(rng.integers(0, 4))])
            else:
                del r[p]
        r = bytes(r)
        if rng.integers(0, 2):
            r = revcomp(r)
        out.append(r)
    return out


def tandem_repeat_reads(K, n_reads=240, flank=20000, unit_len=47, mean_len=6000, seed=256, err=0.0):
    """reads of both orientations over a genome that holds a PERFECT tandem repeat longer than K + period.  Inside it the same
    k-mer recurs every period, so a syncmer is adjacent to itself -- on forward reads as (v+, v+), on reverse ones as (v-, v-); the reference's
    arc counter keeps those as two keys (syncasm.c:256-257 canonicalises by v0 <= v1 only) and emits each with its complement: duplicate
    (v, w) arcs whose order is that of its hash table (syncasm.c:264-282, graph.c:70-83)."""
    rng = np.random.default_rng(seed)
    unit = rand_nohp(rng, unit_len)
    while unit[0] == unit[-1]:
        unit = rand_nohp(rng, unit_len)
    copies = (2 * K + 6 * unit_len) // unit_len + 4
    genome = rand_dna(rng, flank) + unit * copies + rand_dna(rng, flank)
    rep0, rep1 = flank, flank + unit_len * copies
    out = []
    for i in range(n_reads):
        ln = int(np.clip(rng.normal(mean_len, 0.1 * mean_len), 2 * K, len(genome)))
        # two thirds of the reads are laid across the repeat, the rest anywhere
        st = int(rng.integers(max(0, rep0 - ln + K), min(len(genome) - ln, rep1 - K))) if i % 3 else int(rng.integers(0, len(genome) - ln))
        r = bytearray(genome[st:st + ln])
        for p in sorted(rng.integers(0, ln, size=rng.binomial(ln, err)).tolist(), reverse=True):
            r[p] = b"ACGT"[(b"ACGT".index(bytes([r[p]])) + 1 + int(rng.integers(0, 3))) & 3]
        r = bytes(r)
        out.append(revcomp(r) if i % 2 else r)
    return out


# ---- s-mers whose hashes share their TOP WORD but not their low word (r03h) ----
# The fast scan kernel keeps the top 32 bits of every s-mer hash and hashes a position again when top words tie.  On random sequence two
# different s-mers tie with probability 2^-32, so such pairs are built: the mixing function (syncmer.c:116-126) is a bijection on 2S bits and
# is inverted here step by step.

def hash64_py(x, bits):
    M = (1 << bits) - 1
    x = (~x + (x << 21)) & M
    x ^= x >> 24
    x = (x + (x << 3) + (x << 8)) & M
    x ^= x >> 14
    x = (x + (x << 2) + (x << 4)) & M
    x ^= x >> 28
    x = (x + (x << 31)) & M
    return x


def _unxorshift(y, s, bits):
    x = y
    for _ in range(bits // s + 1):
        x = y ^ (x >> s)
    return x


def hash64_inverse(h, bits):
    N = 1 << bits
    x = h * pow((1 << 31) + 1, -1, N) % N
    x = _unxorshift(x, 28, bits)
    x = x * pow(21, -1, N) % N
    x = _unxorshift(x, 14, bits)
    x = x * pow(265, -1, N) % N
    x = _unxorshift(x, 24, bits)
    return (x + 1) * pow((1 << 21) - 1, -1, N) % N      # (~x + (x << 21)) = x (2^21 - 1) - 1


def _code_to_smer(code, S):
    return bytes(b"ACGT"[(code >> (2 * (S - 1 - i))) & 3] for i in range(S))


def _rc_code(code, S):
    r = 0
    for i in range(S):
        r = r << 2 | (3 - ((code >> (2 * i)) & 3))
    return r


def smers_with_top_word(top, n, S=31, start_low=0):
    """n s-mers (no two equal neighbours, forward strand canonical) whose hashes are top << 32 | low, lows ascending"""
    out, low = [], start_low
    while len(out) < n:
        low += 1
        code = hash64_inverse(top << 32 | low, 2 * S)
        f = [(code >> (2 * i)) & 3 for i in range(S)]
        if any(f[i] == f[i + 1] for i in range(S - 1)) or code >= _rc_code(code, S):
            continue
        assert hash64_py(code, 2 * S) == (top << 32 | low)
        out.append((low, _code_to_smer(code, S)))
    return out


def _implant(rng, n, items):
    """homopolymer-free random read of n bases with the given (position, string) pairs written in; neighbours fixed so no run forms"""
    b = bytearray(rand_nohp(rng, n))
    for pos, sm in items:
        b[pos:pos + len(sm)] = sm
    for pos, sm in items:
        for j in (pos - 1, pos + len(sm)):
            if 0 <= j < n:
                taken = {b[j - 1] if j > 0 else 0, b[j + 1] if j + 1 < n else 0}
                if b[j] in taken:
                    b[j] = next(c for c in b"ACGT" if c not in taken)
    for i in range(n - 1):
        assert b[i] != b[i + 1]
    return bytes(b)


def top_word_tie_reads(K, S=31, seed=99):
    """reads in which window minima tie on their top word only: the pair at every distance around the window length, in both orders, with an exact
    duplicate between them, on either strand, and next to the start of the read"""
    rng = np.random.default_rng(seed)
    w = K - S
    sm = smers_with_top_word(3, 4, S)                      # four s-mers, hashes 3 << 32 | low, lows ascending
    a, b_, c, d = (s for _, s in sm)
    reads = []
    for dist in (1, 5, 8, 33, 400, w - 9, w - 8, w - 7, w - 1, w, w + 1, w + 7, w + 8, w + 9, 2 * w - 3):
        for first, second in ((a, b_), (b_, a), (a, a), (c, revcomp(d)), (revcomp(d), c)):
            for p0 in (2600, 2048 - 15):
                reads.append(_implant(rng, 7000, [(p0, first), (p0 + dist + (S if dist < S else 0), second)]))
    for p0 in (3000, 4090):                                 # three in a window: smaller, equal, larger lows in every order
        for trio in ((a, b_, a), (b_, a, b_), (b_, c, a), (a, a, a), (d, c, b_)):
            reads.append(_implant(rng, 9000, [(p0, trio[0]), (p0 + 300, trio[1]), (p0 + 300 + w - 4, trio[2])]))
    reads.append(_implant(rng, 5000, [(0, b_), (w - 2, a), (2 * w, b_)]))
    return reads
