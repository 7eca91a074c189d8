"""CPU: the synthetic HiFi generator is deterministic, counter-based and matches its spec."""
import numpy as np

from oatk_amd.synth import ReadSet


def test_slices_are_consistent_and_deterministic():
    rs = ReadSet(genome_len=50_000, n_reads=500, mean_len=5_000)
    a = rs.as_list(0, 40)
    b = rs.as_list(10, 20)
    assert a[10:30] == b
    rs2 = ReadSet(genome_len=50_000, n_reads=500, mean_len=5_000)
    assert rs2.as_list(0, 40) == a
    seq, off, lens = rs.slice(0, 40, threads=3)
    assert all(int(o) % 64 == 0 for o in off)
    assert [seq[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)] == a


def test_read_statistics():
    rs = ReadSet(genome_len=200_000, n_reads=4000, mean_len=15_000)
    lens = rs.lengths(0, 4000)
    assert 14_800 < lens.mean() < 15_200
    assert 1_300 < lens.std() < 1_700
    assert lens.min() >= 2000 and lens.max() <= 30_000
    reads = rs.as_list(0, 50)
    assert all(set(r) <= set(b"ACGT") for r in reads)
    # reads come from both strands of the genome: an exact 40-mer of most reads occurs in genome or its reverse complement
    g = bytes(b"ACGT"[c] for c in rs.genome)
    gg = g + g[:100]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rc = gg.translate(comp)[::-1]
    fwd = sum(1 for r in reads if r[100:140] in gg)
    rev = sum(1 for r in reads if r[100:140] in rc)
    assert fwd + rev >= 45 and fwd > 5 and rev > 5
