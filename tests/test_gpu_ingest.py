"""GPU: the device-side FASTA / FASTQ record scan (oatk_hip_ingest) -- the reads it finds equal kseq's reading of the same text
(restated in kseq_like below and, end to end, the compiled reference's sr_read of the same FILE), for wrapped and unwrapped FASTA,
CRLF, blank lines, junk before the first header, a last line without newline, four-line FASTQ with quality lines that begin
with '@' / '+', and a text fed in chunks."""
import numpy as np
import pytest

import adversarial as A
import ref_lib as R
from oatk_amd import OatkHipError, pack_reads

pytestmark = pytest.mark.gpu


def kseq_like(text: bytes):
    """(name, seq) records the way kseq_read (kseq.h:192-235) reads FASTA / four-line FASTQ"""
    out, lines, i = [], text.split(b"\n"), 0
    if lines and lines[-1] == b"":
        lines.pop()
    lines = [ln[:-1] if ln.endswith(b"\r") else ln for ln in lines]
    while i < len(lines) and not (lines[i][:1] in (b">", b"@")):
        i += 1
    while i < len(lines):
        name, fq = lines[i][1:].split()[0] if lines[i][1:].split() else b"", lines[i][:1] == b"@"
        i += 1
        seq = b""
        while i < len(lines) and lines[i][:1] not in (b">", b"@", b"+"):
            seq += lines[i]
            i += 1
        if i < len(lines) and lines[i][:1] == b"+":
            i += 1
            q = 0
            while i < len(lines) and q < len(seq):
                q += len(lines[i])
                i += 1
        out.append((name, seq))
    return out


def device_reads(hip, text, fmt=0):
    n, used = hip.ingest_host(text, fmt, True)
    seq, off, lens = hip.fetch("INGEST_SEQ"), hip.fetch("INGEST_OFF"), hip.fetch("INGEST_LEN")
    assert n == len(off) == len(lens) and used == len(text)
    assert np.all(off % 64 == 0)
    return [seq[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)]


def fasta(reads, width=0, eol=b"\n", last_eol=True, blank_every=0, head=b">"):
    t = b""
    for i, r in enumerate(reads):
        t += head + b"read%d some comment" % i + eol
        body = [r[j:j + width] for j in range(0, len(r), width)] if width else [r]
        for j, ln in enumerate(body):
            t += ln + eol
            if blank_every and j % blank_every == 1:
                t += eol
    return t if last_eol else t[:-len(eol)]


def fastq(reads, eol=b"\n", seed=0):
    rng = np.random.default_rng(seed)
    t = b""
    for i, r in enumerate(reads):
        q = bytes(rng.choice(np.frombuffer(b"@+>I5~!", np.uint8), len(r)).tolist())       # quality lines starting with '@', '+', '>'
        t += b"@m84/%d/ccs" % i + eol + r + eol + b"+" + eol + q + eol
    return t


READS = A.hifi_like(40, 20000, 3000, seed=11) + [b"acgtnACGTN" * 30, b"A" * 700, b"C"]


@pytest.mark.parametrize("variant", ["plain", "wrap60", "wrap70_crlf", "blank_lines", "no_last_eol", "junk_first", "at_headers"])
def test_fasta_variants(hip, variant):
    t = {"plain": lambda: fasta(READS), "wrap60": lambda: fasta(READS, 60), "wrap70_crlf": lambda: fasta(READS, 70, b"\r\n"),
         "blank_lines": lambda: fasta(READS, 50, blank_every=3), "no_last_eol": lambda: fasta(READS, 80, last_eol=False),
         "junk_first": lambda: b"\n\n# produced by something\n" + fasta(READS, 61), "at_headers": lambda: fasta(READS, 0, head=b"@")}[variant]()
    got = device_reads(hip, t, 1)                       # FASTA stated: a file of '@' headers is FASTA to kseq if no '+' line follows
    want = [s for _, s in kseq_like(t)]
    assert got == want and got == READS


def test_fastq_four_line(hip):
    for eol in (b"\n", b"\r\n"):
        t = fastq(READS, eol)
        assert device_reads(hip, t) == READS == [s for _, s in kseq_like(t)]
    with pytest.raises(OatkHipError):                   # wrapped FASTQ is refused, not misread
        bad = b"@r0\nACGTACGT\nACGT\n+\nIIIIIIII\nIIII\n@r1\nAC\n+\nII\n@r2\nAC\n+\nII\n"
        hip.ingest_host(bad, 2, True)
    # behind the last whole record: blank lines are fine, a truncated record is refused, never dropped silently
    t = fastq(READS[:5])
    assert device_reads(hip, t + b"\n\r\n") == READS[:5]
    for tail in (b"@r9\nACGT\n+\n", b"@r9\nACGT\n", b"junk"):
        with pytest.raises(OatkHipError):
            hip.ingest_host(t + tail, 2, True)


@pytest.mark.parametrize("kind", ["fasta", "fastq"])
def test_chunked_text_gives_the_same_reads(hip, kind):
    t = fasta(READS, 64) if kind == "fasta" else fastq(READS)
    rng = np.random.default_rng(5)
    got, pos, carry = [], 0, b""
    cuts = sorted(rng.integers(1, len(t) - 1, 9).tolist()) + [len(t)]
    for c in cuts:
        chunk = carry + t[pos:c]
        final = c == len(t)
        n, used = hip.ingest_host(chunk, 1 if kind == "fasta" else 2, final)
        seq, off, lens = hip.fetch("INGEST_SEQ"), hip.fetch("INGEST_OFF"), hip.fetch("INGEST_LEN")
        got += [seq[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)]
        carry, pos = chunk[used:], c
        assert final or used <= len(chunk)
    assert got == READS


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
def test_file_through_device_ingest_equals_reference_sr_read(hip, tmp_path):
    """the FILE the reference reads with kseq, read by the device instead: scan results identical member by member"""
    K, S = 301, 21
    reads = A.hifi_like(120, 30000, 4000, seed=21)
    path = str(tmp_path / "reads.fa")
    with open(path, "wb") as f:
        f.write(fasta(reads, 70))
    db = R.SrDb([path], K, S, 2)
    want = db.flatten()
    n, _ = hip.ingest_host(open(path, "rb").read())
    assert n == len(reads)
    hip.scan_ingested(K, S)
    got = hip.fetch_scan(hip.fetch("INGEST_OFF"))
    for f in ["hoco_l", "n_scm", "hoco_s", "ho_rl", "ho_l_rl", "m_pos", "s_mer", "k_mer"]:
        assert np.array_equal(got[f], want[f]), f
    # and the same as scanning the reads handed over as a packed stream
    seq, off, lens = pack_reads(reads)
    hip.scan_host(seq, off, lens, K, S)
    again = hip.fetch_scan(off)
    for f in ["hoco_l", "n_scm", "m_pos", "s_mer", "k_mer"]:
        assert np.array_equal(again[f], got[f]), f
    db.close()


def wrapped_mixed(reads, eol=b"\n", seed=3):
    """records of every kind kseq reads: FASTA (one line, wrapped), four-line FASTQ, FASTQ with sequence AND quality over several lines whose quality
    lines start with '@', '>' and '+', blank lines inside sequences, text between a finished FASTQ record and the next header"""
    rng = np.random.default_rng(seed)
    t = b"text before the first header" + eol + eol
    for i, r in enumerate(reads):
        kind = i % 4
        if kind == 0:
            t += b">fa%d one line" % i + eol + r + eol
        elif kind == 1:
            w = int(rng.integers(1, 200))
            t += b">fw%d" % i + eol + eol.join(r[j:j + w] for j in range(0, len(r), w)) + eol + eol
        elif kind == 2:
            t += b"@fq%d four lines" % i + eol + r + eol + b"+" + eol + bytes([64]) * len(r) + eol + b"skipped text" + eol
        else:
            a, b = len(r) // 3, 2 * len(r) // 3
            qual = [b"@" + b"I" * (a - 1) if a else b"", b">" * (b - a), b"+" + b"5" * (len(r) - b - 1) if len(r) - b else b""]
            t += b"@fm%d wrapped" % i + eol + r[:a] + eol + eol + r[a:b] + eol + r[b:] + eol + b"+fm%d" % i + eol + eol.join(q for q in qual) + eol
    return t


def kseq_exact(text: bytes):
    """kseq_read itself, character by character as kseq.h:192-235 is written (headers found anywhere; '+' switches to quality; quality read by whole
    lines until it is at least as long as the sequence; -2 on a length mismatch ends the stream)"""
    out, p, n, last = [], 0, len(text), 0

    def line_end(q):
        e = text.find(b"\n", q)
        return n if e < 0 else e

    while True:
        if last == 0:
            while p < n and text[p:p + 1] not in (b">", b"@"):
                p += 1
            if p >= n:
                return out
            p += 1
        e = line_end(p)                                # the header line: name up to white space
        p = min(e + 1, n)
        seq = b""
        c = b""
        while p < n:
            c = text[p:p + 1]
            if c in (b">", b"+", b"@"):
                break
            e = line_end(p)
            ln = text[p:e]
            if ln.endswith(b"\r") and len(seq) + len(ln) > 1:
                ln = ln[:-1]
            seq += ln
            p = min(e + 1, n)
            c = b""
        if c in (b">", b"@"):
            last = 1
            p += 1
        else:
            last = 0
        if c != b"+":
            out.append(seq)
            if p >= n and c == b"":
                return out
            continue
        p = min(line_end(p) + 1, n)
        q = 0
        while True:
            if p >= n:
                break
            e = line_end(p)
            ln = text[p:e]
            if ln.endswith(b"\r") and q + len(ln) > 1:
                ln = ln[:-1]
            q += len(ln)
            p = min(e + 1, n)
            if q >= len(seq):
                break
        if q != len(seq):
            return out                                 # -2: the reader stops
        out.append(seq)


@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
def test_kseq_format_reads_wrapped_fastq_and_mixed_records(hip, eol):
    """OATK_FMT_KSEQ (3): kseq's own reading line by line -- the device extracts first character, length and "a header character inside" per line, the
    classification is walked on the host, lengths / offsets / copy stay on the device -- against kseq_read restated character by character, whole and
    fed in chunks cut at random places; and the two device-only formats answer OATK_E_SPLIT (5) for such text instead of misreading it"""
    t = wrapped_mixed(READS[:37], eol)
    want = kseq_exact(t)
    assert len(want) == 37 and want == [r for r in READS[:37]]
    assert device_reads(hip, t, 3) == want
    rng = np.random.default_rng(8)
    got, pos, carry = [], 0, b""
    cuts = sorted(set(rng.integers(1, len(t) - 1, 12).tolist())) + [len(t)]
    for c in cuts:
        chunk = carry + t[pos:c]
        n, used = hip.ingest_host(chunk, 3, c == len(t))
        seq, off, lens = hip.fetch("INGEST_SEQ"), hip.fetch("INGEST_OFF"), hip.fetch("INGEST_LEN")
        got += [seq[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)]
        carry, pos = chunk[used:], c
    assert got == want
    for fmt in (1, 2):
        with pytest.raises(OatkHipError, match="code 5"):
            hip.ingest_host(t, fmt, True)
    # a quality string longer than its sequence: kseq stops there (-2), the device refuses the text
    with pytest.raises(OatkHipError):
        hip.ingest_host(b"@a" + eol + b"ACGT" + eol + b"+" + eol + b"IIIII" + eol, 3, True)


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
def test_wrapped_mixed_file_and_data_cap_equal_reference_sr_read(hip, tmp_path):
    """a file of wrapped FASTQ and FASTA records through the streamed reader (small windows) against the compiled reference's sr_read of the same file,
    without and with a data cap (syncmer.c:537-541: the read that takes the total to the cap is the last one)"""
    import ctypes as C
    from test_gpu_dropin import host_lib
    K, S = 301, 21
    reads = A.hifi_like(150, 30000, 4000, seed=23)
    path = str(tmp_path / "mixed.fq")
    with open(path, "wb") as f:
        f.write(wrapped_mixed(reads))
    H = host_lib()
    H.oatk_sr_read_files_capped.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_uint64]
    H.oatk_host_debug_window.argtypes = [C.c_uint64]
    total = sum(len(r) for r in reads)
    for cap in (0, total // 3, 1):
        db = R.SrDb([path], K, S, 2, m_data=cap)
        want = db.flatten()
        mine = H.oatk_sr_db_new(K, S)
        files = (C.c_char_p * 1)(path.encode())
        H.oatk_host_debug_window(70000)
        try:
            rc = H.oatk_sr_read_files_capped(hip.h, mine, files, 1, cap)
        finally:
            H.oatk_host_debug_window(0)
        assert rc == 0, hip.L.oatk_hip_last_error(hip.h)
        got_db = R.SrDb.__new__(R.SrDb)
        got_db.K, got_db.S, got_db._h = K, S, mine
        got = got_db.flatten()
        assert got_db.n() == db.n() and (cap == 0) == (db.n() == len(reads)) and (cap != 1 or db.n() == 1)
        for f in ["hoco_l", "n_scm", "sid", "hoco_s", "ho_rl", "m_pos", "s_mer", "k_mer"]:
            assert np.array_equal(got[f], want[f]), (cap, f)
        got_db.close()
        db.close()


def test_empty_and_headers_only(hip):
    assert hip.ingest_host(b"", 0, True) == (0, 0)
    assert device_reads(hip, b">a\n>b\nACGT\n>c\n", 1) == [b"", b"ACGT", b""]


def test_files_plain_and_gzip_through_the_host_helper(hip, tmp_path):
    """oatk_ingest_files (liboatk_host.so): several files, plain and gzip'ed, FASTA without a final newline -- one stream of records"""
    import ctypes as C
    import gzip
    from test_gpu_dropin import host_lib
    H = host_lib()
    H.oatk_ingest_files.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_uint64)]
    a, b, c = READS[:15], READS[15:30], READS[30:]
    p1, p2, p3 = str(tmp_path / "a.fa"), str(tmp_path / "b.fa.gz"), str(tmp_path / "c.fa")
    open(p1, "wb").write(fasta(a, 70, last_eol=False))
    gzip.open(p2, "wb").write(fasta(b, 0))
    open(p3, "wb").write(fasta(c, 61, b"\r\n"))
    files = (C.c_char_p * 3)(p1.encode(), p2.encode(), p3.encode())
    n = C.c_uint64()
    assert H.oatk_ingest_files(hip.h, files, 3, C.byref(n)) == 0 and n.value == len(READS)
    seq, off, lens = hip.fetch("INGEST_SEQ"), hip.fetch("INGEST_OFF"), hip.fetch("INGEST_LEN")
    assert [seq[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)] == READS
    q = str(tmp_path / "r.fastq.gz")
    gzip.open(q, "wb").write(fastq(READS))
    files = (C.c_char_p * 1)(q.encode())
    assert H.oatk_ingest_files(hip.h, files, 1, C.byref(n)) == 0 and n.value == len(READS)
    seq, off, lens = hip.fetch("INGEST_SEQ"), hip.fetch("INGEST_OFF"), hip.fetch("INGEST_LEN")
    assert [seq[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)] == READS


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("window", [0, 5000, 40000, 333333])
def test_streamed_sr_read_files_equals_reference_sr_read(hip, tmp_path, window):
    """oatk_sr_read_files (liboatk_host.so) streams the files through the device in windows -- upload, record scan, syncmer scan, struct filling and
    the append to the assembled batch all overlap -- and must leave exactly the sr_db_t the reference's sr_read leaves for the same files, whatever
    the window (here: records cut by window ends, windows inside one read's sequence line, several files, gzip, CRLF, a file without final
    newline), AND a device batch on which the count continues as if it had been scanned in one go"""
    import ctypes as C
    import gzip
    from test_gpu_dropin import host_lib
    K, S = 301, 21
    reads = A.hifi_like(90, 30000, 4000, seed=23) + [b"acgtnACGTN" * 60, b"A" * 900 + A.rand_dna(np.random.default_rng(1), 700), b"C"]
    a, b, c = reads[:30], reads[30:60], reads[60:]
    p1, p2, p3 = str(tmp_path / "a.fa"), str(tmp_path / "b.fa.gz"), str(tmp_path / "c.fa")
    open(p1, "wb").write(fasta(a, 70, last_eol=False))
    gzip.open(p2, "wb").write(fasta(b, 0))
    open(p3, "wb").write(fasta(c, 61, b"\r\n"))
    db = R.SrDb([p1, p2, p3], K, S, 2)
    want = db.flatten()
    H = host_lib()
    H.oatk_sr_read_files.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_int]
    H.oatk_host_debug_window.argtypes = [C.c_uint64]
    H.oatk_host_set_threads.argtypes = [C.c_int]
    L = R.lib()
    L.refx_srdb_name.restype = C.c_char_p
    L.refx_srdb_name.argtypes = [C.c_void_p, C.c_uint64]
    mine = H.oatk_sr_db_new(K, S)
    files = (C.c_char_p * 3)(p1.encode(), p2.encode(), p3.encode())
    H.oatk_host_debug_window(window)
    H.oatk_host_set_threads(5)
    try:
        rc = H.oatk_sr_read_files(hip.h, mine, files, 3)
    finally:
        H.oatk_host_debug_window(0)
        H.oatk_host_set_threads(0)
    assert rc == 0, hip.L.oatk_hip_last_error(hip.h)
    got_db = R.SrDb.__new__(R.SrDb)
    got_db.K, got_db.S, got_db._h = K, S, mine                      # the reference's own accessors read what liboatk_host.so built
    got = got_db.flatten()
    assert got_db.n() == db.n() == len(reads)
    for f in ["hoco_l", "n_scm", "sid", "hoco_s", "ho_rl", "ho_l_rl", "m_pos", "s_mer", "k_mer"]:
        assert np.array_equal(got[f], want[f]), f
    for i in range(len(reads)):
        assert L.refx_srdb_name(mine, i) == L.refx_srdb_name(db.handle, i)
    # the assembled device batch: counting it gives the reference's table
    hip.count()
    scm = R.ScmDb(db)
    rc_ = scm.flatten()
    dc = hip.fetch_count()
    for f in ("h", "s", "cov", "occ"):
        assert np.array_equal(dc[f], rc_[f]), f
    scm.close()
    got_db.close()                                                   # sr_db_destroy of the reference frees every block
    db.close()
