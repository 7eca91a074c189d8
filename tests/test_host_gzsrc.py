"""CPU: host/gzsrc.c -- a gzip'ed file as a stream of inflated bytes (what the reader of host/ingest_host.c pulls its windows from): the three forms
host/fasta_out.c writes (one member, BGZF, several members), odd buffer sizes, an empty file, bytes behind the last member (ignored like gzread
ignores them), and damage (a flipped byte, a cut file) reported instead of passed on."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from oatk_amd import _lib, synth


@pytest.fixture(scope="module")
def H():
    L = C.CDLL(_lib.HOST_LIB_PATH)
    L.oatk_gzsrc_open.restype = C.c_void_p
    L.oatk_gzsrc_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.oatk_gzsrc_read.restype = C.c_int64
    L.oatk_gzsrc_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.oatk_gzsrc_close.argtypes = [C.c_void_p]
    L.oatk_gzsrc_kind.argtypes = [C.c_void_p]
    L.oatk_gzsrc_tell_in.restype = C.c_uint64
    L.oatk_gzsrc_tell_in.argtypes = [C.c_void_p]
    return L


def slurp(H, path, cap, threads=4):
    rc = C.c_int(0)
    g = H.oatk_gzsrc_open(path.encode(), threads, C.byref(rc))
    assert g, rc.value
    kind = H.oatk_gzsrc_kind(g)
    buf = np.empty(cap, np.uint8)
    out, status = [], 0
    while True:
        n = H.oatk_gzsrc_read(g, buf.ctypes.data, cap)
        if n < 0:
            status = -1
            break
        if n == 0:
            break
        out.append(buf[:n].tobytes())
    consumed = H.oatk_gzsrc_tell_in(g)
    H.oatk_gzsrc_close(g)
    return b"".join(out), status, kind, consumed


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("gz")
    cfg = dict(synth.CONFIG1S)
    cfg.update(n_reads=300, nuclear_len=2_000_000)
    rs = synth.MixReadSet(**cfg)
    seq, off, lens = rs.slice(0, 300)
    paths = {}
    for name, mode in (("plain", synth.FA_PLAIN), ("one", synth.FA_GZ), ("bgzf", synth.FA_BGZF), ("members", synth.FA_GZ_MEMBERS)):
        paths[name] = str(d / (name + (".fa" if mode == 0 else ".fa.gz")))
        synth.write_fasta(paths[name], seq, off, lens, mode=mode, member_bytes=700_000, threads=4)
    return paths, open(paths["plain"], "rb").read(), d


@pytest.mark.parametrize("cap", [1 << 16, 100_003, 1 << 20, 1 << 24])
def test_every_form_inflates_to_the_text(H, files, cap):
    paths, text, _ = files
    for name, kind in (("one", 1), ("bgzf", 2), ("members", 1)):
        assert gzip.open(paths[name]).read() == text                 # the writer, against Python's zlib
        got, status, k, consumed = slurp(H, paths[name], cap)
        assert status == 0 and got == text, name
        assert k == kind and consumed == os.path.getsize(paths[name])


def test_small_buffers_cut_bgzf_members(H, files):
    paths, text, _ = files
    got, status, _, _ = slurp(H, paths["bgzf"], 5000)             # smaller than a BGZF member: the serial form, piece by piece
    assert status == 0 and got == text


def test_empty_and_trailing_bytes(H, files, tmp_path):
    paths, text, _ = files
    e = str(tmp_path / "empty.gz")
    with gzip.open(e, "wb"):
        pass
    assert slurp(H, e, 1 << 16)[:2] == (b"", 0)
    z = str(tmp_path / "zero.gz")
    open(z, "wb").close()
    assert slurp(H, z, 1 << 16)[:2] == (b"", 0)
    t = str(tmp_path / "trail.fa.gz")
    with open(t, "wb") as f:
        f.write(open(paths["one"], "rb").read() + b"\0" * 512)       # padding behind the member (tar, some archivers): gzread ignores it
    got, status, _, _ = slurp(H, t, 1 << 20)
    assert status == 0 and got == text and gzip.open(paths["one"]).read() == text
    m = str(tmp_path / "mixed.fa.gz")
    with open(m, "wb") as f:                                         # a plain member in front of BGZF members and another plain one behind
        f.write(gzip.compress(b">x\nACGT\n") + open(paths["bgzf"], "rb").read() + gzip.compress(b">y\nTTTT\n"))
    got, status, _, _ = slurp(H, m, 1 << 20)
    assert status == 0 and got == b">x\nACGT\n" + text + b">y\nTTTT\n"


def test_damage_is_reported(H, files, tmp_path):
    paths, text, _ = files
    for name in ("one", "bgzf", "members"):
        raw = bytearray(open(paths[name], "rb").read())
        raw[len(raw) // 2] ^= 0x55
        p = str(tmp_path / (name + ".bad.gz"))
        open(p, "wb").write(bytes(raw))
        got, status, _, _ = slurp(H, p, 1 << 20)
        assert status == -1, name
        cut = str(tmp_path / (name + ".cut.gz"))
        open(cut, "wb").write(open(paths[name], "rb").read()[:-100])
        got, status, _, _ = slurp(H, cut, 1 << 20)
        assert status == -1 or (name == "bgzf" and got != text), name       # (a BGZF file cut at a member boundary simply ends early: bgzip's end marker is what tells)


def _bgzf_blocks(data, block=20_000):
    """`data` as BGZF members (RFC 1952 with the BC extra field, SAM spec 4.1), no end marker"""
    import struct
    import zlib
    out = bytearray()
    for a in range(0, len(data), block):
        piece = data[a:a + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(piece) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body
        out += struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
    return bytes(out)


def test_workers_created_in_a_later_round_do_not_run_a_phantom_one(H, tmp_path):
    """three BGZF blocks (three workers start), a plain member, then three hundred blocks (the pool grows in a later call): a worker created
    late began at generation 0, ran a round nobody had started and counted `busy` down once too often -- the reader hung or returned half-filled text."""
    rng = np.random.default_rng(5)
    def text(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT\n", np.uint8), n))
    a, b, c = text(3 * 20_000), text(50_000), text(300 * 20_000)
    p = str(tmp_path / "few_plain_many.gz")
    open(p, "wb").write(_bgzf_blocks(a) + gzip.compress(b) + _bgzf_blocks(c))
    for rep in range(40):
        got, status, _, _ = slurp(H, p, 1 << 24, threads=32)
        assert status == 0 and got == a + b + c, rep
    for cap in (70_000, 1 << 20):                                        # and with windows that cut the runs of blocks into many calls
        got, status, _, _ = slurp(H, p, cap, threads=16)
        assert status == 0 and got == a + b + c, cap


# ---- one member on many threads (host/gzpar.c): every byte equal to zlib's, whatever the member looks like ----

def _reads_text(rng, n_bytes):
    out, tot, i = [], 0, 0
    while tot < n_bytes:
        ln = int(rng.integers(2000, 20000))
        rec = b">r%d a header with words\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), ln)) + b"\n"
        out.append(rec)
        tot += len(rec)
        i += 1
    return b"".join(out)


@pytest.fixture(scope="module")
def big_member(tmp_path_factory):
    d = tmp_path_factory.mktemp("gzpar")
    rng = np.random.default_rng(11)
    text = _reads_text(rng, 24_000_000)
    paths = {}
    for level in (1, 6, 9):
        paths[level] = str(d / ("l%d.fa.gz" % level))
        open(paths[level], "wb").write(gzip.compress(text, compresslevel=level))
    return paths, text, d


@pytest.mark.parametrize("chunk_kb", ["64", "512"])
@pytest.mark.parametrize("cap", [70_001, 1 << 22, 1 << 26])
def test_one_member_on_many_threads_equals_zlib(H, big_member, monkeypatch, cap, chunk_kb):
    paths, text, _ = big_member
    monkeypatch.setenv("OATK_HOST_GZ_PARALLEL", "100000")            # members of 100 kB and more go that way
    monkeypatch.setenv("OATK_HOST_GZ_CHUNK_KB", chunk_kb)            # (many chunks: many boundaries to find and to arrive at)
    for level, p in paths.items():
        for threads in (4, 13):
            got, status, kind, consumed = slurp(H, p, cap, threads=threads)
            assert status == 0 and got == text, (level, threads)
            assert kind == 1 and consumed == os.path.getsize(p)


def test_members_that_cannot_be_entered_in_the_middle(H, tmp_path, monkeypatch):
    """stored blocks (level 0), bytes that are no text (no boundary passes for one: zlib takes the member), a member of fixed-Huffman blocks, text with long runs
    (distances of one, matches of 258), and several large members one after the other"""
    import zlib
    monkeypatch.setenv("OATK_HOST_GZ_PARALLEL", "100000")
    monkeypatch.setenv("OATK_HOST_GZ_CHUNK_KB", "64")
    rng = np.random.default_rng(12)
    text = _reads_text(rng, 3_000_000)
    cases = {
        "stored": (gzip.compress(text, compresslevel=0), text),
        "binary": (gzip.compress(bytes(rng.integers(0, 256, 2_000_000, dtype=np.uint8)) + text), None),
        "runs": (gzip.compress((b"A" * 100_000 + b"\n" + b"ACGT" * 50_000 + b"\n") * 8 + text), None),
        "several": (gzip.compress(text) + gzip.compress(text[::-1].replace(b">", b"N")) + gzip.compress(text), None),
    }
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    cases["fixed"] = (co.compress(text) + co.flush(), text)
    for name, (raw, want) in cases.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(raw)
        want = gzip.open(p).read() if want is None else want
        for cap in (50_000, 1 << 24):
            got, status, _, consumed = slurp(H, p, cap, threads=8)
            assert status == 0 and got == want, (name, cap)
            assert consumed == len(raw)


def test_flush_markers_and_blocks_the_search_does_not_accept(H, tmp_path, monkeypatch):
    """a chunk stops at the first block boundary past its cut, the next one starts at the first boundary its search ACCEPTS (a dynamic block of a thousand symbols that is
    not the last): between the two may lie flush markers (empty stored blocks: pigz -i, Z_FULL_FLUSH / Z_SYNC_FLUSH writers), stored or fixed blocks, short blocks --
    the chain's last chunk goes on over them in order (host/gzpar.c extend_chunk).  Same bytes as zlib whatever the cut, the slots per thread and the buffer"""
    import zlib
    rng = np.random.default_rng(21)
    text = _reads_text(rng, 9_000_000)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts, at, i = [], 0, 0
    while at < len(text):
        n = int(rng.integers(300, 120_000))
        parts.append(co.compress(text[at:at + n]))
        parts.append(co.flush((zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH, zlib.Z_SYNC_FLUSH)[i % 3]))
        at += n
        i += 1
    parts.append(co.flush())
    raw = b"".join(parts)
    p = str(tmp_path / "flushes.gz")
    open(p, "wb").write(raw)
    monkeypatch.setenv("OATK_HOST_GZ_PARALLEL", "100000")
    for chunk_kb, slots in (("32", "1"), ("64", "3"), ("200", "5")):
        monkeypatch.setenv("OATK_HOST_GZ_CHUNK_KB", chunk_kb)
        monkeypatch.setenv("OATK_HOST_GZ_SLOTS", slots)
        for cap in (33_333, 1 << 23):
            got, status, _, consumed = slurp(H, p, cap, threads=6)
            assert status == 0 and got == text, (chunk_kb, slots, cap)
            assert consumed == len(raw)


def test_damage_in_a_member_on_many_threads_is_reported(H, big_member, tmp_path, monkeypatch):
    paths, text, _ = big_member
    monkeypatch.setenv("OATK_HOST_GZ_PARALLEL", "100000")
    monkeypatch.setenv("OATK_HOST_GZ_CHUNK_KB", "256")
    raw = open(paths[6], "rb").read()
    for where in (len(raw) // 7, len(raw) // 2, len(raw) - 5, len(raw) - 2):      # inside the data (twice), in the CRC, in the length
        bad = bytearray(raw)
        bad[where] ^= 0x10
        p = str(tmp_path / ("bad%d.gz" % where))
        open(p, "wb").write(bytes(bad))
        got, status, _, _ = slurp(H, p, 1 << 24, threads=8)
        assert status == -1 or got != text, where
        assert status == -1, where
    for keep in (len(raw) - 9, len(raw) // 2, 100_000):              # the file ends inside the trailer, inside the data
        p = str(tmp_path / ("cut%d.gz" % keep))
        open(p, "wb").write(raw[:keep])
        got, status, _, _ = slurp(H, p, 1 << 24, threads=8)
        assert status == -1, keep
    monkeypatch.setenv("OATK_HOST_GZ_PARALLEL", "0")                 # the switch: zlib on one thread
    got, status, _, _ = slurp(H, paths[6], 1 << 24, threads=8)
    assert status == 0 and got == text


def test_the_position_in_the_file_moves_while_a_member_is_inflated_on_many_threads(H, big_member, monkeypatch):
    """host/ingest_host.c sizes the reads' array and the device's batch by the text a window holds per compressed byte it took (oatk_gzsrc_tell_in before and after): a
    position that stands still through a member turns that into an over-estimate by the file's size (round 5: a box was lost to it)"""
    paths, text, _ = big_member
    monkeypatch.setenv("OATK_HOST_GZ_PARALLEL", "100000")
    monkeypatch.setenv("OATK_HOST_GZ_CHUNK_KB", "256")
    rc = C.c_int(0)
    g = H.oatk_gzsrc_open(paths[6].encode(), 8, C.byref(rc))
    cap = 4 << 20
    buf = np.empty(cap, np.uint8)
    size, last, out, seen = os.path.getsize(paths[6]), 0, 0, []
    while True:
        n = H.oatk_gzsrc_read(g, buf.ctypes.data, cap)
        assert n >= 0
        if n == 0:
            break
        out += n
        at = H.oatk_gzsrc_tell_in(g)
        assert last <= at <= size
        seen.append((out, at))
        last = at
    H.oatk_gzsrc_close(g)
    assert out == len(text) and last == size
    mid = [a for o, a in seen if len(text) // 4 < o < 3 * len(text) // 4]
    assert mid and min(mid) > size // 10 and max(mid) < size, "the position follows the text"
    # what the reader asks of it: a window's text over forty is no more than the compressed bytes the window took, from the second window on
    for (o0, a0), (o1, a1) in zip(seen[1:], seen[2:]):
        if o1 < len(text):
            assert (a1 - a0) * 40 >= (o1 - o0) or a1 == a0, (o0, a0, o1, a1)


def _opened(H, path, cap, threads):
    """(text, members the many-thread reader was opened for) reading `path` in calls of at most `cap` bytes"""
    H.oatk_gzsrc_members_on_many_threads.restype = C.c_uint64
    H.oatk_gzsrc_members_on_many_threads.argtypes = [C.c_void_p]
    rc = C.c_int(0)
    g = H.oatk_gzsrc_open(path.encode(), threads, C.byref(rc))
    assert g
    buf, out = np.empty(cap, np.uint8), bytearray()
    while True:
        n = H.oatk_gzsrc_read(g, buf.ctypes.data, cap)
        assert n >= 0
        if n == 0:
            break
        out += buf[:n].tobytes()
    k = H.oatk_gzsrc_members_on_many_threads(g)
    H.oatk_gzsrc_close(g)
    return bytes(out), int(k)


def test_the_many_thread_reader_is_not_opened_for_what_it_cannot_help(H, tmp_path, monkeypatch):
    """Round 6 (sr_read from BGZF 0.35 -> 1.9 s on the driver's clock).  A BGZF member cut by the end of the caller's buffer goes through the serial path -- and
    that path opened host/gzpar.c on the REST OF THE FILE: threads started, boundaries searched and chunks decoded in later members, all thrown away 64 KiB on, once per call.
    A member that says it is BGZF is zlib's; and in a file of many small plain members only the first one pays for finding that out."""
    rng = np.random.default_rng(23)
    text = _reads_text(rng, 30_000_000)
    monkeypatch.setenv("OATK_HOST_GZ_PARALLEL", "1000000")            # the gate: a megabyte of file left
    p = str(tmp_path / "b.fa.gz")
    open(p, "wb").write(_bgzf_blocks(text, 60_000))
    for cap in (1 << 20, (1 << 22) + 12345):                          # every call ends inside a member
        got, k = _opened(H, p, cap, 16)
        assert got == text and k == 0, (cap, k)
    q = str(tmp_path / "m.fa.gz")
    open(q, "wb").write(b"".join(gzip.compress(text[a:a + 700_000], 6) for a in range(0, len(text), 700_000)))      # 43 members of ~200 kB
    got, k = _opened(H, q, 1 << 22, 16)
    assert got == text and k == 1, k
    r = str(tmp_path / "one.fa.gz")
    open(r, "wb").write(gzip.compress(text, 6))                       # ... and ONE large member still goes the many-thread way
    got, k = _opened(H, r, 1 << 22, 16)
    assert got == text and k == 1, k
