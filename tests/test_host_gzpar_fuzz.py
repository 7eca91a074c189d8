"""CPU: host/gzpar.c under AddressSanitizer + UBSan against zlib on streams made here (tests/c/gzpar_fuzz.c): every zlib strategy, level, window and memLevel, random
flushes, texts with long and with short matches, runs and short periods -- and the same streams damaged (bits flipped, bytes dropped, tails cut, stretches overwritten):
what comes out is zlib's text or an error, a member that is delivered whole is one zlib delivers too (round 5: a distance reaching before the member's first byte was
taken for a window reference), nothing outside the decoder's own memory is touched and nothing hangs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "oatk_amd", "csrc", "host")


def _build(tmp, name, sources):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp / name)
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), "-I" + HOST, "-o", exe] + sources + ["-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this gcc has no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("gzfuzz"), "gzpar_fuzz", [os.path.join(ROOT, "tests", "c", "gzpar_fuzz.c"), os.path.join(HOST, "gzpar.c")])


@pytest.fixture(scope="module")
def file_fuzzer(tmp_path_factory):
    return _build(tmp_path_factory.mktemp("srcfuzz"), "gzsrc_fuzz", [os.path.join(ROOT, "tests", "c", "gzsrc_fuzz.c"), os.path.join(HOST, "gzsrc.c"), os.path.join(HOST, "gzpar.c")])


@pytest.mark.parametrize("seed", [1, 3])
def test_gzpar_equals_zlib_on_made_and_damaged_streams(fuzzer, seed):
    r = subprocess.run([fuzzer, "30", str(seed)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "streams equal to zlib's text" in r.stdout


@pytest.mark.parametrize("seed", [1, 4])
def test_gzsrc_equals_gzread_on_made_and_damaged_files(file_fuzzer, tmp_path, seed):
    """host/gzsrc.c as a whole (tests/c/gzsrc_fuzz.c): files of plain members with every optional header field, runs of BGZF blocks, empty members, bytes behind the
    last member -- and damaged copies: where gzread (what the reference reads through, sstream.c:39-54) reads a file without an error gzsrc delivers the same bytes, and
    gzsrc never reports success with other bytes than gzread's (round 5: a damaged BGZF block whose length field read 0 passed for bgzip's end marker and the rest of
    the file for trailing bytes)"""
    r = subprocess.run([file_fuzzer, "25", str(seed), str(tmp_path)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "files read alike" in r.stdout
