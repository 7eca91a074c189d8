"""CPU: host/gzpar.c under AddressSanitizer + UBSan against zlib on streams made here (tests/c/gzpar_fuzz.c): every zlib strategy, level, window and memLevel, random
flushes, texts with long and with short matches, runs and short periods -- and the same streams damaged (bits flipped, bytes dropped, tails cut, stretches overwritten):
what comes out is zlib's text or an error, a member that is delivered whole is one zlib delivers too (round 5: a distance reaching before the member's first byte was
taken for a window reference), nothing outside the decoder's own memory is touched and nothing hangs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "c", "gzpar_fuzz.c"), os.path.join(ROOT, "oatk_amd", "csrc", "host", "gzpar.c")]


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path_factory.mktemp("gzfuzz") / "gzpar_fuzz")
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "oatk_amd", "csrc", "host"), "-o", exe] + SRC + ["-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this gcc has no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


@pytest.mark.parametrize("seed", [1, 3])
def test_gzpar_equals_zlib_on_made_and_damaged_streams(fuzzer, seed):
    r = subprocess.run([fuzzer, "30", str(seed)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "streams equal to zlib's text" in r.stdout
