"""CPU: the room host/ingest_host.c asks for ahead of the reads (oatk_amd/csrc/host/ingest_estimate.h) -- the arithmetic that, fed a source position that stood still,
promised tens of terabytes in round 5.  tests/c/ingest_estimate_test.c, built here with gcc under UBSan."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_estimates(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "est")
    base = ["gcc", "-O1", "-g", "-Wall", "-I" + os.path.join(ROOT, "oatk_amd", "csrc", "host"), "-o", exe, os.path.join(ROOT, "tests", "c", "ingest_estimate_test.c")]
    r = subprocess.run(base[:3] + ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"] + base[3:], capture_output=True, text=True)
    if r.returncode != 0:
        r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout, r.stderr)
