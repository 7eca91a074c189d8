"""CPU: the error-block search's alignment by matrix rows, the row in BITS (tests/c/rows_bitpar_test.c) -- two bit vectors for a band of up to 640 diagonals, moved one
cell down the target per query base (Myers 1999, Hyyro's diagonal band 2003): a prototype of the next round's kernel, checked against the plain banded matrix after every
row and in the outcome of every resumed call (the closed form of tests/test_oracle_golden.py::test_levdist_resumable_traces_are_the_matrix)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 2])
def test_rows_in_bits_equal_the_matrix(tmp_path, seed):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "bitpar")
    src = os.path.join(ROOT, "tests", "c", "rows_bitpar_test.c")
    r = subprocess.run(["gcc", "-O1", "-g", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe, src], capture_output=True, text=True)
    if r.returncode != 0:
        r = subprocess.run(["gcc", "-O1", "-Wall", "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "150", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), (r.stdout, r.stderr[-2000:])


def test_table_in_bits_equals_the_matrix(tmp_path):
    """the table a long arc is tested with (least cost of fitting the appended string into the target from each position on: tests/c/prof_bitpar_test.c) -- approximate
    matching of the reversed string against the reversed target, multiword, with the carries resolved the way the device resolves them"""
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "prof")
    src = os.path.join(ROOT, "tests", "c", "prof_bitpar_test.c")
    r = subprocess.run(["gcc", "-O1", "-g", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe, src], capture_output=True, text=True)
    if r.returncode != 0:
        r = subprocess.run(["gcc", "-O1", "-Wall", "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "120", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), (r.stdout, r.stderr[-2000:])
