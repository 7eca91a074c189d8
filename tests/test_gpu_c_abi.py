"""GPU: the C ABI consumed from plain C (tests/c/abi_pipeline.c, built here with gcc against include/*.h and the two in-tree libraries):
the whole device pipeline in one process without Python or torch, its figures equal to the same calls made through ctypes."""
import os
import re
import subprocess
import zlib  # noqa: F401

import numpy as np
import pytest

from oatk_amd import _lib
from oatk_amd.synth import ReadSet

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_pipeline_from_plain_c(hip, tmp_path):
    exe = str(tmp_path / "abi_pipeline")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_pipeline.c"),
                    "-L" + libdir, "-loatk_host", "-loatk_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    n, G, L, K, S, c = 600, 40000, 6000, 301, 21, 5
    out = subprocess.run([exe, str(n), str(G), str(L), str(K), str(S), str(c)], check=True, capture_output=True, text=True).stdout
    got = dict(re.findall(r"(\w+)=(\S+)", out))
    # the same through ctypes
    seq, off, lens = ReadSet(G, n, L).slice(0, n, threads=4)
    hip.scan_host(seq, off, lens, K, S)
    hip.count()
    info = hip.info()
    st = hip.stat_raw()
    hip.ec_graph()
    ecs = hip.ec(0.02, c, 0.35)
    hip.consensus(c)
    n_pairs, n_entries = hip.overlap_hist()
    nv, na = hip.asm_graph(c, 0.35)
    want = {"n_occ": info["n_occ"], "n_scm": info["n_scm"], "kmer_unique": st["kmer_unique"], "sum_dist": st["sum_dist"],
            "blocks": int(ecs[0] + ecs[5] + ecs[10]), "corrected": int(ecs[2] + ecs[7]), "pairs": n_pairs, "entries": n_entries, "n_vtx": nv, "n_arc": na}
    for k, v in want.items():
        assert int(got[k]) == int(v), (k, got[k], v)
    assert int(got["ec_kmer"], 16) == fnv(hip.fetch("EC_KMER")) and int(got["ag_arc_w"], 16) == fnv(hip.fetch("AG_ARC_W"))
    assert int(got["cons_rl"], 16) == fnv(hip.fetch("CONS_RL"))
    assert want["blocks"] > 0 and nv > 0 and na > 0 and n_pairs > 0
