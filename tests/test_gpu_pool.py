"""GPU: device memory in pieces (oatk_hip_mem_pool, include/oatk_hip.h) changes where buffers live and how they grow, never what is in them.  Whole test files run
once more in a process of their own with the pool switched on for every handle (OATK_TEST_POOL) and its threshold lowered so that EVERY buffer is an address range
backed by 64 MB pieces (OATK_DEBUG_POOL_MIN) -- scan, assembled batches (buffers that grow and keep what they hold), count, EC graph, correction, assembly graph,
reader: every comparison those files make with the oracle, the golden vectors and the compiled reference.  The allocation log proves the pieces were what ran.
(This is the test that found that an address range given back with hipMemAddressFree must not be: csrc/api.hip, DevBuf::vm_range.)"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("files", [("test_gpu_append.py", "test_gpu_scan.py"), ("test_gpu_ec.py", "test_gpu_ingest.py", "test_gpu_consensus.py"), ("test_gpu_align.py", "test_gpu_asmgraph.py", "test_gpu_levdist.py", "test_gpu_light_graph.py", "test_gpu_overlap.py")],
                         ids=["scan_append", "ec_ingest_consensus", "align_asmgraph"])
def test_files_again_over_pieces(files):
    env = dict(os.environ, OATK_TEST_POOL=str(1 << 30), OATK_DEBUG_POOL_MIN="4096", OATK_DEBUG_ALLOC_LOG="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-s"] + [os.path.join(ROOT, "tests", f) for f in files],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join([ln for ln in (p.stdout + p.stderr).splitlines() if "[oatk alloc]" not in ln][-60:])
    assert p.returncode == 0, tail
    m = re.search(r"(\d+) passed", p.stdout)
    assert m and int(m.group(1)) > 0 and " failed" not in p.stdout, tail
    pieces = len(re.findall(r"\[oatk alloc\] [\d.]+ pieces ", p.stdout + p.stderr))
    assert pieces > 100, "the pool was not what the buffers came from (%d piece events)" % pieces
