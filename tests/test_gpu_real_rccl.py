"""Real RCCL with more than one rank: the sharded path of include/oatk_hip_multi.h -- table merge, sharded error correction, and the tail up to the graph
hand-off -- with ONE DEVICE PER RANK over librccl, against one handle holding all the reads.  Skipped on a box with a single GPU (every test box so far:
there the RCCL branch runs with a world of one, and with several ranks over tests/c/mock_rccl.cpp); the first multi-GPU node that runs the suite runs these.
The checks are the ones of tests/mock_rccl_run.py and tests/test_gpu_multi_tail.py."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import adversarial as A
import test_gpu_multi_c as M
import test_gpu_multi_tail as T
from oatk_amd import _lib

pytestmark = pytest.mark.gpu


def n_devices():
    try:
        return int(_lib.load().oatk_hip_device_count())
    except Exception:          # noqa: BLE001
        return 0


needs_two = pytest.mark.skipif(n_devices() < 2, reason="one GPU: real RCCL refuses two ranks on a device")


def real_comm_factory(world):
    L = _lib.load()
    uid = (C.c_uint8 * 128)()
    assert L.oatk_comm_unique_id(uid) == 0

    def make(rank):
        comm = L.oatk_comm_create(uid, rank, world, rank)           # rank r on device r
        assert comm and L.oatk_comm_backend(comm) == b"rccl" and L.oatk_comm_size(comm) == world
        return comm
    return make


@needs_two
@pytest.mark.parametrize("case", [0, 2, 3])
def test_real_rccl_merge_and_sharded_correction_equal_one_handle(hip, case):
    K, S, c, mk, frac = M.CASES[case]
    reads = mk()
    world = min(len(frac) - 1, n_devices())
    if world < 2:
        pytest.skip("needs as many devices as the case has shards")
    if world != len(frac) - 1:
        frac = [i / world for i in range(world + 1)]
    bounds = [int(round(f * len(reads))) for f in frac]
    out = M.run_ranks(world, real_comm_factory(world), reads, bounds, K, S, c, devices=list(range(world)))
    cnt, st, want = M.single(hip, reads, K, S, c)
    for key, ref in (("MG_H", cnt["h"]), ("MG_S", cnt["s"]), ("MG_COV", cnt["cov"])):
        assert np.array_equal(np.concatenate([o[1][key] for o in out]), ref), key
    for key in ("EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER"):
        assert np.array_equal(np.concatenate([o[5][key] for o in out]), want[key]), key
    assert np.array_equal(np.concatenate([o[5]["MG_EC_COV"] for o in out]), want["EC_SCM_COV"])
    assert all(o[3][:11].tolist() == st[:11].tolist() for o in out)


@needs_two
@pytest.mark.parametrize("case", [0, 2])
def test_real_rccl_tail_up_to_the_graph_hand_off(hip, case):
    K, S, c, mk, frac = T.CASES[case]
    reads = mk()
    world = min(len(frac) - 1, n_devices())
    if world < 2:
        pytest.skip("needs two devices")
    if world != len(frac) - 1:
        frac = [i / world for i in range(world + 1)]
    bounds = [int(round(f * len(reads))) for f in frac]
    out = T.run_ranks(world, None, reads, bounds, K, S, c, 0.35, 0, make_comm=real_comm_factory(world), devices=list(range(world)))
    T.check(out, T.single(hip, reads, K, S, c, 0.35), c, 0)


@needs_two
def test_real_rccl_a_failing_rank_does_not_hang_its_peer(tmp_path):
    """rank 1 fails on request instead of entering a collective; rank 0, inside that collective over real RCCL, must come back: through
    ncclCommGetAsyncError if the library reports the peer's abort, through OATK_COMM_TIMEOUT_S otherwise (api_multi.inc: comm_wait)"""
    here = os.path.dirname(os.path.abspath(__file__))
    code = r'''
import ctypes as C, os, sys, threading, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.dirname(%r))
import adversarial as A
from oatk_amd import HipSyncasm, _lib, pack_reads
L = _lib.load()
reads = A.hifi_like(100, 5000, 1500, seed=9, err=0.004)
uid = (C.c_uint8 * 128)(); assert L.oatk_comm_unique_id(uid) == 0
res = [None, None]
def work(rank):
    h = HipSyncasm(rank); comm = L.oatk_comm_create(uid, rank, 2, rank)
    seq, off, lens = pack_reads(reads[50 * rank:50 * rank + 50]); h.scan_host(seq, off, lens, 101, 11, sid0=50 * rank); h.count()
    t0 = time.time()
    try:
        h.merge_counts(comm); res[rank] = ("ok", time.time() - t0)
    except Exception as ex:
        res[rank] = ("error", time.time() - t0, str(ex))
th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
[t.start() for t in th]; [t.join(timeout=120) for t in th]
assert not any(t.is_alive() for t in th), "a rank hangs"
assert res[0][0] == "error" and res[1][0] == "error" and res[0][1] < 60, res
print("ok", res)
os._exit(0)
''' % (here, here)
    env = dict(os.environ, OATK_DEBUG_FAIL_RANK="1,3", OATK_COMM_TIMEOUT_S="8")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().startswith(b"ok"), (p.returncode, p.stdout[-300:], p.stderr.decode(errors="replace")[-1500:])
