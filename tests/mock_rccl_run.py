"""Child process of tests/test_gpu_multi_c.py::test_rccl_branch_with_several_ranks_over_a_mock_library: OATK_RCCL_LIB points at tests/c/mock_rccl.cpp's
library, so oatk_comm_create makes communicators whose ranks are threads of this process, and the RCCL branch of api_multi.inc -- grouped
send / receive, grouped broadcasts, all-gather, all-reduce -- runs with 2 - 4 ranks on the one GPU.  Compared with one handle holding all reads."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adversarial as A            # noqa: E402
import test_gpu_multi_c as M       # noqa: E402
from oatk_amd import HipSyncasm, _lib  # noqa: E402


def main():
    L = _lib.load()
    hip = HipSyncasm(0)
    done = 0
    for case in (0, 2, 3, 4):                 # 2 ranks; imports; three shards with an empty one (an EMPTY send/receive group); four shards
        K, S, c, mk, frac = M.CASES[case]
        reads = mk()
        bounds = [int(round(f * len(reads))) for f in frac]
        world = len(bounds) - 1
        uid = (C.c_uint8 * 128)()
        assert L.oatk_comm_unique_id(uid) == 0
        comms = {}

        def make(rank, uid=uid, world=world):
            comm = L.oatk_comm_create(uid, rank, world, 0)
            assert comm and L.oatk_comm_backend(comm) == b"rccl" and L.oatk_comm_size(comm) == world
            comms[rank] = comm
            return comm
        out = M.run_ranks(world, make, reads, bounds, K, S, c)
        cnt, st, want = M.single(hip, reads, K, S, c)
        first = 0
        for rank, (ng, mg, local_h, st_r, n_imp, res) in enumerate(out):
            f, no, g = mg["range"]
            assert ng == cnt["n_scm"] and f == first and st_r[:11].tolist() == st[:11].tolist()
            first += no
            l2g = mg["MG_L2G"].astype(np.int64)
            assert np.array_equal(cnt["h"][l2g], local_h) and np.array_equal(cnt["cov"][l2g], mg["MG_LCOV"])
        for key, ref in (("MG_H", cnt["h"]), ("MG_S", cnt["s"]), ("MG_COV", cnt["cov"])):
            assert np.array_equal(np.concatenate([o[1][key] for o in out]), ref), key
        for key in ("EC_N_SCM", "EC_KMER", "EC_MPOS", "EC_SMER"):
            assert np.array_equal(np.concatenate([o[5][key] for o in out]), want[key]), key
        assert np.array_equal(np.concatenate([o[5]["MG_EC_COV"] for o in out]), want["EC_SCM_COV"])
        assert np.array_equal(np.concatenate([o[5]["MG_EC_DEL"] for o in out]), want["EC_SCM_DEL"])
        done += 1
    # forced hash collisions: the k-mer requests and the k-mers travel by send / receive too
    K, S, c = 101, 11, 4
    reads = A.hifi_like(120, 5000, 1500, seed=5, err=0.004)
    uid = (C.c_uint8 * 128)()
    assert L.oatk_comm_unique_id(uid) == 0
    out = M.run_ranks(3, lambda r: L.oatk_comm_create(uid, r, 3, 0), reads, [0, 35, 80, len(reads)], K, S, c, mask=0xFF)
    hip.debug_hash_mask(0xFF)
    cnt, st, want = M.single(hip, reads, K, S, c)
    hip.debug_hash_mask(0xFFFFFFFFFFFFFFFF)
    for key, ref in (("MG_H", cnt["h"]), ("MG_S", cnt["s"]), ("MG_COV", cnt["cov"])):
        assert np.array_equal(np.concatenate([o[1][key] for o in out]), ref), key
    assert np.array_equal(np.concatenate([o[5]["EC_KMER"] for o in out]), want["EC_KMER"])
    # the graph hand-off (include/oatk_hip_multi.h, second half) over the same mock: gather to root by grouped send / receive, the all-gathers of
    # coverage, keys and segments, the all-reduces of the consensus totals, the personalised exchange of the statistics
    import test_gpu_multi_tail as T
    for case in (0, 2, 3):
        K, S, c, mk, frac = T.CASES[case]
        reads = mk()
        bounds = [int(round(f * len(reads))) for f in frac]
        world = len(bounds) - 1
        root = world - 1 if case == 3 else 0
        uid = (C.c_uint8 * 128)()
        assert L.oatk_comm_unique_id(uid) == 0
        out = T.run_ranks(world, None, reads, bounds, K, S, c, 0.35, root, make_comm=lambda r, uid=uid, world=world: L.oatk_comm_create(uid, r, world, 0))
        T.check(out, T.single(hip, reads, K, S, c, 0.35), c, root)
        done += 1
    print("ok %d" % (done + 1))


if __name__ == "__main__":
    main()
