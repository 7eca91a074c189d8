"""Child process of tests/test_gpu_multi_c.py::test_a_rank_that_fails_mid_way_releases_its_peers: three ranks (threads) over tests/c/mock_rccl.cpp; rank 1
fails instead of entering its 5th collective (OATK_DEBUG_FAIL_RANK=1,5, api_multi.inc: comm_enter).  Every rank must RETURN -- the failing one with its
error, the others with an error out of the collective they were waiting in (the failing rank's ncclCommAbort releases them) -- and the communicators
are unusable afterwards."""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adversarial as A            # noqa: E402
from oatk_amd import HipSyncasm, _lib, pack_reads  # noqa: E402


def main():
    L = _lib.load()
    K, S, c = 101, 11, 4
    reads = A.hifi_like(150, 5000, 1500, seed=9, err=0.004)
    bounds = [0, 50, 100, 150]
    uid = (C.c_uint8 * 128)()
    assert L.oatk_comm_unique_id(uid) == 0
    res = [None] * 3

    def work(rank):
        h = HipSyncasm(0)
        comm = L.oatk_comm_create(uid, rank, 3, 0)
        seq, off, lens = pack_reads(reads[bounds[rank]:bounds[rank + 1]])
        h.scan_host(seq, off, lens, K, S, sid0=bounds[rank])
        h.count()
        t0 = time.time()
        try:
            h.merge_counts(comm)
            h.ec_sharded(comm, 0.02, c, 0.35)
            res[rank] = ("ok", time.time() - t0, "")
        except Exception as ex:                          # noqa: BLE001
            again = ""
            try:
                h.merge_counts(comm)
            except Exception as ex2:                     # noqa: BLE001
                again = str(ex2)
            res[rank] = ("error", time.time() - t0, str(ex) + " | again: " + again)
        L.oatk_comm_destroy(comm)
        h.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank hangs in a collective after its peer failed"
    for r, (what, dt, msg) in enumerate(res):
        assert what == "error" and dt < 60, (r, what, dt, msg)
        assert "aborted" in msg.split("| again:")[1], (r, msg)        # the communicator stays unusable
    assert "injected" in res[1][2], res[1]
    print("ok: rank 1 failed on request; ranks 0 and 2 returned after %.2f s and %.2f s" % (res[0][1], res[2][1]))


if __name__ == "__main__":
    main()
