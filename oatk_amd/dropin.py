"""Measurement at the drop-in boundary for bench.py (SURVEY.md 8d timing ii).

time_sr_read_packed   pinned host ASCII -> device -> scan -> the reference's sr_t arrays filled on the host: the contract of sr_read
                      (syncmer.c:487) as liboatk_host.so serves it (oatk_sr_read_packed), PCIe and the per-read mallocs included.
"""
import ctypes as C
import time

from . import _lib


def _host():
    H = C.CDLL(_lib.HOST_LIB_PATH)
    vp = C.c_void_p
    H.oatk_sr_db_new.restype = vp
    H.oatk_sr_db_new.argtypes = [C.c_int, C.c_int]
    H.oatk_sr_read_packed.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp]
    H.oatk_sr_db_clean.argtypes = [vp]
    H.oatk_host_set_arena.argtypes = [C.c_int]
    return H


def time_sr_read_packed(hip, readset, first, n_reads, k, s):
    import torch
    H = _host()
    lens, off, total = readset.layout(first, n_reads)
    pinned = torch.empty(max(total, 64), dtype=torch.uint8).pin_memory()
    seq, off, lens = readset.slice(first, n_reads, out=pinned.numpy())
    bases = int(lens.sum())
    best = arena_best = None
    import os
    os.environ["OATK_HOST_HEAP_TUNE"] = "1"      # the malloc-per-array runs with the heap tuning the library offers as an opt-in (srdb.c); restored after each fill
    for rep in range(4):
        # runs 0, 1: every member array its own malloc'ed block, as the reference's sr_destroy needs them; runs 2, 3: ARENAS, for a caller that owns
        # the destroy functions as the drop-in binary does (include/oatk_syncasm.h)
        H.oatk_host_set_arena(1 if rep >= 2 else 0)
        db = H.oatk_sr_db_new(k, s)
        hip.sync()
        t0 = time.perf_counter()
        rc = H.oatk_sr_read_packed(hip.h, db, seq.ctypes.data, off.ctypes.data, lens.ctypes.data, n_reads, total, None)
        dt = time.perf_counter() - t0
        if rc != 0:
            raise _lib.OatkHipError("oatk_sr_read_packed failed (code %d)" % rc)
        t0 = time.perf_counter()
        H.oatk_sr_db_clean(db)
        C.CDLL(None).free(C.c_void_p(db))
        t_free = time.perf_counter() - t0
        if rep < 2 and (best is None or dt < best[0]):
            best = (dt, t_free)
        if rep >= 2 and (arena_best is None or dt < arena_best[0]):
            arena_best = (dt, t_free)
    H.oatk_host_set_arena(0)
    os.environ.pop("OATK_HOST_HEAP_TUNE", None)
    return {"value": round(bases / best[0] / 1e9, 3), "unit": "Gbases/s", "ms": round(best[0] * 1e3, 1), "ms_free": round(best[1] * 1e3, 1),
            "with_arenas": {"value": round(bases / arena_best[0] / 1e9, 3), "unit": "Gbases/s", "ms": round(arena_best[0] * 1e3, 1), "ms_free": round(arena_best[1] * 1e3, 1),
                            "note": "the reads of a piece share one block (oatk_host_set_arena): for callers that own sr_destroy / sr_db_clean, as the drop-in binary does"},
            "workload": "%d reads (%.2f Gbases) as a packed ASCII stream in pinned host memory -> oatk_sr_read_packed: H2D, scan, D2H of every per-read array, "
                        "sr_db_t filled with the reference's own malloc'ed members (sr_read's contract)" % (n_reads, bases / 1e9)}
