#!/usr/bin/env python3
"""oatk_sr_read_packed in one pass and in pieces (OATK_HOST_PACKED_PIECES=1, host/srdb.c): time, and the structs' contents compared member by member.
    python tools/packed_ab.py [n_reads]"""
import ctypes as C
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oatk_amd import HipSyncasm, dropin  # noqa: E402
from oatk_amd.synth import CONFIGS, ReadSet  # noqa: E402


class Sr(C.Structure):      # include/oatk_syncasm.h: oatk_sr_t
    _fields_ = [("sid", C.c_uint64), ("sname", C.c_void_p), ("hoco_l", C.c_uint32), ("hoco_s", C.c_void_p), ("ho_rl", C.c_void_p), ("ho_l_rl", C.c_void_p),
                ("n_nucl", C.c_void_p), ("n", C.c_uint32), ("m_pos", C.c_void_p), ("s_mer", C.c_void_p), ("k_mer", C.c_void_p)]


class SrDb(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.POINTER(Sr)), ("k", C.c_int), ("s", C.c_int), ("stats", C.c_void_p)]


def digest(db, step):
    d = C.cast(db, C.POINTER(SrDb)).contents
    crc = 0
    for i in range(0, d.n, step):
        r = d.a[i]
        crc = zlib.crc32(np.array([r.sid, r.hoco_l, r.n], np.uint64).tobytes(), crc)
        for ptr, nbytes in ((r.hoco_s, (r.hoco_l + 3) // 4), (r.ho_rl, r.hoco_l), (r.m_pos, 4 * r.n), (r.s_mer, 8 * r.n), (r.k_mer, 8 * r.n)):
            if nbytes:
                crc = zlib.crc32(C.string_at(ptr, nbytes), crc)
    return d.n, crc


n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cfg = dict(CONFIGS["config3"])
cfg["n_reads"] = n_reads
rs = ReadSet(**cfg)
hip = HipSyncasm(0)
H = dropin._host()
lens, off, total = rs.layout(0, n_reads)
pinned = torch.empty(max(total, 64), dtype=torch.uint8).pin_memory()
seq, off, lens = rs.slice(0, n_reads, out=pinned.numpy())
bases = int(lens.sum())
res = {}
for pieces in ("0", "1"):
    os.environ["OATK_HOST_PACKED_PIECES"] = pieces
    for arena in (0, 1):
        H.oatk_host_set_arena(arena)
        ts = []
        for rep in range(3):
            db = H.oatk_sr_db_new(1001, 31)
            hip.sync()
            t0 = time.perf_counter()
            rc = H.oatk_sr_read_packed(hip.h, db, seq.ctypes.data, off.ctypes.data, lens.ctypes.data, n_reads, total, None)
            ts.append(time.perf_counter() - t0)
            assert rc == 0, rc
            if rep == 2:
                res[(pieces, arena)] = digest(db, 7)
            H.oatk_sr_db_clean(db)
            C.CDLL(None).free(C.c_void_p(db))
        print("pieces=%s arenas=%d: %s ms -> best %.1f Gbases/s; digest %s" % (pieces, arena, [round(1e3 * t, 1) for t in ts], bases / min(ts) / 1e9, res[(pieces, arena)]), flush=True)
H.oatk_host_set_arena(0)
assert len(set(res.values())) == 1, res
print("identical structs in all four forms")
