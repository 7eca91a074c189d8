"""CPU: how fast host/gzsrc.c hands out the text of a gzip'ed file -- zlib on one thread (OATK_HOST_GZ_PARALLEL=0) against one member on many threads
(host/gzpar.c), by thread count and chunk size.  python tools/gzbench.py FILE.gz [threads ...]   (OATK_HOST_GZ_CHUNK_KB, OATK_GZPAR_LOG=1 for the phases)"""
import ctypes as C
import os
import resource
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oatk_amd import _lib  # noqa: E402


def lib():
    L = C.CDLL(_lib.HOST_LIB_PATH)
    L.oatk_gzsrc_open.restype = C.c_void_p
    L.oatk_gzsrc_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.oatk_gzsrc_read.restype = C.c_int64
    L.oatk_gzsrc_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.oatk_gzsrc_close.argtypes = [C.c_void_p]
    return L


def run(L, path, threads, buf, check=None):
    rc = C.c_int(0)
    t0 = time.perf_counter()
    g = L.oatk_gzsrc_open(path.encode(), threads, C.byref(rc))
    assert g, rc.value
    tot, crc = 0, 0
    while True:
        n = L.oatk_gzsrc_read(g, buf.ctypes.data, buf.size)
        assert n >= 0, n
        if n == 0:
            break
        if check is not None:
            crc = zlib.crc32(buf[:n], crc)
        tot += n
    L.oatk_gzsrc_close(g)
    return time.perf_counter() - t0, tot, crc


def main():
    path = sys.argv[1]
    threads = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
    L = lib()
    buf = np.empty(256 << 20, dtype=np.uint8)
    buf[:] = 0
    os.environ["OATK_HOST_GZ_PARALLEL"] = "0"
    dt, tot, crc0 = run(L, path, 1, buf, check=True)
    dt, tot, _ = run(L, path, 1, buf)
    print("zlib, one thread:          %7.3f s  %7.1f MB/s of text (%d bytes from %d)" % (dt, tot / dt / 1e6, tot, os.path.getsize(path)))
    os.environ["OATK_HOST_GZ_PARALLEL"] = "1000000"
    for th in threads:
        _, tot1, crc1 = run(L, path, th, buf, check=True)
        assert tot1 == tot and crc1 == crc0, "text differs"
        r0 = resource.getrusage(resource.RUSAGE_SELF)
        dts = [run(L, path, th, buf)[0] for _ in range(3)]
        r1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu = (r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime) / 3
        dt = min(dts)
        print("one member on %3d threads: %7.3f s  %7.1f MB/s of text   (cpu %.2f s a run, of it system %.2f; runs %s)"
              % (th, dt, tot / dt / 1e6, cpu, (r1.ru_stime - r0.ru_stime) / 3, " ".join("%.3f" % x for x in dts)))


if __name__ == "__main__":
    main()
