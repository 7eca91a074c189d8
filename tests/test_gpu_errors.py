"""GPU: the C ABI fails loudly, with a message, on misuse -- and keeps working afterwards (the error behaviour a binding has to rely on)."""
import ctypes as C

import numpy as np
import pytest

import adversarial as A
from oatk_amd import HipSyncasm, OatkHipError, pack_reads

pytestmark = pytest.mark.gpu


def test_calls_out_of_order_and_bad_arguments():
    hip = HipSyncasm(0)
    try:
        with pytest.raises(OatkHipError, match="count before scan"):
            hip.count()
        with pytest.raises(OatkHipError):
            hip.fetch("HOCO_L")
        with pytest.raises(OatkHipError, match="oatk_hip_ingest"):
            hip.fetch("INGEST_OFF")
        with pytest.raises(OatkHipError, match="oatk_hip_ingest"):
            hip.scan_ingested(1001, 31)
        reads = A.hifi_like(40, 20000, 3000, seed=2)
        seq, off, lens = pack_reads(reads)
        for k, s in ((31, 31), (10, 0), (1001, 32), (hip.L.oatk_hip_max_k() + 1, 31)):
            with pytest.raises(OatkHipError, match="out of range"):
                hip.scan_host(seq, off, lens, k, s)
        hip.scan_host(seq, off, lens, 301, 21)
        for what in (hip.ec_graph, lambda: hip.ec(0.02, 4, 0.35), lambda: hip.consensus(1), lambda: hip.ec_mark(4, 0.35)):
            with pytest.raises(OatkHipError):                      # all of them need the count
                what()
        with pytest.raises(OatkHipError):
            hip.fetch("SCM_COV")
        hip.count()
        with pytest.raises(OatkHipError, match="oatk_hip_ec_graph first"):
            hip.ec(0.02, 4, 0.35)                                   # no graph, host or resident
        with pytest.raises(OatkHipError):
            hip.fetch("EC_KMER")
        with pytest.raises(OatkHipError):
            hip.fetch("EG_ARC_V")
        with pytest.raises(OatkHipError):
            hip.ec_correct(0.02)                                    # not marked
        with pytest.raises(OatkHipError, match="ascending|0 <= cap_t0"):
            hip._check(hip.L.oatk_hip_debug_ec_tiers(hip.h, 500, 100), "oatk_hip_debug_ec_tiers")
        hip.ec_graph()
        st = hip.ec(0.02, 4, 0.35)                                  # and after all that the handle still works
        assert int(st[0] + st[5] + st[10]) > 0
        hip.scan_host(seq, off, lens, 301, 21)                      # a new scan invalidates what belonged to the old batch
        with pytest.raises(OatkHipError):
            hip.fetch("EC_KMER")
        with pytest.raises(OatkHipError):
            hip.fetch("CONS_RL")
        raw = hip.stat_raw()                                        # statistics work at any stage
        assert raw["n_syncmers"] == hip.info()["n_occ"] > 0
    finally:
        hip.close()


def test_null_handle_and_no_device():
    from oatk_amd import _lib
    L = _lib.load()
    assert L.oatk_hip_count(None) == _lib.E_NODEV and L.oatk_hip_sync(None) == _lib.E_NODEV
    assert L.oatk_hip_create(10 ** 6) is None                       # no such device: NULL, never a CPU stand-in
