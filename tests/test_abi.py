"""CPU: the C-ABI libraries load and export every symbol their headers declare (no compute: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from oatk_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_hip_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = sum((declared(h, "oatk_hip_") for h in sorted(os.listdir(os.path.join(ROOT, "include"))) if h.startswith("oatk_hip")), [])   # one library
    names += declared("oatk_hip_multi.h", "oatk_comm_")
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), n
    assert set(_lib.EXPORTS) <= set(names)
    assert L.oatk_hip_abi_version() == 1


def test_host_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.HOST_LIB_PATH)
    L = ctypes.CDLL(_lib.HOST_LIB_PATH)
    # (oatk_dropin.h belongs to the static archive that is linked into the CLI: tests/test_cli_fallback.py checks its symbols)
    for hdr in [h for h in os.listdir(os.path.join(ROOT, "include")) if h not in ("oatk_hip.h", "oatk_dropin.h") and h.endswith(".h")]:
        for n in declared(hdr, "oatk_"):
            if n.startswith("oatk_hip_") or n.startswith("oatk_comm_"):
                continue
            assert hasattr(L, n), (hdr, n)


def test_no_silent_fallback_without_a_gpu():
    """without a device the product path raises; it never computes on the CPU"""
    L = _lib.load()
    if L.oatk_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    from oatk_amd import HipSyncasm, OatkHipError
    with pytest.raises(OatkHipError):
        HipSyncasm(0)


def test_product_does_not_import_the_oracle():
    """only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke may touch oracle/"""
    for base, _, files in list(os.walk(os.path.join(ROOT, "oatk_amd"))) + list(os.walk(os.path.join(ROOT, "tools"))) + list(os.walk(os.path.join(ROOT, "include"))):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".hpp", ".inc", ".sh")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "ref_lib" not in txt and "liboatk_oracle" not in txt and "oracle/" not in txt, f


def test_the_scale_model_of_bench_is_plain_arithmetic_on_its_inputs():
    """bench.py `scale_model`: the predicted strong-scaling step is compute(reads / N) + latencies + bytes over links -- more GPUs
    never predict a slower compute part, what is predicted is printed with its parts, and the parts add up."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    tr1, tr8 = [40, 3000, 6, 5 * 10**8, 8, 4 * 10**8, 6, 12 * 10**8], [40, 3000, 6, 7 * 10**7, 8, 39 * 10**7, 6, 15 * 10**7]
    m = b.scale_model(2_000_000, 95.0, tr1, 250_000, 14.0, tr8, 30 * 10**9, 90.0)
    assert m["measured"]["collectives_per_step"] == 60
    prev = None
    for n in ("2", "4", "8"):
        p = m["predicted"][n]
        assert abs(sum(p["parts_ms"].values()) - p["ms_per_step"]) < 0.02
        assert prev is None or p["parts_ms"]["compute"] < prev
        prev = p["parts_ms"]["compute"]
        assert abs(p["speedup_over_1"] - 90.0 / p["ms_per_step"]) < 0.01
