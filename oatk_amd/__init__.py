"""oatk_amd -- MI355X-native (gfx950) implementation of oatk's syncasm hot path.

Device kernels and the C ABI live in oatk_amd/csrc (built into oatk_amd/lib/liboatk_hip.so by
__graft_entry__.build()); this package is the thin Python host mirror used by tests and bench.py.
"""
from ._lib import OatkHipError, READ_ALIGN  # noqa: F401
from .device import HipSyncasm, pack_reads  # noqa: F401
