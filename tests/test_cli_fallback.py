"""CPU: the drop-in syncasm binary (the reference's translation units linked over oatk_amd/lib/liboatk_dropin.a, include/oatk_dropin.h) without a
device.  Every one of the six interposed calls must land in the maintainer's ORIGINAL body and both GFA files must equal the reference CLI's
byte for byte -- the link-level renaming works, nothing is lost, and there is no CPU path of ours behind it."""
import filecmp
import os

import pytest

import adversarial as A
import cli_util as U
import ref_lib as R
from oatk_amd import _lib

pytestmark = pytest.mark.skipif(not U.available(), reason="oracle/_ref CLI binaries not built (needs /root/reference)")


def no_gpu():
    return _lib.load().oatk_hip_device_count() == 0


@pytest.mark.parametrize("env", [{"OATK_DROPIN": "0"}, {}])
def test_dropin_cli_without_device_is_the_reference(tmp_path, env):
    if not env and not no_gpu():
        pytest.skip("a GPU is visible: tests/test_gpu_cli.py covers that")
    reads = A.hifi_like(300, 40000, 6000, seed=31, err=0.001)
    fa = str(tmp_path / "reads.fa")
    R.write_fasta(reads, fa)
    ref, dev = str(tmp_path / "ref"), str(tmp_path / "dev")
    U.run_cli(U.CLI_REF, fa, ref, 301, 6, 4, extra=["-s", "21"])
    e = dict(env)
    e["OATK_DROPIN_LOG"] = "1"
    _, err = U.run_cli(U.CLI_DROPIN, fa, dev, 301, 6, 4, env=e, extra=["-s", "21"])
    for suffix in (".utg.gfa", ".utg.final.gfa"):
        assert os.path.getsize(ref + suffix) > 100
        assert filecmp.cmp(ref + suffix, dev + suffix, shallow=False), suffix
    tab = U.served_table(err)
    assert set(tab) >= {"sr_read", "sr_db_stat", "collect_syncmer_from_reads", "make_syncmer_graph", "read_error_correction", "scg_read_alignment"}
    for f, (dev_calls, _, orig_calls, _) in tab.items():
        assert dev_calls == 0, f
    assert tab["sr_read"][2] == 1 and tab["make_syncmer_graph"][2] == 2 and tab["scg_read_alignment"][2] >= 2


def test_dropin_archive_defines_the_six_symbols_and_the_hooks():
    import subprocess
    out = subprocess.run(["nm", os.path.join(os.path.dirname(_lib.LIB_PATH), "liboatk_dropin.a")], stdout=subprocess.PIPE, check=True).stdout.decode()
    for sym in ("sr_read", "sr_db_stat", "collect_syncmer_from_reads", "make_syncmer_graph", "read_error_correction", "scg_read_alignment", "oatk_dropin_counts"):
        assert (" T %s\n" % sym) in out, sym
    for sym in ("oatk_hook_cons", "oatk_hook_ovl"):
        assert (" B %s\n" % sym) in out or (" D %s\n" % sym) in out, sym
    for sym in ("orig_sr_read", "orig_read_error_correction", "scg_consensus"):
        assert (" U %s\n" % sym) in out, sym
