"""GPU: syncasm()'s `meta` hand-off under the drop-in (run_syncasm.c:306-313).  `oatk` calls syncasm(..., scg_meta), pathfinder_minicircle
then re-aligns the reads and rebuilds the consensus on the structures it was handed (path_finder.c:811-819) and oatk frees them with scg_meta_destroy
(oatk.c:456, syncasm.c:87-92).  oracle/meta_driver.c does that sequence -- twice into the same meta, so scg_meta_clean (run_syncasm.c:307) frees the first
round while the second is live -- linked once over the reference's own objects and once over the drop-in (`make -C oracle ref_meta`).  With the
drop-in the member arrays live in arenas (include/oatk_syncasm.h) and a device batch stays resident behind the structures: the alignments and the
consensus made AFTER syncasm() returned must be the reference's byte for byte, served by the device, and every free must be a valid one (glibc's
MALLOC_CHECK_=3 aborts on an invalid or double free; the address-sanitized build of the same program watches the reference's own objects)."""
import filecmp
import os
import subprocess

import pytest

import adversarial as A
import cli_util as U
import ref_lib as R

REF = os.path.join(U.ROOT, "oracle", "_ref", "meta_ref")
DEV = os.path.join(U.ROOT, "oracle", "_ref", "meta_dropin")
ASAN = os.path.join(U.ROOT, "oracle", "_ref", "meta_dropin_asan")
FILES = (".meta.ra", ".meta.gfa", ".meta2.ra", ".meta2.gfa", ".utg.gfa", ".utg.final.gfa")

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(DEV)), reason="oracle/_ref meta drivers not built")]


def run(binary, out, k, s, c, fa, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([binary, out, str(k), str(s), str(c), "4", fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    return p.returncode, p.stderr.decode(errors="replace")


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    d = tmp_path_factory.mktemp("meta")
    reads = A.hifi_like(300, 50000, 5000, seed=378, err=0.001)
    fa = str(d / "reads.fa")
    R.write_fasta(reads, fa)
    rc, err = run(REF, str(d / "ref"), 301, 21, 6, fa)
    assert rc == 0, err[-2000:]
    return d, fa


@pytest.mark.parametrize("devices", [None, "0,0,0"])
def test_meta_handoff_alignment_and_consensus_after_syncasm_returned(case, devices):
    d, fa = case
    tag = "dev" + (devices or "").replace(",", "")
    env = {"OATK_DROPIN_LOG": "1", "MALLOC_CHECK_": "3", "MALLOC_PERTURB_": "165"}
    if devices:
        env.update({"OATK_DEVICES": devices, "OATK_DEBUG_WINDOW": "250000"})
    rc, err = run(DEV, str(d / tag), 301, 21, 6, fa, env)
    assert rc == 0, err[-3000:]
    for x in FILES:
        assert os.path.getsize(str(d / ("ref" + x))) > 100, x
        assert filecmp.cmp(str(d / ("ref" + x)), str(d / (tag + x)), shallow=False), x
    tab = U.served_table(err)
    for f in ("sr_read", "sr_db_stat", "collect_syncmer_from_reads", "make_syncmer_graph", "read_error_correction", "scg_read_alignment"):
        assert tab[f][2] == 0 and tab[f][0] >= 2, (f, tab[f], err[-3000:])            # both rounds, nothing from an original body
    # the two alignments and the two consensus rounds the DRIVER asked for after syncasm() had returned came from the device as well
    assert tab["scg_read_alignment"][0] >= 4 + 1 + 2
    assert tab["scg_syncmer_consensus"][0] > 40 and tab["scg_syncmer_consensus"][2] == 0 and tab["calc_syncmer_overlap"][2] == 0


@pytest.mark.skipif(not os.path.exists(ASAN), reason="address-sanitized driver not built")
def test_meta_handoff_under_the_address_sanitizer(case):
    """the reference's objects and the driver instrumented, the device libraries as they are: no invalid free, no use after free, no overflow in what the
    reference does with the handed-over structures.  (The HIP runtime does not always come up inside a sanitized process; that is then reported as a skip,
    never as a pass.)"""
    d, fa = case
    env = {"OATK_DROPIN_LOG": "1", "ASAN_OPTIONS": "detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1"}
    rc, err = run(ASAN, str(d / "asan"), 301, 21, 6, fa, env)
    assert "ERROR: AddressSanitizer" not in err, err[-4000:]
    tab = U.served_table(err)
    if rc != 0 or tab.get("sr_read", (0,))[0] == 0:
        pytest.skip("the device path did not come up under the sanitizer (rc %d): %s" % (rc, err[-300:]))
    for x in FILES:
        assert filecmp.cmp(str(d / ("ref" + x)), str(d / ("asan" + x)), shallow=False), x
    assert tab["sr_read"][2] == 0 and tab["read_error_correction"][2] == 0
