"""CPU only: the streamed reader (oatk_sr_read_files: host/ingest_host.c, gzsrc.c, gzpar.c, srdb.c, ingest_estimate.h) END TO END over a stub device, under an
address-space limit -- the harness VERDICT r05 asked for after a reader that extrapolated from a position that stood still touched tens of terabytes on three GPU boxes.

tests/c/stub_device.c stands in for liboatk_hip.so (FASTA record scan + real homopolymer compression + fabricated syncmers, "device" memory counted and capped at 288 GB);
the host library's own sources are linked against it; tests/c/reader_stub_main.c writes a >= 100 MB .fa.gz (ONE member -- the case that hurt -- and BGZF), sets RLIMIT_AS,
reads the file back and checks every read's name, packed bases, run lengths, N list, long runs and syncmer arrays.  What the limit catches was checked by hand when this
was written: with `oatk_gzsrc_tell_in` made to stand still through a member and the estimate's trust test removed (round 5's reader) the run ends here with
"oatk_hip_scan_reserve(3.88e+15 bytes ...) refused" and OATK_E_NOMEM -- in the container, in a second."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "oatk_amd", "csrc", "host")
OUT = os.path.join(ROOT, "tests", "_build", "stub")


def _run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw)


@pytest.fixture(scope="module")
def reader_stub():
    os.makedirs(OUT, exist_ok=True)
    inc = "-I" + os.path.join(ROOT, "include")
    hip, host, exe = (os.path.join(OUT, n) for n in ("liboatk_hip.so", "liboatk_host.so", "reader_stub"))
    src = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".c"))
    stub = os.path.join(ROOT, "tests", "c", "stub_device.c")
    _run(["gcc", "-O2", "-Wall", "-fPIC", "-shared", inc, "-o", hip, stub, "-lpthread"])
    _run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", inc, "-o", host] + src + ["-L" + OUT, "-loatk_hip", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lm", "-ldl", "-lz"])
    # the device entry points the host library names and the reader never reaches (count, EC, graphs, collectives): traps, so that the library loads
    want = {ln.split()[-1] for ln in _run(["nm", "-D", "--undefined-only", host]).stdout.splitlines() if " oatk_" in ln}
    have = {ln.split()[-1] for ln in _run(["nm", "-D", "--defined-only", hip]).stdout.splitlines()}
    traps = os.path.join(OUT, "traps.c")
    with open(traps, "w") as f:
        f.write("#include <stdlib.h>\n" + "".join("int %s(void) { abort(); return 0; }\n" % s for s in sorted(want - have)))
    _run(["gcc", "-O2", "-Wall", "-fPIC", "-shared", inc, "-o", hip, stub, traps, "-lpthread"])
    _run(["gcc", "-O2", "-Wall", inc, "-o", exe, os.path.join(ROOT, "tests", "c", "reader_stub_main.c"), "-L" + OUT, "-loatk_host", "-loatk_hip", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lm"])
    return exe


@pytest.mark.parametrize("mode,arena,reads", [(1, 0, 28000), (2, 1, 28000), (3, 1, 6000), (0, 0, 6000)], ids=["one_member_140MB", "bgzf_140MB_arenas", "several_members", "plain"])
def test_streamed_reader_end_to_end_under_an_address_space_limit(reader_stub, tmp_path, mode, arena, reads):
    path = str(tmp_path / ("reads.fa" + (".gz" if mode else "")))
    p = subprocess.run([reader_stub, path, str(mode), str(reads), "15000", "8", str(arena), "8"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["bad"] == 0 and r["reads"] == reads == r["sr_db_m"]          # every read equal; the reads' array given back to its size
    if reads >= 28000 and mode:
        assert r["file_bytes"] >= 100e6                                    # the size VERDICT r05 names
    assert r["n_bases"] == 7 and r["long_runs"] == 1 and r["syncmers"] > 0
    # what the reader asked the device and the host for: a few windows' worth, not an extrapolation
    assert r["device_biggest_request"] < 3 * r["text_bytes"] + (1 << 30), r
    assert r["host_peak_rss_kb"] * 1024 < 8 * r["text_bytes"] + (2 << 30), r
