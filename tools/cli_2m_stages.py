#!/usr/bin/env python3
"""The drop-in CLI on config 3's FASTA file (2 M reads, 30 GB of text), several runs in a row with OATK_DROPIN_LOG=1: sr_read's stages run by run
(development aid: where the 2.1 - 3.4 s of sr_read go and what varies).   gpurun -- 'python tools/memcap.py --rss-gb 400 --timeout 1500 -- python tools/cli_2m_stages.py [runs] [reads]'
Extra environment for the CLI: CLI_ENV="A=1 B=2"."""
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cli_util as CU  # noqa: E402
from oatk_amd import synth  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
cfg = synth.CONFIGS["config3"]
rs = synth.ReadSet(genome_len=cfg["genome_len"], n_reads=cfg["n_reads"], mean_len=cfg["mean_len"])
seq, off, lens = rs.slice(0, n)
d = tempfile.mkdtemp(prefix="oatk_cli2m_", dir=os.environ.get("TMPDIR", "/tmp"))
fa = os.path.join(d, "reads.fa")
t0 = time.perf_counter()
synth.write_fasta(fa, seq, off, lens, mode=synth.FA_PLAIN)
print("file written in %.1f s (%.1f GB)" % (time.perf_counter() - t0, os.path.getsize(fa) / 1e9), flush=True)
del seq
extra = dict(kv.split("=", 1) for kv in os.environ.get("CLI_ENV", "").split())
try:
    for i in range(runs):
        t, err = CU.run_cli(CU.CLI_DROPIN, fa, os.path.join(d, "out"), 1001, cfg["min_k_cov"], 32, dict({"OATK_DROPIN_LOG": "1"}, **extra))
        print("run %d: %.2f s" % (i, t))
        for ln in err.splitlines():
            if re.search(r"oatk_sr_read_files\]|device text buffer|sr_read  |read_error_correction  |scg_read_alignment  |exit handlers|oatk alloc", ln):
                print("   ", ln.strip()[:330])
        sys.stdout.flush()
finally:
    for fn in os.listdir(d):
        os.unlink(os.path.join(d, fn))
    os.rmdir(d)
