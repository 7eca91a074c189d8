#!/usr/bin/env python3
"""Does VRAM that another process has just given back slow the CLI down (scg_read_alignment 0.49 -> 0.74 - 1.44 s, sr_read 2.08 -> 3.42 s on the driver's clock)?
The surrogate's CLI on a .fa.gz three times: as it is; right after THIS process allocated, touched and freed `gb` GB of VRAM; and while this process still holds them.
    python tools/vram_churn_cli.py [reads] [gb]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oatk_amd import synth
import cli_util as CU

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
gb = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = dict(synth.CONFIG1S); rs = synth.MixReadSet(**cfg)
seq, off, lens = rs.slice(0, n)
d = os.environ.get("TMPDIR", "/tmp")
p = os.path.join(d, "churn.fa.gz")
synth.write_fasta(p, seq, off, lens, mode=synth.FA_BGZF)

EXTRA = {}


def run(tag):
    t, err = CU.run_cli(CU.CLI_DROPIN, p, os.path.join(d, "churn_out"), 1001, cfg["min_k_cov"], 32, dict({"OATK_DROPIN_LOG": "1"}, **EXTRA))
    tab = [l.split("]")[1].split() for l in err.splitlines() if "[M::oatk_dropin]" in l and len(l.split()) >= 6 and l.split()[1] in ("sr_read", "read_error_correction", "scg_read_alignment", "make_syncmer_graph")]
    calls = [l.split("MI355X,")[1].split()[0] for l in err.splitlines() if "scg_read_alignment: MI355X" in l]
    print("%-44s %.2f s: %s | alignment calls %s" % (tag, t, ", ".join("%s %s" % (r[0], r[2]) for r in tab), " ".join(calls)), flush=True)
    for l in err.splitlines():
        if "[ra]" in l: print("      " + l, flush=True)

for extra in ({"OATK_DEBUG_RA_OWN_SLAB": "1"}, {}, {"OATK_DEBUG_RA_OWN_SLAB": "1"}, {}):
    EXTRA = extra
    print("-- child's extra environment:", extra, flush=True)
    run("as it is")
    x = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda").fill_(1) for _ in range(gb)]
    torch.cuda.synchronize()
    run("while this process holds %d GB" % gb)
    del x
    torch.cuda.empty_cache(); torch.cuda.synchronize()
    run("right after this process freed %d GB" % gb)
    run("... and once more")
    time.sleep(3)
    run("three seconds later")
