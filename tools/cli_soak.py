#!/usr/bin/env python3
"""The drop-in CLI again and again on the same files, every run's two GFA files compared with the reference's (development aid: rare differences, rare stalls).
    python tools/cli_soak.py [runs] [reads]      -- the config-1 surrogate as .fa.gz (one member and BGZF in turn)"""
import filecmp, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oatk_amd import synth
import cli_util as CU

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
cfg = dict(synth.CONFIG1S); rs = synth.MixReadSet(**cfg)
seq, off, lens = rs.slice(0, n)
d = tempfile.mkdtemp(prefix="oatk_soak_", dir=os.environ.get("TMPDIR", "/tmp"))
files = {}
for f, mode in (("one", synth.FA_GZ), ("bgzf", synth.FA_BGZF)):
    files[f] = os.path.join(d, f + ".fa.gz")
    synth.write_fasta(files[f], seq, off[:n], lens[:n], mode=mode, member_bytes=200_000_000)
t_ref, _ = CU.run_cli(CU.CLI_REF, files["one"], os.path.join(d, "ref"), 1001, cfg["min_k_cov"], 32)
print("reference %.1f s" % t_ref, flush=True)
bad = 0
ts = []
for i in range(runs):
    f = ("one", "bgzf")[i & 1]
    t, err = CU.run_cli(CU.CLI_DROPIN, files[f], os.path.join(d, "dev"), 1001, cfg["min_k_cov"], 32, {"OATK_DROPIN_LOG": "1"})
    same = all(filecmp.cmp(os.path.join(d, "ref" + x), os.path.join(d, "dev" + x), shallow=False) for x in (".utg.gfa", ".utg.final.gfa"))
    tab = CU.served_table(err)
    orig = {k: v[2] for k, v in tab.items() if v[2] > 0}
    ts.append(t)
    if not same or orig:
        bad += 1
        print("run %d (%s): %.2f s  GFA identical: %s  original bodies: %s" % (i, f, t, same, orig), flush=True)
print("%d runs: %d bad; seconds min %.2f median %.2f max %.2f" % (runs, bad, min(ts), sorted(ts)[len(ts) // 2], max(ts)))
for fn in os.listdir(d):
    os.unlink(os.path.join(d, fn))
os.rmdir(d)
sys.exit(1 if bad else 0)
