#!/usr/bin/env python3
"""The config-1 surrogate (oatk_amd.synth.CONFIG1S) on the device: the resident step with its phase timings, and the syncasm CLI on the .fa.gz in its
three gzip forms with the drop-in's per-function log.  Development aid and the source of bench.py's `config1s` numbers.
    python tools/config1s_run.py [n_reads] [--cli N] [--ref]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from oatk_amd import HipSyncasm, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("n_reads", type=int, nargs="?", default=synth.CONFIG1S["n_reads"])
ap.add_argument("--cli", type=int, default=0, help="reads of the CLI runs (0: none)")
ap.add_argument("--ref", action="store_true", help="also the reference binary")
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--forms", default="plain,one,bgzf,members", help="which file forms the CLI runs on")
args = ap.parse_args()
K, S = 1001, 31
cfg = dict(synth.CONFIG1S)
c = cfg["min_k_cov"]
rs = synth.MixReadSet(**cfg)
seq, off, lens = rs.slice(0, args.n_reads)
bases = int(lens.sum())
hip = HipSyncasm(0)
hip.set_timing(True)
out = {"reads": args.n_reads, "gbases": round(bases / 1e9, 3)}
for it in range(args.steps + 1):
    t0 = time.perf_counter()
    hip.scan_host(seq, off, lens, K, S)
    t1 = time.perf_counter()
    hip.count()
    hip.ec_graph(light_c=c)
    st = hip.ec(0.02, c, 0.35)
    hip.sync()
    t2 = time.perf_counter()
    tm = hip.timing()
    if it:
        print("step %d: scan (host -> device incl.) %.1f ms, count + EC graph + EC %.1f ms; phases %s" % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1),
              {k: round(v, 2) for k, v in tm.items() if v > 0.005}), flush=True)
out["phases_ms"] = {k: round(v, 3) for k, v in tm.items() if v > 0.0005}
out["device_ms"] = round(sum(tm.values()), 2)
out["blocks"] = [int(x) for x in st[:11]]
inf = hip.info()
out["n_occ"], out["n_scm"] = int(inf["n_occ"]), int(inf["n_scm"])
print(json.dumps(out), flush=True)
if args.cli:
    import cli_util as CU
    d = os.environ.get("TMPDIR", "/tmp")
    n = args.cli
    res = {}
    for name, mode, mb in (("plain", synth.FA_PLAIN, 0), ("one", synth.FA_GZ, 0), ("bgzf", synth.FA_BGZF, 0), ("members", synth.FA_GZ_MEMBERS, 200_000_000)):
        if name not in args.forms.split(","):
            continue
        p = os.path.join(d, "c1s_%s.fa%s" % (name, "" if mode == 0 else ".gz"))
        t0 = time.perf_counter()
        synth.write_fasta(p, seq, off[:n], lens[:n], mode=mode, member_bytes=mb)
        tw = time.perf_counter() - t0
        t0 = time.perf_counter()
        if mode:
            os.system("zcat %s > /dev/null" % p)
        tz = time.perf_counter() - t0
        t, err = CU.run_cli(CU.CLI_DROPIN, p, os.path.join(d, "c1s_dev_" + name), K, c, args.threads, {"OATK_DROPIN_LOG": "1"})
        res[name] = {"file_MB": os.path.getsize(p) >> 20, "write_s": round(tw, 2), "zcat_s": round(tz, 2), "dropin_s": round(t, 2)}
        print("== %s: %s" % (name, res[name]))
        print("\n".join(l for l in err.splitlines() if ("oatk_" in l or "oatk::" in l) and "fill_range" not in l)[-4500:], flush=True)
    if args.ref:
        t, _ = CU.run_cli(CU.CLI_REF, os.path.join(d, "c1s_one.fa.gz"), os.path.join(d, "c1s_ref"), K, c, args.threads)
        res["reference_s"] = round(t, 2)
        import filecmp
        res["gfa_identical"] = {nm: all(filecmp.cmp(os.path.join(d, "c1s_ref" + x), os.path.join(d, "c1s_dev_%s%s" % (nm, x)), shallow=False) for x in (".utg.gfa", ".utg.final.gfa"))
                                for nm in args.forms.split(",")}
    print(json.dumps(res))
