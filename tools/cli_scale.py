#!/usr/bin/env python3
"""The syncasm CLI at scale (SURVEY.md 8d timing iii; the north-star comparison): N synthetic HiFi reads of BASELINE.json's config 3 genome as a
FASTA file in shared memory, the drop-in binary with its per-function log, optionally the reference binary on the same file at -t T, GFA md5s.

    python tools/cli_scale.py <n_reads> [--ref] [--threads T] [--workload config3] [--keep]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oatk_amd.synth import CONFIGS, ReadSet  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref")          # noqa: built artefacts of `make ref ref_dropin`

ap = argparse.ArgumentParser()
ap.add_argument("n_reads", type=int)
ap.add_argument("--ref", action="store_true", help="also run the reference binary")
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--workload", default="config3")
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--keep", action="store_true")
ap.add_argument("--devices", default="", help="also run the drop-in over several handles (OATK_DEVICES, e.g. 0,0)")
ap.add_argument("--variants", default="", help="also run the drop-in under these environments, on the same file in the same session: 'A=1,B=2;C=3' (A/B of host-side switches)")
args = ap.parse_args()

cfg = dict(CONFIGS[args.workload])
cfg["n_reads"] = args.n_reads
c = cfg["min_k_cov"]
rs = ReadSet(**cfg)
fa = os.path.join(args.dir, "oatk_cli_%d.fa" % args.n_reads)
t0 = time.perf_counter()
bases = 0
with open(fa, "wb") as f:
    step = 100000
    for first in range(0, args.n_reads, step):
        n = min(step, args.n_reads - first)
        seq, off, lens = rs.slice(first, n)
        bases += int(lens.sum())
        parts = []
        for i in range(n):
            parts.append(b">r%d\n" % (first + i))
            parts.append(seq[int(off[i]):int(off[i]) + int(lens[i])].tobytes())
            parts.append(b"\n")
        f.write(b"".join(parts))
print("wrote %s: %d reads, %.2f Gbases in %.1f s" % (fa, args.n_reads, bases / 1e9, time.perf_counter() - t0), flush=True)


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def run(binary, tag, env=None):
    out = os.path.join(args.dir, "oatk_cli_%s" % tag)
    e = dict(os.environ)
    e.update(env or {})
    t = time.perf_counter()
    print("[cli_scale] start %s" % binary, flush=True)
    p = subprocess.run([os.path.join(BIN, binary), "-k", "1001", "-c", str(c), "-t", str(args.threads), "-o", out, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    dt = time.perf_counter() - t
    err = p.stderr.decode(errors="replace")
    res = {"binary": binary, "rc": p.returncode, "wall_s": round(dt, 2), "gbases_per_s": round(bases / dt / 1e9, 3)}
    for sfx in (".utg.gfa", ".utg.final.gfa"):
        if os.path.exists(out + sfx):
            res["md5" + sfx] = md5(out + sfx)
            res["bytes" + sfx] = os.path.getsize(out + sfx)
            if not args.keep:
                os.unlink(out + sfx)
    return res, err


report = {"n_reads": args.n_reads, "gbases": round(bases / 1e9, 3), "threads": args.threads, "workload": args.workload}
r, err = run("syncasm_dropin", "dev", {"OATK_DROPIN_LOG": "1"})
report["dropin"] = r
report["dropin_log"] = [l for l in err.splitlines() if "oatk_dropin]" in l or "oatk_sr_read_files]" in l]
print("\n".join(l for l in err.splitlines() if "oatk_" in l), flush=True)
if args.devices:
    rm, errm = run("syncasm_dropin", "mul", {"OATK_DROPIN_LOG": "1", "OATK_DEVICES": args.devices})
    report["dropin_several_handles"] = dict(rm, devices=args.devices, same_gfa_as_one_handle=all(rm.get(k) == r.get(k) and rm.get(k) for k in ("md5.utg.gfa", "md5.utg.final.gfa")))
    report["dropin_several_handles_log"] = [l for l in errm.splitlines() if "oatk_dropin]" in l or "oatk_sr_read_files]" in l]
    print("\n".join(l for l in errm.splitlines() if "oatk_" in l), flush=True)
for k, spec in enumerate(v for v in args.variants.split(";") if v):
    env = dict(x.split("=", 1) for x in spec.split(","))
    for rep in range(2):
        rv, errv = run("syncasm_dropin", "var%d" % k, dict(env, OATK_DROPIN_LOG="1"))
        line = [l for l in errv.splitlines() if "oatk_sr_read_files]" in l]
        print("[variant %s] wall %.2f s; %s" % (spec, rv["wall_s"], line[0] if line else ""), flush=True)
        report.setdefault("variants", []).append({"env": spec, "wall_s": rv["wall_s"], "sr_read": line[0] if line else None})
if args.ref:
    r2, _ = run("syncasm", "ref")
    report["reference"] = r2
    report["speedup"] = round(r2["wall_s"] / r["wall_s"], 2)
    report["gfa_identical"] = all(r.get(k) == r2.get(k) and r.get(k) for k in ("md5.utg.gfa", "md5.utg.final.gfa"))
if not args.keep:
    os.unlink(fa)
print(json.dumps(report), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "cli_scale_%d.json" % args.n_reads), "w") as f:
    json.dump(report, f, indent=1)
    f.write("\n")
