#!/usr/bin/env python3
"""What a wavefront step costs in each variant of the solver (oatk_hip_debug_wf_ed / _wg, kernel time by HIP events: OATK_DEBUG_ED_TIME): jobs whose alignment climbs
to the band's edge -- an unrelated query, or one tandem array against another (every p-th diagonal runs on for a few dozen bases at every step).  With few jobs (one per
CU) the kernel's time over the steps of a job is the LATENCY of a step; with many it is the throughput.   python tools/stepbench.py"""
import os, sys, time, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from oatk_amd import HipSyncasm
    rng = np.random.default_rng(5)
    def rand(n): return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tolist())
    def tandem(n, unit, div):
        u = bytearray(unit * (n // len(unit) + 2))[:n]
        for p in np.flatnonzero(rng.random(n) < div): u[p] = b"ACGT"[(b"ACGT".index(bytes([u[p]])) + 1 + int(rng.integers(0, 3))) & 3]
        return bytes(u)
    hip = HipSyncasm(0)
    for name, bw, mk in (("unrelated bw 60", 60, lambda: (rand(3000), rand(3000))), ("unrelated bw 120", 120, lambda: (rand(6000), rand(6000))), ("unrelated bw 250", 250, lambda: (rand(12000), rand(12000))),
                         ("tandem period 37 at 4 % bw 120", 120, lambda: (lambda u: (tandem(6000, u, 0.04), tandem(6000, u[11:] + u[:11], 0.04)))(rand(37))),
                         ("tandem period 171 at 2 % bw 250", 250, lambda: (lambda u: (tandem(12000, u, 0.02), tandem(12000, u[50:] + u[:50], 0.02)))(rand(171)))):
        for n_jobs in (128, 4096):
            jobs = []
            for _ in range(n_jobs):
                t, q = mk()
                jobs.append((t, q, bw, [len(q)]))
            for wg in (0, 1, 2, 6, 8, 16):
                if wg and 2 * bw + 3 > (512 if wg == 8 else 896 if wg == 16 else 256 * wg): continue
                hip.wf_ed(jobs[:8], wg)
                sys.stderr.write("[case] %s | %d jobs | variant %d\n" % (name, n_jobs, wg)); sys.stderr.flush()
                res = hip.wf_ed(jobs, wg)
                sys.stderr.write("[steps] %.1f\n" % np.mean([r[0][0] for r in res])); sys.stderr.flush()
    sys.exit(0)
env = dict(os.environ, OATK_DEBUG_ED_TIME="1")
p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
case, ms = None, None
for ln in p.stderr.splitlines():
    if ln.startswith("[case]"): case = ln[7:]
    elif ln.startswith("[wf_ed_wg]") and case: ms = float(re.search(r"kernel ([0-9.]+) ms", ln).group(1))
    elif ln.startswith("[steps]") and case:
        st = float(ln.split()[1]); nj = int(case.split("|")[1].split()[0])
        print("%-60s kernel %9.3f ms, %5.0f steps per job: %8.0f ns per step of a job, %8.1f ns per step overall" % (case, ms, st, ms * 1e6 / st, ms * 1e6 / st / nj))
        case = None
if p.returncode: print(p.stderr[-3000:])
