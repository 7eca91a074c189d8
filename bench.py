#!/usr/bin/env python3
"""bench.py -- HiFi Gbases/s through syncasm (syncmer + syncerr) at k = 1001 on MI355X: BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config3|config2|config5] [--reads-per-gpu R]

One "step" = one pass of the hot path over one batch of synthetic HiFi reads that is already resident in HBM when the
timed region starts: run_syncasm.c:81-131 without the file parsing --
    scan   sr_read's per-read analysis: homopolymer compression, closed-syncmer selection, k-mer hash   (syncmer.c:243-421)
    count  collect_syncmer_from_reads: syncmer ids, occurrence lists                                     (syncmer.c:1397)
    graph  make_syncmer_graph(sr_db, scm_db, 0, 0.) + the hoco arc overlaps of scg_consensus             (run_syncasm.c:109-117)
    ec     read_error_correction (Levenshtein path search per error block) + update_syncmer_db           (syncerr.c:819)
At N = 1 the workload is BASELINE.json configs[2] (2 M reads x ~15 kb = 30 Gbases, k = 1001, s = 31, -c 30), the configuration
the metric is quoted on; it occupies ~70 GB of the 288 GB.  For N > 1 (launched by torch.distributed.run, one rank per GPU) the default
is BASELINE.json configs[3]: the SAME 2 M reads sharded by record over the N GPUs (rank r owns reads [r*R, (r+1)*R), R = 2 M / N:
"strong" scaling, so the N = 1 -> 8 ratio reads directly against north_star's ">= 6x"), the per-GPU syncmer tables merged by hash range,
the graph built from everybody's candidate pairs, every rank correcting its own reads (include/oatk_hip_multi.h over RCCL); the `weak`
sub-object repeats the step with 2 M reads on EVERY GPU (--scaling weak makes that the headline instead).

Prints ONE JSON line on rank 0.  `value` is the whole step above.  Sub-objects: `scan_count` (the same batch through scan +
count only), `config2` (BASELINE.json configs[1]: 200 k reads, scan + count, and with the EC round), `results_back` (SURVEY 8d
timing ii: pinned host ASCII -> device -> the reference's sr_t arrays filled on the host), `ingest` (from FASTA text, PCIe
included), `cli` (timing iii: the drop-in syncasm CLI against the reference's CLI on the same FASTA file, when both binaries
exist).  `roofline` prices the dominant kernel against the HBM roofline with its duration measured live by HIP events on the
stream the kernels run on; `cpu_baseline` is the compiled reference (oracle/_ref, built from the reference's own sources)
timed on this box's host cores on a bounded sample of the same reads.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
K, S = 1001, 31


# The contract is ONE JSON line on stdout.  Libraries write to file descriptor 1 as they please (RCCL prints a five-line version banner there
# when its communicator goes away), so the real stdout is put aside for the result and descriptor 1 is pointed at stderr for everything else.
_RESULT_OUT = os.fdopen(os.dup(1), "w")
os.dup2(2, 1)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--reads-per-gpu", type=int, default=0, help="override the number of reads each GPU owns")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="at N > 1: strong (default) = the workload's reads divided over the GPUs (BASELINE.json configs[3] for config3); weak = the workload's reads on every GPU")
    ap.add_argument("--no-weak", action="store_true", help="at N > 1 with strong scaling: skip the `weak` sub-object (the workload's reads on every GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; the contract) or gloo (development: several ranks on ONE GPU)")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only: no scan_count / config2 / results_back / ingest / cli legs")
    ap.add_argument("--collectives", default="c", choices=["c", "torch"],
                    help="at N > 1: 'c' = the C entry points over RCCL (include/oatk_hip_multi.h: oatk_hip_merge_counts / oatk_hip_ec_sharded; the communicator's "
                         "128-byte id travels through torch.distributed once), 'torch' = the same exchange steps written over torch.distributed (oatk_amd/multi.py)")
    ap.add_argument("--no-sharded-syncerr", action="store_true",
                    help="at N > 1 stop after scan + count + table merge (the metric then says so); default is the whole step, sharded")
    ap.add_argument("--ec-graph", default="light", choices=["light", "full"],
                    help="light (default; what the drop-in uses): of make_syncmer_graph(0, 0.) only what read_error_correction can use at run_syncasm.c:124's "
                         "thresholds -- arcs between syncmers seen >= c times, one flag for the rest (include/oatk_hip_ec.h); full: every arc.  Same corrected reads.")
    ap.add_argument("--ingest-reads", type=int, default=200000)
    ap.add_argument("--ingest-window", type=int, default=64, help="MiB of text per window of the streamed ingest")
    ap.add_argument("--back-reads", type=int, default=100000)
    ap.add_argument("--cli-reads", type=int, default=2000000, help="reads of the CLI comparison (BASELINE.json quotes the north-star target at 2 M reads; the reference binary takes ~5 min there)")
    ap.add_argument("--cli-threads", type=int, default=32, help="-t of both binaries in the CLI comparison")
    ap.add_argument("--cli-settle-s", type=float, default=6.0, help="seconds between this process giving the device back and the CLI comparisons (see the note in the `cli` object)")
    ap.add_argument("--config1s-reads", type=int, default=200000, help="reads of the config-1 surrogate's resident step (0: leave the `config1s` object out)")
    ap.add_argument("--config1s-cpu-reads", type=int, default=40000, help="reads of the config-1 surrogate the compiled reference's read_error_correction is timed on (its `cpu_baseline`)")
    ap.add_argument("--config1s-cli-reads", type=int, default=100000, help="reads of its CLI comparison from the .fa.gz (0: none)")
    ap.add_argument("--cpu-sample-reads", type=int, default=80000)
    ap.add_argument("--cpu-threads", type=int, default=8)
    ap.add_argument("--dist-timeout-s", type=int, default=600, help="a collective that does not complete within this aborts the run instead of hanging it")
    return ap.parse_args()


def write_fasta(path, seq, off, lens, n):
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b">r%d\n" % i)
            f.write(seq[int(off[i]):int(off[i]) + int(lens[i])].tobytes())
            f.write(b"\n")


def cpu_baseline(readset, first, n_sample, threads, min_k_cov):
    """The compiled reference on host cores: sr_read + collect_syncmer_from_reads, then make_syncmer_graph + scg_consensus +
    read_error_correction (run_syncasm.c:81-131) on a FASTA file of the first n_sample reads of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_lib
    if not ref_lib.available():
        return None
    seq, off, lens = readset.slice(first, n_sample)
    bases = int(lens.sum())
    fd, path = tempfile.mkstemp(suffix=".fa", dir=os.environ.get("TMPDIR", "/tmp"))
    os.close(fd)
    try:
        write_fasta(path, seq, off, lens, n_sample)
        del seq
        t0 = time.perf_counter()
        db = ref_lib.SrDb([path], K, S, threads)
        sc = ref_lib.ScmDb(db)
        dt = time.perf_counter() - t0
        n_scm = sc.n()
        import ec_util
        L = ref_lib.lib()
        t1 = time.perf_counter()
        g = L.refx_make_graph(db.handle, sc.handle, 0, 0.0)
        L.refx_consensus(db.handle, g, 1, 1)
        summ = ec_util.reference_ec(db, sc, g, 0.02, min_k_cov, 0.35, threads=threads)
        dt_ec = time.perf_counter() - t1
        L.refx_scg_destroy(g)
        sc.close()
        db.close()
    finally:
        os.unlink(path)
    return {"value": round(bases / (dt + dt_ec) / 1e9, 4), "unit": "Gbases/s", "cores": threads, "threads": threads, "host": host_cores(), "kind": "reference",
            "sample": "first %d reads of the workload (%.2f Gbases) as FASTA through the compiled reference at -t %d: sr_read + collect_syncmer_from_reads "
                      "(parse included; %.1f s, %d syncmers), then make_syncmer_graph + scg_consensus + read_error_correction -c %d (+%.1f s, %s error blocks)"
                      % (n_sample, bases / 1e9, threads, dt, n_scm, min_k_cov, dt_ec, summ.get("total")),
            "scan_count": {"value": round(bases / dt / 1e9, 4), "unit": "Gbases/s"}}


def host_cores():
    """what the box has: physical cores (distinct (physical id, core id) pairs of /proc/cpuinfo), hardware threads, CPU model -- `cores` in cpu_baseline is
    the number of threads the reference was RUN with (its -t), as the contract defines it; more threads do not help it"""
    phys, model, logical = set(), None, os.cpu_count()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return {"physical_cores": len(phys) or None, "hardware_threads": logical, "cpu": model}


def pmc_traffic(kernel_name, workload, per_gpu):
    """HBM bytes per launch of `kernel_name` from the committed PMC pass of THIS workload (profiles/*_pmc_hbm_<workload>.csv: FETCH_SIZE /
    WRITE_SIZE collected in their own rocprofv3 passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when the
    workload differs from the one that was profiled."""
    import csv
    import glob
    from oatk_amd.synth import CONFIGS
    if workload not in CONFIGS or per_gpu != CONFIGS[workload]["n_reads"]:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_%s.csv" % workload)))
    if not files and workload == "config2":
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01*_pmc_hbm.csv")))
    if not files:
        return None, None
    for row in csv.reader(open(files[-1])):
        if row and kernel_name in row[0]:
            return int((2.0 * float(row[1]) + float(row[2])) * 1024), os.path.basename(files[-1])
    return None, None


def config1s_cpu_baseline(hip, sq, of, ln, n, K, S, c):
    """config1s.cpu_baseline: the compiled reference's read_error_correction (syncerr.c:759-866) on the first n reads of the surrogate at -t 8 and at every core of this
    host, beside the device's on the same reads.  The structs the reference works on are filled from the device scan and count (liboatk_host.so; the reference's own scan of
    n reads would take minutes and is not what is compared); its own make_syncmer_graph and consensus build the graph it corrects on."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import ref_lib as R
        from oatk_amd import _lib
        if not R.available():
            return {"skipped": "oracle/_ref is built only where the reference's sources are"}
        H, L = C.CDLL(_lib.HOST_LIB_PATH), R.lib()
        vp = C.c_void_p
        H.oatk_sr_db_new.restype = vp; H.oatk_sr_db_new.argtypes = [C.c_int, C.c_int]
        H.oatk_sr_read_packed.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp]
        H.oatk_collect_syncmer_from_reads.restype = vp; H.oatk_collect_syncmer_from_reads.argtypes = [vp, vp, C.POINTER(C.c_int)]
        L.refx_make_graph.restype = vp; L.refx_make_graph.argtypes = [vp, vp, C.c_int, C.c_double]
        L.refx_consensus.argtypes = [vp, vp, C.c_int, C.c_int]
        L.refx_ec.argtypes = [vp, vp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        off_n, len_n = np.ascontiguousarray(of[:n]), np.ascontiguousarray(ln[:n])
        nb = int(of[n]) if n < len(of) else int(sq.size)          # (up to where the next read begins: the stream's padding behind a read belongs to it)
        cores = os.cpu_count() or 8
        out = {"kind": "reference", "sample": "the first %d reads (%.2f Gbases): read_error_correction alone, graph built by the reference, scan and count by the device" % (n, int(len_n.sum()) / 1e9),
               "unit": "s", "threads": {}}
        saved = os.dup(2)
        for th in sorted({8, cores}):
            db = H.oatk_sr_db_new(K, S)
            if H.oatk_sr_read_packed(hip.h, db, sq.ctypes.data, off_n.ctypes.data, len_n.ctypes.data, n, nb, None):
                return {"error": "oatk_sr_read_packed failed: %s" % hip.L.oatk_hip_last_error(hip.h)}
            rcc = C.c_int(0)
            scm = H.oatk_collect_syncmer_from_reads(hip.h, db, C.byref(rcc))
            if rcc.value or not scm:
                return {"error": "oatk_collect_syncmer_from_reads failed"}
            g = L.refx_make_graph(db, scm, 0, 0.0)
            L.refx_consensus(db, g, 1, 1)
            devnull = os.open(os.devnull, os.O_WRONLY)
            os.dup2(devnull, 2)
            try:
                t0 = time.perf_counter()
                L.refx_ec(db, g, 0.02, c, c * 10, c, 0.35, th)
                dt = time.perf_counter() - t0
            finally:
                os.dup2(saved, 2); os.close(devnull)
            out["threads"][str(th)] = round(dt, 2)
            L.refx_scg_destroy(g); L.refx_scmdb_destroy(scm); L.refx_srdb_destroy(db)
        os.close(saved)
        # the device on the same reads: mark + solve + refresh, the graph (light) built on the device
        # (twice: the first call sizes the solver's slabs for this read set -- hipMalloc -- and is reported beside the second, which is what a resident step costs)
        times = []
        for _ in range(2):
            hip.scan_host(sq[:nb], off_n, len_n, K, S); hip.count(); hip.ec_graph(light_c=c)
            hip.sync(); t0 = time.perf_counter(); hip.ec(0.02, c, 0.35); hip.sync()
            times.append(round(time.perf_counter() - t0, 4))
        out["device_ec_s"], out["device_ec_first_call_s"] = times[1], times[0]
        out["value"] = out["threads"]["8"]; out["cores"] = 8
        out["device_over_reference_t8"] = round(out["threads"]["8"] / max(out["device_ec_s"], 1e-9), 1)
        out["device_over_reference_all_cores"] = round(out["threads"][str(cores)] / max(out["device_ec_s"], 1e-9), 1)
        return out
    except Exception as ex:      # noqa: BLE001
        return {"error": "%s: %s" % (type(ex).__name__, ex)}


def main():
    args = parse_args()
    import numpy as np
    import torch
    from oatk_amd import HipSyncasm
    from oatk_amd.synth import CONFIGS, ReadSet

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    multi = world > 1 or bool(os.environ.get("OATK_BENCH_FORCE_DIST"))      # (the env switch: exercise the N > 1 code path with a world of one)
    if multi:
        import datetime
        import torch.distributed as dist
        to = datetime.timedelta(seconds=args.dist_timeout_s)
        if args.dist_backend == "gloo":
            local_rank = 0                  # development only: the ranks share GPU 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", timeout=to)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=to)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local_rank)

    cfg = dict(CONFIGS[args.workload])
    c = int(cfg.get("min_k_cov", 30))
    strong = (world > 1 or bool(os.environ.get("OATK_BENCH_FORCE_STRONG"))) and args.scaling == "strong" and not args.reads_per_gpu      # (the env switch: exercise the strong-scaling legs with a world of one)
    per_gpu = args.reads_per_gpu or (cfg["n_reads"] // world if strong else cfg["n_reads"])
    n_workload = cfg["n_reads"]
    cfg["n_reads"] = per_gpu * world
    t_gen = time.perf_counter()
    rs = ReadSet(**cfg)
    first = rank * per_gpu

    # ---- synthetic reads -> HBM (not timed) ----
    seq, off, lens = rs.slice(first, per_gpu)
    bases = int(lens.sum())
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.view(np.int32)).to(dev)
    seq_bytes = int(seq.size)
    del seq
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    hip = HipSyncasm(local_rank)
    hip.set_timing(True)
    merger = sharded = comm = None
    with_ec = True
    collectives = None
    if multi:
        with_ec = not args.no_sharded_syncerr
        hip._check(hip.L.oatk_hip_ec_reserve_import(hip.h, 64 << 20), "oatk_hip_ec_reserve_import")     # k-mers a shard may have to be sent (before the scan)
        if args.collectives == "c" and args.dist_backend == "nccl":
            import ctypes as C
            uid = (C.c_uint8 * 128)()
            box = [None]
            if rank == 0:
                hip._check(hip.L.oatk_comm_unique_id(uid), "oatk_comm_unique_id")
                box[0] = bytes(uid)
            dist.broadcast_object_list(box, src=0)
            uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
            comm = hip.L.oatk_comm_create(uid, rank, world, local_rank)
        if comm:
            collectives = "RCCL from C (oatk_hip_merge_counts / oatk_hip_ec_sharded)"
        else:
            from oatk_amd.multi import CountMerger, ShardedEc
            collectives = "torch.distributed (%s), oatk_amd/multi.py" % args.dist_backend
            if with_ec:
                sharded = ShardedEc(hip, dist, dev)
            else:
                merger = CountMerger(hip, dist, dev)

    def scan_count(seq_t=d_seq, off_t=d_off, len_t=d_len, n=per_gpu, nbytes=seq_bytes, sid0=first):
        hip.scan_device(seq_t.data_ptr(), off_t.data_ptr(), len_t.data_ptr(), n, nbytes, K, S, sid0=sid0)
        hip.count()

    n_imp = [0]

    def step():
        """run_syncasm.c:81-131 on the resident batch"""
        scan_count()
        if comm:                            # count-table merge, graph from everybody's pairs, correction in global ids: two C calls over RCCL
            if not with_ec:
                hip.merge_counts(comm)
                return None
            st_, n_imp[0] = hip.ec_sharded(comm, 0.02, c, 0.35)
            return st_
        if sharded is not None:             # the same over torch.distributed (oatk_amd/multi.py)
            return sharded.run(0.02, c, 0.35)["stats"]
        if merger is not None:
            merger.merge()
            return None
        hip.ec_graph(light_c=c if args.ec_graph == "light" else 0)
        return hip.ec(0.02, c, 0.35)

    def fence():
        hip.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        fence()
        return (time.perf_counter() - t0) / max(steps, 1), r

    # ---- N > 1 over the C entry points: one untimed probe step first.  If any rank's call returns an error (not a crash: those cannot be caught) every
    #      rank switches to the torch.distributed form of the same exchange steps, and the result line says so ----
    if comm and dist is not None:
        ok, why = 1, ""
        try:
            if os.environ.get("OATK_BENCH_FAIL_C_PROBE"):       # (test switch for the fallback below)
                raise RuntimeError("OATK_BENCH_FAIL_C_PROBE")
            step()
        except Exception as ex:             # noqa: BLE001
            ok, why = 0, "%s: %s" % (type(ex).__name__, ex)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            from oatk_amd.multi import CountMerger, ShardedEc
            hip.L.oatk_comm_destroy(comm)
            comm = None
            collectives = "torch.distributed (%s), oatk_amd/multi.py -- the C entry points failed on some rank%s" % (args.dist_backend, (": " + why[:200]) if why else "")
            if with_ec:
                sharded = ShardedEc(hip, dist, dev)
            else:
                merger = CountMerger(hip, dist, dev)

    # ---- the headline: EXACTLY --steps steps between two fences, max over ranks ----
    for _ in range(args.warmup):
        step()
    fence()
    phase_ms = {}
    st = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
        for name, v in hip.timing().items():
            phase_ms[name] = phase_ms.get(name, 0.0) + v
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tb = torch.tensor([bases], dtype=torch.int64, device=dev)
        dist.all_reduce(tb)
        total_bases = int(tb.item())
    else:
        total_bases = bases
    for name in phase_ms:
        phase_ms[name] /= max(args.steps, 1)
    info = hip.info()
    hoco = int(hip.fetch("HOCO_L").astype(np.uint64).sum())
    n_occ, n_scm = info["n_occ"], info["n_scm"]
    ec_summary = None
    if st is not None:
        ec_summary = {"error_blocks": int(st[0] + st[5] + st[10]), "corrected": int(st[2] + st[7]), "uncorrected": int(st[1] + st[6]),
                      "ambiguous": int(st[3] + st[4] + st[8] + st[9]), "blocks_past_first_tier": int(st[11]) if len(st) > 11 else None}
        if sharded is not None:
            ec_summary["imported_kmers_rank0"] = sharded.n_imported
        if comm:
            ec_summary["imported_kmers_rank0"] = n_imp[0]

    extras = {}
    pending_cli_gz, hip_closed = None, False
    hip.set_timing(False)
    if comm and with_ec and not args.no_extras:
        # ---- what follows the EC round in syncasm() with sharded reads: every rank takes part (include/oatk_hip_multi.h, second half) ----
        try:
            after = {}

            def tmax(d):
                t = torch.tensor([d], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            d, _ = timed(lambda: hip.gather_table(comm, 0), args.steps)
            after["gather_table"] = {"ms": round(tmax(d) * 1e3, 3), "workload": "oatk_hip_gather_table: update_syncmer_db's table with every syncmer's occurrences in (sid, idx) order, on rank 0"}
            d, (nv, na) = timed(lambda: hip.asm_graph_sharded(comm, c, 0.35), args.steps)
            after["asm_graph"] = {"ms": round(tmax(d) * 1e3, 3), "n_vtx": nv, "n_arc": na, "workload": "oatk_hip_asm_graph_sharded(-c %d, a 0.35): the same graph on every rank" % c}
            d, _ = timed(lambda: hip.consensus_sharded(comm, c), args.steps)
            after["consensus"] = {"ms": round(tmax(d) * 1e3, 3), "workload": "oatk_hip_consensus_sharded(-c %d): totals all-reduced" % c}
            d, (n_p, n_e) = timed(lambda: hip.overlap_hist_sharded(comm, c), args.steps)
            after["overlap_hist"] = {"ms": round(tmax(d) * 1e3, 3), "pairs": n_p, "workload": "oatk_hip_overlap_hist_sharded(-c %d): tables of the pairs between graph vertices, from weighted segments" % c}
            d, sraw = timed(lambda: hip.stat_sharded(comm), args.steps)
            after["sr_db_stat"] = {"ms": round(tmax(d) * 1e3, 3), "kmer_unique": int(sraw["kmer_unique"]), "workload": "oatk_hip_stat_sharded on the corrected chains"}
            ag = hip.fetch_asm_graph()
            n_scm_all = len(ag["scm_del"])
            su_off = np.zeros(n_scm_all + 1, np.uint64)
            su_off[1:] = np.cumsum(ag["scm_del"] == 0)
            graph = {"n_scm": n_scm_all, "su_off": su_off, "su_uid": np.arange(nv, dtype=np.uint64) << np.uint64(1), "su_pos": np.zeros(nv, np.uint32),
                     "utg_n": np.ones(nv, np.uint32), "idx_p": ag["idx_p"], "idx_n": ag["idx_n"].astype(np.uint64), "arc_w": ag["arc_w"],
                     "arc_ln": np.zeros(na, np.uint64), "arc_del": np.zeros(na, np.uint8)}
            d, (n_aln, n_frg, ast) = timed(lambda: hip.read_alignment(graph), args.steps)
            tot = torch.tensor([n_aln, int(ast[0]), int(ast[2])], dtype=torch.int64, device=dev)
            dist.all_reduce(tot)
            after["read_alignment"] = {"ms": round(tmax(d) * 1e3, 3), "alignments": int(tot[0]), "reads_aligned": int(tot[1]), "reads_over_limits": int(tot[2]),
                                       "workload": "scg_read_alignment: every rank its own %d corrected reads against the %d-vertex graph, no exchange" % (per_gpu, nv)}
            extras["after_syncerr"] = after
        except Exception as ex:             # noqa: BLE001
            extras["after_syncerr"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    if rank == 0 and not args.no_extras:
        # ---- what follows the EC round in syncasm(), on the corrected batch of rank 0 (not part of `value`) ----
        if world == 1 and not multi:
            try:
                after = {}
                d, (nv, na) = timed(lambda: hip.asm_graph(c, 0.35), args.steps)
                after["asm_graph"] = {"ms": round(d * 1e3, 3), "n_vtx": nv, "n_arc": na, "workload": "make_syncmer_graph(-c %d, a 0.35) + asmg_finalize on the corrected chains" % c}
                d, _ = timed(lambda: hip.consensus(c), args.steps)
                after["consensus"] = {"ms": round(d * 1e3, 3), "workload": "oatk_hip_consensus(-c %d): scg_syncmer_consensus' sums for every live syncmer" % c}
                d, _ = timed(hip.overlap_hist, args.steps)
                after["overlap_hist"] = {"ms": round(d * 1e3, 3), "workload": "oatk_hip_overlap_hist: calc_syncmer_overlap's tables for every adjacent pair"}
                ag = hip.fetch_asm_graph()
                n_scm_all = len(ag["scm_del"])
                su_off = np.zeros(n_scm_all + 1, np.uint64)
                su_off[1:] = np.cumsum(ag["scm_del"] == 0)
                graph = {"n_scm": n_scm_all, "su_off": su_off, "su_uid": np.arange(nv, dtype=np.uint64) << np.uint64(1), "su_pos": np.zeros(nv, np.uint32),
                         "utg_n": np.ones(nv, np.uint32), "idx_p": ag["idx_p"], "idx_n": ag["idx_n"].astype(np.uint64), "arc_w": ag["arc_w"],
                         "arc_ln": np.zeros(na, np.uint64), "arc_del": np.zeros(na, np.uint8)}
                d, (n_aln, n_frg, ast) = timed(lambda: hip.read_alignment(graph), args.steps)
                after["read_alignment"] = {"ms": round(d * 1e3, 3), "alignments": n_aln, "fragments": n_frg, "reads_aligned": int(ast[0]), "reads_over_limits": int(ast[2]),
                                           "workload": "scg_read_alignment of the %d corrected reads against the %d-vertex graph (graph upload included)" % (per_gpu, nv)}
                extras["after_syncerr"] = after
            except Exception as ex:         # noqa: BLE001
                extras["after_syncerr"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if not args.no_extras:
        # ---- the same batch through scan + count only (BASELINE.json configs[1]'s shape at this size) ----
        try:
            dsc, _ = timed(scan_count, args.steps)
            if dist is not None:
                t = torch.tensor([dsc], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dsc = float(t.item())
            extras["scan_count"] = {"value": round(total_bases / dsc / 1e9, 3), "unit": "Gbases/s", "ms_per_step": round(dsc * 1e3, 3),
                                    "workload": "the same resident batch through scan + count only%s" % (" (no table merge)" if world > 1 else "")}
        except Exception as ex:             # noqa: BLE001   (an extension must never take the headline down)
            extras["scan_count"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if strong and comm and with_ec and not args.no_extras and not args.no_weak:
        # ---- weak scaling beside it: the workload's reads on EVERY GPU (N x the reads of the headline) ----
        try:
            del d_seq, d_off, d_len
            wcfg = dict(CONFIGS[args.workload])
            w_per = wcfg["n_reads"]
            wcfg["n_reads"] = w_per * world
            rsw = ReadSet(**wcfg)
            sq, of, ln = rsw.slice(rank * w_per, w_per)
            wb = int(ln.sum())
            t_sq, t_of, t_ln = torch.from_numpy(sq).to(dev), torch.from_numpy(of.view(np.int64)).to(dev), torch.from_numpy(ln.view(np.int32)).to(dev)
            nbw = int(sq.size)
            del sq

            def step_weak():
                scan_count(t_sq, t_of, t_ln, w_per, nbw, rank * w_per)
                return hip.ec_sharded(comm, 0.02, c, 0.35)[0]
            step_weak()
            dw, st_w = timed(step_weak, args.steps)
            t = torch.tensor([dw], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tbw = torch.tensor([wb], dtype=torch.int64, device=dev)
            dist.all_reduce(tbw)
            extras["weak"] = {"value": round(int(tbw.item()) / float(t.item()) / 1e9, 3), "unit": "Gbases/s", "ms_per_step": round(float(t.item()) * 1e3, 3), "scaling": "weak",
                              "reads_per_gpu": w_per, "error_blocks": int(st_w[0] + st_w[5] + st_w[10]),
                              "workload": "%s's %d reads on EVERY GPU (%d reads in all): the same sharded step" % (args.workload, w_per, w_per * world)}
            del t_sq, t_of, t_ln
        except Exception as ex:             # noqa: BLE001
            extras["weak"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if world == 1 and not multi and not args.no_extras:
        # ---- the same step with the other EC graph (light <-> full): what the restriction to usable arcs is worth ----
        try:
            other = "full" if args.ec_graph == "light" else "light"

            def step_other():
                scan_count()
                hip.ec_graph(light_c=c if other == "light" else 0)
                return hip.ec(0.02, c, 0.35)
            step_other()
            dso, st_o = timed(step_other, args.steps)
            extras["ec_graph_" + other] = {"value": round(total_bases / dso / 1e9, 3), "unit": "Gbases/s", "ms_per_step": round(dso * 1e3, 3),
                                           "same_statistics": bool(st is not None and st_o is not None and list(st_o[:11]) == list(st[:11])),
                                           "workload": "the headline step with --ec-graph %s" % other}
        except Exception as ex:             # noqa: BLE001
            extras["ec_graph_other"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        # ---- the same step through the code path of N > 1 with a world of ONE: librccl loaded, a communicator from a unique id, the table merge by
        #      hash range (personalised exchange with itself), candidates, segments, refreshed counts to their owner -- what the sharded path costs
        #      before any byte crosses xGMI ----
        try:
            import ctypes as C
            uid = (C.c_uint8 * 128)()
            hip._check(hip.L.oatk_comm_unique_id(uid), "oatk_comm_unique_id")
            comm1 = hip.L.oatk_comm_create(uid, 0, 1, local_rank)
            if not comm1:
                raise RuntimeError("oatk_comm_create failed")

            def step_sharded():
                scan_count()
                hip.merge_counts(comm1)
                return hip.ec_sharded(comm1, 0.02, c, 0.35)[0]
            step_sharded()
            dss, st_s = timed(step_sharded, args.steps)
            traffic = (C.c_uint64 * 8)()
            hip.L.oatk_comm_traffic(comm1, traffic, 1)
            step_sharded()
            hip.L.oatk_comm_traffic(comm1, traffic, 1)
            tr_full = [int(x) for x in traffic]
            extras["sharded_path_world_of_one"] = {"value": round(total_bases / dss / 1e9, 3), "unit": "Gbases/s", "ms_per_step": round(dss * 1e3, 3),
                                                   "same_statistics": bool(st is not None and list(st_s[:11]) == list(st[:11])),
                                                   "workload": "the headline step through oatk_hip_merge_counts + oatk_hip_ec_sharded over an RCCL communicator of one rank"}
            # ---- a prediction to hold the first measured N = 2 / 4 / 8 runs against (strong scaling: the workload's reads divided over the GPUs).  Measured
            #      here: the sharded step of ONE rank at the workload's reads and at an eighth of them (time and what it puts into the collectives, both
            #      taken as a + b x reads); assumed: the link figures below.  Not a measurement of any link. ----
            try:
                n8 = max(per_gpu // 8, 1)
                nb8 = int(off[n8]) if n8 < per_gpu else seq_bytes

                def step_eighth():
                    scan_count(d_seq, d_off, d_len, n8, nb8, first)
                    hip.merge_counts(comm1)
                    return hip.ec_sharded(comm1, 0.02, c, 0.35)[0]
                step_eighth()
                d8, _ = timed(step_eighth, args.steps)
                hip.L.oatk_comm_traffic(comm1, traffic, 1)
                step_eighth()
                hip.L.oatk_comm_traffic(comm1, traffic, 1)
                tr_8 = [int(x) for x in traffic]
                extras["scale_model"] = scale_model(per_gpu, dss * 1e3, tr_full, n8, d8 * 1e3, tr_8, total_bases, dt / args.steps * 1e3)
            except Exception as ex:         # noqa: BLE001
                extras["scale_model"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            hip.L.oatk_comm_destroy(comm1)
        except Exception as ex:             # noqa: BLE001
            extras["sharded_path_world_of_one"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0 and not args.no_extras:
        # ---- BASELINE.json configs[1]: 200 k reads x 15 kb (its own 1 Mb genome), scan + count, and with the EC round ----
        if args.workload != "config2":
            try:
                c2 = dict(CONFIGS["config2"])
                rs2 = ReadSet(**c2)
                sq, of, ln = rs2.slice(0, c2["n_reads"])
                b2 = int(ln.sum())
                t_sq, t_of, t_ln = torch.from_numpy(sq).to(dev), torch.from_numpy(of.view(np.int64)).to(dev), torch.from_numpy(ln.view(np.int32)).to(dev)
                nb2 = int(sq.size)
                del sq
                sc2 = lambda: scan_count(t_sq, t_of, t_ln, c2["n_reads"], nb2, 0)   # noqa: E731

                def full2():
                    sc2()
                    hip.ec_graph(light_c=int(c2["min_k_cov"]) if args.ec_graph == "light" else 0)
                    return hip.ec(0.02, int(c2["min_k_cov"]), 0.35)
                full2()
                d1, _ = timed(sc2, 5)
                d2, s2 = timed(full2, 5)
                extras["config2"] = {"workload": "config2: 200000 reads x ~15 kb (%.2f Gbases, genome 1 Mb), resident in HBM" % (b2 / 1e9),
                                     "scan_count": {"value": round(b2 / d1 / 1e9, 3), "unit": "Gbases/s", "ms_per_step": round(d1 * 1e3, 3)},
                                     "syncmer_syncerr": {"value": round(b2 / d2 / 1e9, 3), "unit": "Gbases/s", "ms_per_step": round(d2 * 1e3, 3),
                                                         "error_blocks": int(s2[0] + s2[5] + s2[10])}}
                del t_sq, t_of, t_ln
            except Exception as ex:         # noqa: BLE001
                extras["config2"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

        # ---- BASELINE.json configs[0] by surrogate: a read set SHAPED like an organelle HiFi data set (oatk_amd/synth.py: CONFIG1S -- two organelles at
        #      thousand-fold coverage in a 256 Mb background at 7x, low-complexity arrays, N, short reads), resident step and the CLI from the .fa.gz ----
        if args.config1s_reads and world == 1:
            try:
                from oatk_amd import synth
                c1 = dict(synth.CONFIG1S)
                n1 = min(args.config1s_reads, c1["n_reads"])
                rs1 = synth.MixReadSet(**c1)
                sq, of, ln = rs1.slice(0, n1)
                b1 = int(ln.sum())
                t_sq, t_of, t_ln = torch.from_numpy(sq).to(dev), torch.from_numpy(of.view(np.int64)).to(dev), torch.from_numpy(ln.view(np.int32)).to(dev)
                nb1 = int(sq.size)
                cc1 = int(c1["min_k_cov"])

                def full1():
                    scan_count(t_sq, t_of, t_ln, n1, nb1, 0)
                    hip.ec_graph(light_c=cc1 if args.ec_graph == "light" else 0)
                    return hip.ec(0.02, cc1, 0.35)
                full1()
                d1s, s1 = timed(full1, 2)
                hip.set_timing(True)                    # (one more step with the phase timers on: they synchronise, so the step above ran without them)
                full1()
                ph1 = {k_: round(v, 3) for k_, v in hip.timing().items() if v > 0.0005}
                hip.set_timing(False)
                inf1 = hip.info()
                extras["config1s"] = {"workload": "config-1 surrogate: %d reads (%.2f Gbases) -- a 154 kb + a 368 kb genome at >= 1000x in a 256 Mb background at ~7x, tandem arrays, "
                                                  "homopolymers > 256, N, 1 %% reads < K; -c %d; resident in HBM" % (n1, b1 / 1e9, cc1),
                                      "syncmer_syncerr": {"value": round(b1 / d1s / 1e9, 3), "unit": "Gbases/s", "ms_per_step": round(d1s * 1e3, 3)},
                                      "phases_ms": ph1, "error_blocks": int(s1[0] + s1[5] + s1[10]), "syncmers": {"occurrences": int(inf1["n_occ"]), "distinct": int(inf1["n_scm"])}}
                del t_sq, t_of, t_ln
                if not args.no_cpu_baseline:
                    extras["config1s"]["cpu_baseline"] = config1s_cpu_baseline(hip, sq, of, ln, min(args.config1s_cpu_reads, n1), K, S, cc1)
                if args.config1s_cli_reads and not args.no_cpu_baseline:
                    pending_cli_gz = (sq, of, ln, min(args.config1s_cli_reads, n1), K, cc1)       # (run at the end, when this process has given the device back)
                del sq
            except Exception as ex:         # noqa: BLE001
                extras.setdefault("config1s", {})["error"] = "%s: %s" % (type(ex).__name__, ex)

        # ---- SURVEY 8d timing (ii): pinned host ASCII -> device -> scan -> the reference's sr_t arrays filled on the host (sr_read's contract) ----
        try:
            from oatk_amd import dropin
            n_b = min(args.back_reads, per_gpu)
            extras["results_back"] = dropin.time_sr_read_packed(hip, rs, first, n_b, K, S)
        except Exception as ex:             # noqa: BLE001
            extras["results_back"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

        # ---- from the TEXT of a FASTA file to syncmers, PCIe included: the text lies in pinned host memory and is streamed through the device in
        #      windows -- the copy of window i + 1 rides beside the record scan + syncmer scan of window i (oatk_scan_text, liboatk_host.so) --,
        #      then the count.  A bounded sample; the reference's reader does 0.4 Gbases/s here. ----
        try:
            import ctypes as C
            from oatk_amd import synth
            H = synth.host_lib()
            H.oatk_scan_text.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64)]
            n_ing = min(args.ingest_reads, per_gpu)
            sq, of, ln = rs.slice(first, n_ing)
            parts = []
            for i in range(n_ing):
                parts.append(b">r%d\n" % i)
                parts.append(sq[int(of[i]):int(of[i]) + int(ln[i])].tobytes())
                parts.append(b"\n")
            text = torch.from_numpy(np.frombuffer(b"".join(parts), dtype=np.uint8).copy()).pin_memory()
            del parts, sq
            ing_bases = int(ln.sum())
            best = None
            for _ in range(3):
                hip.sync()
                nr = C.c_uint64()
                t0 = time.perf_counter()
                rc = H.oatk_scan_text(hip.h, text.data_ptr(), int(text.numel()), 1, K, S, args.ingest_window << 20, C.byref(nr))
                hip.sync()
                t_scan = time.perf_counter() - t0
                if rc != 0 or nr.value != n_ing:
                    raise RuntimeError("oatk_scan_text: code %d, %d reads" % (rc, nr.value))
                t0 = time.perf_counter()
                hip.count()
                hip.sync()
                t_cnt = time.perf_counter() - t0
                if best is None or t_scan + t_cnt < best[0]:
                    best = (t_scan + t_cnt, t_scan, t_cnt)
            extras["ingest"] = {"value": round(ing_bases / best[0] / 1e9, 3), "unit": "Gbases/s",
                                "workload": "text of an unwrapped FASTA file with %d reads (%.2f GB) in pinned host memory -> %d MiB windows over PCIe beside the record "
                                            "scan + syncmer scan of the window before -> count" % (n_ing, text.numel() / 1e9, args.ingest_window),
                                "ms_text_to_scanned_batch": round(best[1] * 1e3, 3), "text_GBs": round(text.numel() / best[1] / 1e9, 2),
                                "ms_count": round(best[2] * 1e3, 3), "syncmers": hip.info()["n_occ"]}
            del text
        except Exception as ex:             # noqa: BLE001
            extras["ingest"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

        # ---- SURVEY 8d timing (iii): the syncasm CLI from a FASTA file, reference binary against the drop-in binary (the reference's own
        #      translation units with the hot-path functions replaced by liboatk_host.so's; built where the reference
        #      sources are).  Same file, same options; both GFA files compared byte for byte. ----
        # The CLI is a process of its own, and it is timed with the device to itself: this process first gives back what it holds there and waits --cli-settle-s.  While a
        # parent holds a hundred GB of VRAM, or for some seconds after it has returned them (the driver clears what comes back), the child's allocations take tenths of a
        # second to seconds each instead of milliseconds -- the same binary on the same file ran 2.1 s, 3.1 s and 5.6 s (tools/vram_churn_cli.py, profiles/r06v_vram_churn.txt);
        # until round 6 these comparisons ran in the middle of the bench with ~100 GB held, and that is what was in their numbers.
        if world == 1 and not args.no_cpu_baseline and (pending_cli_gz or args.cli_reads):
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import cli_util
            import gc
            hip.sync()
            if comm:
                hip.L.oatk_comm_destroy(comm)
                comm = None
            hip.close()
            hip_closed = True
            try:                                # (closures above hold the tensors: their storage is what goes)
                for t_ in (d_seq, d_off, d_len):
                    t_.untyped_storage().resize_(0)
            except (NameError, UnboundLocalError, AttributeError):
                pass
            gc.collect()
            torch.cuda.empty_cache()
            time.sleep(args.cli_settle_s)
            if pending_cli_gz:
                try:
                    extras["config1s"]["cli_fa_gz"] = cli_util.time_cli_gz(*pending_cli_gz, args.cli_threads, settle_s=args.cli_settle_s)
                    extras["config1s"]["cli_fa_gz"]["device_to_itself"] = "this process released its share of the device %.0f s before the first run; %.0f s between the runs (the driver clears what a process gives back while the next one waits: DESIGN.md 11.1)" % (args.cli_settle_s, args.cli_settle_s)
                except Exception as ex:     # noqa: BLE001
                    extras["config1s"]["cli_fa_gz"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
                pending_cli_gz = None
            try:
                extras["cli"] = cli_util.time_cli(rs, first, min(args.cli_reads, per_gpu), K, S, c, args.cli_threads, devices="%d,%d" % (local_rank, local_rank), settle_s=args.cli_settle_s)
                extras["cli"]["device_to_itself"] = "this process released its share of the device %.0f s before the first run; %.0f s between the two drop-in runs" % (args.cli_settle_s, args.cli_settle_s)
            except Exception as ex:         # noqa: BLE001
                extras["cli"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0:
        # ---- roofline of the dominant kernel (by measured time) ----
        alg_bytes = {
            # kernel A: ASCII in, 2-bit hoco_s + ho_rl out (1 + 0.25 rho + rho per raw base)
            "hpc": bases + hoco // 4 + hoco,
            # kernel B: 2-bit hoco_s in, one 20-byte record per syncmer occurrence out (the 8-byte hash comes from kmer_hash_kernel)
            "syncmer": hoco // 4 + 20 * n_occ,
        }
        dom = max(("hpc", "syncmer"), key=lambda k_: phase_ms.get(k_, 0.0))
        kname = {"hpc": "oatk::hpc_pack_kernel", "syncmer": "oatk::syncmer_fast_kernel"}[dom]
        dur_s = phase_ms[dom] / 1e3
        achieved = alg_bytes[dom] / dur_s / 1e9 if dur_s > 0 else 0.0
        traffic, traffic_src = pmc_traffic(kname, args.workload, per_gpu)
        valu_per_64, valu_src = pmc_valu_per_64()
        clk_ghz, clk_src = pmc_clock_ghz()
        # cycles per wave64 VALU instruction weighted by the opcode mix of kernel B's tile loop (tools/isa_mix.py over `hipcc -S`, per-opcode rates measured
        # on the box: profiles/r02c_valu_rates.txt) -- 2.9 for the simple 32-bit forms, 4.5 - 5.5 for 64-bit shifts, v_mad_u64_u32, v_alignbit
        import glob
        mix_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_isa_mix_syncmer_fast.json")))
        mix = json.load(open(mix_files[-1])) if mix_files else None
        cpi = float(mix["cycles_per_valu_instruction_mix"]) if mix else 4.0
        valu_rate = hoco / 64 * valu_per_64 / (phase_ms["syncmer"] / 1e3) / 1e9           # G wave-instructions / s
        # what binds: kernel A is an HBM stream; kernel B issues ~70 integer VALU instructions per position and moves a quarter of a byte -- its HBM
        # fraction is reported because the contract asks for it, the bound that binds is the VALU issue rate (`valu`)
        # (`bound` says what binds the kernel; achieved / peak / frac are the HBM figures the contract asks for either way, `valu` the ones of the bound that binds kernel B)
        roofline = {"bound": "valu issue" if dom == "syncmer" else "hbm", "priced_against": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": alg_bytes[dom], "avg_launch_ms": round(phase_ms[dom], 4),
                    "share_of_step": round(phase_ms[dom] / (dt / args.steps * 1e3), 3),
                    "note": "kernel B reads 0.25 B and hashes one 31-mer per hoco position: it is integer-VALU issue bound, not HBM bound (DESIGN.md 5); "
                            "`valu` prices it against the bound that binds",
                    # wave-instructions issued (PMC count per 64 positions, profiles/) against the issue ceiling of THIS opcode mix: 1024 SIMDs x the measured clock /
                    # (cycles per instruction weighted by the mix, per-opcode rates measured on the box).  (Up to r03g a second figure priced every
                    # instruction at a flat four cycles; the kernel has since run at 1.12 of that "peak", which settles what it was worth.)
                    "valu": {"achieved": round(valu_rate, 1), "peak": round(1024 * clk_ghz / cpi, 1), "unit": "G wave-instr/s",
                             "frac": round(valu_rate * cpi / (1024 * clk_ghz), 3), "clock_GHz": clk_ghz, "clock_source": clk_src, "valu_per_64_positions": valu_per_64, "valu_count_source": valu_src, "cycles_per_instr_mix": cpi,
                             "mix_source": os.path.basename(mix_files[-1]) if mix_files else None},
                    "scan_bytes_per_base": round((alg_bytes["hpc"] + 28 * n_occ) / bases, 4),
                    # the scan of SURVEY.md 8(d) is kernel A + kernel B + the k-mer hash
                    "scan_achieved_GBs": round((alg_bytes["hpc"] + 28 * n_occ) / ((phase_ms["hpc"] + phase_ms["syncmer"] + phase_ms.get("syncmer_n", 0.0) + phase_ms.get("kmer_hash", 0.0)) / 1e3) / 1e9, 2)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(rs, first, min(args.cpu_sample_reads, per_gpu), args.cpu_threads, c)
            # ... and with a thread per physical core ("-t <all physical cores> and -t 8").  The reference hands its reader's batches of 10 000 reads to the
            # threads and analyses when 10 000 x T are in (syncmer.c:504,523,532): a sample occupies sample / 10 000 threads whatever -t says, so the leg means what it says only
            # on >= 10 000 reads per core -- run when the sample allows it (--cpu-sample-reads), said so otherwise (r04's 80 k reads at -t 128 measured -t 8 twice)
            phys = (cpu or {}).get("host", {}).get("physical_cores") or os.cpu_count() or 8
            n_cpu_sample = min(args.cpu_sample_reads, per_gpu)
            if cpu and phys > args.cpu_threads:
                if n_cpu_sample >= 10000 * phys:
                    allc = cpu_baseline(rs, first, n_cpu_sample, phys, c)
                    if allc:
                        cpu["all_physical_cores"] = {"value": allc["value"], "unit": allc["unit"], "cores": phys, "threads": phys, "scan_count": allc["scan_count"], "sample": allc["sample"]}
                else:
                    cpu["all_physical_cores"] = {"skipped": "%d reads keep %d of %d threads busy (10 000 reads per thread and batch, syncmer.c:504,523,532): not a measurement of all cores; "
                                                            "--cpu-sample-reads %d would be" % (n_cpu_sample, max(1, n_cpu_sample // 10000), phys, 10000 * phys),
                                                 "threads_a_sample_of_this_size_occupies": max(1, n_cpu_sample // 10000)}
        what = "syncmer+syncerr" if with_ec else "syncmer scan + count + table merge ONLY (--no-sharded-syncerr)"
        out = {
            "metric": "HiFi Gbases/s through syncasm (%s) at k=1001 s=31" % what,
            "value": round(total_bases * args.steps / dt / 1e9, 3), "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": (("%s sharded over %d GPUs (BASELINE.json configs[3] for config3: %d reads in all), " % (args.workload, world, n_workload)) if strong else "") +
                                   "%s: %d reads x ~%d kb per GPU, k=1001 s=31 -c %d, full hot path of syncasm incl. syncerr (scan + count + EC graph + "
                                   "read_error_correction with its Levenshtein path search), reads resident in HBM" % (args.workload, per_gpu, cfg["mean_len"] // 1000, c)
                       if with_ec else "%s: %d reads x ~%d kb per GPU, scan + count + table merge" % (args.workload, per_gpu, cfg["mean_len"] // 1000),
                       "ec_graph": ("light: arcs between syncmers seen >= %d times + one flag per oriented vertex for the rest (include/oatk_hip_ec.h); same corrected reads "
                                    "as from the full graph (tests/test_gpu_light_graph.py)" % c) if args.ec_graph == "light" else "full: every arc of make_syncmer_graph(0, 0.)",
                       "reads_per_gpu": per_gpu, "bases_per_gpu": bases, "genome_len": cfg["genome_len"],
                       "parallelism": "reads sharded by record, %d rank(s)%s" % (world, ("; table merge by hash range (personalised exchange), candidates + pair segments all-gathered, refreshed counts to their owners: %s" if comm else "; table and pair lists all-gathered, coverage all-reduced (replicated): %s") % collectives if multi else ""),
                       "setup_s_untimed": round(t_gen, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "syncerr": ec_summary,
            "phases_ms": {k_: round(v, 4) for k_, v in phase_ms.items()},
            "syncmers": {"occurrences": n_occ, "distinct": n_scm, "hoco_ratio": round(hoco / bases, 4)},
        }
        out.update(extras)
        print(json.dumps(out), file=_RESULT_OUT, flush=True)
    if comm:
        hip.L.oatk_comm_destroy(comm)
    if not hip_closed:
        hip.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def scale_model(n1, ms1, tr1, n8, ms8, tr8, bases1, ms_headline):
    """Predicted strong-scaling steps at N = 2, 4, 8 from two one-rank measurements of the sharded step (n reads -> ms, traffic as oatk_comm_traffic
    counts it).  Per rank at N GPUs: compute(reads / N) from the line through the two measured points; every collective pays a latency; an exchange's
    bytes leave over N - 1 links at once, 1 / N of them staying home; an all-gather's bytes arrive from N - 1 peers over N - 1 links; an all-reduce
    moves 2 (N - 1) / N of its bytes round a ring.  xGMI: seven links per GPU, 153.6 GB/s each both directions together (the task's figure) = 76.8 GB/s
    one way, taken at 60 %; 40 us per collective (launch + the host's wait for it: comm_wait polls the stream)."""
    link_GBs, eff, lat_ms = 76.8, 0.6, 0.040
    bw = link_GBs * eff * 1e6                   # bytes per ms per link
    slope = (ms1 - ms8) / max(n1 - n8, 1)
    fixed = ms8 - slope * n8

    def lin(a1, a8, n):                         # a + b n through the two points
        b = (a1 - a8) / max(n1 - n8, 1)
        return max(a8 - b * n8 + b * n, 0.0)
    calls = tr1[0] + tr1[2] + tr1[4] + tr1[6]
    out = {"measured": {"reads": [n1, n8], "ms": [round(ms1, 3), round(ms8, 3)], "ms_headline_step_n1": round(ms_headline, 3),
                        "collectives_per_step": calls,
                        "bytes_put_in_per_step": {"small_allgather": [tr1[1], tr8[1]], "allgather": [tr1[3], tr8[3]], "allreduce": [tr1[5], tr8[5]], "exchange": [tr1[7], tr8[7]]}},
           "assumed": {"link_GBs_one_way": link_GBs, "efficiency": eff, "ms_per_collective": lat_ms, "compute_ms": "%.3f + %.6f x reads" % (fixed, slope)},
           "predicted": {}}
    for n in (2, 4, 8):
        r = n1 / n
        comp = fixed + slope * r
        ex, ag, ar = lin(tr1[7], tr8[7], r), lin(tr1[3], tr8[3], r), lin(tr1[5], tr8[5], r)
        # all-reduced arrays are global quantities (they do not shrink with the shard): the measured full-size figure
        ar = max(ar, float(tr1[5]))
        t_ex = (ex / n) / bw                    # per link: a rank's bytes for ONE peer
        t_ag = ag / bw                          # per link: one peer's whole contribution comes in
        t_ar = 2.0 * (n - 1) / n * ar / bw
        t = comp + calls * lat_ms + t_ex + t_ag + t_ar
        out["predicted"][str(n)] = {"ms_per_step": round(t, 2), "Gbases_per_s": round(bases1 / t / 1e6, 1), "speedup_over_1": round(ms_headline / t, 2),
                                    "efficiency": round(ms_headline / t / n, 3),
                                    "parts_ms": {"compute": round(comp, 2), "latencies": round(calls * lat_ms, 2), "exchange": round(t_ex, 3), "allgather": round(t_ag, 3), "allreduce": round(t_ar, 3)}}
    return out


def pmc_clock_ghz(kernel="syncmer_fast_kernel"):
    """the clock kernel B ran at, from the NEWEST committed GRBM_GUI_ACTIVE pass (profiles/*_pmc_clock_config3.csv, tools/pmc_r02.sh); 2.4 (the part's peak) without one"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_clock_config3.csv")), reverse=True):
        try:
            for ln in open(f).read().splitlines()[1:]:
                parts = ln.rsplit(",", 3)
                if len(parts) == 4 and kernel in parts[0]:
                    g = float(parts[3])
                    if 1.0 < g < 2.6:
                        return round(g, 3), os.path.basename(f)
        except (OSError, ValueError, IndexError):
            continue
    return 2.4, "the part's peak engine clock (no GRBM_GUI_ACTIVE pass under profiles/)"


def pmc_valu_per_64():
    """SQ_INSTS_VALU of kernel B per 64 hoco positions from the NEWEST committed PMC pass (profiles/*_pmc_scan.csv; tools/pmc.sh writes the positions the
    pass covered into the file's first line as `positions=<n>`); the constant below only when no pass carries its positions"""
    import glob
    import re
    env = float(os.environ.get("OATK_VALU_PER_64", "0"))
    if env:
        return env, "OATK_VALU_PER_64"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_scan.csv")), reverse=True):
        try:
            lines = open(f).read().splitlines()
            m = re.search(r"positions=(\d+)", lines[0])
            if not m:
                continue
            for ln in lines[1:]:
                parts = ln.rsplit(",", 2)
                if len(parts) == 3 and "syncmer_fast_kernel" in parts[0] and parts[1] == "SQ_INSTS_VALU":
                    return round(float(parts[2]) / (int(m.group(1)) / 64.0), 2), os.path.basename(f)
        except (OSError, ValueError, IndexError):
            continue
    return VALU_PER_64, "constant in bench.py (r03p PMC pass)"


# VALU wave-instructions kernel B issues per 64 hoco positions (profiles/r03p_pmc_scan.csv: SQ_INSTS_VALU 506.69 M over 450 M positions; 543.13 M = 77.2 with tiles of 2048 positions in r03k,
# 545.07 M = 77.5 in r02l / r03e; updated with every PMC pass)
VALU_PER_64 = 72.1

if __name__ == "__main__":
    main()
