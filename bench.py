#!/usr/bin/env python3
"""bench.py -- HiFi Gbases/s through the syncasm hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config3] [--reads-per-gpu R]

One "step" = one pass of the hot path (scan = homopolymer compression + closed-syncmer selection + k-mer
hash; count = syncmer ID assignment) over one batch of synthetic HiFi reads that is already resident in
HBM when the timed region starts.  At N = 1 the workload is BASELINE.json configs[1] (200 k reads x 15 kb,
k = 1001, s = 31, scan + count).  For N > 1 (launched by torch.distributed.run, one rank per GPU) reads are
sharded by record: rank r owns reads [r*R, (r+1)*R) of an N*R-read set ("weak" scaling) and the per-GPU
syncmer tables are merged over RCCL (oatk_amd/multi.py).  The `syncerr` object reports the same batch through the
error-correction round as well (scan + count + EC graph + read correction; sharded: oatk_amd/multi.py ShardedEc).

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against the HBM roofline with its
duration measured live by HIP events on the stream the kernels run on; `cpu_baseline` is the compiled
reference (oracle/_ref, built from the reference's own sources) timed on this box's host cores on a
bounded sample of the same reads.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--reads-per-gpu", type=int, default=0, help="override the number of reads each GPU owns")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; the contract) or gloo (development: several ranks on ONE GPU)")
    ap.add_argument("--no-syncerr", action="store_true", help="skip the extra scan + count + error-correction measurement")
    ap.add_argument("--sharded-syncerr", action="store_true",
                    help="at N > 1 also run the error-correction round sharded over the ranks (oatk_amd/multi.py: ShardedEc).  Off by default: it was "
                         "validated with two ranks sharing one GPU over gloo (tests/test_gpu_sharded_ec.py), not yet on a multi-GPU node over RCCL, "
                         "and a mismatch in a collective would hang the headline measurement with it")
    ap.add_argument("--no-ingest", action="store_true", help="skip the FASTA-text-to-syncmers measurement (device record scan, PCIe included)")
    ap.add_argument("--ingest-reads", type=int, default=50000)
    ap.add_argument("--cpu-sample-reads", type=int, default=80000)
    ap.add_argument("--cpu-threads", type=int, default=8)
    return ap.parse_args()


def cpu_baseline(readset, first, n_sample, k, s, threads, min_k_cov=30):
    """Reference scan + count (sr_read + collect_syncmer_from_reads of the compiled reference) on host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_lib
    if not ref_lib.available():
        return None
    seq, off, lens = readset.slice(first, n_sample)
    bases = int(lens.sum())
    fd, path = tempfile.mkstemp(suffix=".fa", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        with os.fdopen(fd, "wb") as f:
            for i in range(n_sample):
                f.write(b">r%d\n" % i)
                f.write(seq[int(off[i]):int(off[i]) + int(lens[i])].tobytes())
                f.write(b"\n")
        del seq
        t0 = time.perf_counter()
        db = ref_lib.SrDb([path], k, s, threads)
        sc = ref_lib.ScmDb(db)
        dt = time.perf_counter() - t0
        n_scm = sc.n()
        # the error-correction round of syncasm on the same databases (run_syncasm.c:109-124)
        import ec_util
        L = ref_lib.lib()
        t1 = time.perf_counter()
        g = L.refx_make_graph(db.handle, sc.handle, 0, 0.0)
        L.refx_consensus(db.handle, g, 1, 1)
        summ = ec_util.reference_ec(db, sc, g, 0.02, min_k_cov, 0.35, threads=threads)
        dt_ec = time.perf_counter() - t1
        L.refx_scg_destroy(g)
        # the assembly graph of the corrected reads and one alignment of all reads against it (run_syncasm.c:138, alignment.c:596)
        import ctypes as C
        L.refx_ra_new.restype = C.c_void_p
        L.refx_ra_destroy.argtypes = [C.c_void_p]
        L.refx_read_alignment.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        t2 = time.perf_counter()
        g = L.refx_make_graph(db.handle, sc.handle, min_k_cov, 0.35)
        dt_graph = time.perf_counter() - t2
        v = L.refx_ra_new()
        t2 = time.perf_counter()
        L.refx_read_alignment(db.handle, v, g, threads, 0)
        dt_aln = time.perf_counter() - t2
        L.refx_ra_destroy(v)
        L.refx_scg_destroy(g)
        sc.close()
        db.close()
    finally:
        os.unlink(path)
    return {"value": bases / dt / 1e9, "unit": "Gbases/s", "cores": threads, "kind": "reference",
            "sample": "first %d reads of the workload (%.2f Gbases) as FASTA through the compiled reference's sr_read + "
                      "collect_syncmer_from_reads at -t %d, parse included; %.1f s wall, %d syncmers" % (n_sample, bases / 1e9, threads, dt, n_scm),
            "with_syncerr": {"value": bases / (dt + dt_ec) / 1e9, "unit": "Gbases/s",
                             "sample": "the same, then make_syncmer_graph + scg_consensus + read_error_correction (-c %d); +%.1f s wall, %s error blocks"
                                       % (min_k_cov, dt_ec, summ.get("total"))},
            "asm_graph_ms": round(dt_graph * 1e3, 1), "read_alignment_ms": round(dt_aln * 1e3, 1),
            "after_syncerr": "make_syncmer_graph(-c %d, a 0.35) on the corrected sample, then scg_read_alignment of its %d reads against that graph at -t %d"
                             % (min_k_cov, n_sample, threads)}


def pmc_traffic(kernel_name, workload, per_gpu):
    """HBM bytes per launch of `kernel_name` from the committed PMC pass (profiles/*_pmc_hbm.csv: FETCH_SIZE / WRITE_SIZE collected in
    their own rocprofv3 passes on this very workload; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when the
    workload differs from the one that was profiled."""
    import csv
    import glob
    if workload != "config2" or per_gpu != 200000:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.csv")))
    if not files:
        return None
    for row in csv.reader(open(files[-1])):
        if row and row[0].startswith(kernel_name):
            return int((2.0 * float(row[1]) + float(row[2])) * 1024)
    return None


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
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "gloo":
            local_rank = 0                  # development only: the ranks share GPU 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local_rank)

    K, S = 1001, 31
    cfg = dict(CONFIGS[args.workload])
    per_gpu = args.reads_per_gpu or cfg["n_reads"]
    cfg["n_reads"] = per_gpu * world
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

    hip = HipSyncasm(local_rank)
    hip.set_timing(True)
    merger = None
    if world > 1:
        from oatk_amd.multi import CountMerger
        merger = CountMerger(hip, dist, dev)

    def step():
        hip.scan_device(d_seq.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), per_gpu, seq_bytes, K, S, sid0=first)
        hip.count()
        if merger is not None:
            merger.merge()

    def fence():
        hip.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    phase_ms = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
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
    info = hip.info()
    for name in phase_ms:
        phase_ms[name] /= max(args.steps, 1)

    # ---- the same batch through the error-correction round too (syncerr): scan + count + EC graph + read correction, all resident.
    #      Across GPUs every rank builds the graph of ALL reads from the all-gathered adjacent pairs and corrects its own reads. ----
    syncerr = None
    if not args.no_syncerr and (world == 1 or args.sharded_syncerr):
        c = int(cfg.get("min_k_cov", 30))
        sharded = None
        if world > 1:
            from oatk_amd.multi import ShardedEc
            sharded = ShardedEc(hip, dist, dev)

        def step_ec():
            hip.scan_device(d_seq.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), per_gpu, seq_bytes, K, S, sid0=first)
            hip.count()
            if sharded is not None:         # count-table merge, graph from everybody's pairs, correction in global ids (oatk_amd/multi.py)
                return sharded.run(0.02, c, 0.35)["stats"]
            hip.ec_graph()
            return hip.ec(0.02, c, 0.35)

        try:                                # an extension of the headline measurement: it must never take the headline down
            hip.set_timing(False)
            step_ec()
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                st = step_ec()
            fence()
            dt_ec = (time.perf_counter() - t1) / max(args.steps, 1)
            if dist is not None:
                t = torch.tensor([dt_ec], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_ec = float(t.item())
            syncerr = {"value": round(total_bases / dt_ec / 1e9, 3), "unit": "Gbases/s", "ms_per_step": round(dt_ec * 1e3, 3),
                       "workload": "scan + count + %sEC graph (make_syncmer_graph, hoco arc overlaps) + read_error_correction (-c %d, max_edist 0.02, a 0.35)"
                                   % ("count-table merge + all-gather of adjacent pairs + " if world > 1 else "", c),
                       "error_blocks": int(st[0] + st[5] + st[10]), "corrected": int(st[2] + st[7]), "uncorrected": int(st[1] + st[6]),
                       "ambiguous": int(st[3] + st[4] + st[8] + st[9]), "blocks_past_first_tier": int(st[11])}
            if sharded is not None:
                syncerr["imported_kmers_rank0"] = sharded.n_imported
            else:                           # the assembly graph of the corrected reads (run_syncasm.c:138), not part of `value`
                hip.asm_graph(c, 0.35)
                fence()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    nv, na = hip.asm_graph(c, 0.35)
                fence()
                syncerr["asm_graph"] = {"ms": round((time.perf_counter() - t1) / max(args.steps, 1) * 1e3, 3), "n_vtx": nv, "n_arc": na,
                                        "workload": "make_syncmer_graph(-c %d, a 0.35) + asmg_finalize on the corrected chains" % c}
                # what scg_consensus needs of the reads: run-length totals of every live syncmer, distance tables of every adjacent pair
                for name, fn, what in (("consensus", lambda: hip.consensus(c), "oatk_hip_consensus(-c %d): scg_syncmer_consensus' sums for every live syncmer" % c),
                                       ("overlap_hist", hip.overlap_hist, "oatk_hip_overlap_hist: calc_syncmer_overlap's tables for every adjacent pair")):
                    fn()
                    fence()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        fn()
                    fence()
                    syncerr[name] = {"ms": round((time.perf_counter() - t1) / max(args.steps, 1) * 1e3, 3), "workload": what}
                # every corrected read against that graph, one syncmer per vertex (scg_read_alignment before the unitigging)
                ag = hip.fetch_asm_graph()
                n_scm_all = len(ag["scm_del"])
                su_off = np.zeros(n_scm_all + 1, np.uint64)
                su_off[1:] = np.cumsum(ag["scm_del"] == 0)
                graph = {"n_scm": n_scm_all, "su_off": su_off, "su_uid": np.arange(nv, dtype=np.uint64) << np.uint64(1), "su_pos": np.zeros(nv, np.uint32),
                         "utg_n": np.ones(nv, np.uint32), "idx_p": ag["idx_p"], "idx_n": ag["idx_n"].astype(np.uint64), "arc_w": ag["arc_w"],
                         "arc_ln": np.zeros(na, np.uint64), "arc_del": np.zeros(na, np.uint8)}
                hip.read_alignment(graph)
                fence()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    n_aln, n_frg, ast = hip.read_alignment(graph)
                fence()
                syncerr["read_alignment"] = {"ms": round((time.perf_counter() - t1) / max(args.steps, 1) * 1e3, 3), "alignments": n_aln, "fragments": n_frg,
                                             "reads_aligned": int(ast[0]), "reads_over_limits": int(ast[2]),
                                             "workload": "scg_read_alignment of the %d corrected reads against the %d-vertex graph (graph upload included)" % (per_gpu, nv)}
        except Exception as ex:             # noqa: BLE001
            syncerr = {"error": "%s: %s" % (type(ex).__name__, ex)}
        hip.set_timing(True)
        # restore the scan + count state the rest of this report describes
        hip.scan_device(d_seq.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), per_gpu, seq_bytes, K, S, sid0=first)
        hip.count()
        fence()

    # ---- from the TEXT of a FASTA file to syncmers: host -> device copy of the text (PCIe), record scan on the device
    #      (include/oatk_hip_ingest.h), scan.  Rank 0, a bounded sample; the reference's reader does 0.4 Gbases/s here. ----
    ingest = None
    if rank == 0 and not args.no_ingest:
        try:
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
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                d_text = text.to(dev, non_blocking=True)
                torch.cuda.synchronize()
                t_copy = time.perf_counter() - t0
                t0 = time.perf_counter()
                n_found, _ = hip.ingest_device(d_text.data_ptr(), int(d_text.numel()), 1, True)
                hip.sync()
                t_ing = time.perf_counter() - t0
                t0 = time.perf_counter()
                hip.scan_ingested(K, S, sid0=first)
                hip.count()
                hip.sync()
                t_sc = time.perf_counter() - t0
                assert n_found == n_ing
                tot = t_copy + t_ing + t_sc
                if best is None or tot < best[0]:
                    best = (tot, t_copy, t_ing, t_sc)
            ingest = {"value": round(ing_bases / best[0] / 1e9, 3), "unit": "Gbases/s",
                      "workload": "text of an unwrapped FASTA file with %d reads (%.2f GB) in pinned host memory -> PCIe copy -> record scan on the device -> scan + count"
                                  % (n_ing, text.numel() / 1e9),
                      "ms_h2d": round(best[1] * 1e3, 3), "h2d_GBs": round(text.numel() / best[1] / 1e9, 2),
                      "ms_record_scan": round(best[2] * 1e3, 3), "record_scan_GBs_of_text": round(text.numel() / best[2] / 1e9, 2),
                      "ms_scan_count": round(best[3] * 1e3, 3)}
            del text, d_text
        except Exception as ex:             # noqa: BLE001
            ingest = {"error": "%s: %s" % (type(ex).__name__, ex)}
        # restore the scan + count state the rest of this report describes
        hip.scan_device(d_seq.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), per_gpu, seq_bytes, K, S, sid0=first)
        hip.count()
        hip.sync()

    if rank == 0:
        # ---- roofline of the dominant kernel (by measured time) ----
        hoco = int(hip.fetch("HOCO_L").astype(np.uint64).sum())
        n_occ = info["n_occ"]
        alg_bytes = {
            # kernel A: ASCII in, 2-bit hoco_s + ho_rl out (1 + 0.25 rho + rho per raw base)
            "hpc": bases + hoco // 4 + hoco,
            # kernel B: 2-bit hoco_s in, one 20-byte record per syncmer occurrence out (the 8-byte hash comes from kmer_hash_kernel)
            "syncmer": hoco // 4 + 20 * n_occ,
        }
        dom = max(("hpc", "syncmer"), key=lambda k_: phase_ms.get(k_, 0.0))
        dur_s = phase_ms[dom] / 1e3
        achieved = alg_bytes[dom] / dur_s / 1e9 if dur_s > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": {"hpc": "oatk::hpc_pack_kernel", "syncmer": "oatk::syncmer_fast_kernel<4096, true, 256, (-(K-S))&7>"}[dom],
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": pmc_traffic({"hpc": "oatk::hpc_pack_kernel", "syncmer": "void oatk::syncmer_fast_kernel<4096, true"}[dom], args.workload, per_gpu),
                    "algorithmic_bytes_per_launch": alg_bytes[dom], "avg_launch_ms": round(phase_ms[dom], 4),
                    "note": "kernel B is integer-VALU issue bound (PMC profiles/r01q_pmc_scan.csv: 78.7 VALU wave-instructions per 64 hoco positions, 41 of them roll + hash64, ~14 of those 64-bit forms that take two passes; 16 waves per CU), see DESIGN.md 5",
                    # the bound that actually binds kernel B: wave-instructions issued (PMC count per 64 positions, profiles/r01k_pmc_scan.csv)
                    # against 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction
                    "valu": {"achieved": round(hoco / 64 * 78.7 / (phase_ms["syncmer"] / 1e3) / 1e9, 1), "peak": 614.4, "unit": "G wave-instr/s",
                             "frac": round(hoco / 64 * 78.7 / (phase_ms["syncmer"] / 1e3) / 1e9 / 614.4, 3)},
                    "scan_bytes_per_base": round((alg_bytes["hpc"] + 28 * n_occ) / bases, 4),
                    "scan_achieved_GBs": round((alg_bytes["hpc"] + 28 * n_occ) / ((phase_ms["hpc"] + phase_ms["syncmer"]) / 1e3) / 1e9, 2)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(rs, first, min(args.cpu_sample_reads, per_gpu), K, S, args.cpu_threads, int(cfg.get("min_k_cov", 30)))
        out = {
            "metric": "HiFi Gbases/s through syncasm scan+count (closed syncmers, k=1001 s=31)",
            "value": round(total_bases * args.steps / dt / 1e9, 3), "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: %d reads x ~%d kb per GPU, k=1001 s=31, syncmer scan + count, reads resident in HBM"
                                   % (args.workload, per_gpu, cfg["mean_len"] // 1000),
                       "reads_per_gpu": per_gpu, "bases_per_gpu": bases, "genome_len": cfg["genome_len"],
                       "parallelism": "reads sharded by record, %d rank(s)" % world},
            "roofline": roofline, "cpu_baseline": cpu, "syncerr": syncerr, "ingest": ingest,
            "phases_ms": {k_: round(v, 4) for k_, v in phase_ms.items()},
            "syncmers": {"occurrences": n_occ, "distinct": info["n_scm"], "hoco_ratio": round(hoco / bases, 4)},
        }
        print(json.dumps(out), flush=True)
    hip.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
