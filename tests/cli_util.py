"""The syncasm CLI, reference binary against drop-in binary (SURVEY.md 8d timing iii): both are built from the reference's own translation
units by oracle/Makefile (`make ref ref_dropin`) where its sources are, and travel as built artefacts.  The drop-in binary is the same main()
linked over oatk_amd/lib/liboatk_dropin.a (include/oatk_dropin.h).  Used by tests/test_gpu_cli.py, tests/test_cli_fallback.py and the
cpu_baseline leg of bench.py."""
import filecmp
import os
import re
import subprocess
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI_REF = os.path.join(ROOT, "oracle", "_ref", "syncasm")
CLI_DROPIN = os.path.join(ROOT, "oracle", "_ref", "syncasm_dropin")


def available():
    return os.path.exists(CLI_REF) and os.path.exists(CLI_DROPIN)


def run_cli(binary, fasta, out, k, c, threads, env=None, extra=()):
    e = dict(os.environ)
    e.update(env or {})
    t0 = time.perf_counter()
    files = [fasta] if isinstance(fasta, str) else list(fasta)
    p = subprocess.run([binary, "-k", str(k), "-c", str(c), "-t", str(threads), "-o", out] + list(extra) + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        raise RuntimeError("%s exited with %d: %s" % (os.path.basename(binary), p.returncode, p.stderr.decode(errors="replace")[-400:]))
    return dt, p.stderr.decode(errors="replace")


def served_table(stderr_text):
    """the summary OATK_DROPIN_LOG=1 prints at exit -> {function: (device calls, device s, original calls, original s)}"""
    out = {}
    for m in re.finditer(r"\[M::oatk_dropin\] (\w+)\s+(\d+)\s+([\d.]+)\s+(\d+)\s+([\d.]+)", stderr_text):
        out[m.group(1)] = (int(m.group(2)), float(m.group(3)), int(m.group(4)), float(m.group(5)))
    return out


def time_cli(readset, first, n_reads, k, s, c, threads, workdir=None, strict=True, devices=None):
    """strict: the leg FAILS (raises) when the two GFA files differ or when any of the six hot-path calls was served by its original body -- a drop-in
    number is only worth reporting when the device did the work and the bytes are the reference's"""
    if not available():
        return {"skipped": "the CLI binaries are built only where the reference's sources are (make ref ref_dropin)"}
    seq, off, lens = readset.slice(first, n_reads)
    bases = int(lens.sum())
    d = tempfile.mkdtemp(prefix="oatk_cli_", dir=workdir or os.environ.get("TMPDIR", "/tmp"))
    fa = os.path.join(d, "reads.fa")
    try:
        with open(fa, "wb") as f:
            for i in range(n_reads):
                f.write(b">r%d\n" % i)
                f.write(seq[int(off[i]):int(off[i]) + int(lens[i])].tobytes())
                f.write(b"\n")
        del seq
        t_ref, _ = run_cli(CLI_REF, fa, os.path.join(d, "ref"), k, c, threads)
        t_dev, err = run_cli(CLI_DROPIN, fa, os.path.join(d, "dev"), k, c, threads, {"OATK_DROPIN_LOG": "1"})
        same = all(filecmp.cmp(os.path.join(d, "ref" + x), os.path.join(d, "dev" + x), shallow=False) for x in (".utg.gfa", ".utg.final.gfa"))
        tab = served_table(err)
        if strict:
            six = ("sr_read", "sr_db_stat", "collect_syncmer_from_reads", "make_syncmer_graph", "read_error_correction", "scg_read_alignment")
            fell_back = {f: tab[f][2] for f in six if f in tab and tab[f][2] > 0}
            missing = [f for f in six if f not in tab or tab[f][0] == 0]
            if not same or fell_back or missing:
                raise RuntimeError("drop-in CLI run rejected: gfa_identical=%s, calls served by their original bodies %s, calls the device never served %s"
                                   % (same, fell_back, missing))
        multi = None
        if devices:
            # the same file once more with the reads spread over several handles (OATK_DEVICES; include/oatk_multi.h)
            t_m, err_m = run_cli(CLI_DROPIN, fa, os.path.join(d, "mul"), k, c, threads, {"OATK_DROPIN_LOG": "1", "OATK_DEVICES": devices})
            same_m = all(filecmp.cmp(os.path.join(d, "ref" + x), os.path.join(d, "mul" + x), shallow=False) for x in (".utg.gfa", ".utg.final.gfa"))
            tab_m = served_table(err_m)
            orig_m = {f: v[2] for f, v in tab_m.items() if v[2] > 0}
            if strict and (not same_m or orig_m):
                raise RuntimeError("drop-in CLI run over OATK_DEVICES=%s rejected: gfa_identical=%s, original bodies %s" % (devices, same_m, orig_m))
            multi = {"devices": devices, "dropin_s": round(t_m, 2), "speedup": round(t_ref / t_m, 2), "gfa_identical": bool(same_m)}
        m = re.search(r"device context ready after ([\d.]+) s; exit handlers reached at \+([\d.]+) s", err)
        phases = {"device_context_ready_s": float(m.group(1)), "exit_handlers_at_s": float(m.group(2))} if m else None
        return {"reads": n_reads, "gbases": round(bases / 1e9, 3), "threads": threads, "several_handles": multi, "dropin_phases": phases,
                "reference_s": round(t_ref, 2), "dropin_s": round(t_dev, 2), "speedup": round(t_ref / t_dev, 2), "gfa_identical": bool(same),
                "dropin_value": round(bases / t_dev / 1e9, 3), "reference_value": round(bases / t_ref / 1e9, 3), "unit": "Gbases/s",
                "served": {f: {"device_calls": v[0], "device_s": v[1], "original_calls": v[2], "original_s": v[3]} for f, v in tab.items()},
                "workload": "syncasm -k %d -c %d -t %d on a FASTA file of %d reads: process start to exit, parse and both GFA files included; "
                            "the drop-in binary's wall clock includes creating the HIP context" % (k, c, threads, n_reads)}
    finally:
        for fn in os.listdir(d):
            os.unlink(os.path.join(d, fn))
        os.rmdir(d)
