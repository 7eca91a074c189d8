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


def time_cli(readset, first, n_reads, k, s, c, threads, workdir=None, strict=True, devices=None, settle_s=0.0):
    """strict: the leg FAILS (raises) when the two GFA files differ or when any of the six hot-path calls was served by its original body -- a drop-in
    number is only worth reporting when the device did the work and the bytes are the reference's"""
    if not available():
        return {"skipped": "the CLI binaries are built only where the reference's sources are (make ref ref_dropin)"}
    seq, off, lens = readset.slice(first, n_reads)
    bases = int(lens.sum())
    d = tempfile.mkdtemp(prefix="oatk_cli_", dir=workdir or os.environ.get("TMPDIR", "/tmp"))
    fa = os.path.join(d, "reads.fa")
    try:
        from oatk_amd import synth
        synth.write_fasta(fa, seq, off, lens, mode=synth.FA_PLAIN)          # (host/fasta_out.c: 30 GB at 2 M reads, on every host thread)
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
            # (settle_s: the driver clears what the run before gave back -- 80 GB at 2 M reads, ~33 GB/s -- and whoever touches the device next waits for that,
            #  a run is timed with the device to itself)
            if settle_s:
                time.sleep(settle_s)
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


def time_cli_gz(seq, off, lens, n_reads, k, c, threads, workdir=None, forms=("one", "bgzf"), settle_s=0.0):
    """the same comparison on a gzip'ed file (BASELINE.json configs[0] is a .fa.gz): the reference binary on the single-member file, the drop-in binary on
    every form asked for; zcat's time beside it (what the reference's reader waits for); GFA files compared, original bodies counted"""
    if not available():
        return {"skipped": "the CLI binaries are built only where the reference's sources are (make ref ref_dropin)"}
    from oatk_amd import synth
    modes = {"one": synth.FA_GZ, "bgzf": synth.FA_BGZF, "members": synth.FA_GZ_MEMBERS}
    bases = int(lens[:n_reads].sum())
    d = tempfile.mkdtemp(prefix="oatk_cligz_", dir=workdir or os.environ.get("TMPDIR", "/tmp"))
    out = {"reads": n_reads, "gbases": round(bases / 1e9, 3), "threads": threads}
    try:
        for f in forms:
            synth.write_fasta(os.path.join(d, f + ".fa.gz"), seq, off[:n_reads], lens[:n_reads], mode=modes[f], member_bytes=200_000_000)
        t0 = time.perf_counter()
        subprocess.run("zcat %s > /dev/null" % os.path.join(d, forms[0] + ".fa.gz"), shell=True)
        out["zcat_s"] = round(time.perf_counter() - t0, 2)
        out["file_MB"] = os.path.getsize(os.path.join(d, forms[0] + ".fa.gz")) >> 20
        t_ref, _ = run_cli(CLI_REF, os.path.join(d, forms[0] + ".fa.gz"), os.path.join(d, "ref"), k, c, threads)
        out["reference_s"] = round(t_ref, 2)
        for i, f in enumerate(forms):
            if settle_s and i:
                time.sleep(settle_s)            # (see time_cli)
            t_dev, err = run_cli(CLI_DROPIN, os.path.join(d, f + ".fa.gz"), os.path.join(d, f), k, c, threads, {"OATK_DROPIN_LOG": "1"})
            same = all(filecmp.cmp(os.path.join(d, "ref" + x), os.path.join(d, f + x), shallow=False) for x in (".utg.gfa", ".utg.final.gfa"))
            tab = served_table(err)
            m = re.search(r"oatk_sr_read_files\] .*?: ([\d.]+) s \(waiting for the uploader ([\d.]+)", err)
            out[f] = {"dropin_s": round(t_dev, 2), "speedup": round(t_ref / t_dev, 2), "gfa_identical": bool(same),
                      "sr_read_s": float(m.group(1)) if m else None, "waiting_for_the_inflater_s": float(m.group(2)) if m else None,
                      "original_calls": {fn: v[2] for fn, v in tab.items() if v[2] > 0},
                      "device_s": {fn: v[1] for fn, v in tab.items() if v[0] > 0 and v[1] >= 0.01}}
        return out
    finally:
        for fn in os.listdir(d):
            os.unlink(os.path.join(d, fn))
        os.rmdir(d)
