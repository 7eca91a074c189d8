#!/usr/bin/env python3
"""A memory ceiling around a command (a host path that touched tens of TB went to the GPU box three times and cost the round's GPU access).

    python tools/memcap.py [--rss-gb 96] [--avail-frac 0.25] [--timeout S] -- <command ...>

RLIMIT_AS cannot be used around a HIP process (the runtime reserves terabytes of address space at start-up), so the ceiling is a watchdog:
the command runs in a process group of its own; every 50 ms the resident sets of the group's processes are summed (/proc/<pid>/statm) and the machine's
MemAvailable is read; past --rss-gb, or with less than --avail-frac of MemTotal available, the whole group gets SIGKILL and the exit code is 137 with one
line on stderr that says which limit it was.  A memset runs at ~10 GB/s per thread: 64 threads reach 32 GB between two looks, which is why the default
ceiling is far below the box's 3 TB.  --timeout kills the group the same way (exit 124).  Every tools/*.sh that runs on a GPU box goes through this.
"""
import os
import signal
import subprocess
import sys
import time


def group_rss_bytes(pgid, page):
    total = 0
    for d in os.listdir("/proc"):
        if not d.isdigit():
            continue
        try:
            with open("/proc/%s/stat" % d) as f:
                st = f.read()
            # pgrp is the 3rd field after the ")" that ends comm
            if int(st[st.rindex(")") + 2:].split()[2]) != pgid:
                continue
            with open("/proc/%s/statm" % d) as f:
                total += int(f.read().split()[1]) * page
        except (OSError, ValueError, IndexError):
            continue
    return total


def meminfo():
    tot = avail = 0
    with open("/proc/meminfo") as f:
        for line in f:
            if line.startswith("MemTotal:"):
                tot = int(line.split()[1]) * 1024
            elif line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    return tot, avail


def main():
    a = sys.argv[1:]
    rss_gb, frac, timeout = 96.0, 0.25, None
    while a and a[0] != "--":
        if a[0] == "--rss-gb":
            rss_gb = float(a[1]); a = a[2:]
        elif a[0] == "--avail-frac":
            frac = float(a[1]); a = a[2:]
        elif a[0] == "--timeout":
            timeout = float(a[1]); a = a[2:]
        else:
            sys.exit("memcap: unknown option %s" % a[0])
    cmd = a[1:]
    if not cmd:
        sys.exit(__doc__)
    page = os.sysconf("SC_PAGE_SIZE")
    p = subprocess.Popen(cmd, start_new_session=True)
    pgid = p.pid
    t0 = time.time()
    peak = 0
    why = None
    code = None
    n = 0
    while True:
        code = p.poll()
        if code is not None:
            break
        rss = group_rss_bytes(pgid, page)
        peak = max(peak, rss)
        tot, avail = meminfo()
        if rss > rss_gb * 1e9:
            why, code = "resident set of the command's process group %.1f GB > --rss-gb %.0f" % (rss / 1e9, rss_gb), 137
        elif tot and avail < frac * tot and rss > 0.5 * (tot - avail):
            # (only when this command is what holds the memory: a box that is short for other reasons is not this command's doing)
            why, code = "MemAvailable %.1f GB < %.0f %% of %.1f GB with %.1f GB resident here" % (avail / 1e9, frac * 100, tot / 1e9, rss / 1e9), 137
        elif timeout is not None and time.time() - t0 > timeout:
            why, code = "--timeout %.0f s" % timeout, 124
        if why:
            try:
                os.killpg(pgid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            p.wait()
            break
        n += 1
        time.sleep(0.05)
    if why:
        sys.stderr.write("[memcap] KILLED: %s (after %.1f s): %s\n" % (why, time.time() - t0, " ".join(cmd)[:200]))
    else:
        sys.stderr.write("[memcap] rc %d, peak resident set %.2f GB, %.1f s\n" % (code, peak / 1e9, time.time() - t0))
    sys.exit(code if code >= 0 else 128 - code)


if __name__ == "__main__":
    main()
