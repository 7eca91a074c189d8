#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average, like `--stats`.

    python tools/rocpd_stats.py gpurun_out/<tag>/prof/trace_results.db profiles/<name>.csv
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", name)
    m = re.search(r"rocprim::detail::(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|scan_impl|transform_impl|"
                  r"init_lookback_scan_state_kernel|radix_sort_block_sort|radix_sort_single)", name)
    if m:
        tail = "max" if "maximum<" in name else ("u64" if "unsigned long>" in name[:400] else "")
        return "rocprim::" + m.group(1) + ("[" + tail + "]" if tail else "")
    return name[:120]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    agg = {}
    for name, calls, tot, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls
        a[1] += tot
        a[2] += pct
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for k, (calls, tot, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, calls, "%.3f" % tot, "%.3f" % (tot / calls), "%.3f" % pct])
    print(open(sys.argv[2]).read())


if __name__ == "__main__":
    main()
