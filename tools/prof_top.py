#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats CSV with short kernel names.  usage: prof_top.py <kernel_stats.csv> [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows[:n]:
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", r["Name"])
    m = re.search(r"rocprim::detail::(\w+)", name)
    if name.startswith("void rocprim") and m:
        name = "rocprim::" + m.group(1) + " " + re.sub(r".*wrapped_(\w+)_config.*", r"\1", name)[:40]
    print("%-70s calls %4s  tot %9.3f ms  avg %9.3f ms  %5.1f%%" % (name[:70], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
