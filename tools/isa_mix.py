#!/usr/bin/env python3
"""Opcode mix of kernel B's tile loop, priced with the per-opcode issue rates measured on the box (profiles/r04a_valu_rates.txt, tools/ubench/valu_rates.hip):
the VERDICT of round 2 asked for roofline.valu from Sigma n_i c_i instead of a flat 4 cycles per wave64 instruction.  CPU only (hipcc -S).
    python tools/isa_mix.py profiles/r03c_isa_mix_syncmer_fast.json"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "syncmer_fast_kernelILi2048ELb1ELi128ELi6E"
# cycles per wave64 instruction per SIMD, event-derived at 2.4 GHz (profiles/r04a_valu_rates.txt); classes not measured there take the flat 4.0
# cycles per wave64 instruction per SIMD with >= 2 waves resident, event-derived at the MEASURED clock (round 4: profiles/r04a_valu_rates.txt and
# r04a_valu_rates_more.txt, kernels of >= 25 ms; the round-3 table came from 0.2 ms kernels priced at an assumed 2.4 GHz).  Two classes: ~2.3 for the plain
# two-operand forms of xor / and / or / not / add / sub / mov / lshrrev_b32 (and v_mul_f32), ~4.1 - 4.3 for everything else measured -- min / max,
# lshlrev_b32 (!), every three-operand form, every 64-bit operation incl. v_mad_u64_u32 and v_lshl_add_u64, compares, v_cndmask, DPP.
FAST = r"^v_(xor|and|or|not|add|sub|subrev|mov|lshrrev)_(b|u|i)32(_e32|_e64)?$"
RATES = [(FAST, 2.3, "full rate (2.3)"), (r"^v_bitop3_b32|^v_fma", 3.7, "v_bitop3 / fma (3.7)"), (r"^v_mad_u64_u32|^v_add_co|^v_addc|^v_sub_co|^v_subb", 4.3, "v_mad_u64_u32 / carry (4.3)"),
         (r"^v_", 4.15, "half rate (4.15)")]


def main():
    d = tempfile.mkdtemp()
    # (the kernel alone: the header with the one instantiation bench.py's workload runs -- seconds instead of the minutes api.hip takes)
    with open(os.path.join(d, "kb.hip"), "w") as f:
        f.write('#include <hip/hip_runtime.h>\n#include "%s"\ntemplate __global__ void oatk::syncmer_fast_kernel<2048, true, 128, 6>(oatk::SynArgs);\n'
                % os.path.join(ROOT, "oatk_amd/csrc/scan_syncmer_fast.hpp"))
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "kb.s", "kb.hip"], cwd=d, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(d, "kb.s")).read()
    m = re.search(r"^(_ZN4oatk\d+" + KERNEL + r"[A-Za-z0-9_]*):(.*?)\.Lfunc_end", asm, re.S | re.M)
    lines = [ln.strip() for ln in m.group(2).splitlines()]
    ins, labels = [], {}
    for ln in lines:
        if not ln or ln.startswith((";", "//", ".")) and not ln.endswith(":"):
            continue
        if ln.endswith(":"):
            labels[ln[:-1]] = len(ins)
            continue
        ins.append(ln)
    # the tile loop: the backward branch that spans the most instructions
    best = (0, 0, 0)
    for i, ln in enumerate(ins):
        mm = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", ln)
        if mm:
            tgt = labels.get(mm.group(1) or mm.group(2))
            if tgt is not None and tgt < i and i - tgt > best[0]:
                best = (i - tgt, tgt, i)
    body = ins[best[1]:best[2] + 1]
    by = collections.Counter()
    ops = collections.Counter()
    cyc = 0.0
    n_valu = 0
    for ln in body:
        op = ln.split()[0]
        if not op.startswith("v_"):
            continue
        n_valu += 1
        ops[op] += 1
        for pat, c, name in RATES:
            if re.search(pat, op):
                by[name] += 1
                cyc += c
                break
        else:
            by["other (4.0)"] += 1
            cyc += 4.0
    out = {"kernel": m.group(1), "loop_instructions": len(body), "loop_valu": n_valu, "by_class": dict(by.most_common()), "top_opcodes": dict(ops.most_common(20)),
           "cycles_per_valu_instruction_mix": round(cyc / max(n_valu, 1), 3),
           "note": "static count over the tile loop of the instantiation bench.py runs (rare paths inside it included); cycles per class from profiles/r04a_valu_rates.txt"}
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
