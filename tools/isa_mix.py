#!/usr/bin/env python3
"""Opcode mix of kernel B's tile loop, priced with the per-opcode issue rates measured on the box (profiles/r02c_valu_rates.txt, tools/ubench/valu_rates.hip):
the VERDICT of round 2 asked for roofline.valu from Sigma n_i c_i instead of a flat 4 cycles per wave64 instruction.  CPU only (hipcc -S).
    python tools/isa_mix.py profiles/r03c_isa_mix_syncmer_fast.json"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "syncmer_fast_kernelILi2048ELb1ELi128ELi6E"
# cycles per wave64 instruction per SIMD, event-derived at 2.4 GHz (profiles/r02c_valu_rates.txt); classes not measured there take the flat 4.0
RATES = [(r"^v_(xor|and|or|not|add|sub|subrev|mov|cndmask|lshlrev|lshrrev|ashrrev|bfe|bfi|perm|min|max|min3|max3|and_or|or3|xad|lshl_add|add_lshl|lshl_or|add3|xor3)_[a-z]?(b|u|i)?(16|32)?(_e32|_e64|_dpp|_sdwa)?$", 2.9, "32-bit simple"),
         (r"^v_alignbit_b32", 5.15, "v_alignbit_b32"), (r"^v_mul_u32_u24|^v_mad_u32_u24", 4.76, "24-bit multiply"), (r"^v_mul_lo_u32", 5.49, "v_mul_lo_u32"),
         (r"^v_mul_hi_u32", 5.17, "v_mul_hi_u32"), (r"^v_mad_u64_u32", 5.20, "v_mad_u64_u32"), (r"^v_lsh[lr]rev_b64|^v_ashrrev_i64", 4.47, "64-bit shift"),
         (r"^v_lshl_add_u64", 4.82, "v_lshl_add_u64"), (r"^v_add_co|^v_addc|^v_sub_co|^v_subb", 4.72, "carry add"), (r"^v_cmp_.*_[ui]64", 4.68, "64-bit compare"),
         (r"^v_cmp_", 2.9, "32-bit compare"), (r"^v_mov_b64|^v_pk_", 4.0, "64-bit move")]


def main():
    d = tempfile.mkdtemp()
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", os.path.join(ROOT, "oatk_amd/csrc/api.hip"), "--save-temps", "-o", "x.o"], cwd=d, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(d, "api-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
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
           "note": "static count over the tile loop of the instantiation bench.py runs (rare paths inside it included); cycles per class from profiles/r02c_valu_rates.txt"}
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
