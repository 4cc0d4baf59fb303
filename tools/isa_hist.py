#!/usr/bin/env python3
"""Instruction-class histogram of the loops of one kernel in a gfx950 assembly file (hipcc -S
--cuda-device-only).  usage: isa_hist.py file.s <substring of the mangled kernel name> [min_body]

A loop is a backward branch to a label; its body is the text between the label and the branch.
Prints, per loop with at least `min_body` instructions, the number of instructions by class and
the most frequent mnemonics, so that the VALU instructions that are not floating-point arithmetic
(moves, selects, compares, conversions, integer address work) can be named."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_body = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path).read().splitlines()
# the kernel's text: from its label to its s_endpgm / .Lfunc_end
start = next(i for i, l in enumerate(lines) if re.match(r"^[_\w]+:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
insts = []
for l in body:
    s = l.split(";")[0].strip()
    if not s or s.startswith("."):
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(insts)
        continue
    m = re.match(r"^(\.?\w+):", s)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    insts.append(s)


def klass(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac_f32",
                      "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")):
        return "valu fp arith"
    if op.startswith(("v_max_f64", "v_min_f64", "v_max_f32", "v_min_f32")):
        return "valu fp min/max"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_div", "v_rndne", "v_fract", "v_trig", "v_ldexp", "v_frexp", "v_sin", "v_cos", "v_log", "v_exp")):
        return "valu fp other"
    if op.startswith("v_cmp"):
        return "valu compare"
    if op.startswith("v_cndmask"):
        return "valu select"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap")):
        return "valu move"
    if op.startswith("v_cvt"):
        return "valu convert"
    if op.startswith(("v_mfma", "v_smfmac")):
        return "mfma"
    if op.startswith("v_"):
        return "valu int/bit"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "vmem store"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_nop"):
        return "s_nop"
    return "salu"


loops = []
for i, s in enumerate(insts):
    m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", s)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i, m.group(1)))
print(f"kernel text: {len(insts)} instructions, {len(loops)} loops")
for lo, hi, lab in loops:
    n = hi - lo + 1
    if n < min_body:
        continue
    ops = [s.split()[0] for s in insts[lo:hi + 1]]
    by = collections.Counter(klass(o) for o in ops)
    valu = sum(v for k, v in by.items() if k.startswith("valu") or k == "mfma")
    print(f"\nloop {lab}: {n} instructions, {valu} VALU")
    for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
        print(f"  {k:18s} {v:5d}")
    top = collections.Counter(ops).most_common(24)
    print("  top mnemonics: " + ", ".join(f"{o} {c}" for o, c in top))
