#!/usr/bin/env python3
"""Rough VALU issue-cycle count of a range of an AMDGPU .s listing (gfx950, measured with scratch/ubench/valu_rate.hip):
plain 2-operand 32-bit VOP1/VOP2 forms 2 cycles, everything else (VOP3, compares, DPP, 64-bit, perm, dot4, sad, mul) 4."""
import re, sys, collections
fast = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_not_b32", "v_add_f32", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
def cost(op, line):
    base = re.sub(r"_e(32|64)$", "", op)
    if op.endswith("_e64") or "dpp" in op or "sdwa" in op or "row_" in line or "quad_perm" in line: return 4
    return 2 if base in fast else 4
lo, hi = int(sys.argv[2]), int(sys.argv[3])
tot = collections.Counter(); n = collections.Counter()
for i, line in enumerate(open(sys.argv[1]), 1):
    if i < lo or i > hi: continue
    m = re.match(r"\s+(v_\w+)", line)
    if not m: continue
    op = m.group(1); c = cost(op, line)
    tot[re.sub(r"_e(32|64)$", "", op)] += c; n[re.sub(r"_e(32|64)$", "", op)] += 1
print("VALU instr", sum(n.values()), "cycles", sum(tot.values()))
for k, v in tot.most_common(25): print("  %-22s n=%4d cycles=%5d" % (k, n[k], v))
