"""Static VALU cost estimate of a span of gfx950 assembly, using the per-instruction issue costs
measured by tools/ubench (ns per wave-instruction per SIMD, MI355X).  Debug tool.

    python3 tools/asm_cost.py file.s first_line last_line
"""
import re, sys
from collections import Counter

FULL = 1.0    # v_add/sub/mul_f32, v_and/or/xor, v_mov, v_add_u32 ...
FMA = 1.3
HALF = 1.8    # v_cmp, v_cndmask, v_min/max, cvt, int mul, packed f32, VOP3 shifts, SGPR-operand ops
TRANS = 3.45
F64 = 2.2


def cost(ins, ops):
    if ins.startswith(("v_sqrt", "v_rcp", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans", TRANS
    if "f64" in ins and not ins.startswith("v_cvt"):
        return "f64", F64
    if ins.startswith(("v_pk_",)):
        return "pk", HALF
    if ins.startswith(("v_cmp", "v_cndmask", "v_min", "v_max", "v_med3", "v_cvt", "v_mul_lo", "v_mul_hi", "v_mad_u64",
                       "v_div_", "v_lshl_add", "v_add3", "v_alignbit", "v_lshlrev_b64", "v_readlane", "v_readfirstlane",
                       "v_bfe", "v_bfi", "v_and_or", "v_lshl_or", "v_ldexp", "v_frexp", "v_mbcnt", "v_cmpx")):
        return "half", HALF
    if ins.startswith(("v_fma", "v_fmac", "v_mad_f32", "v_mac_f32")):
        return "fma", FMA
    if re.search(r"\bs\d+\b|\bs\[\d+:\d+\]", ops) and not ins.startswith("v_mov"):
        return "sgpr-operand", HALF
    return "full", FULL


def main():
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lines = open(f).read().split("\n")[a - 1:b]
    cnt, tot = Counter(), Counter()
    other = Counter()
    for ln in lines:
        ln = ln.split(";")[0].strip()
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        parts = ln.split(None, 1)
        ins = parts[0]
        ops = parts[1] if len(parts) > 1 else ""
        if ins.startswith("v_"):
            k, c = cost(ins, ops)
            cnt[k] += 1
            tot[k] += c
        else:
            other[ins.split("_")[0] + "_" + (ins.split("_")[1] if "_" in ins else "")] += 1
    n = sum(cnt.values())
    print("VALU instructions: %d, estimated issue cost %.1f ns" % (n, sum(tot.values())))
    for k in sorted(cnt, key=lambda k: -tot[k]):
        print("  %-14s %4d  %.1f" % (k, cnt[k], tot[k]))
    print("non-VALU:", dict(other))


if __name__ == "__main__":
    main()
