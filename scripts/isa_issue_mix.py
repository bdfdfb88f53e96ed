"""Static issue-cost mix of a kernel's vector instructions, priced with the measured table of
profiles/r03_valu_issue_price_list.json (scripts/ubench/valu_issue2.hip):
  2 cycles/wave64 instruction per SIMD : v_fma / v_mul / v_add / v_sub / v_fmac / v_mov / v_add_u32 / v_and ... with VGPR, inline-constant
                                          or literal operands only
  4 cycles                             : ANY vector instruction with an SGPR (or vcc / exec as data) operand, DPP forms, v_cmp*, v_cndmask,
                                          v_max / v_min, shifts, v_cvt, v_pk_* (two results), f64, v_readlane / v_writelane
  8 cycles                             : v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos
  MFMA                                 : 32 (f32 16x16x4), 16 (bf16 16x16x32)
usage: python scripts/isa_issue_mix.py file.s 'substring of the kernel symbol' [more substrings]"""
import json, re, sys

TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
HALF_OPS = ("v_cmp", "v_cndmask", "v_max", "v_min", "v_med3", "v_lshl", "v_lshr", "v_ashr", "v_cvt", "v_pk_", "v_readlane", "v_writelane",
            "v_readfirstlane", "v_bfe", "v_bfi", "v_perm", "v_alignbit", "v_mbcnt", "v_mad_u64", "v_mul_lo", "v_mul_hi", "v_ldexp", "v_frexp",
            "v_rndne", "v_floor", "v_ceil", "v_trunc", "v_fract", "v_mad_u32", "v_mad_i32", "v_lshl_add", "v_add_lshl", "v_or3", "v_and_or",
            "v_xad", "v_accvgpr", "v_permlane", "v_swap", "v_div", "v_sad", "v_bcnt", "v_ffb", "v_not", "v_bfrev", "v_add3", "v_xor3")
SGPR = re.compile(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|ttmp\d+)(?![\w])")


def classify(line):
    parts = line.split(None, 1)
    op = parts[0]
    args = parts[1] if len(parts) > 1 else ""
    args = args.split(";")[0]
    if op.startswith("v_mfma"):
        return "mfma_f32" if op.endswith("_f32") and "bf16" not in op and "f16" not in op else "mfma_bf16"
    if op.startswith(TRANS):
        return "transcendental"
    if "_f64" in op:
        return "half_other"
    if "dpp" in op or " row_" in line or "quad_perm" in line or "row_newbcast" in line:
        return "half_dpp"
    if op.startswith(HALF_OPS):
        return "half_other"
    # destination of VOP3 compares etc. handled above; any SGPR-class source operand halves the rate
    srcs = args.split(",")[1:] if "," in args else []
    if any(SGPR.search(a) for a in srcs):
        return "half_sgpr_operand"
    return "full"


COST = {"full": 2, "half_sgpr_operand": 4, "half_dpp": 4, "half_other": 4, "transcendental": 8, "mfma_f32": 32, "mfma_bf16": 16}


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    text = open(path).read().splitlines()
    out = {}
    for pat in pats:
        start = next(i for i, l in enumerate(text) if l.endswith(":") is False and re.match(r"^_Z\S*:", l) and pat in l)
        end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm"))
        counts = {k: 0 for k in COST}
        salu = smem = vmem = lds = 0
        for l in text[start:end]:
            l = l.strip()
            if not l or l.startswith((";", ".")) or l.endswith(":"):
                continue
            op = l.split()[0]
            if op.startswith("v_"):
                counts[classify(l)] += 1
            elif op.startswith(("s_load", "s_buffer_load")):
                smem += 1
            elif op.startswith("s_"):
                salu += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                vmem += 1
            elif op.startswith("ds_"):
                lds += 1
        nv = sum(v for k, v in counts.items() if not k.startswith("mfma"))
        cyc = sum(counts[k] * COST[k] for k in counts if not k.startswith("mfma"))
        out[pat] = {"static_vector_instructions": counts, "valu_total": nv, "modelled_issue_cycles_per_valu": round(cyc / max(nv, 1), 3),
                    "share_half_rate_or_slower": round(1 - counts["full"] / max(nv, 1), 3), "salu": salu, "smem": smem, "vmem": vmem, "lds": lds}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
