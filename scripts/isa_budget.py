#!/usr/bin/env python3
"""Static instruction mix of selected kernels from hipcc -S output (gfx950), grouped by class, next to the issue cost of each class measured by
scripts/ubench/valu_rates.hip (cycles per wave64 instruction per SIMD).  usage: isa_budget.py file.s kernel_substring ..."""
import collections, re, sys
COST = [("v_rcp_f64|v_rsq_f64|v_sqrt_f64", 16.2, "f64 transcendental"), ("v_rcp_f32|v_sqrt_f32|v_rsq_f32", 8.2, "f32 transcendental"),
        (r"v_\w+_f64", 4.2, "f64 arithmetic / compare / convert"), ("v_pk_", 4.2, "packed f32"), ("v_fma_f32|v_fmac_f32", 3.6, "f32 fma"),
        ("v_cmp|v_cvt|v_min|v_max|v_floor|v_mul_lo|v_mad_u|v_mul_hi|v_div_|v_cndmask|v_lshl_add_u64|v_mad_i", 4.2, "compare / convert / min-max / integer mul / division helpers"),
        ("v_add_f32|v_sub_f32|v_mul_f32|v_mov_b32|v_add_u32|v_sub_u32|v_and|v_or|v_xor|v_lshl|v_lshr|v_ashr|v_add_co|v_addc|v_sub_co|v_subrev|v_not|v_bfe|v_add3|v_lshl_or|v_and_or|v_or3|v_readlane|v_accvgpr", 2.3, "f32 add / mul, moves, 32-bit integer and logic")]
src = open(sys.argv[1]).read().splitlines()
for name in sys.argv[2:]:
    start = next(i for i, l in enumerate(src) if re.match(r"^_ZN.*%s.*:\s" % name, l + " "))
    end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end") or ".amdhsa_kernel" in src[i])
    ops = collections.Counter(l.split()[0] for l in src[start:end] if re.match(r"^\s+(v_|s_|global_|buffer_|ds_|flat_|scratch_)", l))
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    print("%s: %d instructions in the image (all paths, fast and exact), %d VALU, %d SALU, %d memory / LDS" % (
        name, sum(ops.values()), sum(valu.values()), sum(v for k, v in ops.items() if k.startswith("s_")),
        sum(v for k, v in ops.items() if not k.startswith(("v_", "s_")))))
    left = dict(valu)
    tot_c = 0.0
    for pat, cost, label in COST:
        hit = {k: v for k, v in left.items() if re.match(pat, k)}
        for k in hit:
            del left[k]
        n = sum(hit.values())
        tot_c += n * cost
        top = ", ".join("%s x%d" % kv for kv in sorted(hit.items(), key=lambda kv: -kv[1])[:6])
        print("   %-86s %4d x %4.1f cycles   (%s)" % (label, n, cost, top))
    if left:
        n = sum(left.values())
        tot_c += n * 4.2
        print("   %-86s %4d x  4.2 cycles   (%s)" % ("other VALU", n, ", ".join("%s x%d" % kv for kv in sorted(left.items(), key=lambda kv: -kv[1])[:8])))
    print("   -> %.0f issue cycles if every VALU instruction of the image ran once (the dynamic count per pixel / voxel is in the PMC summaries)" % tot_c)
