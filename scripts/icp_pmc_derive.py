#!/usr/bin/env python3
"""Derived figures from a scripts/pmc_summary.py summary of the path-B NN kernels (k_count_inliers, k_icp_iter, k_find_corr):
VALU issue share, wait shares, L1 / L2 hit rates, per-query instruction counts.   usage: icp_pmc_derive.py <summary.txt> [queries per launch of k_count_inliers]"""
import re, sys
k, out = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"^(k_[a-z_0-9]+)", line)
    if m:
        k = m.group(1); out.setdefault(k, {}); continue
    m = re.match(r"^\s+([A-Za-z_0-9]+)(?: \(under the counters\))?\s+n=\s*(\d+)\s+mean\s+([0-9.]+)", line)
    if m and k:
        out[k][m.group(1)] = float(m.group(3))
for k in ("k_count_inliers", "k_icp_iter", "k_find_corr"):
    c = out.get(k)
    if not c:
        continue
    dur = c.get("DURATION_NS", 0) * 1e-9
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8.0 / max(c.get("DURATION_NS", 1), 1)            # GHz
    simd_cycles = 1024 * dur * clk * 1e9
    g = lambda n: c.get(n, float("nan"))
    print(k)
    print("   duration %.1f us (under the counters), clock %.2f GHz, waves %.0f" % (dur * 1e6, clk, g("SQ_WAVES")))
    print("   VALU instructions per wave %.0f; VALU issue share = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles) = %.2f" % (g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_ACTIVE_INST_VALU") * 4 / simd_cycles))
    print("   wave cycles: waiting for anything %.2f, waiting for an instruction to issue %.2f, any instruction active %.2f" % (
        g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
    # SQ_WAVE_CYCLES (like SQ_WAIT_* and SQ_ACTIVE_INST_*) counts QUAD-cycles (MI355X_MICROARCH.md): x 4 for resident waves per SIMD.  The round-5 files
    # (r05f_icp_pmc_*) printed this figure without the factor -- their "1.1 - 1.6 waves per SIMD" are 4.4 - 6.4, at a launch bound of 6.
    print("   occupancy = SQ_WAVE_CYCLES x 4 / (1024 SIMDs x cycles) = %.1f waves per SIMD" % (g("SQ_WAVE_CYCLES") * 4.0 / simd_cycles))
    print("   per wave: VMEM reads %.0f, LDS instructions %.0f (bank-conflict cycles / LDS instruction %.2f), SMEM %.0f" % (
        g("SQ_INSTS_VMEM_RD") / g("SQ_WAVES"), g("SQ_INSTS_LDS") / g("SQ_WAVES"), g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_INSTS_LDS"), 1), g("SQ_INSTS_SMEM") / g("SQ_WAVES")))
    print("   L1 (TCP): %.1f M accesses, %.1f M go on to L2 -> hit rate %.3f;  L2 (TCC): hit rate %.3f of %.1f M requests" % (
        g("TCP_TOTAL_CACHE_ACCESSES_sum") / 1e6, g("TCP_TCC_READ_REQ_sum") / 1e6, 1 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"),
        g("TCC_HIT_sum") / max(g("TCC_REQ_sum"), 1), g("TCC_REQ_sum") / 1e6))
    if "FETCH_SIZE" in c:
        print("   HBM: FETCH_SIZE %.1f MB (x2 on gfx950: %.1f MB), WRITE_SIZE %.1f MB per launch" % (g("FETCH_SIZE") / 1024, 2 * g("FETCH_SIZE") / 1024, g("WRITE_SIZE") / 1024))
