#!/usr/bin/env python
"""tools/summarize_sq.py TAG [SUMMARY_DIR] -- the SQ counter logs of tools/pmc_sq.sh (lines "kernel counter value-per-launch")
as ONE machine-readable file, profiles/TAG_sq.json, which bench.py reads for `roofline.valu`:

  {workload: {kernel: {"valu_busy", "waiting", "valu_per_wave_layer", "salu_per_wave_layer", "waves_per_simd", ...}}}

valu_busy  = SQ_ACTIVE_INST_VALU x waves_per_simd / SQ_WAVE_CYCLES   (one VALU instruction at a time per SIMD)
waiting    = SQ_WAIT_ANY / SQ_WAVE_CYCLES
waves_per_simd = SQ_WAVES / (256 CUs x 4 SIMDs) for the persistent grids (one grid = the resident waves), at least 1
per wave-layer = instructions / (columns x lanes per column / 64 x 137 layers); lanes per column = the kernel's NGP
template argument, `columns` = 100 000 (the profiled batch)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOGS = {"sq_headline.log": "clear_homogeneous_ecckd32", "sq_tripleclouds.log": "tripleclouds_ecckd32", "sq_mcica.log": "mcica_ecckd32",
        "sq_rrtmg.log": "mcica_rrtmg", "sq_spartacus.log": "spartacus_ecckd32_sp"}
NCOL, NLEV = 100000, 137


def parse(path):
    k = {}
    for line in open(path):
        m = re.match(r"^(\S.*\S)\s+(SQ_[A-Z_]+)\s+([0-9.eE+-]+)\s*$", line)
        if m:
            k.setdefault(m.group(1).replace("ecrad::", ""), {})[m.group(2)] = float(m.group(3))
    return k


def summarize(counters):
    out = {}
    for kernel, c in counters.items():
        if "SQ_WAVE_CYCLES" not in c:
            continue
        waves = c.get("SQ_WAVES", 0.0)
        wps = max(1, int(round(waves / 1024.0))) if waves and waves <= 8 * 1024 else None
        rec = {"waiting": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], "valu_insts": c.get("SQ_INSTS_VALU"), "salu_insts": c.get("SQ_INSTS_SALU"),
               "vmem_insts": c.get("SQ_INSTS_VMEM"), "lds_insts": c.get("SQ_INSTS_LDS"), "waves_per_simd": wps}
        if wps and "SQ_ACTIVE_INST_VALU" in c:
            busy = c["SQ_ACTIVE_INST_VALU"] * wps / c["SQ_WAVE_CYCLES"]
            rec["valu_busy"] = busy if busy <= 1.0 else None      # (not a persistent grid: the occupancy guess does not hold)
        m = re.search(r"<[A-Za-z]+, (\d+)", kernel)
        if m and c.get("SQ_INSTS_VALU") and any(t in kernel for t in ("ica_kernel", "tc_kernel", "spartacus_sw", "spartacus_lw")):
            wave_layers = NCOL * int(m.group(1)) / 64.0 * NLEV
            rec["valu_per_wave_layer"] = c["SQ_INSTS_VALU"] / wave_layers
            rec["salu_per_wave_layer"] = c.get("SQ_INSTS_SALU", 0.0) / wave_layers
        out[kernel] = rec
    return out


def main():
    tag = sys.argv[1]
    d = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"{tag}_summaries")
    res = {}
    for log, workload in LOGS.items():
        p = os.path.join(d, log)
        if os.path.exists(p):
            s = summarize(parse(p))
            if s:
                res[workload] = s
    out = os.path.join(ROOT, "profiles", f"{tag}_sq.json")
    json.dump({"source": f"tools/pmc_sq.sh via tools/profile_round.sh {tag} (two rocprofv3 --pmc passes per workload)", "columns": NCOL,
               "workloads": res}, open(out, "w"), indent=1)
    for w, ks in res.items():
        for k, r in ks.items():
            print(f"{w:28s} {k:55s} busy {r.get('valu_busy') or 0:.2f} wait {r['waiting']:.2f} valu/wl {r.get('valu_per_wave_layer') or 0:.0f} "
                  f"salu/wl {r.get('salu_per_wave_layer') or 0:.0f}")


if __name__ == "__main__":
    main()
