#!/usr/bin/env python
"""Summarise a tools/profile.sh output directory (rocprofv3 rocpd .db files) into profiles/<tag>.md.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
come from separate --pmc passes, are in KiB, and are corrected with factors CALIBRATED on this
build's own access width (8 bytes per lane) using tools/hbm_calibrate (known byte counts)."""
import json
import os
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return list(con.execute(sql))
    finally:
        con.close()


def short(name):
    name = name.replace("ecrad::", "").replace("void ", "")
    return name.split("(")[0]


def main(tag):
    src = os.path.join("gpurun_out", tag)
    out = [f"# rocprofv3 summary `{tag}`", ""]
    bench = None
    bj = os.path.join(src, "bench.json")
    if os.path.exists(bj):
        for line in open(bj):
            if line.startswith("{"):
                bench = json.loads(line)
    if bench:
        out += ["## bench.py line (un-profiled run, same box)", "", "```json", json.dumps(bench), "```", ""]
    ncol = bench["config"]["columns_per_gpu_per_step"] if bench else 100000
    stats = q(os.path.join(src, "stats", "stats_results.db"),
              "select name, total_calls, total_duration, average, percentage from top_kernels")
    wl = bench["config"]["workload"] if bench else "clear_homogeneous_ecckd32"
    extra = "" if wl == "clear_homogeneous_ecckd32" else f" --workload {wl} --ncol {ncol}"
    out += [f"## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only --no-host-mode{extra}`", "",
            "(`--no-host-mode`: without the PCIe-inclusive measurements, whose column tiles and small batches are launches of the same kernels)", "",
            "| kernel | calls | total (ms) | average (ms) | % |", "|---|---|---|---|---|"]
    for n, c, t, a, p in stats:
        out.append(f"| `{short(n)}` | {c} | {t/1e3:.2f} | {a/1e3:.3f} | {p:.2f} |")
    out.append("")
    # calibration
    cal_f = dict((short(k), (v, n)) for k, v, n in q(os.path.join(src, "cal_fetch", "cal_results.db"),
                 "select kernel_name, sum(value), count(*) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name"))
    cal_w = dict((short(k), (v, n)) for k, v, n in q(os.path.join(src, "cal_write", "cal_results.db"),
                 "select kernel_name, sum(value), count(*) from counters_collection where counter_name='WRITE_SIZE' group by kernel_name"))
    known_kib = 2 * 1024 * 1024       # 2 GiB per launch
    f_corr = known_kib * cal_f["read8"][1] / cal_f["read8"][0]
    w_corr = known_kib * cal_w["write8"][1] / cal_w["write8"][0]
    out += ["## Counter calibration (tools/hbm_calibrate: 2 GiB streamed at 8 B/lane, known byte count)", "",
            f"* `read8`: FETCH_SIZE reports {cal_f['read8'][0]/cal_f['read8'][1]:.1f} KiB per launch for {known_kib} KiB read -> correction x{f_corr:.3f}",
            f"* `write8`: WRITE_SIZE reports {cal_w['write8'][0]/cal_w['write8'][1]:.1f} KiB per launch for {known_kib} KiB written -> correction x{w_corr:.3f}",
            ""]
    fetch = dict((k, (v, n)) for k, v, n in q(os.path.join(src, "fetch", "fetch_results.db"),
                 "select kernel_name, sum(value), count(*) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name"))
    write = dict((k, (v, n)) for k, v, n in q(os.path.join(src, "write", "write_results.db"),
                 "select kernel_name, sum(value), count(*) from counters_collection where counter_name='WRITE_SIZE' group by kernel_name"))
    avg = dict((n, a) for n, c, t, a, p in stats)
    out += ["## HBM traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, corrected)", "",
            f"| kernel | read GB | written GB | total GB | KB/column ({ncol} columns) | avg duration (ms) | HBM GB/s |",
            "|---|---|---|---|---|---|---|"]
    traffic, calls = {}, {}
    for k in fetch:
        if k not in write or "ecrad" not in k:
            continue
        r = fetch[k][0] / fetch[k][1] * f_corr * 1024 / 1e9
        w = write[k][0] / write[k][1] * w_corr * 1024 / 1e9
        ms = avg.get(k, 0) / 1e3      # top_kernels durations are in microseconds
        traffic[short(k)] = (r + w) * 1e9
        calls[short(k)] = fetch[k][1]
        out.append(f"| `{short(k)}` | {r:.2f} | {w:.2f} | {r+w:.2f} | {(r+w)*1e6/ncol:.1f} | {ms:.3f} | {((r+w)/(ms*1e-3)) if ms else 0:.0f} |")
    out.append("")
    os.makedirs("profiles", exist_ok=True)
    path = os.path.join("profiles", f"{tag}.md")
    open(path, "w").write("\n".join(out) + "\n")
    json.dump({"traffic_bytes_per_launch": traffic, "launches_profiled": calls, "fetch_correction": f_corr, "write_correction": w_corr,
               "columns": ncol, "workload": bench["config"]["workload"] if bench else None,
               # columns one launch of the solver kernels covers (a call may run as several column tiles)
               "columns_per_launch": ncol // max(1, bench["roofline"].get("column_tiles", 1)) if bench else ncol},
              open(os.path.join("profiles", f"{tag}_traffic.json"), "w"), indent=1)
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
