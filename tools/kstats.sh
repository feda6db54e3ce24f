#!/bin/bash
# tools/kstats.sh WORKLOAD -- per-kernel average durations (rocprofv3 --kernel-trace --stats) of one bench workload
export TMPDIR=/tmp
out=gpurun_out/kstats_$1; rm -rf $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out -o k -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --no-host-mode --workload $1 > /dev/null 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("$out/*.db")[0]
for r in sqlite3.connect(db).execute("select name,total_calls,average from top_kernels"): print("%-72s %3d %8.2f ms" % (r[0][:72], r[1], r[2]/1e3))
PY
