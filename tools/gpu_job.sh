#!/bin/bash
out=gpurun_out/r04_bt; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCP|TCC|TA|TD|SQ|SQC|GRBM)_[A-Za-z0-9_]+" | sort -u > $out/counters.txt
wc -l $out/counters.txt
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --headline-only --no-host-mode --workload mcica_rrtmg --ncol 100000"
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set -d $out/pmc_$tag -o pmc -- python bench.py $ARGS > $out/log_$tag.txt 2>&1
  python - <<PY
import sqlite3,glob
for db in glob.glob("$out/pmc_$tag/**/*.db", recursive=True):
    try:
        con=sqlite3.connect(db)
        tabs=[r[0] for r in con.execute("select name from sqlite_master where type='view' or type='table'")]
        t=[x for x in tabs if x.startswith('counters_collection')]
        rows=con.execute("select kernel_name, counter_name, sum(value), count(*) from %s group by kernel_name, counter_name" % t[0]).fetchall()
        for k,c,v,n in rows:
            if 'taumol' in k or 'StageD, 64' in k: print(k[:60], c, v, n)
    except Exception as e: print('ERR', db, e)
PY
done 2>&1 | tee $out/pmc.log
find gpurun_out -name "*.db" -delete
