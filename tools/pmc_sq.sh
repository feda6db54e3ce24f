#!/bin/bash
# tools/pmc_sq.sh TAG [bench args] -- SQ-level counters for the spectral kernels (own rocprofv3 pass)
TAG=${1:-sq}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU -d $OUT/sq -o sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-mode "$@" > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS SQ_WAVES SQ_BUSY_CYCLES -d $OUT/sq2 -o sq2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-mode "$@" > $OUT/sq2.log 2>&1
python - <<PY
import sqlite3,glob
for db in glob.glob("$OUT/sq*/*.db"):
    con=sqlite3.connect(db)
    rows=con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    for k,c,v,n in rows:
        if any(t in k for t in ('ica_kernel','tc_kernel','spartacus','taumol','generator','optics_dump')):
            print(k.split('(')[0].replace('void ecrad::',''), c, v/n)
PY
