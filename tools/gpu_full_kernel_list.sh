#!/bin/bash
# Complete per-kernel listing of the headline step (every kernel, incl. ATen), plus the (kernel, workgroups) table.
# usage (GPU box): tools/gpu_full_kernel_list.sh PREFIX
R=$PWD; P=$R/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o t -- python $R/bench.py --no_prof --no_secondary --no_legs --cpu_baseline off --pmc off --steps 8 --warmup 4 > /tmp/kt.log 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $R/tools/prof_db_summary.py $DB 2 400 --by-grid > ${P}_kernel_stats_full.md 2>&1
python - "$DB" > ${P}_rocpd_kernels_columns.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
print([r[1] for r in db.execute('pragma table_info(kernels)').fetchall()])
PY
tail -n 3 /tmp/kt.log > ${P}_kt_log_tail.txt
