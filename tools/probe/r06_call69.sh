mkdir -p gpurun_out/r06
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_sc -o t -- python $R/bench.py --no_prof --no_secondary --no_legs --cpu_baseline off --pmc off --steps 6 --warmup 4 > /tmp/sc.log 2>&1
db=$(find /tmp/prof_sc -name "*.db" | head -1)
python $R/tools/stream_chains.py $db 3 14 6 > $R/gpurun_out/r06/stream_chains.md 2>&1
python $R/tools/prof_db_summary.py $db 2 70 > $R/gpurun_out/r06/kernel_stats_front.md 2>&1
tail -3 /tmp/sc.log
head -60 $R/gpurun_out/r06/stream_chains.md
