#!/bin/bash
# Round-end artefacts on the GPU box: the bench line, the rocprofv3 kernel-trace summary of the same command and the PMC
# HBM-traffic summary (two separate --pmc passes, no other trace domain).  Databases stay under /tmp on the box; only the
# summaries go to gpurun_out/.  usage: tools/gpu_profile.sh PREFIX
R=$PWD; P=$R/gpurun_out/$1
python bench.py --steps 20 --warmup 5 > ${P}_bench.json 2> ${P}_bench.err
tail -c 600 ${P}_bench.err > ${P}_bench.err.tail; rm -f ${P}_bench.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no_prof --no_secondary --no_legs --cpu_baseline off"
# kernel trace: the one-stream form first (every kernel alone on the chip: what bench.py's roofline pass prices and what its
# per-kernel durations agree with), then the default command (object front on its side stream: durations stretched by co-running)
SG_STREAM_GROUPS= timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o t -- $B --steps 8 --warmup 4 > /tmp/kt.log 2>&1
python $R/tools/prof_db_summary.py $(find /tmp/prof_kt -name "*.db" | head -1) > ${P}_kernel_stats.md 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt2 -o t -- $B --steps 8 --warmup 4 > /tmp/kt2.log 2>&1
python $R/tools/prof_db_summary.py $(find /tmp/prof_kt2 -name "*.db" | head -1) > ${P}_kernel_stats_side_streams.md 2>&1
python $R/tools/stream_chains.py $(find /tmp/prof_kt2 -name "*.db" | head -1) 3 14 6 > ${P}_stream_chains.md 2>&1
rm -f /tmp/launch.log
SG_STREAM_GROUPS= SG_LAUNCH_LOG=/tmp/launch.log SG_GRAPHS=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_f -o f -- $B --steps 2 --warmup 1 > /tmp/pf.log 2>&1
SG_STREAM_GROUPS= SG_GRAPHS=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_w -o w -- $B --steps 2 --warmup 1 > /tmp/pw.log 2>&1
python $R/tools/pmc_db_summary.py $(find /tmp/prof_f -name "*.db" | head -1) $(find /tmp/prof_w -name "*.db" | head -1) --json ${P}_pmc_traffic.json --launch-log /tmp/launch.log > ${P}_pmc_traffic.md 2>&1
for f in /tmp/kt.log /tmp/pf.log /tmp/pw.log; do tail -n 3 $f; done > ${P}_rocprof_logs.txt 2>&1; true
