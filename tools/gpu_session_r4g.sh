#!/bin/bash
# round-4 GPU session G: kernel tables of the c5 / c4 legs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$R/gpurun_out
for leg in c5 c4; do
  rm -rf /tmp/prof_$leg
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$leg -o $leg -- python $R/tools/run_leg.py $leg 8 ) > $O/r4g_rocprof_$leg.log 2>&1
  DB=$(find /tmp/prof_$leg -name "*.db" | head -1)
  python tools/prof_db_summary.py "$DB" 2 70 > $O/r4g_${leg}_kernel_stats.md 2>&1
  python tools/run_leg.py $leg 10 > $O/r4g_${leg}_speed.txt 2>&1
done
grep -h "images/s" $O/r4g_c5_speed.txt $O/r4g_c4_speed.txt; head -30 $O/r4g_c5_kernel_stats.md | cut -c1-140; head -24 $O/r4g_c4_kernel_stats.md | cut -c1-140
