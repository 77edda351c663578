mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
SG_COND_FOLD=$v python tools/run_leg.py c5 8 2>/dev/null | tail -1
done
leg=c5
rm -rf gpurun_out/r06/eff_$leg gpurun_out/r06/eff_$leg.log
SG_GRAPHS=0 SG_LAUNCH_LOG=gpurun_out/r06/eff_$leg.log timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r06/eff_$leg -o eff -- python tools/run_leg.py $leg 4 > gpurun_out/r06/eff_$leg.out 2>&1
db=$(find gpurun_out/r06/eff_$leg -name "*.db" | head -1)
python tools/launch_eff.py $db gpurun_out/r06/eff_$leg.log 0 40 > gpurun_out/r06/launch_eff_${leg}_fold.md 2>&1
python tools/trace_list.py $db 70 > gpurun_out/r06/c5_kernel_stats_fold.md 2>&1
rm -rf gpurun_out/r06/eff_$leg
grep -n "true, true\|cond_\|TapNK<128, true" gpurun_out/r06/launch_eff_${leg}_fold.md gpurun_out/r06/c5_kernel_stats_fold.md | head -20
