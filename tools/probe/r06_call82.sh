mkdir -p gpurun_out/r06
for i in 1 2 3; do
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06/suite_final_$i.log 2>&1
echo "run $i rc=$?"; tail -1 gpurun_out/r06/suite_final_$i.log
done
python bench.py --steps 20 --warmup 5 --no_secondary --no_legs --cpu_baseline off --pmc off --no_prof 2>/dev/null | tail -1 | cut -c1-200
