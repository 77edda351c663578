mkdir -p gpurun_out/r06j
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06j/smoke.log 2>&1
bash tools/gpu_profile.sh r06j/r06
bash tools/gpu_full_kernel_list.sh r06j/r06 2>/dev/null || true
