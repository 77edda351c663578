mkdir -p gpurun_out/r06
python tools/probe/overlap_probe.py > gpurun_out/r06/overlap_probe.txt 2>&1
python tools/bench_gemm_classes.py --only Gres --bgemm --sweep w43_kfold=0,256,128 --iters 20 > gpurun_out/r06/gemm_kfold_nsub1.md 2>&1
python tools/bench_gemm_classes.py --only Gres --bgemm --opt w43_nsub=2 --sweep w43_kfold=0,256,128 --iters 20 > gpurun_out/r06/gemm_kfold_nsub2.md 2>&1
for kf in 0 256 128; do SG_W43_KFOLD=$kf python -m pytest tests/test_gpu_parity.py -q -x -k "winograd_f43_trunk" > gpurun_out/r06/f43_err_kfold$kf.log 2>&1; cp gpurun_out/winograd_f43_errors.json gpurun_out/r06/f43_errors_kfold$kf.json; done
SG_DIST_BACKEND=gloo SG_SHARE_GPU=1 SG_BENCH_STACKS=6 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 1 --warmup 3 --batch_per_gpu 2 --no_secondary --no_legs --cpu_baseline off --no_prof > gpurun_out/r06/dp2_stacks.out 2> gpurun_out/r06/dp2_stacks.err
tail -3 gpurun_out/r06/overlap_probe.txt
