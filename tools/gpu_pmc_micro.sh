#!/bin/bash
# PMC passes over the conv micro-benchmark (instruction mix, TA / LDS pressure).  usage: tools/gpu_pmc_micro.sh PREFIX 'shape filter'
R=$PWD; P=$R/gpurun_out/$1; F="$2"
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/bench_conv.py $F"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d /tmp/pa -o a -- $B > /tmp/pa.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAIT_ANY --kernel-trace -d /tmp/pb -o b -- $B > /tmp/pb.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/pc -o c -- $B > /tmp/pc.log 2>&1
python $R/tools/pmc_raw_summary.py $(find /tmp/pa -name "*.db" | head -1) $(find /tmp/pb -name "*.db" | head -1) $(find /tmp/pc -name "*.db" | head -1) > ${P}_pmc_micro.md 2>&1
tail -n 2 /tmp/pa.log /tmp/pb.log /tmp/pc.log > ${P}_pmc_logs.txt 2>&1; true
