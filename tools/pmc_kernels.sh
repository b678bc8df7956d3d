#!/bin/bash
# SQ counters per kernel of one serial bench step (two passes of <= 8 SQ counters); tools/pmc_kernels.py prints the ratios.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp; rm -rf $O/pmck1 $O/pmck2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $O/pmck1 -o m -- python $R/bench.py --steps 2 --warmup 1 ${PMC_LEGS:---no-legs} --serial-branch ${PMC_BENCH_ARGS} > $O/pmck1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmck2 -o m -- python $R/bench.py --steps 2 --warmup 1 ${PMC_LEGS:---no-legs} --serial-branch ${PMC_BENCH_ARGS} > $O/pmck2.log 2>&1
ls $O/pmck1 $O/pmck2
