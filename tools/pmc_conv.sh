cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp; rm -rf $O/pmcc1 $O/pmcc2
cd /tmp
XFH_CONV_CFG=4 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $O/pmcc1 -o m -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 3 --warmup 1 --cpu-frames 0 --match-iters 3 > $O/pmcc1.log 2>&1
XFH_CONV_CFG=4 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmcc2 -o m -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 3 --warmup 1 --cpu-frames 0 --match-iters 3 > $O/pmcc2.log 2>&1
ls $O/pmcc1 $O/pmcc2
