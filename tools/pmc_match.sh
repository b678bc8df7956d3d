cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_[A-Z_0-9]+)\b" | sort -u | tr '\n' ' ' > $O/pmc_list.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $O/pmc1 -o m -- python $GRAFT_REPO_ROOT/tools/match_only.py 10 > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -o m -- python $GRAFT_REPO_ROOT/tools/match_only.py 10 > $O/pmc2.log 2>&1
ls -R $O/pmc1 | head; tail -3 $O/pmc1.log
