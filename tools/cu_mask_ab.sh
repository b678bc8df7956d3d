#!/bin/bash
# A/B of CU-masked streams (hipExtStreamCreateWithCUMask) for the four sub-batch ctx of the bench step: all CUs for everybody (the
# default) against fixed partitions, same box, same process order
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; : > $O/cu_mask_ab.txt
export XFEAT_HIP_LIB=${GRAFT_REPO_ROOT:-/root/repo}/xfeatslam_amd/libxfeat_hip_knobs.so   # XFH_CU_MASKS exists in the debug build only (make knobs)
run() { echo "== XFH_CU_MASKS=${XFH_CU_MASKS:-<unset>} $*" >> $O/cu_mask_ab.txt
  ( timeout 200 python bench.py --no-legs --steps 30 "$@" ) 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   frames/s %.0f  ms/step %.3f  step frac %.4f' % (d['value'], d['ms_per_step'], d['step_roofline']['frac']))" >> $O/cu_mask_ab.txt; }
unset XFH_CU_MASKS; run; run --streams 2 --batch 128
export XFH_CU_MASKS="0-127,128-255"; run; run --streams 2 --batch 128
export XFH_CU_MASKS="0-63,64-127,128-191,192-255"; run
export XFH_CU_MASKS="0-191,64-255"; run
export XFH_CU_MASKS="0-223,32-255,0-223,32-255"; run
unset XFH_CU_MASKS; run
cat $O/cu_mask_ab.txt
