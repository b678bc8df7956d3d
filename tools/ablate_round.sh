#!/bin/bash
# serial kernel trace of one 256-frame step for each prebuilt ablation variant of the library (gpurun_out/abl/libxfeat_hip_N.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cp xfeatslam_amd/libxfeat_hip.so /tmp/lib_orig.so
for n in $ABL; do
  cp tools/ab/libxfeat_hip_$n.so xfeatslam_amd/libxfeat_hip.so
  rm -rf $O/prof_abl
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof_abl -o t -- python $R/bench.py --streams 1 --batch 256 --serial-branch --no-legs --steps 4 --warmup 1 ) > $O/prof_abl.log 2>&1
  echo "== variant $n"; python tools/kstat.py $(ls $O/prof_abl/*kernel_trace.csv | head -1) | grep -E "$PAT"
done
cp /tmp/lib_orig.so xfeatslam_amd/libxfeat_hip.so
