#!/bin/bash
# A/B of tools/ab/A.so and B.so on one box: single 720x1280 and 480x640 frames (tools/b1_loop.py) and small batches of VGA frames
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp xfeatslam_amd/libxfeat_hip.so /tmp/keep.so
for r in 1 2; do for v in A B; do cp tools/ab/$v.so xfeatslam_amd/libxfeat_hip.so; echo "== $v"; python tools/b1_loop.py 720 1280; python tools/b1_loop.py; done; done
cp /tmp/keep.so xfeatslam_amd/libxfeat_hip.so
AB_BATCHES="${AB_BATCHES:-2 4 8 16 32}" bash tools/ab_small_batches.sh 2>/dev/null
