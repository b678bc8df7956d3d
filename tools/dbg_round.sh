#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; cp xfeatslam_amd/libxfeat_hip.so /tmp/lib_orig.so; cp tools/ab/libxfeat_hip_7.so xfeatslam_amd/libxfeat_hip.so
timeout 120 python bench.py --streams 1 --batch 256 --serial-branch --no-legs --steps 1 --warmup 0 2>&1 | grep "conv4" | head -80
cp /tmp/lib_orig.so xfeatslam_amd/libxfeat_hip.so
