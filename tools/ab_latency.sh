cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp xfeatslam_amd/libxfeat_hip.so /tmp/keep.so
for r in 1 2; do for v in tools/ab/v_*.so; do cp $v xfeatslam_amd/libxfeat_hip.so; echo "== $v"; timeout 200 python tools/latency_check.py 2>&1 | grep "xfh_extract"; done; done
cp /tmp/keep.so xfeatslam_amd/libxfeat_hip.so
