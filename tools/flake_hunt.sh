#!/bin/bash
# does tests/test_gpu_extract.py::test_host_visible_batch_pipeline fail on THIS box, and with which test knob does it stop failing?
# usage: N0=100 KNOBS=1 bash tools/flake_hunt.sh   (KNOBS=1 also runs XFH_NO_RIDE / XFH_NO_NMS_HEAT / XFH_SELECT_LEGACY; those exist in the debug build
# only -- make -C xfeatslam_amd/csrc knobs -> libxfeat_hip_knobs.so, loaded here through XFEAT_HIP_LIB)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { # name, iterations, env...
  local name=$1 n=$2; shift 2; local f=0
  for i in $(seq 1 $n); do env "$@" timeout 120 python -m pytest tests/test_gpu_extract.py -m gpu -q -x -k "host_visible_batch_pipeline" > /tmp/o.log 2>&1; if grep -q failed /tmp/o.log; then f=$((f+1)); grep -E "records_equal" /tmp/o.log | head -3; fi; done
  echo "$name: $f failures in $n"
}
run default ${N0:-60} X=1
if [ -n "$KNOBS" ]; then
  K=$PWD/xfeatslam_amd/libxfeat_hip_knobs.so
  run no_ride ${N1:-60} XFEAT_HIP_LIB=$K XFH_NO_RIDE=1
  run no_nms_heat ${N1:-60} XFEAT_HIP_LIB=$K XFH_NO_NMS_HEAT=1
  run select_legacy ${N1:-60} XFEAT_HIP_LIB=$K XFH_SELECT_LEGACY=1
fi
