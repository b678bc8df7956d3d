#!/bin/bash
# host-visible pipeline (in-order lanes): shapes x HW-queue count x second stream per lane; then a timeline of one shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_extract.py -m gpu -q -k "pipeline or lanes" 2>&1 | tail -5 ) > $O/pytest_pipe.log
: > $O/host_shapes.log
SH="64x3,64x4,64x6,32x4,32x8,43x6,128x2,128x3,256x2"
for q in default 8 16; do for ser in 0 1; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  if [ $ser = 1 ]; then export XFH_LANE_SERIAL=1 XFH_PROBE_SERIAL=1; else unset XFH_LANE_SERIAL XFH_PROBE_SERIAL; fi
  timeout 200 python tools/host_batch_probe.py 512 "$SH" >> $O/host_shapes.log 2>&1
done; done
unset XFH_LANE_SERIAL XFH_PROBE_SERIAL; export GPU_MAX_HW_QUEUES=8
rm -rf $O/tl_64x4
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl_64x4 -o t -- python $R/tools/host_batch_probe.py 512 64x4 ) > $O/tl_64x4.log 2>&1
cat $O/pytest_pipe.log $O/host_shapes.log
