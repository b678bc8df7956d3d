#!/bin/bash
# single-frame latency loop: extraction + select parity tests, per-kernel times at B = 1, host-API latency, timeline -> gpurun_out/lat_*.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_extract.py tests/test_gpu_select.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
( timeout 200 python tools/select_timing.py 2>&1 | grep gain ) | tee $O/lat_select.log
( timeout 200 python tools/latency_check.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -6 ) | tee $O/lat_host_api.log
( timeout 100 python tools/b1_loop.py; timeout 100 python tools/b1_loop.py 720 1280 ) | tee $O/lat_b1.log
bash tools/b1_timeline.sh 2>&1 | tail -40 | tee $O/lat_timeline.log
[ -x tools/probes/chain_probe ] && ( timeout 120 tools/probes/chain_probe | tee $O/lat_chain_probe.log | grep "G= 75 work= 600" )
