#!/bin/bash
# single-frame latency loop: extraction parity tests, per-kernel times at B = 1, host-API latency
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_extract.py tests/test_gpu_select.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 200 python tools/select_timing.py 2>&1 | grep gain
timeout 200 python tools/latency_check.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -6
timeout 300 python bench.py --batch 1 --streams 1 --steps 300 --warmup 20 --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('B=1 back to back: %.1f us/frame' % (d['ms_per_step']*1e3))"
bash tools/b1_timeline.sh 2>&1 | tail -40
