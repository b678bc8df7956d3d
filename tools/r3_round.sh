#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 ) > $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
bash tools/pmc_calib.sh
bash tools/cu_mask_ab.sh
