#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known-byte-count kernels (MI355X_MICROARCH.md, HBM): separate --pmc passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/calib_fetch $O/calib_write
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -o c -- python $R/tools/pmc_calib.py ) > $O/calib_fetch.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -o c -- python $R/tools/pmc_calib.py ) > $O/calib_write.log 2>&1
tail -1 $O/calib_fetch.log $O/calib_write.log
