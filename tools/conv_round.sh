#!/bin/bash
# parity tests of the extraction + a serial per-kernel trace of one 256-frame step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_extract.py -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_extract.log
rm -rf $O/prof_serial
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_serial -o t -- python $R/bench.py --streams 1 --batch 256 --serial-branch --no-legs --steps 6 --warmup 2 ) > $O/prof_serial.log 2>&1
python tools/kstat.py $(ls $O/prof_serial/*kernel_trace.csv | head -1) > $O/kstat_serial.txt
( timeout 300 python bench.py --no-legs --steps 30 ) 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['step_roofline']['frac'])" > $O/bench_quick.txt
cat $O/pytest_extract.log; head -32 $O/kstat_serial.txt; cat $O/bench_quick.txt
