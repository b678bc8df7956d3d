#!/bin/bash
# One GPU-box session: parity tests, smoke, the default bench, a sub-batch sweep, the RCCL path on one GPU, rocprofv3
# kernel-trace of the default bench command, separate PMC passes for HBM traffic, and a serial (one ctx, no second stream)
# kernel trace for the per-kernel roofline table.  Everything lands in gpurun_out/; tools/summarize_profiles.py and
# tools/roofline_table.py turn it into profiles/.   usage: TAG=r02 bash tools/gpu_round.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 ) > $O/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
: > $O/bench_sweep.txt
for cfg in "1 1" "1 8" "1 64" "1 256" "2 128" "3 85" "4 64" "4 128" "8 32"; do set -- $cfg
  ( timeout 300 python bench.py --streams $1 --batch $2 --steps 30 --no-legs ) 2> $O/bench_S$1_B$2.err | tail -1 > $O/bench_S$1_B$2.json
  python - "$O/bench_S$1_B$2.json" >> $O/bench_sweep.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]
print(c["sub_batches_in_flight"], c["frames_per_gpu_per_step"] // c["sub_batches_in_flight"], round(d["value"]), round(d["ms_per_step"], 3),
      round(d["step_roofline"]["frac"], 4), round(d["roofline"]["frac"], 4), round(d["roofline"]["in_timed_region"]["frac"], 4))
PY
done
( timeout 400 python bench.py --force-comm --steps 20 --cpu-frames 0 --only-match-leg ) 2> $O/bench_dist1.err | tail -1 > $O/bench_dist1.json
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/prof_serial
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 20 --cpu-frames 0 ) > $O/bench_prof.json 2> $O/bench_prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --match-iters 10 --cpu-frames 0 --only-match-leg ) > $O/pmc_fetch.json 2> $O/pmc_fetch.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --match-iters 10 --cpu-frames 0 --only-match-leg ) > $O/pmc_write.json 2> $O/pmc_write.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_serial -o bench -- python $R/bench.py --streams 1 --batch 256 --serial-branch --no-bn-leg --steps 10 --warmup 3 --match-iters 10 --cpu-frames 0 ) > $O/bench_prof_serial.json 2> $O/bench_prof_serial.err
bash tools/pmc_calib.sh > $O/pmc_calib.out 2>&1
( timeout 200 python tools/host_batch_probe.py 512 "64x4,64x3,32x4,128x2" ) > $O/host_batch_probe.log 2>&1
( timeout 100 python tools/pcie_probe.py ) > $O/pcie_probe.log 2>&1
python tools/queue_view.py $(ls $O/prof/*kernel_trace.csv | head -1) > $O/queue_view.txt 2>&1
ls $O/prof $O/pmc_fetch $O/pmc_write $O/prof_serial
echo round done
