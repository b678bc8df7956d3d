#!/bin/bash
# Round-6 GPU session (one box): parity + soak tests, smoke, the default bench, the probes, serial kernel traces at B = 256 and B = 64 (the batch the
# bench line's roofline.frac is measured at), SQ counters per kernel at both, HBM traffic counters.  Everything lands in gpurun_out/; afterwards:
#   python tools/summarize_profiles.py r06; python tools/pmc_kernels.py r06; python tools/roofline_table.py r06 prof_serial; ... (tools/r6_post.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 ) > $O/pytest_gpu.log
echo "[$(date +%T)] done: ( timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -" >> $O/round_times.log
( timeout 400 python -m pytest tests/test_gpu_campaign.py tests/test_gpu_hazard.py -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" ) > $O/campaign_gpu.log
( timeout 250 tools/probes/mnn_tail_probe 300 ) > $O/mnn_tail_probe.log 2>&1
( timeout 100 tools/probes/dist_probe; python tools/dist_probe.py; python tools/dist_probe.py 1000 | tail -1 ) > $O/dist_probe.log 2>&1
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
echo "[$(date +%T)] done: ( timeout 120 python -c 'import __graft_entry__ as g; g.smok" >> $O/round_times.log
( timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
echo "[$(date +%T)] done: ( timeout 600 python bench.py ) > O/bench_default.json 2> " >> $O/round_times.log
( timeout 200 tools/probes/mnn_seg_probe 200 ) > $O/mnn_seg_probe.log 2>&1
echo "[$(date +%T)] done: ( timeout 200 tools/probes/mnn_seg_probe 200 ) > O/mnn_seg_" >> $O/round_times.log
( timeout 200 tools/probes/mnn_probe 200 ) > $O/mnn_probe.log 2>&1
echo "[$(date +%T)] done: ( timeout 200 tools/probes/mnn_probe 200 ) > O/mnn_probe.lo" >> $O/round_times.log
( timeout 100 tools/probes/pipe_probe ) > $O/pipe_probe.log 2>&1
echo "[$(date +%T)] done: ( timeout 100 tools/probes/pipe_probe ) > O/pipe_probe.log " >> $O/round_times.log
( timeout 300 bash tools/gemm_b2b.sh ) > /dev/null 2>&1          # -> gemm_b2b.md: the match GEMM back to back from an idle GPU, events and rocprofv3 side by side
echo "[$(date +%T)] done: ( timeout 300 bash tools/gemm_b2b.sh ) > /dev/null 2>&1     " >> $O/round_times.log
( timeout 100 python tools/b1_modes.py ) > $O/b1_modes.log 2>&1
( timeout 200 bash tools/b1_timeline.sh ) > $O/b1_timeline.log 2>&1
echo "[$(date +%T)] done: ( timeout 100 python tools/b1_modes.py ) > O/b1_modes.log 2" >> $O/round_times.log
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/prof_serial $O/prof_serial64
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 20 --match-warm-ms 10 --cpu-frames 0 ) > $O/bench_prof.json 2> $O/bench_prof.err
echo "[$(date +%T)] done: ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --" >> $O/round_times.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_serial -o bench -- python $R/bench.py --streams 1 --batch 256 --serial-branch --only-match-leg --steps 10 --warmup 3 --cpu-frames 0 ) > $O/bench_prof_serial.json 2> $O/bench_prof_serial.err
echo "[$(date +%T)] done: ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-f" >> $O/round_times.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_serial64 -o bench -- python $R/bench.py --streams 1 --batch 64 --serial-branch --only-match-leg --steps 20 --warmup 3 --cpu-frames 0 ) > $O/bench_prof_serial64.json 2> $O/bench_prof_serial64.err
echo "[$(date +%T)] done: ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-f" >> $O/round_times.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --match-iters 10 --match-warm-ms 0 --cpu-frames 0 --only-match-leg ) > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "[$(date +%T)] done: ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETC" >> $O/round_times.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --match-iters 10 --match-warm-ms 0 --cpu-frames 0 --only-match-leg ) > $O/pmc_write.json 2> $O/pmc_write.err
echo "[$(date +%T)] done: ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRIT" >> $O/round_times.log
bash tools/pmc_calib.sh > $O/pmc_calib.out 2>&1
echo "[$(date +%T)] done: bash tools/pmc_calib.sh > O/pmc_calib.out 2>&1" >> $O/round_times.log
PMC_LEGS="--only-match-leg --match-iters 10 --match-warm-ms 0 --cpu-frames 0" PMC_BENCH_ARGS="--streams 1 --batch 64" bash tools/pmc_kernels.sh > $O/pmck.out 2>&1
echo "[$(date +%T)] done: PMC_LEGS='--only-match-leg --match-iters 10 --match-warm-ms " >> $O/round_times.log
rm -rf $O/pmck1_B64 $O/pmck2_B64; mv $O/pmck1 $O/pmck1_B64; mv $O/pmck2 $O/pmck2_B64
PMC_BENCH_ARGS="--streams 1 --batch 256" bash tools/pmc_kernels.sh >> $O/pmck.out 2>&1
echo "[$(date +%T)] done: PMC_BENCH_ARGS='--streams 1 --batch 256' bash tools/pmc_kern" >> $O/round_times.log
rm -rf $O/pmck1_B256 $O/pmck2_B256; mv $O/pmck1 $O/pmck1_B256; mv $O/pmck2 $O/pmck2_B256
( timeout 200 python tools/host_batch_probe.py 512 "64x4,64x5,64x6,64x8,48x6,32x8" ) > $O/host_batch_probe.log 2>&1
echo "[$(date +%T)] done: ( timeout 200 python tools/host_batch_probe.py 512 '64x4,64x" >> $O/round_times.log
python tools/queue_view.py $(ls $O/prof/*kernel_trace.csv | head -1) > $O/queue_view.txt 2>&1
ls $O/prof $O/prof_serial $O/prof_serial64 $O/pmck1_B256 $O/pmck1_B64 | head -40
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -2
# post-processing here (the raw traces exceed what gpurun copies back): profiles/r06_* of THIS copy -> gpurun_out/profiles_r06/, raw traces dropped
bash tools/r6_post.sh > $O/r6_post.log 2>&1
mkdir -p $O/profiles_r06; cp profiles/r06_* profiles/pmc_traffic.json $O/profiles_r06/ 2>/dev/null
rm -rf $O/prof $O/prof_serial $O/prof_serial64 $O/pmc_fetch $O/pmc_write $O/pmck1_B64 $O/pmck2_B64 $O/pmck1_B256 $O/pmck2_B256 $O/prof_gemm $O/calib_fetch $O/calib_write $O/prof_b1
du -sh $O | tail -1
echo round done
