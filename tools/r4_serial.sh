cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/prof_serial $O/prof_serial64
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_serial -o bench -- python $R/bench.py --streams 1 --batch 256 --serial-branch --only-match-leg --steps 10 --warmup 3 --match-iters 10 --cpu-frames 0 ) > $O/bench_prof_serial.json 2> $O/bench_prof_serial.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_serial64 -o bench -- python $R/bench.py --streams 1 --batch 64 --serial-branch --only-match-leg --steps 20 --warmup 3 --match-iters 10 --cpu-frames 0 ) > $O/bench_prof_serial64.json 2> $O/bench_prof_serial64.err
( timeout 300 python tools/host_batch_probe.py 512 "64x4,64x5,64x6,64x8,48x6,32x8" ) > $O/host_batch_probe.log 2>&1
cat $O/host_batch_probe.log
