#!/bin/bash
# One GPU-box session: parity tests, smoke, bench at several batch sizes, torch/RCCL interop on
# one GPU, rocprofv3 kernel-trace (+ separate PMC passes for HBM traffic) of the bench command.
# Everything lands in gpurun_out/; tools/summarize_profiles.py turns it into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 300 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
for B in ${BENCH_BATCHES:-1 8 16 32 64 128 256}; do for S in 1 3; do
  ( timeout 300 python bench.py --batch $B --streams $S --steps 20 --cpu-frames 0 --match-iters 10 ) > $O/bench_B${B}_S$S.json 2> $O/bench_B${B}_S$S.err
done; done
( XFH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --cpu-frames 0 ) > $O/bench_dist1.json 2> $O/bench_dist1.err
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 20 --cpu-frames 0 ) > $O/bench_prof.json 2> $O/bench_prof.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 5 --warmup 2 --match-iters 10 --cpu-frames 0 ) > $O/pmc_fetch.json 2> $O/pmc_fetch.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --steps 5 --warmup 2 --match-iters 10 --cpu-frames 0 ) > $O/pmc_write.json 2> $O/pmc_write.err
rm -rf $O/prof_serial
( cd /tmp && XFH_AUX_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_serial -o bench -- python $R/bench.py --steps 10 --warmup 3 --match-iters 10 --cpu-frames 0 ) > $O/bench_prof_serial.json 2> $O/bench_prof_serial.err
ls $O/prof $O/pmc_fetch $O/pmc_write
echo round done
