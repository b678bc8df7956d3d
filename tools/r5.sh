cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -2
python bench.py --cpu-frames 4 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 1500 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
for S in 1 3; do XFH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --streams $S --cpu-frames 0 2>gpurun_out/dist_S$S.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('dist S=%d fps %.0f ms/step %.3f'%(d['config']['sub_batches_in_flight'],d['value'],d['ms_per_step']))"; tail -2 gpurun_out/dist_S$S.err; done
