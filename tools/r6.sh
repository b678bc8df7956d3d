cd $GRAFT_REPO_ROOT
for S in 1 3; do
XFH_FORCE_DIST=1 python -X faulthandler -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --streams $S --cpu-frames 0 > gpurun_out/d$S.out 2> gpurun_out/d$S.err; echo "S=$S rc=$?"; head -c 300 gpurun_out/d$S.out; echo; grep -v "Warning\|amdgpu.ids\|socket.cpp\|return func" gpurun_out/d$S.err | tail -20
done
