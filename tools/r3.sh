cd $GRAFT_REPO_ROOT; O=gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for cfg in 0 1; do for S in 1 2 3; do echo "cfg=$cfg S=$S B=16"; XFH_CONV_CFG=$cfg python bench.py --batch 16 --streams $S --steps 30 --cpu-frames 0 --match-iters 10 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(' fps %.0f ms/step %.3f conv %.1f us frac %.2f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac']))"; done; done
for S in 2 4; do echo "cfg=0 S=$S B=8"; python bench.py --batch 8 --streams $S --steps 30 --cpu-frames 0 --match-iters 10 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(' fps %.0f ms/step %.3f'%(d['value'],d['ms_per_step']))"; done
XFH_CONV_CFG=1 python tools/gpu_stage_check.py --timing-only 2>&1 | grep -E "conv layer  (3|7|16|17)|extract B" | head -12
