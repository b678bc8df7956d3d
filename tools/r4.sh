cd $GRAFT_REPO_ROOT
for cfg in 0 1 2 3 4; do echo "cfg=$cfg"; XFH_CONV_CFG=$cfg python -m pytest tests/test_gpu_extract.py -m gpu -q -x -k "oracle and 480" 2>&1 | tail -1; for S in 1 3; do XFH_CONV_CFG=$cfg python bench.py --batch 16 --streams $S --steps 30 --cpu-frames 0 --match-iters 10 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(' S=%d fps %.0f ms/step %.3f conv %.1f us frac %.2f'%(d['config']['sub_batches_in_flight'],d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac']))"; done; done
