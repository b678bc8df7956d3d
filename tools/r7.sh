cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -4
python tools/gpu_stage_check.py --timing-only 2>&1 | grep -E "extract B|SELECT|NMS|HEADS|DESC|PREPROC|CONV_" | head -30
python bench.py --cpu-frames 0 --match-iters 20 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('default fps %.0f ms/step %.3f'%(d['value'],d['ms_per_step']))"
python bench.py --cpu-frames 0 --match-iters 20 --batch 1 --streams 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('B1S1 fps %.0f ms/step %.3f'%(d['value'],d['ms_per_step']))"
