cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp xfeatslam_amd/libxfeat_hip.so /tmp/keep.so
for B in ${AB_BATCHES:-2 3 4 8}; do for v in A B; do
  cp tools/ab/$v.so xfeatslam_amd/libxfeat_hip.so
  python bench.py --steps 200 --no-legs --batch $B --streams 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', $B, round(d['value']), round(d['ms_per_step'],4))"
done; done
cp /tmp/keep.so xfeatslam_amd/libxfeat_hip.so
