#!/bin/bash
# A/B of two builds of libxfeat_hip.so on ONE box (box-to-box spread is ~2 %): tools/ab/A.so and tools/ab/B.so are swapped in
# alternately; prints frames/s of `bench.py --no-legs $AB_ARGS` for each run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp xfeatslam_amd/libxfeat_hip.so /tmp/keep.so
for r in 1 2 3; do for v in A B; do
  cp tools/ab/$v.so xfeatslam_amd/libxfeat_hip.so
  python bench.py --steps 40 --no-legs $AB_ARGS | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3))"
done; done
cp /tmp/keep.so xfeatslam_amd/libxfeat_hip.so
