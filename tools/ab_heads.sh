#!/bin/bash
# Same-box A/B of two COMMITS (each with its own bench.py and its own build): tools/ab/wt_r4 = a git worktree of 1a10945 (round 4's last commit, the
# driver's 31 431 frames/s) against the working tree (the driver's headline went 31 431 -> 30 506 between rounds 4 and 5 on two
# different boxes).  Prepare here:  git worktree add -f tools/ab/wt_r4 1a10945 && make -C tools/ab/wt_r4/xfeatslam_amd/csrc -j6
# Prints frames/s of `bench.py --steps 150 --warmup 20 --no-legs` for each run, alternating, six rounds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd)
for r in 1 2 3 4 5 6; do
  for v in wt_r4 HEAD; do
    d=$R; [ $v = wt_r4 ] && d=$R/tools/ab/wt_r4
    ( cd $d && python bench.py --steps 150 --warmup 20 --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms per step')" )
  done
done
