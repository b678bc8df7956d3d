#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 ) > $O/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -30 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -5 $O/bench_default.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["in_timed_region"]["frac"], d["step_roofline"]["frac"])
print(json.dumps(d.get("host_visible"), indent=0)[:1500]); print(d.get("configs3_one_gpu")); print(d["single_frame"]); print(d["host_api"]); print(d["match"]["us_per_call"], d["match"]["roofline"]["frac"], d["match"]["roofline"]["steady_state"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("libtorch_ops")); print(d["parity"])
PY
