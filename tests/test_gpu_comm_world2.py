"""BASELINE.json configs[3] with MORE THAN ONE RANK: a global batch is sharded frame i -> rank i mod R over R
processes, each with its own ctx, and the records reach the consumer through xfh_comm_* -- ncclAllGather, the ncclSend / ncclRecv
group (r * nb offsets, both roots), and the compact exchange (size all-gather, exact counts) -- i.e. every line of
csrc/comm.cpp that a world of one rank never executes.  The box has ONE GPU and real RCCL refuses two ranks on one device, so the
worker processes load the TEST-ONLY librccl stand-in of tests/stubs (same symbols; bytes move through a shared file) through the
dlopen search comm.cpp performs anyway.  Rank 0's gathered records, unsharded to global frame order, must equal the records a
single serial ctx produces for the same frames bit for bit, and the oracle's keypoints / descriptors."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, joined_desc_diff, kp_set, records_equal
from xfeatslam_amd import capi, dist as xd, weights as WT

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tests", "workers"))


def _run_world(world, n, out_dir):
    stub_dir = os.path.join(ROOT, "tests", "stubs")
    if not os.path.exists(os.path.join(stub_dir, "librccl.so.1")):
        subprocess.check_call(["make", "-C", stub_dir, "-s"])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ)
    env["XFH_RCCL_LIB"] = os.path.join(stub_dir, "librccl.so.1")           # explicit (comm.cpp: no fallback behind it); round 5 went through LD_LIBRARY_PATH
    worker = os.path.join(ROOT, "tests", "workers", "comm_world_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(n), str(out_dir)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r] if r < len(outs) else ''}"
    return outs


@pytest.mark.parametrize("world,n", [(2, 7), (3, 4)])
def test_sharded_extract_and_gathers_world_n(gpu_lib, oracle_mod, tmp_path, world, n):
    import comm_world_worker as Wk
    from xfeatslam_amd.extractor import Context
    outs = _run_world(world, n, tmp_path)
    assert any("rccl_stub: TEST-ONLY" in o for o in outs)
    NF, H, W = Wk.NF, Wk.H, Wk.W
    blob = WT.pack_blob(WT.make_synthetic(1234, 6.0))
    ref = Context(nfeatures=NF, max_height=H, max_width=W, max_batch=n, flags=capi.FLAG_SERIAL_BRANCH)
    ref.load_weights(blob)
    rec = ref.rec_bytes
    plan = xd.ShardPlan(n, 0, world)
    want = {}
    for step in range(Wk.STEPS):
        fr = Wk.step_frames(n, step)
        raw = np.empty(n * rec, np.uint8)
        capi.check(gpu_lib.xfh_extract_batch(ref.h, fr.ctypes.data, n, H, W, 0, 64, raw.ctypes.data), ref.h)
        want[step] = raw.reshape(n, rec)
    # all-gather: every rank holds the records of all ranks, rank-major; unsharded = the serial ctx' records in frame order
    for step in range(Wk.STEPS):
        views = [np.load(tmp_path / f"allgather_s{step}_r{r}.npy") for r in range(world)]
        for r in range(1, world):
            assert records_equal(ref, views[0], views[r], world * plan.slots), (step, r)
        got = plan.unshard_bytes(views[0], rec)
        for i in range(n):
            assert records_equal(ref, got[i], want[step][i], 1), (step, i)
    # gather to either root = the all-gather's bytes of the last step
    last = np.load(tmp_path / f"allgather_s{Wk.STEPS - 1}_r0.npy")
    for root in range(world):
        assert records_equal(ref, np.load(tmp_path / f"root{root}.npy"), last, world * plan.slots), root
    # compact gather of step 1 (it holds the frame without keypoints), unpacked on the host = the padded records
    comp = plan.unshard_bytes(np.load(tmp_path / "compact.npy"), rec)
    nvs = []
    for i in range(n):
        a, b = ref.parse_records(np.ascontiguousarray(comp[i]), 1)[0], ref.parse_records(np.ascontiguousarray(want[1][i]), 1)[0]
        assert a[2:4] == b[2:4] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), i
        nvs.append(b[2])
    assert nvs[n // 2] == 0 and sum(nvs) > 0
    sizes = np.load(tmp_path / "compact_sizes.npy")
    for r in range(world):
        rows = sum(ref.parse_records(np.ascontiguousarray(want[1][i]), 1)[0][2] for i in xd.ShardPlan(n, r, world).local)
        S = plan.slots
        assert sizes[r] == 256 + ((S * 16 + 255) & ~255) + ((rows * 28 + 255) & ~255) + rows * 256, r
    # and the serial ctx itself against the oracle (the parity anchor of the records that travelled)
    orc = oracle_mod.Oracle(blob)
    fr = Wk.step_frames(n, 0)
    for i in range(n):
        hk, hd, hnv, hmono, _ = ref.parse_records(np.ascontiguousarray(want[0][i]), 1)[0]
        ok, od, onv, omono = orc.extract(fr[i], NF, (0, 64))
        assert (hnv, hmono) == (onv, omono) and kp_set(hk) == kp_set(ok), i
        dd, _, ncommon = joined_desc_diff(ok, od, hk, hd)
        assert ncommon == hnv and dd <= 1e-4, (i, dd)
    ref.close()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_runs_with_n_ranks(gpu_lib, tmp_path, world):
    """bench.py --gpus N exactly as the driver launches it (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), all ranks on
    GPU 0 over the librccl stand-in, small sizes, N = 2 and N = 8 (the node the scaling run uses: its first real launch should be boring):
    the multi-rank line carries the headline, the other two gather forms, the extraction alone (no exchange) and the configs[3] leg (one
    1280x720 frame per rank + gather, all three forms)."""
    import json
    stub_dir = os.path.join(ROOT, "tests", "stubs")
    if not os.path.exists(os.path.join(stub_dir, "librccl.so.1")):
        subprocess.check_call(["make", "-C", stub_dir, "-s"])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env["XFH_RCCL_LIB"] = os.path.join(stub_dir, "librccl.so.1")           # explicit (comm.cpp: no fallback behind it)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "3", "--streams", "2",
                                       "--height", "96", "--width", "128", "--cpu-frames", "0", "--match-iters", "5", "--match-pairs", "2", "--host-steps", "3"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{outs[r][1][-3000:] if r < len(outs) else ''}"
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert all(outs[r][0].strip() == "" for r in range(1, world))                     # only rank 0 prints
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["value"] > 0 and line["config"]["frames_per_gpu_per_step"] == 6
    assert set(line["gather_forms"]) >= {"root", "compact"} and all(line["gather_forms"][f]["frames_per_s"] > 0 for f in ("root", "compact"))
    assert line["extract_only"]["frames_per_s"] > 0
    c3 = line["configs3"]
    assert c3["n_ranks"] == world and set(c3["per_gather_form"]) == {"allgather", "root", "compact"}
    assert all(v["frames_per_s"] > 0 and v["one_step_latency_ms"] > 0 for v in c3["per_gather_form"].values())
    assert line["roofline"]["frac"] > 0 and "in_timed_region" in line["roofline"] and line["host_visible"]["value"] > 0
    assert line["match"]["batched"]["pair_lists_equal_pair_by_pair_calls"] is True
    # the first real multi-GPU curve should explain itself: the partitioning is named, every multi-rank sub-key agrees on the world size, the flat scalars
    # the driver keeps carry the headline next to the extraction alone, and what rank 0 holds after the exchange equals a serial ctx's records
    cfg = line["config"]
    assert cfg["parallelism"] == f"frames x{world}"
    assert c3["n_ranks"] == world and line["exchange_check"]["ranks"] == world
    assert line["exchange_check"]["equal_to_serial_ctx"] is True and line["exchange_check"]["records_checked"] == 2 * world
    assert cfg["exchange_equal_serial_ctx"] is True
    assert cfg["extract_only_frames_per_s"] == line["extract_only"]["frames_per_s"] and 0 <= 1 - line["value"] / cfg["extract_only_frames_per_s"] == cfg["gather_cost_frac"] or cfg["gather_cost_frac"] < 0
    for form in ("root", "compact"):
        assert cfg[f"gather_{form}_frames_per_s"] == line["gather_forms"][form]["frames_per_s"]
    for form in ("allgather", "root", "compact"):
        assert cfg[f"configs3_{form}_frames_per_s"] > 0
    assert cfg["host_visible_frames_per_s"] == line["host_visible"]["value"] and cfg["match_us_per_call"] == line["match"]["us_per_call"]
    assert "gfx950" in cfg["library"]
    # round 6: the line says which librccl moved the records and in which form, within the first 20 config keys (what the driver's record keeps)
    first20 = list(cfg)[:20]
    for key in ("rccl", "gather_form", "extract_only_frames_per_s", "gather_cost_frac", "exchange_equal_serial_ctx", "step_mfma_frac", "match_us_per_call"):
        assert key in first20, (key, first20)
    for key in ("parity_keypoint_sets_equal", "parity_match_pairs_equal", "match_paced_30hz_us"):          # legs this small run switches off (--cpu-frames 0, few match calls)
        assert key in first20 or key not in cfg, (key, first20)
    assert "tests/stubs/librccl.so.1" in cfg["rccl"] and cfg["gather_form"] == "allgather"
    assert all(len(k) <= 32 for k in cfg), [k for k in cfg if len(k) > 32]
