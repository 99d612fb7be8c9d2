"""GPU: rehearsal of bench.py's multi-rank control flow on the one card of a gpurun box — two ranks launched by
torch.distributed.run exactly as the driver does, gloo instead of RCCL (RCCL refuses two ranks on one device), both on
cuda:0: rank-0 broadcast, per-step gradient all-reduce of the flat buffer, barrier-bracketed timing, max/sum
reductions of the results, ONE JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_json_line():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, TABGNN_DIST_BACKEND="gloo", TABGNN_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch-size", "512"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                              # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and d["value"] > 0
    # whole-job aggregate: both ranks' edges over the max-over-ranks time
    assert abs(d["value"] - 2 * d["config"]["edges_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02
    assert "cpu_baseline" not in d and "end_to_end" not in d            # rank 0 at N = 1 only
    # bucket-ready exchange: the head's and layer 1's gradient ranges left from autograd pre-hooks during the backward
    assert d["collective"]["overlapped"] and 0 < d["collective"]["overlapped_bytes_per_step"] < d["collective"]["bytes_per_step"]


def test_bench_world1_rccl_allreduce_runs_on_hardware():
    """The DEFAULT backend ("nccl" = RCCL) with one rank: `init_process_group`, the rank-0 broadcast and the bucketed
    all-reduce of the flat gradient buffer all execute on the MI355X (TABGNN_FORCE_ALLREDUCE=1 switches the collective
    path on at world size 1), and the step still trains: same JSON contract, `parallelism` dp1."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, TABGNN_FORCE_ALLREDUCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", TABGNN_LOG_COLLECTIVES="1")
    env.pop("TABGNN_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch-size", "512", "--no-cpu-baseline", "--no-e2e", "--no-extras"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    assert d["collective"]["backend"] == "nccl" and d["collective"]["all_reduce_calls"] >= 3       # one per step
    assert d["collective"]["bytes_per_step"] > 30e6                                               # 8.2 M fp32 parameters
    assert d["collective"]["overlapped"]                                                          # (RCCL calls issued inside the backward)


def test_overlapped_exchange_trains_like_the_post_backward_exchange():
    """Two ranks (gloo, one card), three steps of bench.py's model at B = 256 with TABGNN_NO_ALLREDUCE_OVERLAP=1 and
    without: the loss trajectories agree bit for bit (same sums in the same order; only WHEN a range is exchanged moves)."""
    res = []
    for off in ("1", "0"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, TABGNN_DIST_BACKEND="gloo", TABGNN_ONE_DEVICE="1", TABGNN_NO_ALLREDUCE_OVERLAP=off,
                   TABGNN_BENCH_LOSSES="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
               "--warmup", "1", "--batch-size", "256"]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        res.append(d)
    assert not res[0]["collective"]["overlapped"] and res[1]["collective"]["overlapped"]
    assert res[0]["losses"] == res[1]["losses"] and len(res[0]["losses"]) == 4, (res[0]["losses"], res[1]["losses"])
