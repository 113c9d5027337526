"""bench.py under torch.distributed.run with two ranks (the driver's N>1 launch line). On a box with >= 2 GPUs this is the real thing: one rank
per device, `nccl` (= RCCL) process group, no test hooks. On the 1-GPU test box both ranks are squeezed onto device 0 and rendezvous over gloo
(RCCL refuses two ranks on one device). Checks the contract of the result line: one JSON line from rank 0, n_gpus = 2, whole-job value = sum of
both ranks' work over the max elapsed."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_one_json_line(tmp_path):
    import torch
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env.update(SRBA_BENCH_DEVICE="0", SRBA_BENCH_BACKEND="gloo")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--n-kf", "1500", "--cpu-seconds", "0", "--cache-dir", str(tmp_path)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["capsules_per_gpu"] == 1499 and "replicas x2" in d["config"]["parallelism"]
    per_rank = d["config"]["lm_trials_per_step_per_gpu"]
    total_per_step = d["value"] * d["ms_per_step"] * 1e-3
    assert 1.6 * per_rank < total_per_step < 2.4 * per_rank      # two different maps of the same size
    assert d["roofline"]["frac"] > 0 and d["roofline"]["kernel_ms_samples"] == 3


@pytest.mark.gpu
def test_bench_eight_ranks_line_comes_out(tmp_path):
    """The driver's N = 8 launch line. With fewer than 8 devices the ranks share device 0 over gloo (a readiness check of the launch / harvest / aggregate logic, not a
    scaling measurement): one JSON line, n_gpus = 8, whole-job value = the eight maps' trials over the slowest rank's time, inside a few minutes."""
    import time
    import torch
    env = dict(os.environ)
    if torch.cuda.device_count() < 8:
        env.update(SRBA_BENCH_DEVICE="0", SRBA_BENCH_BACKEND="gloo")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--n-kf", "500", "--cpu-seconds", "0", "--cache-dir", str(tmp_path)]
    t0 = time.time(); p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT); dt = time.time() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and dt < 600
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and "replicas x8" in d["config"]["parallelism"] and d["cpu_baseline"] is None and "secondary_workloads" not in d
    per_rank = d["config"]["lm_trials_per_step_per_gpu"]; total_per_step = d["value"] * d["ms_per_step"] * 1e-3
    assert 6.4 * per_rank < total_per_step < 9.6 * per_rank


@pytest.mark.gpu
def test_sharded_map_sweep_two_ranks_equals_one_rank(tmp_path):
    """bench.py --workload sweep: ONE map sharded over two ranks (each a batch per round on its GPU, shared edges exchanged per round) must do the work of the one-rank sweep of the
    same map -- the same LM trials in total (the two schedules are the same sequential schedule) and the same whole-map squared error afterwards."""
    import torch
    def run(n):
        env = dict(os.environ)
        if torch.cuda.device_count() < n: env.update(SRBA_BENCH_DEVICE="0", SRBA_BENCH_BACKEND="gloo")
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "sweep", "--sweep-kf", "500", "--steps", "1", "--warmup", "0"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]; assert len(lines) == 1
        return json.loads(lines[0])
    one, two = run(1), run(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    c1, c2 = one["config"], two["config"]
    assert c1["rounds"] == c2["rounds"] and c1["windows_per_step"] == c2["windows_per_step"] == 499 and c2["windows_this_rank"] in (249, 250)
    assert c1["shared_edges"] == 0 and c2["shared_edges"] > 0 and c2["exchange_bytes_per_step"] > 0
    t1, t2 = one["value"] * one["ms_per_step"] * 1e-3, two["value"] * two["ms_per_step"] * 1e-3
    # The protocol itself is exact (tests/test_sweep.py: bit for bit over gloo with one numeric back-end). On the GPU the kernel variant a window runs on depends on the batch it
    # arrives in (k_lm_run / k_lm_run_lean / k_lm_run2 / the speculative single-capsule run are chosen by class counts) and the variants agree to rounding, not bit for bit: the
    # windows part at rounding-floor decisions -- the same map to 1e-6, trial counts within a few per cent
    assert abs(t1 - t2) < 0.1 * t1
    e1, e2 = c1["overall_sqr_error_after_1_sweeps"], c2["overall_sqr_error_after_1_sweeps"]
    assert abs(e1 - e2) <= 1e-6 * e1 and e1 < c1["overall_sqr_error_before"]
