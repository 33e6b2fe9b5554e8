"""The N > 1 path of bench.py with the HIP solver in front of the collective, on a box with ONE GPU: `python bench.py --gpus 2`
launches itself (no external launcher), both ranks share device 0 (BPMPC_BENCH_ONE_DEVICE=1; RCCL refuses two ranks on one device,
so the all-gather goes over gloo with device tensors - sharding, explicit-stream ordering, overlap and consistency checks are the
code the 8-GPU RCCL run executes)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra, env_extra=None, timeout=900):
    env = dict(os.environ, BPMPC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-sample", "0"] + list(extra),
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_two_ranks_weak_and_strong():
    one = _bench("--gpus", "1", "--batch", "32", "--intervals", "30")
    two = _bench("--gpus", "2", "--batch", "32", "--intervals", "30")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["global_batch"] == 64
    rep = two["config"]["job_report"]
    assert rep["failures"] == 0 and rep["gather_consistent"]
    assert two["config"]["node_linearizations_per_step"] == 2 * one["config"]["node_linearizations_per_step"]
    # strong scaling: 50 problems in contiguous slices of 25; the job solves the problems 0..49 of the generator, whatever the rank count
    s1 = _bench("--gpus", "1", "--scaling", "strong", "--global-batch", "50", "--intervals", "30")
    s2 = _bench("--gpus", "2", "--scaling", "strong", "--global-batch", "50", "--intervals", "30")
    assert s2["scaling"] == "strong" and s2["config"]["global_batch"] == 50 and s2["config"]["problems_on_rank0"] == 25
    a, b = s1["config"]["job_report"], s2["config"]["job_report"]
    assert b["gather_consistent"] and b["failures"] == 0
    assert abs(a["merit_sum"] - b["merit_sum"]) <= 1e-9 * abs(a["merit_sum"]) and abs(a["dynamics_sse_sum"] - b["dynamics_sse_sum"]) <= 1e-9 * a["dynamics_sse_sum"]


@pytest.mark.timeout(1200)
def test_three_ranks_uneven_shards_and_rccl_single_rank():
    s3 = _bench("--gpus", "3", "--scaling", "strong", "--global-batch", "50", "--intervals", "30")     # shards 17 + 17 + 16: the short one is padded
    assert s3["config"]["job_report"]["gather_consistent"] and s3["config"]["problems_on_rank0"] == 17
    g3 = _bench("--gpus", "3", "--scaling", "strong", "--global-batch", "50", "--intervals", "30", "--gather", "root")   # the same job, gathered to rank 0 only
    assert g3["config"]["job_report"]["gather_consistent"] and "gather to rank 0" in g3["config"]["parallelism"]
    assert g3["config"]["job_report"]["merit_sum"] == s3["config"]["job_report"]["merit_sum"]
    r1 = _bench("--gpus", "1", "--batch", "16", "--intervals", "30", env_extra={"BPMPC_BENCH_FORCE_DIST": "1", "BPMPC_BENCH_ONE_DEVICE": "0"})   # RCCL itself, one rank
    assert r1["config"]["job_report"]["gather_consistent"] and "nccl" in r1["config"]["parallelism"]


@pytest.mark.timeout(900)
def test_bench_line_carries_the_write_roof_and_the_ddp_solver_has_a_line_of_its_own():
    """Round 6: `roofline.write_roof` (the lineariser's store pattern as a write-only stream, measured in the bench process: the calibration of `frac` for a
    kernel whose traffic is stores) and `bench.py --solver ddp` (the reference's second solver; `metric` says so, the kernel table has the roll-outs)."""
    one = _bench("--gpus", "1", "--batch", "32", "--intervals", "30", "--no-fused", env_extra={"BPMPC_BENCH_ONE_DEVICE": "0"})
    wr = one["roofline"]["write_roof"]
    assert wr["pattern_GBs"] and wr["pattern_GBs"] > 500 and 0 < wr["frac_of_write_roof"] < 1.5, wr
    assert one["config"]["gait_start"] == 0.0 and one["config"]["solver"] == "sqp"
    ddp = _bench("--gpus", "1", "--batch", "16", "--intervals", "30", "--solver", "ddp", "--no-fused", env_extra={"BPMPC_BENCH_ONE_DEVICE": "0"})
    assert ddp["metric"].startswith("DDP (ILQR) MPC solves/s") and ddp["config"]["solver"] == "ddp" and ddp["value"] > 0
    assert {"ddp_rollout", "ddp_search", "linearize", "riccati"} <= set(ddp["kernel_ms_per_step"])
    assert sum(ddp["config"]["step_lengths"].values()) == 16 and ddp["config"]["job_report"]["failures"] <= 1
