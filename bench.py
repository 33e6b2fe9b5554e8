#!/usr/bin/env python3
"""bench.py - MPC solves/s of the H1 trot workload (BASELINE.json configs[1]) on N MI355X.

One "step" = one MPC solve (task.info: 1 SQP iteration = linearise every shooting node, eliminate the equality
constraints, Riccati sweep, filter line search, step) of a batch of independent problems per GPU, inputs already resident
in HBM.  Ranks own disjoint problem slices (weak scaling: 256 problems per GPU); the only collective is the final
all-gather of the optimal trajectories over RCCL, which is inside the timed step for N > 1.
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (linearisation sweep) and, at N = 1,
`cpu_baseline` (the single-thread C++ oracle timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# 26 444 algorithmic bytes of one materialised node linearisation (788 in + 25 656 out).  The cost
# cross term P (nu x nx, 3 872 B) is structurally zero for this problem and is no longer written by the kernel (the buffer
# is zero-filled once at allocation), so it is not counted: 788 in + 21 784 out.
BYTES_PER_NODE_H1 = 26444 - 3872
BYTES_PER_NODE_24 = 30748 - 24 * 24 * 8      # nx = nu = 24 class, same rule
HBM_PEAK_GBS = 8000.0              # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="problems per GPU")
    ap.add_argument("--intervals", type=int, default=100, help="horizon in shooting intervals of dt = 0.015 s")
    ap.add_argument("--cpu-sample", type=int, default=256, help="problems solved by the CPU baseline (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not wrap kernels in HIP events")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class inside the timed region (default: the linearisation kernel only)")
    ap.add_argument("--robot", default="h1", choices=["h1", "openloong"],
                    help="h1 = the headline workload (nx = nu = 22); openloong = the 24/24 class of BASELINE.json configs[3] (informational)")
    ap.add_argument("--gait", default="trot", help="gait template of the workload (headline: trot)")
    ap.add_argument("--chunks", type=int, default=0, help="horizon chunks of the linearise/project || Riccati pipeline (0 = library default, 1 = off)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get("BPMPC_BENCH_FORCE_DIST") == "1"   # the latter exercises RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios

    itf = scenarios.interface(args.robot)
    B, NI = args.batch, args.intervals
    prob = scenarios.trot_problem(itf, batch=B, n_intervals=NI, offset=rank * B, gait=args.gait)
    max_nodes = NI + 16
    stream = torch.cuda.current_stream().cuda_stream
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=max_nodes, profile=(0 if args.no_profile else (1 if args.profile_all else 2)), device=local, stream=stream,
                           pipeline_chunks=args.chunks)
    lay = mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    n_nodes = lay["n_nodes_max"]
    kinds = mpc.read("g_kind")[:n_nodes]
    n_intermediate = int((kinds == 0).sum())
    nx, nu = itf.stateDim, itf.inputDim

    # result buffers owned by torch so that RCCL can gather them
    # (one flat block per rank, [x | u], so that a step needs a single collective)
    n_x, n_u = B * (max_nodes + 1) * nx, B * max_nodes * nu
    xu_loc = torch.empty(n_x + n_u, dtype=torch.float64, device="cuda")
    x_loc = xu_loc[:n_x].view(B, max_nodes + 1, nx)
    u_loc = xu_loc[n_x:].view(B, max_nodes, nu)
    if use_dist:
        xu_all = torch.empty(world * (n_x + n_u), dtype=torch.float64, device="cuda")

    pending = []           # the all-gather of step i runs on RCCL's stream while step i + 1 is being solved

    def drain():
        while pending:
            pending.pop().wait()      # the compute stream waits for the collective (x_loc / u_loc may be overwritten afterwards)

    def step():
        mpc.reset()        # device-side restore of the cold-start iterate: every step solves the same problems
        mpc.enqueue()      # 1 SQP iteration per problem, all on the GPU
        drain()
        mpc.export_trajectories(x_loc.data_ptr(), u_loc.data_ptr())
        if use_dist:
            pending.append(dist.all_gather_into_tensor(xu_all, xu_loc, async_op=True))

    def fence():
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    mpc.synchronize()
    for k in ("prepare", "linearize", "project_lu", "project", "riccati", "linesearch"):
        mpc.kernel_time(k, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    mpc.synchronize()

    t, x, u, _, stats = mpc.fetch()
    ok = sum(1 for s in stats if s.status == 0)
    lin_timed = mpc.kernel_time("linearize", reset=False)         # HIP events on the launch stream, over exactly the timed steps
    kt_steps = args.steps
    if not args.no_profile and not args.profile_all:
        # per-kernel breakdown from a short extra pass with every kernel class timed (each event pair costs 1-2 us of stream time,
        # so the timed region itself only carries the pair around the roofline kernel)
        mpc.set_profile(1)
        for k in ("linearize", "project_lu", "project", "riccati", "linesearch"):
            mpc.kernel_time(k, reset=True)
        extra = kt_steps = max(3, min(args.steps, 10))
        for _ in range(extra):
            step()
        fence()
        mpc.synchronize()
    # whole-job report: one 32-byte all-reduce of {merit, dynamics SSE, equality SSE, failures}, outside the timed region
    report = [sum(s.merit_after for s in stats), sum(s.dynamics_sse_after for s in stats), sum(s.equality_sse_after for s in stats), float(len(stats) - ok)]
    if use_dist:
        rep = torch.tensor(report, dtype=torch.float64, device="cuda")
        dist.all_reduce(rep, op=dist.ReduceOp.SUM)
        report = [float(v) for v in rep.tolist()]
        gathered_ok = bool(torch.equal(xu_all[rank * (n_x + n_u):(rank + 1) * (n_x + n_u)], xu_loc))     # the gathered block of this rank is its own result
    else:
        gathered_ok = True
    ktimes = {k: mpc.kernel_time(k, reset=False) for k in ("linearize", "project_lu", "project", "riccati", "linesearch")}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        lin_ms, lin_n = lin_timed
        roofline = None
        if lin_n > 0:
            # the horizon is linearised in `launches_per_step` chunk launches: an average launch covers that share of the nodes
            launches_per_step = lin_n / args.steps
            avg_s = 1e-3 * lin_ms / lin_n
            bytes_per_node = BYTES_PER_NODE_H1 if args.robot == "h1" else BYTES_PER_NODE_24
            alg_bytes = bytes_per_node * B * n_intermediate / launches_per_step
            achieved = alg_bytes / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "linearize_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if tj.get("batch") == B and tj.get("intervals") == NI and (args.robot, args.gait) == ("h1", "trot"):
                        traffic = tj.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"kernel": "k_linearize_fast<%d>" % (10 if args.robot == "h1" else 12), "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "avg_launch_us": round(1e6 * avg_s, 2),
                        "algorithmic_bytes_per_launch": round(alg_bytes), "launches_per_step": launches_per_step,
                        "node_linearizations_per_s": round(B * n_intermediate / launches_per_step / avg_s, 1)}
        out = {"metric": "MPC solves/s (%s, horizon=%d)" % ("H1" if args.robot == "h1" else args.robot, NI), "value": round(value, 2), "unit": "solves/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s %s, horizon=%d intervals (dt 0.015), batch=%d perturbed initial states per GPU, "
                                      "cold start, 1 SQP iteration (%s)" % ("Unitree H1" if args.robot == "h1" else "OpenLoong (nx = nu = 24)", args.gait, NI, B,
                                                                             "BASELINE.json configs[1]" if (args.robot, args.gait) == ("h1", "trot")
                                                                             else "not the headline workload"),
                          "global_batch": world * B, "shooting_nodes": n_nodes, "intermediate_nodes": n_intermediate, "nx": nx, "nu": nu,
                          "parallelism": "problem-sharded x%d, all-gather of trajectories overlapped with the next solve" % world, "accepted_steps": ok, "job_report": {"merit_sum": report[0], "dynamics_sse_sum": report[1], "equality_sse_sum": report[2], "failures": int(report[3]),
                                                                "gather_consistent": gathered_ok}},
               "ms_per_solve": round(ms_per_step / B, 6),
               "kernel_ms_per_step": {k: round(v[0] / max(1, kt_steps), 4) for k, v in ktimes.items()},
               "roofline": roofline}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(prob, min(args.cpu_sample, B), x, u, stats, args.robot)
        print(json.dumps(out))
    if use_dist:
        # the gathered block of this rank must equal its local result
        assert torch.equal(xu_all[rank * (n_x + n_u):(rank + 1) * (n_x + n_u)], xu_loc)
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(prob, sample, x_gpu, u_gpu, stats, robot="h1"):
    """Single-thread C++ oracle (a port of the same SQP iteration; the reference itself cannot be built here) on the
    first `sample` problems of the same workload; doubles as an end-to-end parity check of the timed run."""
    import numpy as np
    from tests import oracle_bridge as ob
    from oracle import reference_py as rp
    m, om = ob.model(robot), ob.oracle(robot)
    s = m["sqp"]
    pre = []
    for b in range(sample):
        nodes = ob.oracle_nodes(prob, b, robot=robot)
        xi, ui = rp.cold_start(m, nodes, prob["x0"][b])
        pre.append((nodes, xi, ui))
    worst = 0.0
    t0 = time.perf_counter()
    sols = [om.solve(nodes, prob["x0"][b], xi, ui, iterations=1, g_max=s["g_max"], g_min=s["g_min"], delta_tol=s["deltaTol"])
            for b, (nodes, xi, ui) in enumerate(pre)]
    dt = time.perf_counter() - t0
    for b, (xo, uo, _, _) in enumerate(sols):
        n = stats[b].n_nodes
        worst = max(worst, float(np.abs(x_gpu[b, :n + 1] - xo).max()))
    # for context, the reference's thread count (sqp.nThreads 3, task.info:68), here as three problems in flight (the C library
    # releases the GIL); `value` stays the single-thread figure
    threads3 = None
    if sample >= 6:
        from concurrent.futures import ThreadPoolExecutor
        sub = pre[:min(sample, 48)]
        with ThreadPoolExecutor(3) as pool:
            t1 = time.perf_counter()
            list(pool.map(lambda it: om.solve(it[1][0], prob["x0"][it[0]], it[1][1], it[1][2], iterations=1, g_max=s["g_max"], g_min=s["g_min"],
                                              delta_tol=s["deltaTol"]), enumerate(sub)))
            threads3 = round(len(sub) / (time.perf_counter() - t1), 3)
    try:
        cpu_name = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu_name = "unknown"
    return {"value": round(sample / dt, 3), "unit": "solves/s", "cores": 1, "kind": "port", "ms_per_solve": round(1e3 * dt / sample, 3),
            "sample": "%d of the same problems (horizon and SQP iteration count as on the GPU), solve only, "
                      "reference pre-pass excluded" % sample,
            "host_cpu": cpu_name, "host_cores": os.cpu_count(), "max_abs_x_diff_vs_gpu": worst, "value_3_threads": threads3}


if __name__ == "__main__":
    main()
