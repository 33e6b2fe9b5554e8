#!/usr/bin/env python3
"""bench.py - MPC solves/s of the H1 trot workload (BASELINE.json configs[1]) on N MI355X.

One "step" = one MPC solve (task.info: 1 SQP iteration = linearise every shooting node, eliminate the equality
constraints, Riccati sweep, filter line search, step) of a batch of independent problems per GPU, inputs already resident
in HBM.  Ranks own disjoint problem slices; the only collective is the all-gather of the optimal trajectories over RCCL
(bipedal_control_amd.distributed.TrajectoryGather), which is inside the timed step for N > 1 and overlaps with the next solve.

  python bench.py                      N = 1, configs[1]: H1 trot, horizon 100, batch 256
  python bench.py --gpus 8             spawns 8 ranks itself (one per GPU, torch.distributed.run on 127.0.0.1); weak scaling, 256 per GPU
  python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8     the same, launched from outside
  python bench.py --gpus 8 --scaling strong --global-batch 4096                configs[2]: 4096 problems split into contiguous slices
  python bench.py --gpus 8 --workload gait-sweep                               configs[4]: 8 gaits x 512 commands, horizon 150, gaits split over ranks
  python bench.py --robot g1 --gait standing_trot --batch 1024                 configs[3]: G1 walk (self-defined configuration)

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (linearisation sweep) and, at N = 1,
`cpu_baseline` (the single-thread C++ oracle timed on this box's host cores).
"""
import argparse
import json
import os
import re
import socket
import subprocess
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
# FP64 peak: half of the guide's FP32 vector peak (157.3 TFLOP/s = 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz) - a wave64 v_fma_f64
# issues in 4 cycles (16 lanes per clock and SIMD); v_mfma_f64_16x16x4_f64 runs at the same 32 FLOP/clk/SIMD (measured 64 cycles,
# tools/probes/mfma_f64_probe.hip), so vector and matrix FP64 peaks coincide
FP64_PEAK_TFLOPS = 78.6
FP64_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9      # FP64 VALU lane-operations per second at full issue rate
ROBOT_LABEL = {"h1": "Unitree H1", "openloong": "OpenLoong (nx = nu = 24)", "g1": "Unitree G1 (nx = nu = 24; self-defined configuration, not reference parity)",
               "hunter": "Hunter (the reference's configuration with positionErrorGain 20)",
               "h1:hard": "Unitree H1 with useHardFrictionConeConstraint (cones as inequality constraints, sqp.inequalityConstraintMu / Delta)",
               "hunter:hard": "Hunter with useHardFrictionConeConstraint"}
KERNEL_CLASSES = ("linearize", "project_lu", "project", "riccati", "linesearch")
DDP_KERNEL_CLASSES = ("linearize", "project_lu", "project", "riccati", "ddp_rollout", "ddp_search")      # --solver ddp: the line search is roll-outs + cost / decision


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle", type=int, default=32, help="untimed steps BEFORE the --warmup steps: after an idle period the first ~20 steps run 2 .. 3 %% slow "
                    "(clocks; measured with and without kernel events: 0.7805 against 0.7637 ms), whatever --warmup the caller picks")
    ap.add_argument("--batch", type=int, default=256, help="problems per GPU (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="strong: --global-batch problems split over the ranks")
    ap.add_argument("--global-batch", type=int, default=4096, help="total problems with --scaling strong (BASELINE.json configs[2])")
    ap.add_argument("--workload", default="trot", choices=["trot", "gait-sweep"],
                    help="trot: perturbed initial states on one gait (configs[1..3]); gait-sweep: 8 gaits x 512 velocity commands, "
                         "horizon 150, generated on the device, gaits split over the ranks (configs[4])")
    ap.add_argument("--intervals", type=int, default=None, help="horizon in shooting intervals of dt = 0.015 s (default 100; gait-sweep 150)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="problems solved by the CPU baseline (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not wrap kernels in HIP events")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class inside the timed region (default: the linearisation kernel only)")
    ap.add_argument("--robot", default="h1", choices=["h1", "openloong", "g1", "hunter", "h1:hard", "hunter:hard"],
                    help="h1 = the headline workload (nx = nu = 22); g1 = BASELINE.json configs[3] (nx = nu = 24, self-defined configuration); "
                         "openloong = the reference's own 12-joint robot")
    ap.add_argument("--gait", default=None, help="gait template of the trot workload (default: trot; g1: standing_trot = \"walk\")")
    ap.add_argument("--gait-start", type=float, default=0.0,
                    help="time at which the gait template of the trot workload is inserted (default 0 = SURVEY 8(d) config 2 to the letter: the template tiled "
                         "from t = 0, the solve starts on a mode switch; -1.225 = the rounds 1-4 scenario, t0 = 0 falls mid-swing)")
    ap.add_argument("--no-fused", action="store_true", help="skip the second timed region (fused solve mode)")
    ap.add_argument("--gather", default="auto", choices=["auto", "all", "root"],
                    help="collective of the solved trajectories per step: all-gather (every rank holds every block) or gather to rank 0; auto = "
                         "root for --scaling strong and --workload gait-sweep (the north-star's \"final gather\": one job, one owner of the "
                         "result), all for weak scaling (independent per-GPU batches, the contract's default line)")
    ap.add_argument("--gather-report", action="store_true",
                    help="report what RCCL chose for the per-solve gather (algorithm, protocol, transports, channels) in config.distributed.collective: "
                         "switches NCCL_DEBUG=INFO on before the process group exists and parses rank 0's log after the run (off by default: the log costs time)")
    ap.add_argument("--gather-algo", default=None, help="passed through as NCCL_ALGO (e.g. Ring, Tree) before the process group exists; recorded in the line")
    ap.add_argument("--gather-proto", default=None, help="passed through as NCCL_PROTO (LL, LL128, Simple); recorded in the line")
    ap.add_argument("--solver", default="sqp", choices=["sqp", "ddp"],
                    help="sqp: the reference's SqpMpc (the headline metric); ddp: its second solver, GaussNewtonDDP_MPC as configured (one ILQR iteration, "
                         "line search over eight policy roll-outs per problem): a line of its own, `metric` says DDP")
    ap.add_argument("--chunks", type=int, default=0, help="horizon chunks of the linearise/project || Riccati pipeline (0 = library default, 1 = off)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU, rendezvous on
    127.0.0.1.  Rank 0's JSON line passes through on stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def default_gather(scaling, workload):
    """`--gather auto`: what the job needs at the least - see the option's help."""
    return "root" if (scaling == "strong" or workload == "gait-sweep") else "all"


def main():
    args = parse_args()
    if args.gather == "auto":
        args.gather = default_gather(args.scaling, args.workload)
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # BPMPC_BENCH_ONE_DEVICE=1: every rank on device 0 (a 1-GPU box exercising the N > 1 code path; RCCL refuses two ranks on one
    # device, so that mode gathers over gloo - the transport differs, sharding / stream ordering / overlap logic are the same)
    one_device = os.environ.get("BPMPC_BENCH_ONE_DEVICE") == "1"
    device = 0 if one_device else local
    torch.cuda.set_device(device)
    use_dist = world > 1 or os.environ.get("BPMPC_BENCH_FORCE_DIST") == "1"   # the latter exercises RCCL with one rank
    backend = os.environ.get("BPMPC_BENCH_BACKEND", "gloo" if (one_device and world > 1) else "nccl")
    nccl_log = None
    if use_dist and backend == "nccl" and (args.gather_report or args.gather_algo or args.gather_proto):
        from bipedal_control_amd import distributed as bd0
        if args.gather_report:
            nccl_log = os.path.join(os.environ.get("TMPDIR", "/tmp"), "bpmpc_nccl_%d_rank%d.log" % (os.getpid(), rank))
            os.environ.update(bd0.nccl_debug_env(nccl_log, args.gather_algo, args.gather_proto))
        else:
            if args.gather_algo:
                os.environ["NCCL_ALGO"] = args.gather_algo
            if args.gather_proto:
                os.environ["NCCL_PROTO"] = args.gather_proto
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import bipedal_control_amd as bp
    from bipedal_control_amd import distributed as bd, scenarios

    itf = scenarios.interface(args.robot)
    nx, nu = itf.stateDim, itf.inputDim
    gait = args.gait or ("standing_trot" if args.robot == "g1" else "trot")
    sweep = args.workload == "gait-sweep"
    NI = args.intervals or (150 if sweep else 100)
    # ---- which problems does this rank own?
    if sweep:
        names, lib = scenarios.gait_library(itf)
        glo, ghi = bd.shard_range(len(lib), world, rank)          # one gait per GPU at N = 8
        cp = scenarios.gait_sweep_commands(itf, list(range(glo, ghi)), n_intervals=NI)
        per_gait = len(cp["cmd_vel"]) // max(1, ghi - glo) if ghi > glo else 512
        B = len(cp["x0"])
        total = len(lib) * per_gait
        capacity = bd.shard_capacity(len(lib), world) * per_gait
        scaling = "strong"
        max_nodes = NI + 2 * int(NI * scenarios.DT / 0.25 + 2) + 16     # shortest period of the library is 0.5 s: two events per 0.25 s at most
    else:
        if args.scaling == "strong":
            total = args.global_batch
            lo, hi = bd.shard_range(total, world, rank)
            capacity = bd.shard_capacity(total, world)
        else:
            total = world * args.batch
            lo, hi = rank * args.batch, (rank + 1) * args.batch
            capacity = args.batch
        B = hi - lo
        scaling = args.scaling
        prob = scenarios.trot_problem(itf, batch=B, n_intervals=NI, offset=lo, gait=gait, gait_start=args.gait_start)
        max_nodes = NI + 16
        if args.solver == "ddp":
            max_nodes = 2 * NI + 32      # a DDP solution lives on the time points of its roll-out (70 .. 130 at this horizon): the arrays must hold them
    if B < 1:
        raise SystemExit("rank %d owns no problems (more ranks than work units)" % rank)

    # The solver runs on an explicit torch stream: torch.distributed orders a collective only against torch's CURRENT stream, so
    # solve -> export -> all-gather -> next export are ordered by running everything under `with torch.cuda.stream(s)`.
    s = torch.cuda.Stream(device=device)
    # The timed region runs the MATERIALISED formulation (the lineariser writes the reference's complete per-node LQ model): that is the
    # sweep the north-star asks to profile and the one the roofline unit (algorithmic bytes per node linearisation) is defined on.
    # The engine's default, fused solve mode (the lineariser leaves only what the solve reads; identical solution bits) is timed in a
    # second region of the same length and reported as `fused` ("report both").
    ddp = args.solver == "ddp"
    if ddp and sweep:
        raise SystemExit("--solver ddp runs the trot workload")
    global KERNEL_CLASSES
    if ddp:
        KERNEL_CLASSES = DDP_KERNEL_CLASSES
    mpc = (bp.BatchedDdpMpc if ddp else bp.BatchedSqpMpc)(itf, max_batch=B, max_nodes=max_nodes, profile=(0 if args.no_profile else (1 if args.profile_all else 2)), device=device,
                                                          stream=s.cuda_stream, pipeline_chunks=args.chunks, materialize_lq=True)
    if sweep:
        lay = mpc.setup_commands(cp["t0"], cp["x0"], cp["gaits"], cp["gait_of_problem"], cp["gait_start"], cp["cmd_vel"], horizon=cp["horizon"])
    else:
        lay = mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    n_nodes = lay["n_nodes_max"]
    # intermediate (= linearised) nodes of this rank: per grid, times the problems on it
    kinds = mpc.read("g_kind").reshape(B, max_nodes)          # [grid][node]; grid index < n_grids <= B
    g_nodes = mpc.read("g_nodes")[:lay["n_grids"]].astype(int)
    p_grid = mpc.read("p_grid")[:B].astype(int)
    inter_of_grid = np.array([int((kinds[g, :g_nodes[g]] == 0).sum()) for g in range(lay["n_grids"])])
    n_intermediate_total = int(inter_of_grid[p_grid].sum())     # node linearisations per launch on this rank

    with torch.cuda.stream(s):
        gather = bd.TrajectoryGather(capacity, max_nodes, nx, nu, torch.device("cuda", device), mode=args.gather)
    if capacity != B:       # a short shard exports into the head of its block: x and u sub-blocks of B problems are contiguous only together with
        x_dst = torch.zeros(B * (max_nodes + 1) * nx, dtype=torch.float64, device="cuda")   # the padding, so go through a staging copy
        u_dst = torch.zeros(B * max_nodes * nu, dtype=torch.float64, device="cuda")

    def step():
        mpc.reset()        # device-side restore of the cold-start iterate: every step solves the same problems
        mpc.enqueue()      # 1 SQP iteration per problem, all on the GPU
        gather.drain()     # the previous all-gather has read the block
        if capacity == B:
            mpc.export_trajectories(gather.x_local.data_ptr(), gather.u_local.data_ptr())
        else:
            mpc.export_trajectories(x_dst.data_ptr(), u_dst.data_ptr())
            gather.x_local[:B].copy_(x_dst.view(B, max_nodes + 1, nx))
            gather.u_local[:B].copy_(u_dst.view(B, max_nodes, nu))
        gather.launch()

    def fence(barrier=True):
        """Opening bracket of a timed region: every rank idle and aligned.  Closing bracket (barrier=False): this rank's work drained - the
        clock is read per rank and the MAX over ranks is taken afterwards, so no collective sits inside the timed region beyond the
        per-step gather itself."""
        gather.drain()
        if use_dist and barrier:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(s):
        for _ in range(max(0, args.settle)):
            step()
        for _ in range(args.warmup):
            step()
        fence()
        mpc.synchronize()
        for k in ("prepare",) + KERNEL_CLASSES:
            mpc.kernel_time(k, reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence(barrier=False)
        elapsed = time.perf_counter() - t0
        rank_ms = elapsed / args.steps * 1e3                    # this rank's own clock: the line reports min / max over the ranks (stragglers)
        if use_dist:
            elapsed = float(bd.reduce_stats([elapsed], op="max")[0])
            rank_ms_min = -float(bd.reduce_stats([-rank_ms], op="max")[0])
        else:
            rank_ms_min = rank_ms
        mpc.synchronize()

        t, x, u, _, stats = mpc.fetch()
        ok = sum(1 for st in stats if st.status == 0)
        if ddp:     # status 1 = no step length passed the Armijo test, the baseline roll-out (new gains, no feedforward increment) is the solution: a valid solve
            ok = sum(1 for st in stats if st.status in (0, 1))
        lin_timed = mpc.kernel_time("linearize", reset=False)         # HIP events attached to the kernel's dispatches on the launch stream, over exactly the timed steps
        # error bar of `value`: the same region of `steps` steps four more times (the line's value stays the FIRST region, the contract's)
        spread = [elapsed / args.steps * 1e3]
        if world == 1:
            mpc.set_profile(0)
            for _ in range(4):
                fence()
                ts0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                fence(barrier=False)
                spread.append((time.perf_counter() - ts0) / args.steps * 1e3)
            mpc.set_profile(0 if args.no_profile else (1 if args.profile_all else 2))
        kt_steps = args.steps
        if not args.no_profile and not args.profile_all:
            # per-kernel breakdown from a short extra pass with every kernel class timed (each event pair costs 1-2 us of stream time,
            # so the timed region itself only carries the pair around the roofline kernel)
            mpc.set_profile(1)
            for k in KERNEL_CLASSES:
                mpc.kernel_time(k, reset=True)
            extra = kt_steps = max(3, min(args.steps, 10))
            for _ in range(extra):
                step()
            fence()
            mpc.synchronize()
        # whole-job report: one 40-byte all-reduce of {merit, dynamics SSE, equality SSE, failures, node
        # linearisations}, outside the timed region
        report = [sum(st.merit_after for st in stats), sum(st.dynamics_sse_after for st in stats), sum(st.equality_sse_after for st in stats),
                  float(len(stats) - ok), float(n_intermediate_total)]
        gathered_ok = gather.own_block_consistent()
        if use_dist:
            report = [float(v) for v in bd.reduce_stats(report).tolist()]
            # every rank holds everybody's result: rank 0's copy of the last rank's block must be what that rank computed
            # (checksum of the BIT PATTERNS, folded to 52 bits so that it travels exactly in a double: a floating-point sum of the same numbers
            #  depends on the reduction order torch picks for the alignment of the view - 1 ulp apart between the local block and its gathered copy)
            def bit_checksum(tv):
                w = tv.contiguous().view(torch.int64)
                return float(int(((w & 0xFFFFFFFF).sum() + (w >> 32).sum()).item()) % (1 << 52))
            probe = torch.zeros(2, dtype=torch.float64)
            if rank == world - 1:
                probe = torch.tensor([bit_checksum(gather.x_local), bit_checksum(gather.u_local)], dtype=torch.float64)
            probe = bd.reduce_stats(probe.tolist())
            if args.gather == "all" or rank == 0:       # (gather to root: only rank 0 holds the other ranks' blocks)
                xb, ub = gather.block(world - 1)
                gathered_ok = gathered_ok and bit_checksum(xb) == float(probe[0]) and bit_checksum(ub) == float(probe[1])
        ktimes = {k: mpc.kernel_time(k, reset=False) for k in KERNEL_CLASSES}
        # ---- second timed region: the fused solve mode, same problems, same number of steps, no kernel carries an event pair
        fused = None
        if not args.no_fused:
            mpc.set_materialize(False)
            mpc.set_profile(0)
            for _ in range(max(1, args.warmup)):
                step()
            fence()
            tf0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence(barrier=False)
            f_elapsed = time.perf_counter() - tf0
            if use_dist:
                f_elapsed = float(bd.reduce_stats([f_elapsed], op="max")[0])
            _, xf, uf, _, _ = mpc.fetch()
            same = bool(np.array_equal(xf, x) and np.array_equal(uf, u))
            if use_dist:
                same = bool(bd.reduce_stats([0.0 if same else 1.0])[0] == 0.0)
            fused = {"value": round(total * args.steps / f_elapsed, 2), "unit": "solves/s", "ms_per_step": round(1e3 * f_elapsed / args.steps, 4),
                     "solution_bits_equal_materialised": same, "hbm_bytes_per_step": None}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total * args.steps / elapsed
        lin_ms, lin_n = lin_timed
        roofline = None
        if lin_n > 0:
            # the horizon is linearised in `launches_per_step` chunk launches: an average launch covers that share of the nodes
            launches_per_step = lin_n / args.steps
            avg_s = 1e-3 * lin_ms / lin_n
            bytes_per_node = BYTES_PER_NODE_H1 if nx == 22 else BYTES_PER_NODE_24
            alg_bytes = bytes_per_node * n_intermediate_total / launches_per_step
            achieved = alg_bytes / avg_s / 1e9
            # HBM bytes seen by the PMC counters for this workload, when a committed pass of the same command exists (builder run, not
            # measured in this process: counters need rocprofv3 around the whole command)
            prof = None if ddp else committed_traffic(args.robot, gait, sweep, B, NI)      # (the committed counter passes are of the SQP command)
            traffic = prof["kernels"].get("linearize_materialised") if prof else None
            if fused is not None and prof:
                fused["hbm_bytes_per_step"] = prof.get("fused_hbm_bytes_per_step")
                fused["materialised_hbm_bytes_per_step"] = prof.get("materialised_hbm_bytes_per_step")
                fused["hbm_bytes_source"] = prof["source"]
            roofline = {"kernel": "k_linearize_fast<%d, true, true%s>" % (nx - 12, ", ILQR" if ddp else ""), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_stale": bool(prof and prof.get("stale")), "traffic_source": (prof["source"] if prof else None),
                        "avg_launch_us": round(1e6 * avg_s, 2),
                        "timing": "HIP events attached to the kernel's dispatch on its launch stream (hipExtLaunchKernelGGL start / stop events), every "
                                  "launch of the timed region: the kernel's duration, the figure the rocprofv3 kernel trace reports",
                        "algorithmic_bytes_per_launch": round(alg_bytes), "launches_per_step": launches_per_step,
                        "node_linearizations_per_s": round(n_intermediate_total / launches_per_step / avg_s, 1), "measured_on": "rank 0"}
            # calibration of `frac` for a kernel whose HBM traffic is 97 % stores: what a write-only stream of the same bytes achieves on this
            # device, measured here, outside the timed region (tools/probes/write_roof.hip: the lineariser's own store pattern without arithmetic)
            roofline["write_roof"] = write_roof(B, int(round(n_intermediate_total / max(1.0, launches_per_step) / B)), nx, achieved)
        headline = (args.robot, gait, sweep, NI) == ("h1", "trot", False, 100)
        if sweep:
            wl = "%s gait-library sweep: %d gaits (%s) x %d velocity commands, horizon=%d intervals (dt 0.015), generated on the device, " \
                 "gaits split over the ranks, cold start, 1 SQP iteration (BASELINE.json configs[4])" % (ROBOT_LABEL[args.robot], len(lib), ", ".join(names), per_gait, NI)
        else:
            wl = "%s %s (template tiled from t = %g), horizon=%d intervals (dt 0.015), %s perturbed initial states, cold start, 1 SQP iteration (%s)" % (
                ROBOT_LABEL[args.robot], gait, args.gait_start, NI, ("batch=%d per GPU" % args.batch) if scaling == "weak" else ("global batch %d in contiguous slices" % total),
                "the shape of BASELINE.json configs[1] through the reference's second solver, GaussNewtonDDP_MPC: 1 ILQR iteration, %d policy roll-outs per problem in the line search" % (
                    1 + sum(1 for k in range(16) if 0.5 ** k >= 1e-2)) if ddp else
                "BASELINE.json configs[1]" if headline and scaling == "weak" and args.batch == 256 else
                "BASELINE.json configs[2]" if headline and scaling == "strong" and total == 4096 else
                "BASELINE.json configs[3]" if (args.robot, gait, NI) == ("g1", "standing_trot", 100) else "not the headline workload")
        out = {"metric": ("DDP (ILQR) " if ddp else "") + "MPC solves/s (%s, horizon=%d)" % ({"h1": "H1", "g1": "G1", "h1:hard": "H1 hard cones"}.get(args.robot, args.robot), NI), "value": round(value, 2), "unit": "solves/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "settle_steps": max(0, args.settle), "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": wl, "gait_start": (None if sweep else args.gait_start), "solver": args.solver, "global_batch": total, "problems_on_rank0": B, "shooting_nodes": n_nodes,
                          "node_linearizations_per_step": int(report[4]), "nx": nx, "nu": nu,
                          "parallelism": "problem-sharded x%d, one %s of trajectories per solve (%s), overlapped with the next solve" % (
                              world, "all-gather" if args.gather == "all" else "gather to rank 0", backend if use_dist else "none at N = 1"),
                          "accepted_steps_rank0": ok,
                          "distributed": {"backend": (dist.get_backend() if use_dist else None), "world_size_seen": (dist.get_world_size() if use_dist else 1),
                                          "gather": args.gather if use_dist else None, "devices": ("one shared device" if one_device and world > 1 else "one per rank"),
                                          "rank_ms_per_step_min": round(rank_ms_min, 4), "rank_ms_per_step_max": round(elapsed / args.steps * 1e3, 4),
                                          "gather_algo_requested": args.gather_algo, "gather_proto_requested": args.gather_proto,
                                          "collective": collective_report(nccl_log, backend if use_dist else None, args.gather_report)},
                          "job_report": {"merit_sum": report[0], "dynamics_sse_sum": report[1], "equality_sse_sum": report[2], "failures": int(report[3]),
                                         "gather_consistent": bool(gathered_ok)}},
               "ms_per_solve": round(ms_per_step / max(1, total // world), 6),
               "kernel_ms_per_step": {k: round(v[0] / max(1, kt_steps), 4) for k, v in ktimes.items()},
               "timing_spread": {"regions": len(spread), "steps_per_region": args.steps, "ms_per_step_min": round(min(spread), 4),
                                 "ms_per_step_median": round(float(np.median(spread)), 4), "ms_per_step_max": round(max(spread), 4),
                                 "note": "region 1 carries the events of the roofline kernel and is `ms_per_step`; regions 2.. run without them"},
               "roofline": roofline, "fused": fused}
        kms = out["kernel_ms_per_step"]
        n_all_nodes = int(sum(g_nodes[p_grid])) if world == 1 else None
        if world == 1:
            out["roofline_fp64"] = roofline_fp64("h1" if nx == 22 else "g1", kms, n_intermediate_total, n_all_nodes, n_all_nodes,
                                                 applicable=headline and scaling == "weak" and args.batch == 256 and not ddp)
            nut_mean = float(mpc.read("nut").reshape(B, max_nodes)[:, :n_nodes][kinds[p_grid][:, :n_nodes] == 0].mean())
            out["roofline_all"] = roofline_all(nx, nu, nut_mean, kms, n_intermediate_total, n_all_nodes, out.get("roofline_fp64"),
                                               None if ddp else committed_traffic(args.robot, gait, sweep, B, NI))
        if roofline is not None:
            # what binds the roofline kernel: the largest of the fractions of the three roofs it could sit under, "latency" when none of them
            # is half used (then the kernel waits: low occupancy / dependent chains, see roofline_fp64.<kernel>.wave_time_split)
            fr = {"hbm": roofline["frac"]}
            e = (out.get("roofline_fp64") or {}).get("linearize") or {}
            if "issue_frac" in e:
                fr["fp64-issue"] = e["issue_frac"]
                fr["mfma"] = e["mfma_frac"]
            wr = (roofline.get("write_roof") or {}).get("frac_of_write_roof")
            if wr:      # the share of what a write-only stream with this kernel's store pattern reaches on this device (97 % of its traffic is stores)
                fr["hbm-write-roof"] = wr
            top = max(fr, key=fr.get)
            roofline["bound"] = ("hbm" if top.startswith("hbm") else top) if fr[top] >= 0.5 else "latency"
            roofline["bound_fracs"] = fr
            roofline["bound_note"] = ("`frac` stays achieved / HBM peak (the roof the north-star names); hbm-write-roof = achieved / the measured rate of a write-only stream with "
                                      "the kernel's own store pattern (write_roof); fp64-issue / mfma fractions use executed-instruction "
                                      "counts from profiles/%s (builder run) at this run's kernel time" % sq_counters_file())
        if ddp:
            out["config"]["step_lengths"] = {str(a): int(sum(1 for st in stats if st.step_size == a)) for a in sorted({st.step_size for st in stats}, reverse=True)}
            out["config"]["roll_out_points_mean"] = round(float(np.mean([st.n_nodes + 1 for st in stats])), 1)
        if world == 1 and args.cpu_sample > 0 and ddp:
            out["cpu_baseline"] = cpu_baseline_ddp(prob, min(args.cpu_sample, 2), t, x, stats, args.robot)
        elif world == 1 and args.cpu_sample > 0:
            if sweep:
                out["cpu_baseline"] = cpu_baseline_sweep(itf, cp, min(args.cpu_sample, 16), x, stats, args.robot)
            else:
                out["cpu_baseline"] = cpu_baseline(prob, min(args.cpu_sample, B), x, u, stats, args.robot)
                if ":" not in args.robot:       # (the lane-emulation build of the test tier loads the soft-cone model)
                    try:                    # a context figure must never cost the measured line (stale test library, missing compiler, ...)
                        out["cpu_baseline_analytic"] = cpu_baseline_analytic(prob, min(args.cpu_sample, B), x, stats, args.robot)
                    except Exception as e:
                        out["cpu_baseline_analytic"] = {"value": None, "why": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out), flush=True)
    if use_dist:
        assert gathered_ok, "gathered trajectories differ from the local result"
        dist.barrier()
        dist.destroy_process_group()


def sq_counters_file():
    """File name (under profiles/) of the committed SQ counter summary of the headline command: profiles/traffic_index.json["sq_counters"]."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic_index.json"))).get("sq_counters", "r03_sq_counters.json")
    except Exception:
        return "r03_sq_counters.json"


def collective_report(nccl_log, backend, requested):
    """config.distributed.collective: what the backend chose for the per-solve gather (rank 0's RCCL INFO log, bench.py --gather-report)."""
    if not requested:
        return {"reported": False, "why": "run with --gather-report (NCCL_DEBUG=INFO costs time: off in the default line)"}
    if backend != "nccl" or not nccl_log:
        return {"reported": False, "why": "backend %s: no RCCL algorithm to report (RCCL refuses two ranks on one device; the one-device dry runs gather over gloo)" % backend}
    try:
        from bipedal_control_amd.distributed import parse_nccl_debug
        with open(nccl_log) as f:
            rep = parse_nccl_debug(f.read())
        rep["reported"] = True
        rep["log"] = nccl_log
        return rep
    except Exception as e:
        return {"reported": False, "why": "%s: %s" % (type(e).__name__, e)}


def write_roof(batch, nodes_per_problem, nx, achieved_gbs):
    """What a write-only stream reaches on this device (tools/probes/write_roof.hip, built by bipedal_control_amd.build.build_probes): a plain fill
    and the lineariser's own store pattern - 4 nodes per wave, three (four at nx = 24) 8-byte store instructions per output row through its role
    pointers - with no arithmetic and the same bytes per launch.  `frac_of_write_roof` = the lineariser's achieved rate / the pattern's."""
    import ctypes
    lib = os.path.join(ROOT, "tools", "probes", "libwrite_roof.so")
    if not os.path.exists(lib):
        return {"pattern_GBs": None, "why": "tools/probes/libwrite_roof.so is not built (python -m bipedal_control_amd.build)"}
    try:
        h = ctypes.CDLL(lib)
        out = (ctypes.c_double * 9)()
        rc = h.write_roof_measure(ctypes.c_int(int(batch)), ctypes.c_int(int(nodes_per_problem)), ctypes.c_int(int(nx)), out)
        if rc != 0:
            return {"pattern_GBs": None, "why": "write_roof_measure returned %d" % rc}
        return {"pattern_GBs": round(out[2], 1), "pattern_free_occupancy_GBs": round(out[3], 1), "fill16_GBs": round(out[0], 1), "fill8_GBs": round(out[1], 1),
                "bytes_per_launch": int(out[4]), "pattern_us": round(1e3 * out[5], 2), "frac_of_write_roof": round(achieved_gbs / out[2], 4) if out[2] > 0 else None,
                "note": "store pattern of k_linearize_fast (materialised) without arithmetic at its geometry (256-thread workgroups, 76 KB of LDS); "
                        "measured in this process after the timed region; `frac` above stays achieved / 8 TB/s"}
    except Exception as e:        # a calibration figure must never cost the measured line
        return {"pattern_GBs": None, "why": "%s: %s" % (type(e).__name__, e)}


def committed_traffic(robot, gait, sweep, batch, intervals):
    """HBM bytes per launch from the committed PMC passes (tools/collect_profiles.sh, FETCH_SIZE x 2 + WRITE_SIZE as the guide prescribes)
    of the bench command with this workload, or None.  profiles/traffic_index.json maps a workload key to the summary file."""
    try:
        index = json.load(open(os.path.join(ROOT, "profiles", "traffic_index.json")))
        key = "%s:%s:%s:%d:%d" % (robot, "gait-sweep" if sweep else gait, "sweep" if sweep else "trot", batch, intervals)
        name = index.get(key)
        if not name:
            return None
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None
    # the counters were collected on some build of the kernels: they describe THIS library only if the sources are still the same
    try:
        from bipedal_control_amd.build import csrc_hash
        now = csrc_hash()
    except Exception:
        now = None
    if tj.get("csrc_hash") != now:
        return {"stale": True, "source": "profiles/%s was collected on other kernel sources (csrc_hash %s, now %s): not reported" % (name, str(tj.get("csrc_hash"))[:12], str(now)[:12]),
                "kernels": {}, "materialised_hbm_bytes_per_step": None, "fused_hbm_bytes_per_step": None}
    ks = tj.get("all_kernels", {})

    def pick(prefix, suffix=""):
        for k, v in ks.items():
            if k.startswith(prefix) and k.endswith(suffix):
                return v.get("hbm_bytes_per_launch")
        return None
    def pick_sum(prefix):          # the sweep is one launch (k_riccati_fast*) or two (k_riccati_wave*, k_riccati_rollout) per step
        vals = [v.get("hbm_bytes_per_launch") for k, v in ks.items() if k.startswith(prefix) and v.get("hbm_bytes_per_launch")]
        return sum(vals) if vals else None
    def pick_lin(mat):             # k_linearize_fast<NJ, MAT[, CHAIN]>: the second template argument says materialised / fused
        for k, v in ks.items():
            m = re.match(r"k_linearize_fast<\d+, (true|false)", k)
            if m and (m.group(1) == "true") == mat:
                return v.get("hbm_bytes_per_launch")
        return None
    kernels = {"linearize_materialised": pick_lin(True), "linearize_fused": pick_lin(False),
               "project_lu": pick("k_project_lu"), "project": pick("k_project_fast"), "riccati": pick_sum("k_riccati"), "linesearch": pick("k_trial_fast")}
    return {"stale": False, "source": "profiles/%s (builder run of the same command under rocprofv3 --pmc; not measured in this process)" % name, "kernels": kernels,
            "materialised_hbm_bytes_per_step": tj.get("materialised_hbm_bytes_per_step"), "fused_hbm_bytes_per_step": tj.get("fused_hbm_bytes_per_step")}


def roofline_all(nx, nu, nut_mean, kernel_ms, n_lin_nodes, n_nodes_total, fp64, prof):
    """One roofline entry per hot kernel class of the MATERIALISED step (DESIGN.md section 4 defines the units):
      algorithmic bytes per unit = what the stage must read and write if every operand moved exactly once, in the reference's own data
      model (dense node matrices; nut = mean number of reduced inputs of this workload):
        linearize   node inputs + the materialised LQ model without the structurally zero cost cross term
        project_lu  C, D, e (16 rows each) in; Px, Pu, Pe out
        project     A, B, b, Q, R, q, r, Px, Pu, Pe in; projected At, Bt, bt, Qt, Rt, Pt, qt, rt out (Bt, Rt, Pt, rt at nut columns / rows)
        riccati     projected model + Px, Pu, Pe in; K, dx, du out
        linesearch  x, u, dx, du, x_next, dx_next, xref, swing references in; 3 sums out
      traffic_ratio = PMC bytes of the committed pass / algorithmic bytes (> 1: re-reads, padding, scratch; < 1: the kernel exchanges its
      operands PACKED - joint rows of [Px | Pe | Pu] only, projected model in 16-column blocks up to nx + 1 + nut - and touches fewer bytes
      than the dense reference unit).  hbm_util = counter bytes / kernel time / 8 TB/s: the share of the HBM roof the kernel really uses.
      frac = algorithmic bytes / kernel time / 8 TB/s for every kernel (one definition, comparable across kernels and rounds); where
      traffic_ratio < 1 the dense unit credits bytes the kernel never touches: hbm_util and packed_bytes_per_unit (counter bytes / units,
      from the committed counter pass - not measured in this process) say what it really moves.  bound: largest of hbm_util (hbm frac without counters) / fp64-issue / mfma if
      >= 0.5, else latency."""
    d = 8.0
    proj_model = nx * nx + nx * nut_mean + nx + nx * nx + nx + nut_mean * nut_mean + nut_mean * nx + nut_mean
    pxe = nu * nx + nu * nu + nu
    units = {
        "linearize": (n_lin_nodes, float(BYTES_PER_NODE_H1 if nx == 22 else BYTES_PER_NODE_24), "linearize_materialised"),
        "project_lu": (n_lin_nodes, d * (16 * (nx + nu + 1) + pxe), "project_lu"),
        "project": (n_lin_nodes, d * (nx * nx + nx * nu + nx + nx * nx + nu * nu + nx + nu + pxe + proj_model), "project"),
        "riccati": (n_nodes_total, d * (proj_model + pxe + nu * nx + nx + nu), "riccati"),
        "linesearch": (n_nodes_total, d * (3 * nx + 2 * nu + 2 * nx + 8 + 3), "linesearch"),
    }
    out = {"peak": HBM_PEAK_GBS, "unit": "GB/s", "mean_reduced_inputs": round(nut_mean, 3), "traffic_source": prof["source"] if prof else None}
    dominant = max((k for k in units if kernel_ms.get(k)), key=lambda k: kernel_ms[k], default=None)
    for cls, (n_units, bytes_per_unit, pkey) in units.items():
        ms = kernel_ms.get(cls)
        if not ms or not n_units:
            continue
        alg = bytes_per_unit * n_units
        tr = (prof or {}).get("kernels", {}).get(pkey)
        util = (tr / (1e-3 * ms) / 1e9 / HBM_PEAK_GBS) if tr else None
        packed = bool(tr) and tr < alg
        ach = alg / (1e-3 * ms) / 1e9          # ONE definition of frac for every kernel and round: algorithmic bytes / time
        fr = {"hbm": round(util if util is not None else ach / HBM_PEAK_GBS, 4)}
        e = (fp64 or {}).get(cls) or {}
        if "issue_frac" in e:
            fr["fp64-issue"], fr["mfma"] = e["issue_frac"], e["mfma_frac"]
        top = max(fr, key=fr.get)
        out[cls] = {"ms": ms, "algorithmic_bytes_per_unit": round(bytes_per_unit), "units": int(n_units), "achieved": round(ach, 1),
                    "frac": round(ach / HBM_PEAK_GBS, 4), "frac_basis": "algorithmic bytes",
                    "bound_note": "hbm_util / traffic / packed_bytes_per_unit come from the committed counter pass (builder run, not measured in this process)" if tr else None,
                    "packed_bytes_per_unit": round(tr / n_units) if packed else None, "hbm_util": round(util, 4) if util is not None else None,
                    "bound": top if fr[top] >= 0.5 else "latency", "bound_fracs": fr, "traffic": tr,
                    "traffic_ratio": round(tr / alg, 3) if tr else None, "dominant": cls == dominant}
    return out


def roofline_fp64(robot, kernel_ms, n_lin_nodes, n_nodes_total, n_stages_total, applicable):
    """FP64 view of the five hot kernels: none of them is near the HBM roof, so what binds them?
      restatement_flops    operation count of the CPU restatement for the work of one launch (profiles/flop_counts.json: instrumented
                           scalar for the lineariser, standard dense-algebra counts behind it) - NOT a bound for the lineariser: the
                           restatement differentiates in forward mode over 44 directions, the kernel uses analytic derivatives
      executed             from the committed SQ counter passes of this same command (profiles/r03_sq_counters.json): VALU
                           lane-operations (SQ_INSTS_VALU x 64) and FP64 MFMAs (x 2048 flop) per launch
      issue_frac           executed lane-operations / launch time / the FP64 issue rate (16 lanes per clock and SIMD): the fraction of
                           the vector pipe's FP64 issue slots the kernel fills - an upper bound of its FP64 utilisation, since every
                           VALU instruction is counted as one FP64 slot
      mfma_frac            MFMA flop / launch time / 78.6 TFLOP/s
    Durations are this run's HIP-event times; the counter file only applies to the headline workload."""
    cpath, fpath = os.path.join(ROOT, "profiles", sq_counters_file()), os.path.join(ROOT, "profiles", "flop_counts.json")
    if not (os.path.exists(cpath) and os.path.exists(fpath)):
        return None
    try:
        counters, flops = json.load(open(cpath)), json.load(open(fpath))["robots"][robot]
    except Exception:
        return None
    classes = {"linearize": ("k_linearize_fast", flops["node_linearization"] * n_lin_nodes),
               "project_lu": ("k_project_lu", flops["lu_projection"] * n_lin_nodes),
               "project": ("k_project_fast", flops["change_of_variables"] * n_lin_nodes),
               "riccati": ("k_riccati_fast", flops["riccati_stage"] * n_stages_total),
               "linesearch": ("k_trial_fast", 2 * flops["flow_map"] * n_nodes_total + flops["ee_kinematics"] * n_nodes_total)}
    out = {"peak_tflops": FP64_PEAK_TFLOPS, "note": "restatement_flops = CPU restatement's operation count (forward-mode AD for the lineariser: not a bound); "
                                                     "issue_frac = executed VALU lane-ops / time / FP64 issue rate; counters from profiles/" + sq_counters_file()}
    for cls, (prefix, rflops) in classes.items():
        ms = kernel_ms.get(cls)
        if not ms:
            continue
        e = {"ms": ms, "restatement_flops": int(rflops), "restatement_tflops": round(rflops / (1e-3 * ms) / 1e12, 2)}
        names = sorted((k for k in counters if k.startswith(prefix)), key=lambda k: (re.match(r"k_linearize_fast<\d+, false", k) is not None, k))     # the timed lineariser is <NJ, true, ..>
        ck = counters[names[0]] if (applicable and names) else None
        if ck:
            lane_ops = ck.get("SQ_INSTS_VALU", 0.0) * 64.0
            mfma = ck.get("SQ_INSTS_VALU_MFMA_F64", 0.0)
            e.update({"executed_valu_lane_ops": int(lane_ops), "executed_mfma_f64": int(mfma),
                      "issue_frac": round(lane_ops / (1e-3 * ms) / FP64_LANE_OPS_PER_S, 4),
                      "mfma_frac": round(mfma * 2048.0 / (1e-3 * ms) / (FP64_PEAK_TFLOPS * 1e12), 4),
                      "wave_time_split": {"active": ck.get("active"), "parked_on_waitcnt_or_barrier": ck.get("stall_parked"), "issue_stall": ck.get("stall_issue")},
                      "lds_bank_conflict_frac": ck.get("lds_conflict_frac"), "vgpr": ck.get("vgpr"), "lds_bytes": ck.get("lds_bytes")})
        out[cls] = e
    return out


def _host_cpu():
    try:
        return [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        return "unknown"


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
    return {"value": round(sample / dt, 3), "unit": "solves/s", "cores": 1, "kind": "port", "ms_per_solve": round(1e3 * dt / sample, 3),
            "sample": "%d of the same problems (horizon and SQP iteration count as on the GPU), solve only, "
                      "reference pre-pass excluded" % sample,
            "note": "the port differentiates with 44-direction dual numbers where the reference runs CppAD-generated sparse code: likely 2-4x slower than "
                    "the reference's own LQ approximation; a large GPU/CPU ratio says nothing about kernel quality",
            "host_cpu": _host_cpu(), "host_cores": os.cpu_count(), "max_abs_x_diff_vs_gpu": worst, "value_3_threads": threads3}


def cpu_baseline_ddp(prob, sample, t_gpu, x_gpu, stats, robot="h1"):
    """--solver ddp: the restatement of the ILQR iteration (oracle/ddp_py.py: numpy over the C++ oracle's LQ model, python roll-outs) on the first
    `sample` problems; doubles as a parity check of the timed run.  A python port: context only."""
    import numpy as np
    from tests import oracle_bridge as ob
    from oracle import ddp_py, reference_py as rp
    m, om = ob.model(robot), ob.oracle(robot)
    sched = prob["schedule"]
    worst, spent = 0.0, 0.0
    for b in range(sample):
        nodes = ob.oracle_nodes(prob, b, robot=robot)
        x_nom, u_nom = rp.cold_start(m, nodes, prob["x0"][b])
        sc = sched[b] if isinstance(sched, list) else sched
        tt = prob["targets"][b if len(prob["targets"]) > 1 else 0]
        t0 = time.perf_counter()
        ref = ddp_py.ilqr_iteration(om, m, nodes, prob["x0"][b], x_nom, u_nom, list(map(float, sc.eventTimes)), list(map(int, sc.modeSequence)),
                                    np.asarray(tt.timeTrajectory), np.asarray(tt.stateTrajectory), m["ddp"], m["rollout"])
        spent += time.perf_counter() - t0
        n = len(ref["times"])
        if stats[b].n_nodes == n - 1:
            worst = max(worst, float(np.abs(x_gpu[b, :n] - ref["states"]).max()), float(np.abs(t_gpu[b, :n] - ref["times"]).max()))
        else:
            worst = float("inf")
    return {"value": round(sample / spent, 4), "unit": "solves/s", "cores": 1, "kind": "port", "ms_per_solve": round(1e3 * spent / sample, 1),
            "sample": "%d of the same problems, one ILQR iteration each incl. its eight roll-outs" % sample,
            "note": "python / numpy restatement (oracle/ddp_py.py) over the C++ oracle's LQ model: a checker, not a tuned CPU solver",
            "host_cpu": _host_cpu(), "host_cores": os.cpu_count(), "max_abs_diff_vs_gpu": worst}


def cpu_baseline_analytic(prob, sample, x_gpu, stats, robot="h1"):
    """A fairer single-thread CPU figure than `cpu_baseline`: the engine's own kernel bodies with ANALYTIC derivatives
    (kernels/{centroidal_eval,node_lq,project_node,riccati}.h) compiled by g++ -O3 -march=native on this box through the lane-emulation
    shim of the CPU test tier (tests/hostemu: test infrastructure, never linked into the product) - a stand-in for the reference's
    CppAD-generated sparse code, where the C++ oracle differentiates with 44-direction dual numbers.  Per problem: one QP step
    (linearise, FullPivLU projection, Riccati) and the evaluation of the full-step trial on every node (all problems of the benchmark
    accept alpha = 1; the filter decision itself is a handful of comparisons).  `value` of the bench line is unaffected."""
    import ctypes as C
    import numpy as np
    from tests import oracle_bridge as ob
    from tests.hostemu import build_hostemu
    from oracle import reference_py as rp
    try:
        lib = C.CDLL(build_hostemu.build(optimised=True, outdir=os.environ.get("TMPDIR", "/tmp")))      # not into tests/hostemu: a concurrent pytest builds there
    except Exception as e:      # no compiler on this box
        return {"value": None, "why": "could not build tests/hostemu with -O3 -march=native: %s" % e}
    lib.emu_model_create.restype = C.c_void_p
    d = os.path.join(ROOT, "assets", robot)
    urdf = {"h1": "h1_mpc.urdf", "openloong": "openloong_mpc.urdf", "g1": "g1_mpc.urdf", "hunter": "hunter_mpc.urdf"}[robot]
    h = C.c_void_p(lib.emu_model_create(os.path.join(d, urdf).encode(), os.path.join(d, "task.info").encode(), os.path.join(d, "reference.info").encode()))
    if not h:
        return {"value": None, "why": "emu_model_create failed"}
    m = ob.model(robot)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    f = lambda a: a.ctypes.data_as(dp)      # noqa: E731
    pre = []
    for b in range(sample):
        nodes = ob.oracle_nodes(prob, b, robot=robot)
        xi, ui = rp.cold_start(m, nodes, prob["x0"][b])
        c = lambda a, t=float: np.ascontiguousarray(a, t)      # noqa: E731
        pre.append((int(nodes["N"]), c(nodes["kind"], np.int32), c(nodes["dt"]), c(nodes["mode"], np.int32), c(nodes["zref"]), c(nodes["zdref"]),
                    c(nodes["xref"]), c(prob["x0"][b]), c(xi), c(ui)))
    nxx = pre[0][8].shape[1]
    worst, failures = 0.0, 0
    outs = [(np.zeros_like(p[8]), np.zeros_like(p[9]), np.zeros((p[0], nxx, nxx))) for p in pre]
    pb, pa = np.zeros(3), np.zeros(3)
    t0 = time.perf_counter()
    for (N, kind, dt, mode, zr, zd, xr, x0, xi, ui), (xn, un, K) in zip(pre, outs):
        failures += int(lib.emu_solve_iteration(h, N, kind.ctypes.data_as(ip), f(dt), mode.ctypes.data_as(ip), f(zr), f(zd), f(xr), f(x0), f(xi), f(ui),
                                                f(xn), f(un), f(K), f(pb), f(pa)) != 0)
    spent = time.perf_counter() - t0
    for b, (xn, _, _) in enumerate(outs):
        if stats[b].step_size == 1.0:
            worst = max(worst, float(np.abs(x_gpu[b, :stats[b].n_nodes + 1] - xn).max()))
    lib.emu_model_destroy(h)
    return {"value": round(sample / spent, 3), "unit": "solves/s", "cores": 1, "kind": "port", "ms_per_solve": round(1e3 * spent / sample, 3),
            "sample": "%d of the same problems: QP step with analytic derivatives + full-step trial evaluation per problem, reference pre-pass excluded" % sample,
            "what": "the engine's reference kernel bodies under lane emulation (tests/hostemu, g++ -O3 -march=native), one thread",
            "host_cpu": _host_cpu(), "host_cores": os.cpu_count(), "failures": failures, "max_abs_x_diff_vs_gpu": worst}


def cpu_baseline_sweep(itf, cp, sample, x_gpu, stats, robot):
    """The same for the gait-library sweep: `sample` problems spread over the rank's gaits, rebuilt on the host through the product's
    GaitSchedule / cmdVelToTargetTrajectories mirror and solved by the single-thread oracle."""
    import numpy as np
    from bipedal_control_amd import scenarios
    from tests import oracle_bridge as ob
    B = len(cp["x0"])
    picks = [int(i) for i in np.linspace(0, B - 1, sample)]
    worst, spent = 0.0, 0.0
    for b in picks:
        hp = scenarios.commands_problem_on_host(itf, cp, b)
        t0 = time.perf_counter()
        xo, uo, _, _ = ob.oracle_solve_like(hp, 0, robot=robot)
        spent += time.perf_counter() - t0
        n = stats[b].n_nodes
        worst = max(worst, float(np.abs(x_gpu[b, :n + 1] - xo).max()))
    return {"value": round(len(picks) / spent, 3), "unit": "solves/s", "cores": 1, "kind": "port", "ms_per_solve": round(1e3 * spent / len(picks), 3),
            "sample": "%d problems spread over the gaits of the sweep (same horizon, 1 SQP iteration), oracle pre-pass included" % len(picks),
            "host_cpu": _host_cpu(), "host_cores": os.cpu_count(), "max_abs_x_diff_vs_gpu": worst}


if __name__ == "__main__":
    main()
