# Long closed-loop run on the device: B robots, T ticks of the 50 Hz MPC loop (reference generation -> warm start -> SQP iteration ->
# policy rollout), states never leave the GPU.  Prints tracking statistics; fails on any solver failure or non-finite state.
import sys, time, numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
itf = sc.h1_interface()
B, NI = 256, 67                      # the reference's own horizon: 1.0 s
horizon, period = NI * sc.DT, 0.02
tm = [bp.loadModeSequenceTemplate(sc.H1["gait"], g) for g in ("trot", "standing_trot", "flying_trot")]
rng = np.random.default_rng(7)
gop = rng.integers(0, 3, B).astype(np.int32)
cmd = np.stack([rng.uniform(-0.2, 0.5, B), rng.uniform(-0.1, 0.1, B), np.zeros(B), rng.uniform(-0.3, 0.3, B)], axis=1)
x0 = sc.perturbed_initial_states(itf, B)
mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=sc.max_nodes_for(NI, horizon), return_gains=True)
mpc.setup_commands(0.0, x0, tm, gop, sc.GAIT_START, cmd, horizon=horizon)
t_wall = time.perf_counter()
worst_h, fails = [], 0
for k in range(T):
    mpc.enqueue()
    last = k % 100 == 99 or k == T - 1
    out = mpc.rollout(period, fetch=last)           # the end states stay on the device; read back only for the statistics
    if last:
        xe, ue, st = out
        _, x, u, _, stats = mpc.fetch()
        fails += sum(1 for s in stats if s.status != 0)
        assert np.isfinite(xe).all(), "non-finite state at tick %d" % k
        worst_h.append((float(xe[:, 8].min()), float(xe[:, 8].max())))
    mpc.setup_commands((k + 1) * period, None, tm, gop, sc.GAIT_START, cmd, horizon=horizon, from_previous=True)
wall = time.perf_counter() - t_wall
v_body = xe[:, 0:2]                 # normalised linear momentum = com velocity (world)
yaw = xe[:, 9]
v_cmd_world = np.stack([np.cos(yaw) * cmd[:, 0] - np.sin(yaw) * cmd[:, 1], np.sin(yaw) * cmd[:, 0] + np.cos(yaw) * cmd[:, 1]], axis=1)
print("%d robots x %d ticks (%.0f s simulated) in %.2f s wall = %.2f ms per tick; failures %d; base height over the run %.3f..%.3f m; "
      "final |v_com - v_cmd| mean %.3f max %.3f m/s; yaw travelled mean %.2f rad (commanded %.2f)"
      % (B, T, T * period, wall, 1e3 * wall / T, fails, min(h[0] for h in worst_h), max(h[1] for h in worst_h),
         float(np.linalg.norm(v_body - v_cmd_world, axis=1).mean()), float(np.linalg.norm(v_body - v_cmd_world, axis=1).max()),
         float(np.mean(np.abs(yaw - x0[:, 9]))), float(np.mean(np.abs(cmd[:, 3])) * T * period)))
assert fails == 0
