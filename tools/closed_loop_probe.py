# Device-resident closed loop (SURVEY.md section 8(f) ranks 1-3 together): B robots, every tick = device-side reference generation
# with the warm start shifted from the previous solve + one SQP iteration + policy rollout over the MPC period.
import time, numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
itf = sc.h1_interface()
B, NI = 256, 100
horizon = NI * sc.DT
tm = [bp.loadModeSequenceTemplate(sc.H1["gait"], g) for g in ("trot", "standing_trot", "flying_trot")]
rng = np.random.default_rng(1)
gop = rng.integers(0, 3, B).astype(np.int32)
cmd = np.stack([rng.uniform(-0.3, 0.5, B), rng.uniform(-0.1, 0.1, B), np.zeros(B), rng.uniform(-0.3, 0.3, B)], axis=1)
x0 = sc.perturbed_initial_states(itf, B)
mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=sc.max_nodes_for(NI, horizon), return_gains=True)
period = 1.0 / 50.0
for dur in (1.0 / 400.0, period):
    mpc.setup_commands(0.0, x0, tm, gop, sc.GAIT_START, cmd, horizon=horizon); mpc.enqueue(); mpc.synchronize()
    mpc.rollout(dur)
    t = time.perf_counter()
    for _ in range(20): xe, ue, st = mpc.rollout(dur)
    print("rollout of %d problems over %.4f s: %.3f ms per call, accepted steps %d..%d, rejected max %d" % (B, dur, 1e3 * (time.perf_counter() - t) / 20, st[:, 0].min(), st[:, 0].max(), st[:, 1].max()))
ticks = 50
tt = dict(setup=0.0, solve=0.0, rollout=0.0)
mpc.setup_commands(0.0, x0, tm, gop, sc.GAIT_START, cmd, horizon=horizon)
for k in range(ticks):
    t = time.perf_counter(); mpc.enqueue(); mpc.synchronize(); tt["solve"] += time.perf_counter() - t
    t = time.perf_counter(); out = mpc.rollout(period, fetch=(k == ticks - 1)); tt["rollout"] += time.perf_counter() - t
    if out is not None: xe, ue, st = out
    t = time.perf_counter(); mpc.setup_commands((k + 1) * period, None, tm, gop, sc.GAIT_START, cmd, horizon=horizon, from_previous=True); tt["setup"] += time.perf_counter() - t
tot = sum(tt.values())
print("closed loop, %d robots x %d ticks of %.0f ms: %.2f ms per tick (setup %.2f, solve %.2f, rollout %.2f) = %.0f robot-ticks/s; base height %.3f..%.3f m, forward speed mean %.2f m/s"
      % (B, ticks, 1e3 * period, 1e3 * tot / ticks, 1e3 * tt["setup"] / ticks, 1e3 * tt["solve"] / ticks, 1e3 * tt["rollout"] / ticks, B * ticks / tot,
         xe[:, 8].min(), xe[:, 8].max(), float(np.mean(xe[:, 0]))))
