# how often the filter line search back-tracks in a device-resident closed loop (256 robots, three gaits, 100 ticks): share of solves with a step size < 1
import numpy as np, bipedal_control_amd as bp
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
mpc.setup_commands(0.0, x0, tm, gop, sc.GAIT_START, cmd, horizon=horizon)
hist = []
for k in range(100):
    mpc.enqueue(); mpc.synchronize()
    st = mpc.fetch()[-1]
    hist.append([s.step_size for s in st])
    mpc.rollout(period, fetch=False)
    mpc.setup_commands((k + 1) * period, None, tm, gop, sc.GAIT_START, cmd, horizon=horizon, from_previous=True)
h = np.array(hist)
print("ticks x robots", h.shape, " full steps %.4f, back-tracked %.4f, no step %.4f; ticks in which any robot back-tracked: %d of %d" % ((h == 1.0).mean(), ((h < 1.0) & (h > 0)).mean(), (h == 0).mean(), int(((h < 1.0).any(axis=1)).sum()), h.shape[0]))
print("per tick share of back-tracking robots (first 20 ticks):", np.round((h < 1.0).mean(axis=1)[:20], 3))
