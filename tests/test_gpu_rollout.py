"""GPU tier: batched policy rollout (bpmpc_solver_rollout) against the oracle's restatement of
MRT_BASE::rolloutPolicy / TimeTriggeredRollout / odeint's controlled dopri5 (oracle/reference_py.py time_triggered_rollout),
both driven by the SAME solution (the GPU's x, u, K), so only the rollout itself is compared.
  end state                1e-9  (the step sequence must coincide: accepted / rejected counts are compared exactly)
  policy at the end point  1e-8 relative to max(1, |u|)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios
    from tests import oracle_bridge as ob
    from oracle import reference_py as rp
    return bp, scenarios, ob, rp, scenarios.h1_interface()


def _oracle_rollout(ob, rp, prob, b, x, u, K, n, t_start, x_start, duration, robot="h1"):
    m, om = ob.model(robot), ob.oracle(robot)
    nodes = ob.oracle_nodes(prob, b, robot=robot)
    assert nodes["N"] == n
    tp, xp, uff, KK = rp.primal_solution_arrays(nodes, x[b, :n + 1], u[b, :n], K[b, :n])
    sched = prob["schedule"][b] if isinstance(prob["schedule"], list) else prob["schedule"]
    ev = [float(e) for e in sched.eventTimes]
    ctrl = lambda t, xx: rp.linear_controller_input(tp, uff, KK, t, xx)   # noqa: E731
    return rp.time_triggered_rollout(lambda xx, uu: om.flow_map(xx, uu), ctrl, t_start, x_start, t_start + duration, ev, m["rollout"])


@pytest.mark.parametrize("gait", ["trot", "flying_trot"])
def test_rollout_matches_oracle(ctx, gait):
    bp, sc, ob, rp, itf = ctx
    B = 5
    prob = sc.trot_problem(itf, batch=B, n_intervals=40, gait=gait)
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=64, return_gains=True)
    t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    n = st[0].n_nodes
    rng = np.random.default_rng(5)
    worst = 0.0
    for duration, t_start in ((0.0025, 0.0), (0.02, 0.0), (0.02, 0.013), (0.25, 0.0), (0.3, 0.05)):
        xs = prob["x0"] + 2e-3 * rng.standard_normal(prob["x0"].shape)
        x_end, u_end, steps = mpc.rollout(duration, t_start=t_start, x_start=xs)
        for b in range(B):
            r = _oracle_rollout(ob, rp, prob, b, x, u, K, n, t_start, xs[b], duration)
            assert (int(steps[b, 0]), int(steps[b, 1])) == (r["accepted"], r["rejected"]), (duration, b, steps[b], r["accepted"], r["rejected"])
            ex = np.abs(x_end[b] - r["states"][-1]).max()
            eu = np.abs(u_end[b] - r["inputs"][-1]).max() / max(1.0, np.abs(r["inputs"][-1]).max())
            worst = max(worst, ex, eu)
            assert ex < 1e-9 and eu < 1e-8, (duration, t_start, b, ex, eu)
        if duration >= 0.25:
            assert any(len(_oracle_rollout(ob, rp, prob, b, x, u, K, n, t_start, xs[b], duration)["post_event_indices"]) > 0 for b in range(1))
    # defaults: from the initial time and measured state of the solve
    x_end, u_end, steps = mpc.rollout(0.02)
    r = _oracle_rollout(ob, rp, prob, 2, x, u, K, n, float(np.broadcast_to(prob["t0"], (B,))[2]), prob["x0"][2], 0.02)
    assert np.abs(x_end[2] - r["states"][-1]).max() < 1e-9


def test_closed_loop_stays_on_the_device(ctx):
    """solve -> rollout over one MPC period -> next solve from the rolled-out states (x0 = NULL) with the shifted warm start: the
    device-resident loop equals the loop that carries the states through the host, bit for bit."""
    bp, sc, ob, rp, itf = ctx
    B, NI, period = 4, 40, 0.02
    horizon = NI * sc.DT
    tm = [bp.loadModeSequenceTemplate(sc.H1["gait"], g) for g in ("trot", "standing_trot")]
    gop = np.array([0, 1, 0, 1], np.int32)
    cmd = np.array([(0.3, 0, 0, 0.1), (0.2, 0, 0, 0.0), (-0.2, 0.05, 0, 0.0), (0.0, 0, 0, 0.3)], float)
    x0 = sc.perturbed_initial_states(itf, B)
    dev = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=64, return_gains=True)
    host = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=64, return_gains=True)
    xm = x0.copy()
    for tick in range(4):
        t0 = tick * period
        dev.setup_commands(t0, x0 if tick == 0 else None, tm, gop, sc.GAIT_START, cmd, horizon=horizon, from_previous=tick > 0)
        host.setup_commands(t0, xm, tm, gop, sc.GAIT_START, cmd, horizon=horizon, from_previous=tick > 0)
        dev.enqueue(); host.enqueue()
        _, x1, u1, _, s1 = dev.fetch()
        _, x2, u2, _, s2 = host.fetch()
        assert np.array_equal(x1, x2) and np.array_equal(u1, u2), tick
        xm, _, st2 = host.rollout(period)
        if tick < 3:
            assert dev.rollout(period, fetch=False) is None            # only enqueued: the end states stay on the device
            continue
        xe1, _, st1 = dev.rollout(period)
        assert np.array_equal(xe1, xm) and np.array_equal(st1, st2)
        assert np.isfinite(xm).all() and np.abs(xm[:, 8] - x0[:, 8]).max() < 0.1          # the robots keep standing
    assert st1[:, 0].min() >= 1


def test_rollout_needs_a_solution(ctx):
    bp, sc, ob, rp, itf = ctx
    prob = sc.trot_problem(itf, batch=2, n_intervals=20)
    mpc = bp.BatchedSqpMpc(itf, max_batch=2, max_nodes=32)
    mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    with pytest.raises(bp.BpmpcError):
        mpc.rollout(0.02)
    with pytest.raises(bp.BpmpcError):
        mpc.setup_commands(0.0, None, [], -1, 0.0, (0, 0, 0, 0), horizon=prob["horizon"])   # no rollout yet


def test_openloong_device_prepass_and_rollout(ctx):
    """nx = nu = 24: 32 lanes per problem in the rollout kernel; device-side reference generation on the other model files."""
    bp, sc, ob, rp, _ = ctx
    itf = sc.interface("openloong")
    B, NI = 3, 30
    horizon = NI * sc.DT
    x0 = sc.perturbed_initial_states(itf, B)
    tm = [bp.loadModeSequenceTemplate(sc.OPENLOONG["gait"], "standing_trot")]
    cmd = (0.2, 0.0, 0.0, 0.1)
    sched = sc.gait_schedule(itf, "standing_trot", 0.0, horizon)
    targets = [itf.cmdVelToTargetTrajectories(cmd, 0.0, x0[b], horizon) for b in range(B)]
    prob = dict(t0=0.0, x0=x0, schedule=sched, targets=targets, horizon=horizon)
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=48, return_gains=True)
    mpc.setup_commands(0.0, x0, tm, 0, sc.GAIT_START, cmd, horizon=horizon)
    mpc.enqueue()
    t, x, u, K, st = mpc.fetch(gains=True)
    n = st[0].n_nodes
    for b in range(B):
        xo, uo, _, _ = ob.oracle_solve_like(prob, b, robot="openloong")
        assert np.abs(x[b, :n + 1] - xo).max() / max(1.0, np.abs(xo).max()) < 1e-8
    for duration in (0.02, 0.2):
        xs = x0 + 1e-3
        x_end, u_end, steps = mpc.rollout(duration, t_start=0.0, x_start=xs)
        for b in range(B):
            r = _oracle_rollout(ob, rp, prob, b, x, u, K, n, 0.0, xs[b], duration, robot="openloong")
            assert (int(steps[b, 0]), int(steps[b, 1])) == (r["accepted"], r["rejected"])
            assert np.abs(x_end[b] - r["states"][-1]).max() < 1e-9
            assert np.abs(u_end[b] - r["inputs"][-1]).max() / max(1.0, np.abs(r["inputs"][-1]).max()) < 1e-8


def test_rollout_degenerate_windows(ctx):
    """Zero duration returns the start state without a step; a negative duration is an argument error; a window that reaches past
    the end of the solution keeps using the last segment of the controller (clamped interpolation), like the oracle."""
    bp, sc, ob, rp, itf = ctx
    prob = sc.trot_problem(itf, batch=2, n_intervals=20)
    mpc = bp.BatchedSqpMpc(itf, max_batch=2, max_nodes=32, return_gains=True)
    t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    n = st[0].n_nodes
    xe, ue, steps = mpc.rollout(0.0)
    assert np.array_equal(xe, prob["x0"]) and not steps.any()
    with pytest.raises(bp.BpmpcError):
        mpc.rollout(-0.01)
    xs = prob["x0"] + 1e-3
    t_start = prob["horizon"] - 0.01                              # 10 ms before the end of the solution, 30 ms window
    xe, ue, steps = mpc.rollout(0.03, t_start=t_start, x_start=xs)
    for b in range(2):
        r = _oracle_rollout(ob, rp, prob, b, x, u, K, n, t_start, xs[b], 0.03)
        assert (int(steps[b, 0]), int(steps[b, 1])) == (r["accepted"], r["rejected"]) and np.abs(xe[b] - r["states"][-1]).max() < 1e-9


def test_rollout_with_dpp_tree_walks_agrees_with_the_table_walks(ctx, monkeypatch):
    """Round 6: on a robot of two serial legs whose coordinates fit one 16-lane row (H1, Hunter) the roll-out's flow map runs its tree walks by DPP between
    neighbouring lanes (k_rollout<NJ, true>, the lineariser's CHAIN form); BPMPC_LIN_TABLES=1 keeps the walks over LDS tables.  The same steps - identical
    accepted / rejected counts -, end states equal to rounding (the whole-robot sums associate differently)."""
    bp, sc, ob, rp, itf = ctx
    B = 6
    prob = sc.trot_problem(itf, batch=B, n_intervals=40)
    out = []
    for tables in ("0", "1"):
        monkeypatch.setenv("BPMPC_LIN_TABLES", tables)
        mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=64, return_gains=True)
        mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
        out.append(mpc.rollout(0.3, t_start=0.01, x_start=prob["x0"] + 1e-3))
    (xa, ua, sa), (xb, ub, sb) = out
    assert np.array_equal(sa, sb), (sa, sb)
    assert np.abs(xa - xb).max() < 1e-10 and np.abs(ua - ub).max() < 1e-8 * max(1.0, np.abs(ub).max())
