"""GPU tier: device-side reference generation (bpmpc_solver_setup_commands) against the host
pre-pass (bpmpc_solver_setup fed by GaitSchedule / cmdVelToTargetTrajectories, itself checked against the oracle in
tests/test_reference_prepass.py) and against the oracle's solve.
  node tables (kind, mode, dt, start, zref, zdref, node count)   bit-identical (contraction is off in the device code)
  targets / xref / initial iterate                               1e-13 (device sin / cos may differ in the last place)
  solve output x, u                                              1e-9 between the two paths, 1e-8 against the oracle"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GAITS = ["stance", "trot", "standing_trot", "flying_trot"]


@pytest.fixture(scope="module")
def ctx():
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios
    from tests import oracle_bridge as ob
    itf = scenarios.h1_interface()
    tm = [bp.loadModeSequenceTemplate(scenarios.H1["gait"], g) for g in GAITS[1:]]
    return bp, scenarios, ob, itf, tm


def _problem(sc, itf, t0s, gaits, cmds, n_intervals):
    """One problem per (t0, gait, command), described both ways: host schedules / targets and device gait ids / commands."""
    horizon = n_intervals * sc.DT
    rows = [(t0, g, c) for t0 in t0s for g in gaits for c in cmds]
    nb = len(rows)
    x0 = sc.perturbed_initial_states(itf, nb)
    sched = {(t0, g): sc.gait_schedule(itf, g, t0, horizon) for t0 in t0s for g in gaits}
    schedules = [sched[(t0, g)] for t0, g, _ in rows]
    targets = [itf.cmdVelToTargetTrajectories(c, t0, x0[b], horizon) for b, (t0, g, c) in enumerate(rows)]
    return dict(t0=np.array([r[0] for r in rows]), x0=x0, schedule=schedules, targets=targets, horizon=horizon,
                gait_of_problem=np.array([GAITS.index(r[1]) - 1 for r in rows], np.int32), gait_start=np.full(nb, sc.GAIT_START),
                cmd_vel=np.array([r[2] for r in rows], float))


def _tables(mpc, nb):
    N = mpc.max_nodes
    pg = mpc.read("p_grid").astype(int)[:nb]
    out = {}
    for name, w in (("g_kind", 1), ("g_mode", 1), ("g_dt", 1), ("g_start", 1), ("g_zref", 4), ("g_zdref", 4)):
        a = mpc.read(name).reshape(mpc.max_batch, N, w)
        out[name] = a[pg]
    out["nodes"] = mpc.read("g_nodes").astype(int)[pg]
    nx = mpc.nx
    out["xref"] = mpc.read("xref").reshape(mpc.max_batch, N, nx)[:nb]
    out["x"] = mpc.read("x").reshape(mpc.max_batch, N + 1, nx)[:nb]
    out["u"] = mpc.read("u").reshape(mpc.max_batch, N, nx)[:nb]
    return out


def test_tables_bit_identical_and_solve_matches(ctx):
    bp, sc, ob, itf, tm = ctx
    cmds = [(0.3, 0.0, 0.0, 0.0), (-0.2, 0.1, 0.0, 0.25)]
    prob = _problem(sc, itf, [0.0, 0.13, 1.234567], GAITS, cmds, 60)
    nb = len(prob["t0"])
    host = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=96)
    dev = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=96)
    lh = host.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    ld = dev.setup_commands(prob["t0"], prob["x0"], tm, prob["gait_of_problem"], prob["gait_start"], prob["cmd_vel"], horizon=prob["horizon"])
    assert lh == ld and ld["n_grids"] == 3 * len(GAITS)
    th, td = _tables(host, nb), _tables(dev, nb)
    for b in range(nb):
        n = th["nodes"][b]
        assert n == td["nodes"][b]
        for name in ("g_kind", "g_mode", "g_dt", "g_start", "g_zref", "g_zdref"):
            assert np.array_equal(th[name][b], td[name][b]), (name, b)        # whole stride, padding included
        assert np.abs(th["xref"][b, :n] - td["xref"][b, :n]).max() < 1e-13
        assert np.abs(th["x"][b, :n + 1] - td["x"][b, :n + 1]).max() < 1e-13 and np.abs(th["u"][b, :n] - td["u"][b, :n]).max() < 1e-11
    host.enqueue(); dev.enqueue()
    t1, x1, u1, _, s1 = host.fetch()
    t2, x2, u2, _, s2 = dev.fetch()
    assert np.array_equal(t1, t2)
    for b in range(nb):
        n = s1[b].n_nodes
        assert s2[b].n_nodes == n and s1[b].step_size == s2[b].step_size
        assert np.abs(x1[b, :n + 1] - x2[b, :n + 1]).max() < 1e-9 and np.abs(u1[b, :n] - u2[b, :n]).max() < 1e-9 * max(1.0, np.abs(u1[b]).max())
    for b in (1, 9, nb - 1):                                                  # and against the oracle's own pre-pass + solve
        xo, uo, _, _ = ob.oracle_solve_like(prob, b)
        n = s2[b].n_nodes
        assert np.abs(x2[b, :n + 1] - xo).max() / max(1.0, np.abs(xo).max()) < 1e-8
        assert np.abs(u2[b, :n] - uo).max() / max(1.0, np.abs(uo).max()) < 1e-8


def test_receding_horizon_on_device_matches_host_path(ctx):
    bp, sc, ob, itf, tm = ctx
    nb, NI, tick = 6, 40, 0.02
    horizon = NI * sc.DT
    gait_names = ["trot", "standing_trot", "flying_trot"] * 2
    gop = np.array([GAITS.index(g) - 1 for g in gait_names], np.int32)
    cmd = np.array([(0.3, 0.0, 0.0, 0.1)] * 3 + [(-0.1, 0.05, 0.0, -0.2)] * 3)
    host = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=64, return_gains=True)
    dev = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=64, return_gains=True)
    x_meas = sc.perturbed_initial_states(itf, nb)
    for it in range(4):
        t0 = it * tick
        scheds = [sc.gait_schedule(itf, g, t0, horizon) for g in gait_names]
        targets = [itf.cmdVelToTargetTrajectories(tuple(cmd[b]), t0, x_meas[b], horizon) for b in range(nb)]
        if it == 0:
            host.setup(t0, x_meas, scheds, targets, horizon=horizon)
        else:
            host.setup_from_previous(t0, x_meas, scheds, targets, horizon=horizon)
        dev.setup_commands(t0, x_meas, tm, gop, sc.GAIT_START, cmd, horizon=horizon, from_previous=it > 0)
        host.enqueue(); dev.enqueue()
        t1, x1, u1, K1, s1 = host.fetch(gains=True)
        t2, x2, u2, K2, s2 = dev.fetch(gains=True)
        assert np.array_equal(t1, t2)
        nxt = np.zeros_like(x_meas)
        for b in range(nb):
            n = s1[b].n_nodes
            assert s2[b].n_nodes == n
            assert np.abs(x1[b, :n + 1] - x2[b, :n + 1]).max() < 1e-9, (it, b)
            assert np.abs(u1[b, :n] - u2[b, :n]).max() < 1e-9 * max(1.0, np.abs(u1[b]).max()), (it, b)
            nxt[b] = x1[b, 1] + 1e-3 * np.sin(np.arange(x1.shape[2]) + b + it)
        x_meas = nxt


def test_errors_are_reported_like_the_host_path(ctx):
    bp, sc, ob, itf, tm = ctx
    x0 = sc.perturbed_initial_states(itf, 2)
    mpc = bp.BatchedSqpMpc(itf, max_batch=2, max_nodes=16)
    with pytest.raises(bp.BpmpcError) as e:                                   # grid longer than max_nodes
        mpc.setup_commands(0.0, x0, tm, 0, sc.GAIT_START, (0.3, 0, 0, 0), horizon=30 * sc.DT)
    assert e.value.status == -6 and "max_nodes" in str(e.value)
    with pytest.raises(bp.BpmpcError):                                        # nothing usable is left behind
        mpc.enqueue()
    fast = bp.ModeSequenceTemplate(np.array([0.0, 0.002, 0.004]), np.array([1, 2], np.int32))
    with pytest.raises(bp.BpmpcError) as e:                                   # 2 ms phases over 3 horizons: more events than the device holds
        bp.BatchedSqpMpc(itf, max_batch=2, max_nodes=512).setup_commands(0.0, x0, [fast], 0, -0.1, (0.3, 0, 0, 0), horizon=0.45)
    assert "capacity" in str(e.value)
    with pytest.raises(bp.BpmpcError):                                        # template index out of range
        mpc.setup_commands(0.0, x0, tm, 7, sc.GAIT_START, (0.3, 0, 0, 0), horizon=10 * sc.DT)
    # and a valid call afterwards works
    mpc.setup_commands(0.0, x0, tm, -1, 0.0, (0.0, 0, 0, 0), horizon=10 * sc.DT)
    mpc.enqueue()
    _, x, _, _, st = mpc.fetch()
    assert st[0].n_nodes == 10 and np.isfinite(x[:, :11]).all()


def test_goal_pose_targets_match_host_path(ctx):
    bp, sc, ob, itf, tm = ctx
    nb, horizon = 5, 30 * sc.DT
    x0 = sc.perturbed_initial_states(itf, nb)
    goals = np.array([(1.0, 0.2, 0.0, 0.3), (-0.5, 0.0, 0.0, 0.0), (0.05, -0.4, 0.0, -1.0), (2.0, 2.0, 0.0, 0.1), (0.0, 0.0, 0.0, 0.0)])
    sched = sc.gait_schedule(itf, "trot", 0.5, horizon)
    targets = [itf.goalToTargetTrajectories(tuple(goals[b]), 0.5, x0[b]) for b in range(nb)]
    host = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=48)
    dev = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=48)
    host.setup(0.5, x0, sched, targets, horizon=horizon)
    dev.setup_commands(0.5, x0, tm, 0, sc.GAIT_START, goals, horizon=horizon, goal=True)
    th, td = _tables(host, nb), _tables(dev, nb)
    n = th["nodes"][0]
    assert np.array_equal(th["nodes"], td["nodes"]) and np.array_equal(th["g_zref"], td["g_zref"])
    assert np.abs(th["xref"][:, :n] - td["xref"][:, :n]).max() < 1e-13
    host.enqueue(); dev.enqueue()
    _, x1, u1, _, _ = host.fetch()
    _, x2, u2, _, _ = dev.fetch()
    assert np.abs(x1[:, :n + 1] - x2[:, :n + 1]).max() < 1e-9 and np.abs(u1[:, :n] - u2[:, :n]).max() < 1e-9 * max(1.0, np.abs(u1).max())


def test_config5_gait_library_sweep_full_size(ctx):
    """BASELINE.json configs[4] per-GPU shape: 512 problems, horizon 150, the reference's 4 gaits + 4 synthetic trot variants
    (periods 0.5 / 0.6 / 0.9 / 1.0 s), commands on a v_x x omega_z grid - generated on the device.
    Full-size properties (every problem solves, solutions do not depend on batch neighbours) and the oracle on a sample."""
    bp, sc, ob, itf, tm = ctx
    trot = tm[0]
    variants = [bp.ModeSequenceTemplate(np.asarray(trot.switchingTimes) * (T / 0.7), np.asarray(trot.modeSequence)) for T in (0.5, 0.6, 0.9, 1.0)]
    stance = bp.loadModeSequenceTemplate(sc.H1["gait"], "stance")
    lib = [stance] + list(tm) + variants                                   # 8 templates
    cmds = [(vx, 0.0, 0.0, wz) for vx in np.linspace(-0.5, 0.5, 8) for wz in np.linspace(-0.3, 0.3, 8)]      # 64 commands per gait
    NI = 150
    horizon = NI * sc.DT
    B = len(lib) * len(cmds)
    assert B == 512
    gop = np.repeat(np.arange(len(lib)), len(cmds)).astype(np.int32)
    cmd = np.array(cmds * len(lib))
    x0 = sc.perturbed_initial_states(itf, B)
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=200)
    lay = mpc.setup_commands(0.0, x0, lib, gop, sc.GAIT_START, cmd, horizon=horizon)
    assert lay["n_grids"] == len(lib) and lay["n_nodes_max"] <= 200
    mpc.enqueue()
    t, x, u, _, st = mpc.fetch()
    assert all(s.status == 0 and s.step_size > 0 for s in st)
    viol = np.array([np.sqrt(s.dynamics_sse_after + s.equality_sse_after) for s in st])
    viol0 = np.array([np.sqrt(s.dynamics_sse_before + s.equality_sse_before) for s in st])
    assert np.all(viol < viol0) and np.isfinite(x[:, :NI]).all()
    # a sub-batch in another order gives the same solutions: to rounding against the batch of 512 (four-wave Riccati sweep, two problems
    # per CU; a batch with one problem per CU runs the eight-wave sweep of riccati_mfma8.h - same mathematics, another elimination
    # order), bitwise between two batches on the same sweep
    sub = [511, 130, 64, 7]
    mpc2 = bp.BatchedSqpMpc(itf, max_batch=4, max_nodes=200)
    mpc2.setup_commands(0.0, x0[sub], lib, gop[sub], sc.GAIT_START, cmd[sub], horizon=horizon)
    mpc2.enqueue()
    _, x2, u2, _, _ = mpc2.fetch()
    for j, i in enumerate(sub):
        assert np.abs(x2[j] - x[i]).max() < 1e-10 * max(1.0, np.abs(x[i]).max()) and np.abs(u2[j] - u[i]).max() < 1e-10 * max(1.0, np.abs(u[i]).max())
    sub3 = [7, 64, 300, 130, 511]
    mpc3 = bp.BatchedSqpMpc(itf, max_batch=5, max_nodes=200)
    mpc3.setup_commands(0.0, x0[sub3], lib, gop[sub3], sc.GAIT_START, cmd[sub3], horizon=horizon)
    mpc3.enqueue()
    _, x3, u3, _, _ = mpc3.fetch()
    for j, i in enumerate(sub):
        assert np.array_equal(x2[j], x3[sub3.index(i)]) and np.array_equal(u2[j], u3[sub3.index(i)])
    # oracle on one problem per synthetic variant: schedule from the host GaitSchedule fed with the same scaled template
    for i in (4 * 64 + 5, 7 * 64 + 63):
        gs = bp.GaitSchedule(itf)
        gs.insertModeSequenceTemplate(lib[gop[i]], sc.GAIT_START, 2 * horizon)
        sched = gs.getModeSchedule(-horizon, 2 * horizon)
        prob = dict(t0=0.0, x0=x0[i:i + 1], schedule=sched, targets=[itf.cmdVelToTargetTrajectories(tuple(cmd[i]), 0.0, x0[i], horizon)], horizon=horizon)
        xo, uo, _, _ = ob.oracle_solve_like(prob, 0)
        n = st[i].n_nodes
        assert xo.shape[0] == n + 1
        assert np.abs(x[i, :n + 1] - xo).max() / max(1.0, np.abs(xo).max()) < 1e-8 and np.abs(u[i, :n] - uo).max() / max(1.0, np.abs(uo).max()) < 1e-8


def test_long_running_gait_is_tiled_without_growing(ctx):
    """t0 = 512.3 s: the trot has been tiled for ~1470 events since its insertion, far more than the device keeps (448); what lies in
    front of the window is dropped while tiling, the tables still equal the host pre-pass bit for bit."""
    bp, sc, ob, itf, tm = ctx
    prob = _problem(sc, itf, [512.3], ["trot", "flying_trot"], [(0.3, 0.0, 0.0, 0.1)], 40)
    nb = len(prob["t0"])
    assert len(prob["schedule"][0].eventTimes) < 30          # the host keeps only the window, too
    host = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=64)
    dev = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=64)
    host.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    dev.setup_commands(prob["t0"], prob["x0"], tm, prob["gait_of_problem"], prob["gait_start"], prob["cmd_vel"], horizon=prob["horizon"])
    th, td = _tables(host, nb), _tables(dev, nb)
    for name in ("g_kind", "g_mode", "g_dt", "g_start", "g_zref", "g_zdref", "nodes"):
        assert np.array_equal(th[name], td[name]), name
    host.enqueue(); dev.enqueue()
    _, x1, u1, _, s1 = host.fetch()
    _, x2, u2, _, s2 = dev.fetch()
    n = s1[0].n_nodes
    assert np.abs(x1[:, :n + 1] - x2[:, :n + 1]).max() < 1e-9 and all(s.status == 0 for s in s2)
