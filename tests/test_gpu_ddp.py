"""The DDP slice (bpmpc_settings.solver = BPMPC_SOLVER_DDP: one GaussNewtonDDP / ILQR iteration, the reference's second solver,
ocs2_bipedal_robot_ros/src/BipedalRobotDdpMpcNode.cpp:70-71) on the GPU against its restatement oracle/ddp_py.py: the policy of the backward
pass (gains, feedforward increment), the performance indices of every step length, the accepted step and the solution on the roll-out's own
time points.  Tolerances: the policy 1e-8 relative (two different exact solutions of the constrained stage problems: pivoted elimination
against the oracle's restatement of it + dense algebra), trajectories 1e-7 on identical sequences of accepted ODE45 steps."""
import numpy as np
import pytest

import bipedal_control_amd as bp
from bipedal_control_amd import scenarios
from oracle import ddp_py, reference_py as rp
from tests import oracle_bridge as ob


def _oracle(prob, b, x0, robot="h1"):
    m, om = ob.model(robot), ob.oracle(robot)
    nodes = ob.oracle_nodes(prob, b, robot=robot)
    x_nom, u_nom = rp.cold_start(m, nodes, x0)
    sched = prob["schedule"][b] if isinstance(prob["schedule"], list) else prob["schedule"]
    ev, ms = list(map(float, sched.eventTimes)), list(map(int, sched.modeSequence))
    tt = prob["targets"][b if len(prob["targets"]) > 1 else 0]
    return nodes, ddp_py.ilqr_iteration(om, m, nodes, x0, x_nom, u_nom, ev, ms, np.asarray(tt.timeTrajectory), np.asarray(tt.stateTrajectory), m["ddp"], m["rollout"])


def _check(prob, nodes_cap, robot="h1"):
    itf = scenarios.interface(robot)
    B = prob["x0"].shape[0]
    mpc = bp.BatchedDdpMpc(itf, B, nodes_cap)
    t, x, u, K, stats = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    lff = mpc.read("ddp_lff").reshape(B, nodes_cap, itf.inputDim)
    upd = mpc.read("ddp_update_is")
    for b in range(B):
        nodes, ref = _oracle(prob, b, prob["x0"][b], robot)
        N = int(nodes["N"])
        scale_k = max(1.0, float(np.abs(ref["K"]).max()))
        assert np.abs(K[b, :N] - ref["K"]).max() < 1e-8 * scale_k, "gains"
        assert np.abs(lff[b, :N] - ref["lff"]).max() < 1e-8 * max(1.0, float(np.abs(ref["lff"]).max())), "feedforward increment"
        assert abs(upd[b] - ref["update_is"]) < 1e-8 * max(1.0, ref["update_is"])
        st = stats[b]
        assert st.status == (0 if ref["alpha"] > 0 else 1) and st.step_size == ref["alpha"]
        assert abs(st.merit_before - ref["merit0"]) < 1e-8 * max(1.0, abs(ref["merit0"]))
        n = len(ref["times"])
        assert st.n_nodes == n - 1, (st.n_nodes, n)
        # the same sequence of accepted steps (checked above: the same count); the step lengths themselves come out of the controller's
        # 0.9 err^(-1/5) with err a difference of nearly equal numbers: 1e-9 relative between the two implementations
        assert np.abs(t[b, :n] - ref["times"]).max() < 1e-7
        assert np.abs(x[b, :n] - ref["states"]).max() < 1e-7
        assert np.abs(u[b, :n - 1] - ref["inputs"][:n - 1]).max() < 1e-6 * max(1.0, float(np.abs(ref["inputs"]).max()))
    return stats


@pytest.mark.gpu
def test_ddp_stance_config1_policy_defined_by_the_engine_matches_its_restatement():
    itf = scenarios.h1_interface()
    prob = scenarios.stance_problem(itf, 20)                 # BASELINE.json configs[0]: H1 stance, horizon 20
    x0 = np.repeat(prob["x0"], 3, axis=0)
    x0[1, 8] -= 0.03; x0[1, 0] += 0.05                        # a height error and a forward momentum
    x0[2, 12:] += 0.04 * np.sin(np.arange(itf.stateDim - 12))
    prob = dict(prob, x0=x0, targets=prob["targets"] * 3)
    stats = _check(prob, 40)
    assert all(s.step_size > 0 for s in stats[1:])


@pytest.mark.gpu
def test_ddp_trot_policy_defined_by_the_engine_where_D_is_rank_deficient_matches_its_restatement():
    """While a foot stands D has 6 zero-velocity rows of rank 5: upstream's Hm-weighted projectors do not exist there and the policy is DEFINED by the
    engine's pivoted elimination, which oracle/ddp_py.py restates (constrained_stage(method="lu")) - this compares the product with the restatement of
    its own definition.  The full-row-rank case is pinned independently (pseudo-inverse route) in tests/test_ddp_oracle.py."""
    itf = scenarios.h1_interface()
    prob = scenarios.trot_problem(itf, batch=2, n_intervals=40, gait_start=0.0)
    _check(prob, 60)


@pytest.mark.gpu
@pytest.mark.parametrize("robot,gait", [("g1", "standing_trot"), ("hunter", "trot")])
def test_ddp_on_the_other_robots_policy_defined_by_the_engine_matches_its_restatement(robot, gait):
    """nx = nu = 24 (Unitree G1: the kernels' second instantiation, six joints per leg) and Hunter (`positionErrorGain 20`: the position term of the
    zero-velocity rows enters e, so the constrained stage problems have inconsistent dependent rows - the case the pivoted elimination defines)."""
    itf = scenarios.interface(robot)
    prob = scenarios.trot_problem(itf, batch=2, n_intervals=30, gait=gait)
    _check(prob, 48, robot)


@pytest.mark.gpu
def test_ddp_refuses_what_the_slice_does_not_implement():
    itf = scenarios.h1_interface()
    with pytest.raises(bp.BpmpcError):
        bp.BatchedDdpMpc(itf, 1, 40, sqp_iterations=2)        # later iterations live on the roll-out's adaptive grid


@pytest.mark.gpu
def test_ddp_receding_horizon_tick_warm_starts_from_the_roll_out_of_the_previous_one():
    """MPC loop (mpc.coldStart false): the second run's nominal INPUTS are the previous solution - a FeedforwardController on the roll-out's own
    time points (ddp.useFeedbackPolicy false) - interpolated onto the new grid (oracle/reference_py.py warm_start_from_previous with
    feedback = False), its nominal STATES the roll-out of that controller from the measured state (round 6; [OCS2-upstream, recalled]
    GaussNewtonDDP::rolloutInitialTrajectory; oracle/ddp_py.py nominal_rollout)."""
    itf = scenarios.h1_interface()
    prob = scenarios.stance_problem(itf, 20)
    x0 = prob["x0"].copy(); x0[0, 8] -= 0.02; x0[0, 1] += 0.03
    prob = dict(prob, x0=x0)
    cap = 40
    mpc = bp.BatchedDdpMpc(itf, 1, cap)
    t1, x1, u1, _, st1 = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    n1 = st1[0].n_nodes + 1
    # next tick: 15 ms later, the measured state is the solution's state there (interpolated) plus a disturbance
    tn = 0.015
    j = int(np.searchsorted(t1[0, :n1], tn, side="left")) - 1
    a = (t1[0, j + 1] - tn) / (t1[0, j + 1] - t1[0, j])
    xm = a * x1[0, j] + (1 - a) * x1[0, j + 1]
    xm[2] += 0.01
    t2, x2, u2, _, st2 = mpc.advance(tn, xm[None, :], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    # the oracle's second tick
    m, om = ob.model("h1"), ob.oracle("h1")
    prob2 = dict(prob, t0=tn, x0=xm[None, :])
    nodes2 = ob.oracle_nodes(prob2, 0)
    prev = dict(N=n1 - 1, times=t1[0, :n1].copy(), kind=np.zeros(n1 - 1, np.int32))
    x_sh, u_nom = rp.warm_start_from_previous(m, nodes2, xm, prev, x1[0, :n1], u1[0, :n1 - 1], np.zeros((n1 - 1, m["nu"], m["nx"])), feedback=False)
    sched = prob["schedule"]
    ev, ms = list(map(float, sched.eventTimes)), list(map(int, sched.modeSequence))
    # round 6: the nominal STATES are the roll-out of that input trajectory from the measured state (what GaussNewtonDDP_MPC does), not the shifted solution
    x_nom = ddp_py.nominal_rollout(om, nodes2, xm, x_sh, u_nom, ev, m["rollout"])
    x_init = mpc.read("x_init").reshape(1, cap + 1, m["nx"])      # the initial iterate of the tick as the device built it (after the run `x` holds the solution)
    assert np.abs(x_init[0, :int(nodes2["N"]) + 1] - x_nom).max() < 1e-7, np.abs(x_init[0, :int(nodes2["N"]) + 1] - x_nom).max()
    assert np.abs(x_nom - x_sh).max() > 1e-4                       # (and it is not the shifted solution)
    tt = prob["targets"][0]
    ref = ddp_py.ilqr_iteration(om, m, nodes2, xm, x_nom, u_nom, ev, ms, np.asarray(tt.timeTrajectory), np.asarray(tt.stateTrajectory), m["ddp"], m["rollout"])
    n = len(ref["times"])
    assert st2[0].n_nodes == n - 1 and st2[0].step_size == ref["alpha"]
    assert abs(st2[0].merit_before - ref["merit0"]) < 1e-7 * max(1.0, abs(ref["merit0"]))
    assert np.abs(t2[0, :n] - ref["times"]).max() < 1e-7 and np.abs(x2[0, :n] - ref["states"]).max() < 1e-6


@pytest.mark.gpu
def test_ddp_fast_kernels_agree_with_the_reference_kernel_bodies():
    """Round 6: the backward pass runs on the kernels of the SQP path (the fast lineariser in its ILQR form, structured elimination, change of variables on
    the matrix cores, the regime's Riccati sweep) and the line search rolls all step lengths out in one launch; `reference_kernels = 1` keeps the lane-emulated
    bodies of round 5.  Same policy to rounding, identical decisions."""
    itf = scenarios.h1_interface()
    prob = scenarios.trot_problem(itf, batch=6, n_intervals=40, gait_start=0.0)
    out = []
    for ref in (False, True):
        mpc = bp.BatchedDdpMpc(itf, 6, 60, reference_kernels=ref)
        t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
        out.append((t, x, u, K, st, mpc.read("ddp_lff").copy()))
    (t0, x0, u0, K0, s0, l0), (t1, x1, u1, K1, s1, l1) = out
    assert np.abs(K0 - K1).max() < 1e-9 * max(1.0, np.abs(K1).max()) and np.abs(l0 - l1).max() < 1e-9 * max(1.0, np.abs(l1).max())
    for b in range(6):
        assert s0[b].n_nodes == s1[b].n_nodes and s0[b].step_size == s1[b].step_size and s0[b].status == s1[b].status
        n = s0[b].n_nodes + 1
        assert np.abs(t0[b, :n] - t1[b, :n]).max() < 1e-8 and np.abs(x0[b, :n] - x1[b, :n]).max() < 1e-7


@pytest.mark.gpu
def test_ddp_full_size_config2_properties():
    """BASELINE.json configs[1] shape (256 x horizon 100) through the DDP solver: every problem returns a roll-out over its whole horizon from its measured
    state, the accepted performance index satisfies the Armijo condition against the baseline, rejected problems return the baseline, and the batch is
    independent of its neighbours (a problem solved alone gives the same bits)."""
    itf = scenarios.h1_interface()
    prob = scenarios.trot_problem(itf, batch=256, n_intervals=100, gait_start=0.0)
    cap = 232       # the solution arrays hold max_nodes + 1 time points: a roll-out of this horizon records 70 .. 130 (a longer one is reported as status 3)
    mpc = bp.BatchedDdpMpc(itf, 256, cap)
    t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    upd = mpc.read("ddp_update_is")
    # status 3: the BASELINE roll-out (the new gains without the feedforward increment) did not fit the record - for two or three of the 256 perturbed
    # states it is stiff (> 233 accepted ODE45 steps where the others take ~80): no baseline, no Armijo test, the nominal trajectories stay (documented)
    assert all(s.status in (0, 1, 3) for s in st), sorted({s.status for s in st})
    assert sum(s.status == 3 for s in st) <= 5
    assert sum(s.status == 0 for s in st) >= 200
    for b in range(256):
        if st[b].status == 3:
            continue
        n = st[b].n_nodes + 1
        assert n >= 3 and abs(t[b, 0] - prob["t0"]) <= 1e-6 + 1e-12      # (the first time point is the weakEpsilon-nudged begin of the first interval)
        assert abs(t[b, n - 1] - (prob["t0"] + prob["horizon"])) < 1e-9
        assert np.all(np.diff(t[b, :n]) > 0)
        assert np.array_equal(x[b, 0], prob["x0"][b])
        assert np.isfinite(x[b, :n]).all() and np.isfinite(u[b, :n - 1]).all()
        if st[b].status == 0:
            assert st[b].step_size in (1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625)
            assert st[b].merit_after < st[b].merit_before - 1e-4 * st[b].step_size * upd[b] + 1e-12
        else:
            assert st[b].step_size == 0.0 and st[b].merit_after == st[b].merit_before
    # batch independence: problem 17 alone
    one = dict(prob, x0=prob["x0"][17:18], targets=[prob["targets"][17 if len(prob["targets"]) > 1 else 0]])
    m1 = bp.BatchedDdpMpc(itf, 1, cap)
    t1, x1, u1, _, s1 = m1.run(one["t0"], one["x0"], one["schedule"], one["targets"], horizon=one["horizon"])
    n = s1[0].n_nodes + 1
    assert s1[0].n_nodes == st[17].n_nodes and s1[0].step_size == st[17].step_size
    assert np.abs(x1[0, :n] - x[17, :n]).max() < 1e-9      # (another Riccati sweep kernel serves a batch of one: same policy to rounding)


@pytest.mark.gpu
def test_ddp_solution_on_its_own_time_points_refuses_grid_based_calls():
    """Advisor r05: after a DDP solve x / u live on the roll-out's adaptive time points; the calls that would read them against the shooting grid - the
    policy roll-out, the constraint values, a second iteration on the same setup - are refused instead of returning silently wrong numbers; reset
    (and any new setup) puts a nominal trajectory back on the grid."""
    itf = scenarios.h1_interface()
    prob = scenarios.stance_problem(itf, 20)
    mpc = bp.BatchedDdpMpc(itf, 1, 40)
    mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    for call in (lambda: mpc.rollout(0.01), lambda: mpc.constraint_values(), lambda: mpc.enqueue()):
        with pytest.raises(bp.BpmpcError) as e:
            call()
        assert e.value.status == -3, e.value                    # BPMPC_ERR_UNSUPPORTED (include/bpmpc.h)
    mpc.reset()
    mpc.enqueue(); mpc.synchronize()                            # a first iteration again
    _, x2, _, _, st2 = mpc.fetch()
    assert st2[0].n_nodes >= 2 and np.isfinite(x2[0, :st2[0].n_nodes + 1]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["h1", "g1"])
def test_ilqr_lineariser_lq_model_matches_oracle(robot):
    """The ILQR form of the fast lineariser (kernels/linearize_fast.h, ILQR = true; materialised) against the oracle's Euler-discretised LQ model
    (oracle/ddp_py.py euler_lq: A = I + dt A_c, B = dt B_c from the dual-number flow map; cost and constraint rows of the transcription) plus the
    DIAGONAL_SHIFT on R: every block to 1e-11, b = 0, at a generic (warm) iterate, stance and swing modes."""
    itf = scenarios.interface(robot)
    prob = scenarios.trot_problem(itf, batch=2, n_intervals=30, gait=("standing_trot" if robot == "g1" else "trot"))
    cap = 48
    nx = nu = itf.stateDim
    # a generic iterate: one SQP iteration first (its x, u are then fed to the DDP handle as the warm start)
    sqp = bp.BatchedSqpMpc(itf, 2, cap)
    _, xs, us, _, _ = sqp.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    mpc = bp.BatchedDdpMpc(itf, 2, cap, materialize_lq=True)
    lay = mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], warm_x=xs, warm_u=us)
    mpc.stage("linearize"); mpc.synchronize()
    x = mpc.read("x").reshape(2, cap + 1, nx); u = mpc.read("u").reshape(2, cap, nu)
    shapes = dict(A=(nx, nx), B=(nx, nu), b=(nx,), Q=(nx, nx), R=(nu, nu), q=(nx,), r=(nu,), C=(16, nx), D=(16, nu), e=(16,))
    dev = {k: mpc.read(k).reshape(2, cap, *s) for k, s in shapes.items()}
    nc = mpc.read("nc").reshape(2, cap)
    om, m = ob.oracle(robot), ob.model(robot)
    shift = float(m["ddp"]["hessianCorrectionMultiple"])
    worst = {}
    rel = lambda a, o: float(np.abs(np.asarray(a) - np.asarray(o)).max() / max(1.0, np.abs(np.asarray(o)).max()))      # noqa: E731
    for b in range(2):
        nodes = ob.oracle_nodes(prob, b, robot=robot)
        lq = ddp_py.euler_lq(om, nodes, x[b], u[b])
        for k in range(int(nodes["N"])):
            if nodes["kind"][k] == 1:
                assert np.array_equal(dev["A"][b, k], np.eye(nx)) and not dev["b"][b, k].any()
                continue
            n = lq["C"][k].shape[0]
            assert int(nc[b, k]) == n
            ref = dict(A=lq["A"][k], B=lq["B"][k], b=np.zeros(nx), Q=lq["Q"][k], R=lq["R"][k] + shift * np.eye(nu), q=lq["q"][k], r=lq["r"][k])
            for name, o in ref.items():
                worst[name] = max(worst.get(name, 0.0), rel(dev[name][b, k], o))
            worst["C"] = max(worst.get("C", 0.0), rel(dev["C"][b, k, :n], lq["C"][k])); worst["D"] = max(worst.get("D", 0.0), rel(dev["D"][b, k, :n], lq["D"][k]))
            worst["e"] = max(worst.get("e", 0.0), rel(dev["e"][b, k, :n], lq["e"][k]))
    assert max(worst.values()) < 1e-11, worst
