"""One named test per piece of [OCS2-upstream] behaviour that this repository RECALLS instead of reading (OCS2, odeint, Eigen and HPIPM are
not vendored in the reference and cannot be fetched here; DESIGN.md section 5 lists the items).  Each test pins the CHOSEN behaviour of the
oracle - the thing the HIP path is held to - so that a correction, once someone can check the real source, is one visible diff: change the
oracle, update the test named after the item, and the GPU parity tests follow.  These tests do not claim the choice is right."""
import numpy as np
import pytest

from oracle import oracle_py, reference_py as rp
from tests import oracle_bridge as ob


def test_recalled_time_discretization_merges_nodes_closer_than_dt_min():
    """timeDiscretizationWithEvents: a node that would land within dt_min (10 x numeric_limits epsilon-scale LIMIT) of the previous
    one overwrites it instead of being appended (an event 1e-12 before a grid point leaves ONE node pair, not a sliver interval)."""
    g = rp.time_discretization_with_events(0.0, 0.06, 0.015, [0.03 - 1e-12])
    times = [t for t, _ in g]
    kinds = [e for _, e in g]
    assert kinds.count(rp.EVENT_PRE) == 1 and kinds.count(rp.EVENT_POST) == 1
    assert min(b - a for a, b in zip(times, times[1:]) if b > a) > 1e-3            # no sliver interval survives
    # an event exactly on a grid point: pre / post pair at that time, grid continues from the event time
    g2 = rp.time_discretization_with_events(0.0, 0.06, 0.015, [0.03])
    assert [round(t, 12) for t, _ in g2] == [0.0, 0.015, 0.03, 0.03, 0.045, 0.06]
    # the grid restarts at an event: steps of dt are counted from the event, the last interval is the remainder
    g3 = rp.time_discretization_with_events(0.0, 0.06, 0.015, [0.02])
    assert [round(t, 12) for t, _ in g3] == [0.0, 0.015, 0.02, 0.02, 0.035, 0.05, 0.06]


def test_recalled_interval_nudges_use_weak_epsilon():
    """getIntervalStart / getIntervalEnd: post-event nodes start weakEpsilon (1e-6 scale) late, pre-event nodes end weakEpsilon early;
    the mode of an interval is looked up at its (nudged) start; a time exactly on an event belongs to the EARLIER mode."""
    assert rp.interval_start((0.35, rp.EVENT_POST)) == 0.35 + rp.WEAK_EPS and rp.interval_end((0.35, rp.EVENT_PRE)) == 0.35 - rp.WEAK_EPS
    assert rp.interval_start((0.35, rp.EVENT_NONE)) == 0.35
    assert rp.mode_at_time([0.35], [1, 2], 0.35) == 1 and rp.mode_at_time([0.35], [1, 2], 0.35 + 1e-9) == 2


def test_recalled_event_node_performance_has_no_dt_factor():
    """An event node contributes |x - x_next|^2 to the dynamics SSE WITHOUT the interval's dt factor (intermediate nodes: dt x)."""
    m, om = ob.h1_model(), ob.h1_oracle()
    x = m["initial_state"]
    xn = x.copy(); xn[8] += 0.1
    u = rp.weight_compensating_input(m, 3)
    z = np.zeros(4)
    ev = om.node_lq(1, 0.015, x, u, xn, x, 3, z, z)
    assert abs(ev["perf"][1] - 0.01) < 1e-15 and ev["perf"][0] == 0.0 and ev["perf"][2] == 0.0 and ev["nc"] == 0
    im = om.node_lq(0, 0.015, x, u, xn, x, 3, z, z)
    assert abs(im["perf"][1] - 0.015 * float(im["b"] @ im["b"])) < 1e-15


def test_recalled_full_piv_lu_rank_threshold_and_first_pivot_rule():
    """Eigen::FullPivLU as used by luConstraintProjection: complete pivoting takes the FIRST maximum in column-major order; the rank
    counts pivots above eps * min(rows, cols) * |largest pivot|; free variables of solve() are zero."""
    D = np.zeros((3, 5)); D[0, 1] = 2.0; D[1, 3] = 2.0; D[2, 0] = 1.0; D[2, 1] = 1e-17        # two equal maxima: (0,1) comes first column-major
    C = np.zeros((3, 4)); e = np.array([1.0, 2.0, 3.0])
    Px, Pu, Pe, rank = oracle_py.lu_projection(C, D, e)
    assert rank == 3 and Pu.shape == (5, 2) and np.allclose(D @ Pu, 0.0, atol=1e-15) and np.allclose(D @ Pe + e, 0.0, atol=1e-15)
    assert Pe[2] == 0.0 and Pe[4] == 0.0                                                    # free variables stay zero
    D2 = np.array([[1.0, 0.0, 0.0], [0.0, 1e-17, 0.0]])                                     # second pivot below eps * 2 * 1
    assert oracle_py.lu_projection(np.zeros((2, 2)), D2, np.zeros(2))[3] == 1               # rank 1
    D3 = np.array([[1.0, 0.0, 0.0], [0.0, 1e-15, 0.0]])                                     # above the threshold 4.4e-16
    assert oracle_py.lu_projection(np.zeros((2, 2)), D3, np.zeros(2))[3] == 2


def _stance_solve(**kw):
    from bipedal_control_amd import scenarios as sc
    itf = sc.h1_interface()
    prob = sc.stance_problem(itf, 8)
    m, om = ob.h1_model(), ob.h1_oracle()
    nodes = ob.oracle_nodes(prob, 0)
    xi, ui = rp.cold_start(m, nodes, prob["x0"][0])
    return m, om, nodes, prob, xi, ui


def test_recalled_filter_line_search_acceptance_rules():
    """FilterLinesearch::acceptStep as restated: violation > g_max -> accept iff it shrinks by (1 - gamma_c); both violations < g_min and
    a descent direction -> Armijo on the merit; otherwise accept iff merit OR violation improves.  With g_max tiny the first branch
    decides, with g_min huge the Armijo branch does - the reported step sizes differ accordingly on the same problem."""
    m, om, nodes, prob, xi, ui = _stance_solve()
    s = m["sqp"]
    a = om.solve(nodes, prob["x0"][0], xi, ui, iterations=1, g_max=s["g_max"], g_min=s["g_min"], delta_tol=s["deltaTol"])[3]
    assert a[0][3] == 1.0 and a[0][10] == 1                                                # full step at the first trial
    # Armijo branch: violation counted as negligible on both sides (g_min above it): the step must still decrease the merit
    b = om.solve(nodes, prob["x0"][0], xi, ui, iterations=1, g_max=1e9, g_min=1e9, delta_tol=s["deltaTol"])[3]
    assert b[0][3] > 0.0 and b[0][4] < b[0][0] + 1e-4 * b[0][3] * b[0][7] + 1e-15


def test_recalled_backtracking_stops_below_delta_tol():
    """SqpSolver::takeStep: after a rejected trial, alpha is halved and the search gives up (step size 0) as soon as alpha |du| and
    alpha |dx| are both below deltaTol - it does not walk down to alpha_min (advisor finding r01)."""
    m, om, nodes, prob, xi, ui = _stance_solve()
    # an always-rejecting filter: g_max < 0 sends every trial into the first branch (violation > g_max), gamma_c = 1 demands a
    # violation below zero.  With deltaTol above the step norm the first rejected trial ends the search; with deltaTol = 0 the search
    # walks down to alpha_min = 1e-4: 1, 1/2, ... 2^-13 = 14 trials
    st = om.solve(nodes, prob["x0"][0], xi, ui, iterations=1, g_max=-1.0, g_min=-1.0, gamma_c=1.0, delta_tol=1e9)[3]
    assert st[0][10] == 1 and st[0][3] == 0.0                                              # one trial, rejected, search abandoned
    st2 = om.solve(nodes, prob["x0"][0], xi, ui, iterations=1, g_max=-1.0, g_min=-1.0, gamma_c=1.0, delta_tol=0.0)[3]
    assert st2[0][10] == 14 and st2[0][3] == 0.0
    # in between: the search stops at the first alpha with alpha |dx| and alpha |du| both below deltaTol
    dx, du, _ = om.qp_step(nodes, prob["x0"][0], xi, ui)
    nrm = max(np.linalg.norm(dx), np.linalg.norm(du))
    st3 = om.solve(nodes, prob["x0"][0], xi, ui, iterations=1, g_max=-1.0, g_min=-1.0, gamma_c=1.0, delta_tol=nrm / 6.0)[3]
    assert st3[0][10] == 3                                                                 # 1, 1/2, 1/4 tried; 1/8 |d| < |d| / 6


def test_recalled_rollout_first_interval_begin_is_nudged_too():
    """RolloutBase::findActiveModesTimeInterval: EVERY begin time, the first included, is moved by weakEpsilon (never past its end)."""
    iv = rp.find_active_modes_time_interval(0.1, 0.5, [0.0, 0.3, 0.7])
    assert iv == [(0.1 + rp.WEAK_EPS, 0.3), (0.3 + rp.WEAK_EPS, 0.5)]
    assert rp.find_active_modes_time_interval(0.3, 0.3 + 0.5 * rp.WEAK_EPS, [])[0][0] == 0.3 + 0.5 * rp.WEAK_EPS   # clipped at the end


def test_recalled_odeint_controller_constants():
    """boost::numeric::odeint controlled dopri5: error norm max |e_i| / (abs + rel (|x_i| + dt |dxdt_i|)) (a_x = a_dxdt = 1); reject -> dt x
    max(0.9 err^(-1/3), 0.2); accept with err < 0.5 -> dt x 0.9 max(err, 5^-5)^(-1/5), i.e. growth capped at 4.5 x."""
    f = lambda t, x: -50.0 * x                                                              # noqa: E731
    seen = []
    x, t, acc, rej = rp.integrate_adaptive_dopri5(f, np.array([1.0]), 0.0, 0.2, 0.015, 1e-5, 1e-3, lambda x_, t_: seen.append(t_))
    steps = np.diff(seen)
    assert acc == len(steps) and abs(x[0] - np.exp(-10.0)) < 1e-4
    assert np.all(steps[1:] / steps[:-1] <= 4.5 + 1e-12)
    # a first step that is far too long is rejected and shrunk by at most a factor 5 per rejection
    seen3 = []
    _, _, acc3, rej3 = rp.integrate_adaptive_dopri5(f, np.array([1.0]), 0.0, 0.2, 0.2, 1e-5, 1e-3, lambda x_, t_: seen3.append(t_))
    assert rej3 >= 1 and seen3[1] - seen3[0] >= 0.2 * 0.2 ** rej3 - 1e-15
    # a problem with a vanishing error estimate grows by exactly 0.9 * 5 = 4.5 per accepted step until the end time clips it
    seen2 = []
    rp.integrate_adaptive_dopri5(lambda t, x: 0.0 * x, np.array([1.0]), 0.0, 1.0, 0.01, 1e-5, 1e-3, lambda x_, t_: seen2.append(t_))
    d2 = np.diff(seen2)
    assert abs(d2[1] / d2[0] - 4.5) < 1e-12


def test_recalled_warm_start_interpolation_limits():
    """SqpSolver::initializeStateInputTrajectories with a previous PrimalSolution: a node is interpolated only while its interval start
    lies at or before the SECOND-TO-LAST previous time and its interval end at or before the LAST; beyond that the initializer is used."""
    m, om, nodes, prob, xi, ui = _stance_solve()
    x1, u1, K1, _ = om.solve(nodes, prob["x0"][0], xi, ui, iterations=1, g_max=1e-2, g_min=1e-6, delta_tol=1e-4)
    from bipedal_control_amd import scenarios as sc
    itf = sc.h1_interface()
    shift = 0.02
    prob2 = dict(prob, t0=shift, targets=[bp_t for bp_t in prob["targets"]])
    nodes2 = ob.oracle_nodes(prob2, 0)
    xw, uw = rp.warm_start_from_previous(m, nodes2, x1[1], nodes, x1, u1, K1)
    unom = rp.weight_compensating_input(m, 3)
    tp_last, tp_second_last = nodes["times"][-1], nodes["times"][-2]
    for i in range(nodes2["N"]):
        t_start, t_end = nodes2["times"][i], nodes2["times"][i + 1]
        inside = t_start <= tp_second_last and t_end <= tp_last
        if not inside:
            assert np.array_equal(uw[i], unom), i                                          # initializer: weight compensation, x_next = x
            assert np.array_equal(xw[i + 1], xw[i]), i
    assert any(not (nodes2["times"][i] <= tp_second_last and nodes2["times"][i + 1] <= tp_last) for i in range(nodes2["N"]))


def test_recalled_qp_regularisation_is_a_named_option():
    """HPIPM's reg_prim (1e-12 added to the diagonal of the stage Hessians before factorisation, [OCS2-upstream] hpipm_catkin settings) is
    NOT applied by default; oracle and product offer it as `reg_prim` so that its effect is a measured number."""
    m, om, nodes, prob, xi, ui = _stance_solve()
    dx0, du0, K0 = om.qp_step(nodes, prob["x0"][0], xi, ui)
    dx1, du1, K1 = om.qp_step(nodes, prob["x0"][0], xi, ui, reg_prim=1e-12)
    rel = lambda a, b: float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))              # noqa: E731
    # measured: 5e-9 on du (the last stages have H = R~ only, dt x 5e-3 on the force block: 1e-12 / 7.5e-5), below the 1e-8 of the solve parity
    assert 1e-10 < rel(du1, du0) < 2e-8 and rel(dx1, dx0) < 2e-8 and rel(K1, K0) < 1e-7


# ---- the DDP slice (oracle/ddp_py.py; GaussNewtonDDP with algorithm ILQR, BipedalRobotDdpMpcNode.cpp:70-71, task.info:115-156) ----------------
def _ddp_case(n=6):
    from bipedal_control_amd import scenarios
    from oracle import ddp_py
    itf = scenarios.h1_interface()
    prob = scenarios.stance_problem(itf, n)
    m, om = ob.h1_model(), ob.h1_oracle()
    nodes = ob.oracle_nodes(prob, 0)
    x0 = prob["x0"][0].copy(); x0[8] -= 0.02; x0[0] += 0.03
    x_nom, u_nom = rp.cold_start(m, nodes, x0)
    return ddp_py, prob, m, om, nodes, x0, x_nom, u_nom


def test_recalled_ilqr_discretises_the_continuous_model_with_one_euler_step():
    """ILQR::discreteLQWorker: A = I + dt A_c, B = dt B_c with the flow-map Jacobians at the nominal point (NOT the RK2 sensitivities of the
    multiple-shooting transcription), the cost is scaled by dt exactly as there, the constraints are not scaled."""
    ddp_py, prob, m, om, nodes, x0, x_nom, u_nom = _ddp_case()
    lq = ddp_py.euler_lq(om, nodes, x_nom, u_nom)
    k, dt = 2, float(nodes["dt"][2])
    _, Ac, Bc = om.flow_map(x_nom[k], u_nom[k], lin=True)
    assert np.array_equal(lq["A"][k], np.eye(m["nx"]) + dt * Ac) and np.array_equal(lq["B"][k], dt * Bc)
    ms = om.node_lq(0, dt, x_nom[k], u_nom[k], x_nom[k + 1], nodes["xref"][k], int(nodes["mode"][k]), nodes["zref"][k], nodes["zdref"][k])
    assert np.abs(ms["A"] - lq["A"][k]).max() > 1e-6                                   # the transcription's RK2 matrix is another one
    assert np.array_equal(ms["Q"], lq["Q"][k]) and np.array_equal(ms["R"], lq["R"][k]) and np.array_equal(ms["C"][:ms["nc"]], lq["C"][k])


def test_recalled_ddp_has_no_dynamics_bias_and_no_terminal_cost():
    """The nominal trajectory of a DDP iteration is treated as a roll-out: the LQ model carries no defect term b (the Initializer's constant-state
    trajectory of a cold start included), and this problem has no final cost: the recursion starts from S_N = 0, s_N = 0."""
    ddp_py, prob, m, om, nodes, x0, x_nom, u_nom = _ddp_case()
    lq = ddp_py.euler_lq(om, nodes, x_nom, u_nom)
    assert "b" not in lq
    N = int(nodes["N"])
    K, lff, S0, s0 = ddp_py.backward_pass(lq, nodes, 0.0)
    # the last stage sees only its own cost: its unconstrained part of the policy does not depend on anything behind it
    Kl, ll = ddp_py.constrained_stage(lq["R"][N - 1], lq["P"][N - 1], lq["r"][N - 1], lq["C"][N - 1], lq["D"][N - 1], lq["e"][N - 1])
    assert np.array_equal(K[N - 1], Kl) and np.array_equal(lff[N - 1], ll)


def test_recalled_hessian_correction_diagonal_shift_is_unconditional():
    """hessian_correction::shiftHessian with DIAGONAL_SHIFT adds hessianCorrectionMultiple (task.info:152: 1e-5) to EVERY diagonal entry of
    Hm = R + B' S B, whether Hm is positive definite or not; it enters the stage problem like an input weight (the value function is the one
    of the shifted problem)."""
    ddp_py, prob, m, om, nodes, x0, x_nom, u_nom = _ddp_case()
    assert m["ddp"]["hessianCorrectionStrategy"] == "DIAGONAL_SHIFT" and m["ddp"]["hessianCorrectionMultiple"] == 1e-5
    lq = ddp_py.euler_lq(om, nodes, x_nom, u_nom)
    K0, l0, _, _ = ddp_py.backward_pass(lq, nodes, 0.0)
    K1, l1, _, _ = ddp_py.backward_pass(lq, nodes, 1e-5)
    lq2 = dict(lq, R=[R + 1e-5 * np.eye(m["nu"]) if nodes["kind"][k] == 0 else R for k, R in enumerate(lq["R"])])
    K2, l2, _, _ = ddp_py.backward_pass(lq2, nodes, 0.0)
    assert np.abs(K1 - K2).max() < 1e-9 * np.abs(K1).max() and np.abs(l1 - l2).max() < 1e-9 * max(1.0, np.abs(l1).max())
    assert np.abs(K1 - K0).max() > 0.0                                                 # (and it is not a no-op on a positive definite Hm)


def test_recalled_ddp_line_search_baseline_is_the_roll_out_with_step_length_zero():
    """LineSearchStrategy::run: the merit every step length is compared with belongs to a roll-out of the NEW feedback gains with no
    feedforward increment (step length 0) from the measured state - not to the nominal trajectories (after a cold start those are the
    Initializer's constant state, which is no trajectory of the system); Armijo: merit < baseline - 1e-4 * alpha * int |lff|^2 dt; the
    performance index is the trapezoidal integral of the cost over the roll-out's own (adaptive, ODE45) time points; no accepted step: the
    baseline roll-out is the solution."""
    ddp_py, prob, m, om, nodes, x0, x_nom, u_nom = _ddp_case(12)
    sched = prob["schedule"]
    ev, ms = list(map(float, sched.eventTimes)), list(map(int, sched.modeSequence))
    tt = prob["targets"][0]
    args = (ev, ms, np.asarray(tt.timeTrajectory), np.asarray(tt.stateTrajectory))
    r = ddp_py.ilqr_iteration(om, m, nodes, x0, x_nom, u_nom, *args, m["ddp"], m["rollout"])
    tp, xp, uff, KK = rp.primal_solution_arrays(nodes, x_nom, u_nom, r["K"])
    nominal = ddp_py.trajectory_cost(om, m, tp, xp, uff + np.einsum("kij,kj->ki", KK, xp), *args)
    assert abs(nominal - r["merit0"]) > 1e-3 * r["merit0"]                              # the baseline is NOT the nominal trajectories' index
    # merits approach the baseline as the step length goes to zero
    assert abs(r["merits"][-1] - r["merit0"]) < abs(r["merits"][0] - r["merit0"])
    assert ddp_py.ARMIJO_COEFFICIENT == 1e-4 and ddp_py.CONTRACTION_RATE == 0.5
    # an impossible Armijo test leaves the baseline roll-out as the solution
    old = ddp_py.ARMIJO_COEFFICIENT
    try:
        ddp_py.ARMIJO_COEFFICIENT = 1e12
        r0 = ddp_py.ilqr_iteration(om, m, nodes, x0, x_nom, u_nom, *args, m["ddp"], m["rollout"])
    finally:
        ddp_py.ARMIJO_COEFFICIENT = old
    assert r0["alpha"] == 0.0 and abs(ddp_py.trajectory_cost(om, m, r0["times"], r0["states"], r0["inputs"], *args) - r0["merit0"]) < 1e-12


def test_recalled_ddp_stance_rows_have_no_full_row_rank():
    """What upstream's projection (Hm-weighted pseudo-inverse of D: needs full row rank) cannot do for this robot: two contact points on a rigid
    foot give 6 zero-velocity rows of rank 5 - double support: 12 rows of rank 10, single support: 14 rows of rank 13 -, and in single support
    at a perturbed nominal point the dependent row of [C | D] is not even consistent (its state part does not vanish).  The slice DEFINES the
    policy there by the pivoted elimination of the SQP path (oracle_lu_projection)."""
    from bipedal_control_amd import scenarios
    ddp_py, prob, m, om, nodes, x0, x_nom, u_nom = _ddp_case()
    lq = ddp_py.euler_lq(om, nodes, x_nom, u_nom)
    D = lq["D"][1]
    assert D.shape[0] == 12 and np.linalg.matrix_rank(D) == 10
    prob = scenarios.trot_problem(scenarios.h1_interface(), batch=1, n_intervals=6, gait_start=0.0)
    nodes = ob.oracle_nodes(prob, 0)
    x_nom, u_nom = rp.cold_start(m, nodes, prob["x0"][0])
    lq = ddp_py.euler_lq(om, nodes, x_nom, u_nom)
    D = lq["D"][1]
    assert D.shape[0] == 14 and np.linalg.matrix_rank(D) == 13
    w = np.linalg.svd(D)[0][:, 13:]                                                     # the dependent combination of the rows
    assert np.abs(w.T @ D).max() < 1e-12 and np.abs(w.T @ lq["C"][1]).max() > 1e-3      # ... which the state part of the rows does not share


def test_recalled_ddp_warm_tick_nominal_states_are_the_roll_out_of_the_previous_controller():
    """[OCS2-upstream, recalled] GaussNewtonDDP::rolloutInitialTrajectory: with a controller from the previous run the nominal trajectories of a tick are
    that controller ROLLED OUT from the measured state - not the previous solution shifted in time.  Chosen form here (oracle/ddp_py.py nominal_rollout,
    csrc/k_ddp.hip k_ddp_nominal): the shifted input trajectory as a FeedforwardController, TimeTriggeredRollout, LinearInterpolation onto the shooting
    grid.  What it buys: the nominal trajectory starts AT the measured state and satisfies the dynamics (ILQR carries no dynamics bias), where the
    shifted solution keeps the disturbance as a defect at its first node."""
    import numpy as np
    from bipedal_control_amd import scenarios
    from oracle import ddp_py, reference_py as rp
    from tests import oracle_bridge as ob
    itf = scenarios.h1_interface()
    prob = scenarios.stance_problem(itf, 12)
    m, om = ob.model("h1"), ob.oracle("h1")
    nodes = ob.oracle_nodes(prob, 0)
    N = int(nodes["N"])
    x0 = prob["x0"][0].copy(); x0[8] -= 0.02
    xc, uc = rp.cold_start(m, nodes, x0)
    measured = x0.copy(); measured[2] += 0.05; measured[12] += 0.03            # a disturbance since the last tick
    x_nom = ddp_py.nominal_rollout(om, nodes, measured, xc, uc, [], m["rollout"])
    assert np.array_equal(x_nom[0], measured)

    def defect(x):      # one explicit Euler step per interval is enough to tell a trajectory of the dynamics from one that is not
        return max(float(np.abs(x[k + 1] - x[k] - nodes["dt"][k] * om.flow_map(x[k], uc[k])).max()) for k in range(N))
    shifted = xc.copy(); shifted[0] = measured
    assert defect(x_nom) < 0.2 * defect(shifted)      # (what remains is the difference between one Euler step and the ODE45 roll-out)
