"""ORACLE - test infrastructure, NOT product code: one iteration of the reference's second solver, GaussNewtonDDP with `algorithm ILQR`
(ocs2_bipedal_robot_ros/src/BipedalRobotDdpMpcNode.cpp:70-71: GaussNewtonDDP_MPC(mpcSettings, ddpSettings, getRollout(), ocp, initializer);
settings task.info:115-156), restated in numpy on top of the C++ oracle's model functions.

PARITY STATUS: UNPINNED, like the SQP oracle - /root/reference holds neither ocs2_ddp nor a vector of its output; every step below is
[OCS2-upstream, recalled] and has a named test in tests/test_recalled_behaviours.py.  What is restated, and what is NOT:

  restated (one ILQR iteration = ddp.maxNumIterations 1, the configured value)
    1. nominal trajectories: cold start = the Initializer on the time grid (x = x0, u = weight compensation; InitializerRollout)
    2. LQ approximation at every node of the nominal time trajectory, continuous time (LinearQuadraticApproximator: flow-map Jacobians,
       Gauss-Newton cost incl. the soft friction cones, state-input equality constraints)
    3. ILQR::discreteLQWorker: Euler discretisation  A = I + dt A_c, B = dt B_c, cost x dt, constraints unchanged, no dynamics bias
    4. hessian_correction::shiftHessian, DIAGONAL_SHIFT: Hm = R + B' S B gets hessianCorrectionMultiple on every diagonal entry, always
    5. the equality-constrained minimisation of the stage's Q-function (upstream: Hm-weighted projectors DmDagger / null projector, which need
       D of full row rank - there every exact method gives the same policy; D of this robot NEVER has it while a foot stands (two contact
       points on a rigid foot: 6 zero-velocity rows of rank 5): upstream's projectors do not exist and the policy is DEFINED by the pivoted
       elimination of the SQP path) and the discrete Riccati
       recursion from S_N = 0 (no terminal cost in this problem)
    6. LINE_SEARCH: step lengths maxStepLength * contractionRate^i >= minStepLength, each a TimeTriggeredRollout (rollout block of
       task.info: ODE45, adaptive) of the closed loop u = u_nom + alpha lff + K (x - x_nom) from the measured state; performance index =
       trapezoidal integral of the cost over the roll-out's own time points (no state-only constraints here: merit = cost); the baseline
       is the roll-out with step length 0 (new gains, no feedforward increment); Armijo test merit < merit_baseline - armijoCoefficient
       alpha int |lff|^2 dt; the largest accepted step wins; none: the baseline roll-out is the solution
    7. the primal solution is the accepted roll-out on ITS time points with a FeedforwardController (ddp.useFeedbackPolicy false)
  NOT restated
    * SLQ / the continuous-time backward pass (backwardPassIntegratorType ODE45 belongs to it; ILQR does not integrate Riccati equations)
    * later iterations on the adaptive roll-out grid (maxNumIterations > 1), the constraint penalty schedule (constraintPenaltyInitialValue /
      IncreaseRate act on state-only constraints, of which this problem has none), LEVENBERG_MARQUARDT, multi-threaded line search order
      (upstream evaluates the step lengths concurrently and keeps the largest accepted one: the result is the same)
"""
import numpy as np

from oracle import reference_py as rp

ARMIJO_COEFFICIENT = 1e-4          # [OCS2-upstream] line_search::Settings default
CONTRACTION_RATE = 0.5             # [OCS2-upstream] line_search::Settings default


def euler_lq(om, nodes, x, u):
    """Steps 2 + 3 per node: dict of lists A, B, Q, R, P, q, r, c, C, D, e (constraints cut to their nc rows)."""
    N, nx, nu = int(nodes["N"]), om.nx, om.nu
    out = {k: [] for k in ("A", "B", "Q", "R", "P", "q", "r", "c", "C", "D", "e")}
    for k in range(N):
        if nodes["kind"][k] == 1:                      # event node: identity jump map, no input, no cost (as the SQP transcription)
            lq = dict(A=np.eye(nx), B=np.zeros((nx, nu)), Q=np.zeros((nx, nx)), R=np.zeros((nu, nu)), P=np.zeros((nu, nx)), q=np.zeros(nx),
                      r=np.zeros(nu), c=0.0, C=np.zeros((0, nx)), D=np.zeros((0, nu)), e=np.zeros(0))
        else:
            dt = float(nodes["dt"][k])
            o = om.node_lq(0, dt, x[k], u[k], x[k + 1], nodes["xref"][k], int(nodes["mode"][k]), nodes["zref"][k], nodes["zdref"][k])
            _, Ac, Bc = om.flow_map(x[k], u[k], lin=True)
            nc = o["nc"]
            lq = dict(A=np.eye(nx) + dt * Ac, B=dt * Bc, Q=o["Q"], R=o["R"], P=o["P"], q=o["q"], r=o["r"], c=o["c"], C=o["C"][:nc], D=o["D"][:nc],
                      e=o["e"][:nc])                   # (cost x dt comes with node_lq: the transcription scales it the same way)
        for key in out:
            out[key].append(lq[key])
    return out


def constrained_stage(Hm, G, g, C, D, e, method="lu"):
    """min_du 0.5 du' Hm du + du' (G dx + g)  s.t.  C dx + D du + e = 0  ->  du = K dx + lff.
    method "lu": the parametrisation du = Px dx + Pe + Pu w of the SQP path (FullPivLU restatement of the C++ oracle, oracle_lu_projection) - it
    also DEFINES the answer where D loses row rank (double support: 12 rows of rank 10, single support: 14 of rank 13; the dependent rows
    of [C | D | e] are dropped by the pivoting), where upstream's Hm-weighted pseudo-inverse does not exist.  method "pinv": pseudo-inverse + null space, the textbook solution;
    with D of full row rank both are the unique constrained minimiser (tests/test_ddp_oracle.py)."""
    nu = Hm.shape[0]
    if D.shape[0] == 0:
        Kp, lp, Z = np.zeros((nu, C.shape[1])), np.zeros(nu), np.eye(nu)
    elif method == "lu":
        from oracle import oracle_py
        Kp, Z, lp, _ = oracle_py.lu_projection(C, D, e)
    else:
        U, s, Vt = np.linalg.svd(D)
        rank = int(np.sum(s > 1e-9 * max(1.0, s[0])))
        Dp = Vt[:rank].T @ np.diag(1.0 / s[:rank]) @ U[:, :rank].T
        Z = Vt[rank:].T
        Kp, lp = -Dp @ C, -Dp @ e                      # particular solution
    if Z.shape[1] == 0:
        return Kp, lp
    Hz = Z.T @ Hm @ Z
    Kz = -np.linalg.solve(Hz, Z.T @ (G + Hm @ Kp))
    lz = -np.linalg.solve(Hz, Z.T @ (g + Hm @ lp))
    return Kp + Z @ Kz, lp + Z @ lz


def backward_pass(lq, nodes, shift):
    """Steps 4 + 5: gains K[k], feedforward lff[k] (zero at event nodes), and the value function at node 0."""
    N = int(nodes["N"])
    nx, nu = lq["A"][0].shape[0], lq["B"][0].shape[1]
    S, s = np.zeros((nx, nx)), np.zeros(nx)
    K, lff = np.zeros((N, nu, nx)), np.zeros((N, nu))
    for k in range(N - 1, -1, -1):
        A, B = lq["A"][k], lq["B"][k]
        if nodes["kind"][k] == 1:
            S, s = A.T @ S @ A, A.T @ s                # identity: unchanged
            continue
        Hm = lq["R"][k] + B.T @ S @ B + shift * np.eye(nu)
        G = lq["P"][k] + B.T @ S @ A
        g = lq["r"][k] + B.T @ s
        K[k], lff[k] = constrained_stage(Hm, G, g, lq["C"][k], lq["D"][k], lq["e"][k])
        Sn = lq["Q"][k] + A.T @ S @ A + K[k].T @ Hm @ K[k] + K[k].T @ G + G.T @ K[k]
        s = lq["q"][k] + A.T @ s + K[k].T @ Hm @ lff[k] + K[k].T @ g + G.T @ lff[k]
        S = 0.5 * (Sn + Sn.T)
    return K, lff, S, s


def cost_rate(om, model, t, x, u, event_times, mode_sequence, target_times, target_states):
    """Intermediate cost L(t, x, u) of the problem (tracking + soft cones): the node metric of the transcription with dt = 1."""
    mode = rp.mode_at_time(event_times, mode_sequence, t)
    xref = rp.interpolate_target(target_times, target_states, t)
    z = np.zeros(4)
    return float(om.node_perf(0, 1.0, x, u, x, xref, mode, z, z)[0])


def trajectory_cost(om, model, times, xs, us, event_times, mode_sequence, target_times, target_states):
    c = np.array([cost_rate(om, model, t, x, u, event_times, mode_sequence, target_times, target_states) for t, x, u in zip(times, xs, us)])
    return float(np.sum(0.5 * (c[1:] + c[:-1]) * np.diff(times)))          # trapezoidalIntegration


def step_lengths(ddp):
    out, a = [], float(ddp["maxStepLength"])
    while a >= float(ddp["minStepLength"]):
        out.append(a)
        a *= CONTRACTION_RATE
    return out


def nominal_rollout(om, nodes, x_measured, x_shifted, u_shifted, event_times, rollout):
    """Nominal trajectories of a warm MPC tick.  [OCS2-upstream, recalled] GaussNewtonDDP::rolloutInitialTrajectory: the controller of the previous run
    (ddp.useFeedbackPolicy false: a FeedforwardController, here the input trajectory already shifted onto the new grid - warm_start_from_previous with
    feedback=False -, the initializer's input beyond its end) is rolled out from the measured state with TimeTriggeredRollout; the state trajectory of
    that roll-out is the nominal one.  The backward pass of this engine works on the shooting grid: node k takes LinearInterpolation(t_k) of the
    roll-out (the first node the measured state).  Follows csrc/k_ddp.hip k_ddp_nominal / solver.hip ddp_nominal_rollout.  Returns x_nom [N + 1, nx]."""
    N = int(nodes["N"])
    tp = np.asarray(nodes["times"], float)
    tpa, _, uff, KK = rp.primal_solution_arrays(nodes, x_shifted, u_shifted, np.zeros((N, om.nu, om.nx)))
    ctrl = lambda t, x: rp.linear_controller_input(tpa, uff, KK, t, x)      # K = 0: the interpolated input trajectory
    ro = rp.time_triggered_rollout(lambda x, u: om.flow_map(x, u), ctrl, float(tp[0]), x_measured, float(tp[-1]), list(event_times), rollout)
    x_nom = np.array(x_shifted, float)
    for k in range(N + 1):
        i, al = rp.time_segment(ro["times"], float(tp[k]))
        x_nom[k] = al * ro["states"][i] + (1.0 - al) * ro["states"][i + 1]
    x_nom[0] = x_measured
    return x_nom


def ilqr_iteration(om, model, nodes, x_measured, x_nom, u_nom, event_times, mode_sequence, target_times, target_states, ddp, rollout):
    """One GaussNewtonDDP / ILQR iteration.  Returns dict(alpha, times, states, inputs, K, lff, merit0, merits, update_is)."""
    N = int(nodes["N"])
    lq = euler_lq(om, nodes, x_nom, u_nom)
    K, lff, S0, s0 = backward_pass(lq, nodes, float(ddp["hessianCorrectionMultiple"]))
    update_is = float(sum(nodes["dt"][k] * lff[k] @ lff[k] for k in range(N) if nodes["kind"][k] == 0))
    tp = np.asarray(nodes["times"], float)
    flow = lambda x, u: om.flow_map(x, u)

    def closed_loop(alpha):
        tpa, _, uff, KK = rp.primal_solution_arrays(nodes, x_nom, u_nom + alpha * lff, K)
        ctrl = lambda t, x: rp.linear_controller_input(tpa, uff, KK, t, x)
        ro = rp.time_triggered_rollout(flow, ctrl, float(tp[0]), x_measured, float(tp[-1]), list(event_times), rollout)
        return ro, trajectory_cost(om, model, ro["times"], ro["states"], ro["inputs"], event_times, mode_sequence, target_times, target_states)

    base, merit0 = closed_loop(0.0)        # the baseline of the search: the roll-out under the new gains with NO feedforward increment (step length 0)
    best, merits = None, []
    for alpha in step_lengths(ddp):
        ro, merit = closed_loop(alpha)
        merits.append(merit)
        if best is None and merit < merit0 - ARMIJO_COEFFICIENT * alpha * update_is:
            best = (alpha, ro)
    alpha, ro = best if best is not None else (0.0, base)
    return dict(alpha=alpha, times=ro["times"], states=ro["states"], inputs=ro["inputs"], K=K, lff=lff, merit0=merit0, merits=merits, update_is=update_is)
