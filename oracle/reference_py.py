"""ORACLE (test infrastructure, not product code): pure-Python restatement of the host-side pre-pass of one MPC
solve - rows a10, a11, a12 of SURVEY.md section 8 plus the time grid and target trajectories (A.5-A.8).

Each function cites the reference file:line it follows (paths relative to /root/reference).  [OCS2-upstream]
marks behaviour of un-vendored OCS2 (`ocs2_core/misc/Lookup.h`, `ocs2_oc/oc_data/TimeDiscretization.cpp`,
`ocs2_core/misc/LinearInterpolation.h`, `ocs2_core/reference/ModeSchedule.cpp`) restated from the published sources.
Parity status: UNPINNED (no reference fixtures exist); pinned by the hand-derived known answers of SURVEY.md
section 8(c)(4) in tests/test_reference_prepass.py.
"""
import bisect
import math

import numpy as np

STANCE = 3
EVENT_NONE, EVENT_PRE, EVENT_POST = 0, 1, 2
LIMIT_EPS = 1e-9  # [OCS2-upstream] numeric_traits::limitEpsilon
WEAK_EPS = 1e-6   # [OCS2-upstream] numeric_traits::weakEpsilon


def mode_flags(mode):
    """modeNumber2StanceLeg, include/ocs2_bipedal_robot/gait/MotionPhaseDefinition.h:57-76."""
    return {0: (False, False, False, False), 1: (True, True, False, False), 2: (False, False, True, True),
            3: (True, True, True, True)}[mode]


def find_index_in_time_array(times, t):
    """[OCS2-upstream] lookup::findIndexInTimeArray = std::lower_bound."""
    return bisect.bisect_left(times, t)


def mode_at_time(event_times, modes, t):
    """[OCS2-upstream] ModeSchedule::modeAtTime: t exactly on an event belongs to the earlier mode."""
    return modes[find_index_in_time_array(event_times, t)]


class GaitSchedule:
    """src/gait/GaitSchedule.cpp:40-137."""

    def __init__(self, event_times, mode_sequence, template, phase_transition_stance_time):
        self.event_times = list(event_times)
        self.mode_sequence = list(mode_sequence)
        self.template = (list(template[0]), list(template[1]))
        self.phase_transition_stance_time = phase_transition_stance_time

    def insert_mode_sequence_template(self, template, start_time, final_time):  # :46-73
        self.template = (list(template[0]), list(template[1]))
        ev, ms = self.event_times, self.mode_sequence
        index = bisect.bisect_left(ev, start_time)
        if index < len(ev):
            del ev[index:]
            del ms[index + 1:]
        stance_time = self.phase_transition_stance_time
        if ms and ms[-1] == STANCE:
            stance_time = 0.0
        if stance_time > 0.0:
            ev.append(start_time)
            ms.append(STANCE)
        self._tile(start_time + stance_time, final_time)

    def get_mode_schedule(self, lower, upper):  # :78-102
        ev, ms = self.event_times, self.mode_sequence
        index = bisect.bisect_left(ev, lower)
        if index > 0:
            del ev[:index - 1]
            del ms[:index - 1]
            ms[0] = STANCE
        tiling_start = upper if not ev else ev[-1]
        del ev[-1:]
        del ms[-1:]
        self._tile(tiling_start, upper)
        return list(ev), list(ms)

    def _tile(self, start_time, final_time):  # :107-137
        ev, ms = self.event_times, self.mode_sequence
        t_times, t_modes = self.template
        if len(t_modes) == 0:
            return
        if ev and start_time <= ev[-1]:
            raise RuntimeError("The initial time for template-tiling is not greater than the last event time.")
        ev.append(start_time)
        while ev[-1] < final_time:
            for i in range(len(t_modes)):
                ms.append(t_modes[i])
                ev.append(ev[-1] + (t_times[i + 1] - t_times[i]))
        ms.append(STANCE)


class CubicSpline:
    """src/foot_planner/CubicSpline.cpp:38-72."""

    def __init__(self, start, end):  # nodes are (time, position, velocity)
        self.t0, self.t1 = start[0], end[0]
        self.dt = end[0] - start[0]
        dp = end[1] - start[1]
        dv = end[2] - start[2]
        self.dc0 = 0.0
        self.dc1 = start[2]
        self.dc2 = -(3.0 * start[2] + dv)
        self.dc3 = 2.0 * start[2] + dv
        self.c0 = self.dc0 * self.dt + start[1]
        self.c1 = self.dc1 * self.dt
        self.c2 = self.dc2 * self.dt + 3.0 * dp
        self.c3 = self.dc3 * self.dt - 2.0 * dp

    def position(self, t):
        tn = (t - self.t0) / self.dt
        return self.c3 * tn * tn * tn + self.c2 * tn * tn + self.c1 * tn + self.c0

    def velocity(self, t):
        tn = (t - self.t0) / self.dt
        return (3.0 * self.c3 * tn * tn + 2.0 * self.c2 * tn + self.c1) / self.dt


class SplineCpg:
    """src/foot_planner/SplineCpg.cpp:38-53."""

    def __init__(self, lift_off, mid_height, touch_down):
        self.mid_time = (lift_off[0] + touch_down[0]) / 2
        self.left = CubicSpline(lift_off, (self.mid_time, mid_height, 0.0))
        self.right = CubicSpline((self.mid_time, mid_height, 0.0), touch_down)

    def position(self, t):
        return self.left.position(t) if t < self.mid_time else self.right.position(t)

    def velocity(self, t):
        return self.left.velocity(t) if t < self.mid_time else self.right.velocity(t)


class SwingTrajectoryPlanner:
    """src/foot_planner/SwingTrajectoryPlanner.cpp:50-219 (terrain height 0, SwitchedModelReferenceManager.cpp:66-67)."""

    def __init__(self, cfg, num_feet=4):
        self.cfg = cfg
        self.num_feet = num_feet
        self.events = None
        self.traj = None

    def update(self, event_times, mode_sequence, terrain_height=0.0):
        n = len(mode_sequence)
        flags = [[mode_flags(mode_sequence[p])[j] for p in range(n)] for j in range(self.num_feet)]
        self.traj = []
        for j in range(self.num_feet):
            row = []
            for p in range(n):
                if not flags[j][p]:
                    start, final = self._find_index(p, flags[j])
                    if start < 0:
                        raise RuntimeError("The time of take-off for the first swing of the EE with ID %d is not defined." % j)
                    if final >= n - 1:
                        raise RuntimeError("The time of touch-down for the last swing of the EE with ID %d is not defined." % j)
                    t_start, t_final = event_times[start], event_times[final]
                    scaling = min(1.0, (t_final - t_start) / self.cfg["swingTimeScale"])
                    lift_off = (t_start, terrain_height, scaling * self.cfg["liftOffVelocity"])
                    touch_down = (t_final, terrain_height, scaling * self.cfg["touchDownVelocity"])
                    mid = min(terrain_height, terrain_height) + scaling * self.cfg["swingHeight"]
                    row.append(SplineCpg(lift_off, mid, touch_down))
                else:
                    row.append(SplineCpg((0.0, terrain_height, 0.0), terrain_height, (1.0, terrain_height, 0.0)))
            self.traj.append(row)
        self.events = list(event_times)

    @staticmethod
    def _find_index(index, flag):  # :159-186
        n = len(flag)
        start = -1
        for ip in range(index - 1, -1, -1):
            if flag[ip]:
                start = ip
                break
        final = n - 1
        for ip in range(index + 1, n):
            if flag[ip]:
                final = ip - 1
                break
        return start, final

    def z_velocity(self, leg, t):  # :52-55
        return self.traj[leg][find_index_in_time_array(self.events, t)].velocity(t)

    def z_position(self, leg, t):  # :57-60
        return self.traj[leg][find_index_in_time_array(self.events, t)].position(t)


def time_discretization_with_events(t0, tf, dt, event_times, dt_min=10.0 * LIMIT_EPS):
    """[OCS2-upstream] ocs2_oc/oc_data/TimeDiscretization.cpp timeDiscretizationWithEvents.
    Returns [(time, event)], event in {NONE, PRE, POST}."""
    assert dt > 0 and tf > t0
    out = [(t0, EVENT_NONE)]
    next_event = find_index_in_time_array(event_times, t0)
    t_next = t0
    while out[-1][0] < tf:
        t_next = t_next + dt
        ev = EVENT_NONE
        if next_event < len(event_times) and t_next >= event_times[next_event]:
            t_next = event_times[next_event]
            ev = EVENT_PRE
            next_event += 1
        if t_next >= tf:
            t_next = tf
            ev = EVENT_NONE
        if t_next > out[-1][0] + dt_min:
            out.append((t_next, ev))
        else:  # points are close together -> overwrite the old point
            out[-1] = (t_next, ev)
        if ev == EVENT_PRE:
            out.append((t_next, EVENT_POST))
    return out


def interval_start(node):
    """[OCS2-upstream] getIntervalStart: post-event nodes are nudged +weakEpsilon."""
    return node[0] + WEAK_EPS if node[1] == EVENT_POST else node[0]


def interval_end(node):
    """[OCS2-upstream] getIntervalEnd: pre-event nodes are nudged -weakEpsilon."""
    return node[0] - WEAK_EPS if node[1] == EVENT_PRE else node[0]


def rot_zyx(zyx):
    z, y, x = zyx
    cz, sz, cy, sy, cx, sx = math.cos(z), math.sin(z), math.cos(y), math.sin(y), math.cos(x), math.sin(x)
    return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                     [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                     [-sy, cy * sx, cy * cx]])


def target_pose_to_target_trajectories(model, target_pose, t_now, x_now, t_reach):
    """bipedal_controllers/src/TargetTrajectoriesPublisher.cpp:40-58."""
    nx = model["nx"]
    cur = np.array(x_now[6:12], float)
    cur[2] = model["com_height"]
    cur[4] = 0.0
    cur[5] = 0.0
    xs = np.zeros((2, nx))
    xs[0, 6:12] = cur
    xs[0, 12:] = model["default_joint_state"]
    xs[1, 6:12] = target_pose
    xs[1, 12:] = model["default_joint_state"]
    return np.array([t_now, t_reach]), xs


def cmd_vel_to_target_trajectories(model, cmd_vel, t_now, x_now, time_to_target):
    """bipedal_controllers/src/TargetTrajectoriesPublisher.cpp:76-99 (TIME_TO_TARGET passed explicitly)."""
    cur = np.array(x_now[6:12], float)
    vrot = rot_zyx(cur[3:6]) @ np.asarray(cmd_vel[:3], float)
    target = np.array([cur[0] + vrot[0] * time_to_target, cur[1] + vrot[1] * time_to_target, model["com_height"],
                       cur[3] + cmd_vel[3] * time_to_target, 0.0, 0.0])
    times, xs = target_pose_to_target_trajectories(model, target, t_now, x_now, t_now + time_to_target)
    xs[0, 0:3] = vrot
    xs[1, 0:3] = vrot
    return times, xs


def goal_to_target_trajectories(model, goal, t_now, x_now):
    """bipedal_controllers/src/TargetTrajectoriesPublisher.cpp:30-38,60-74."""
    cur = np.array(x_now[6:12], float)
    target = np.array([goal[0], goal[1], model["com_height"], goal[3], 0.0, 0.0])
    d = target - cur
    t_rot = abs(d[3]) / model["target_rotation_velocity"]
    t_dis = math.sqrt(d[0] * d[0] + d[1] * d[1]) / model["target_displacement_velocity"]
    return target_pose_to_target_trajectories(model, target, t_now, x_now, t_now + max(t_rot, t_dis))


def interpolate_target(times, xs, t):
    """[OCS2-upstream] TargetTrajectories::getDesiredState -> LinearInterpolation::interpolate (clamped)."""
    n = len(times)
    if n == 1 or t <= times[0]:
        return np.array(xs[0])
    if t >= times[-1]:
        return np.array(xs[-1])
    i = bisect.bisect_left(times, t) - 1
    alpha = (times[i + 1] - t) / (times[i + 1] - times[i])
    return alpha * xs[i] + (1.0 - alpha) * xs[i + 1]


def node_arrays(model, t0, tf, dt, event_times, mode_sequence, target_times, target_states, planner=None):
    """Everything the solver needs per shooting interval: kind (0 intermediate / 1 event),
    interval start time ti, duration, mode id, swing references and x_ref, plus the node times."""
    grid = time_discretization_with_events(t0, tf, dt, event_times)
    if planner is None:
        raise ValueError("planner required")
    N = len(grid) - 1
    nx = model["nx"]
    kind = np.zeros(N, np.int32)
    ti = np.zeros(N)
    dts = np.zeros(N)
    mode = np.zeros(N, np.int32)
    zref = np.zeros((N, 4))
    zdref = np.zeros((N, 4))
    xref = np.zeros((N, nx))
    for k in range(N):
        if grid[k][1] == EVENT_PRE:
            kind[k] = 1
            ti[k] = grid[k][0]
            mode[k] = mode_at_time(event_times, mode_sequence, ti[k])
            xref[k] = interpolate_target(target_times, target_states, ti[k])
            continue
        ti[k] = interval_start(grid[k])
        dts[k] = interval_end(grid[k + 1]) - ti[k]
        mode[k] = mode_at_time(event_times, mode_sequence, ti[k])
        for j in range(4):
            zref[k, j] = planner.z_position(j, ti[k])
            zdref[k, j] = planner.z_velocity(j, ti[k])
        xref[k] = interpolate_target(target_times, target_states, ti[k])
    times = np.array([g[0] for g in grid])
    return dict(N=N, kind=kind, ti=ti, dt=dts, mode=mode, zref=zref, zdref=zdref, xref=xref, times=times,
                events=np.array([g[1] for g in grid], np.int32))


def weight_compensating_input(model, mode):
    """include/ocs2_bipedal_robot/common/utils.h:63-76."""
    flags = mode_flags(mode)
    n = sum(flags)
    u = np.zeros(model["nu"])
    if n > 0:
        w = model["robot_mass"] * 9.81
        for i in range(4):
            if flags[i]:
                u[3 * i + 2] = w / n
    return u


def cold_start(model, nodes, x0):
    """[OCS2-upstream] SqpSolver::initializeStateInputTrajectories without a previous solution +
    BipedalRobotInitializer::compute (src/initialization/BipedalRobotInitializer.cpp:56-63)."""
    N = nodes["N"]
    x = np.tile(np.asarray(x0, float), (N + 1, 1))
    u = np.zeros((N, model["nu"]))
    for k in range(N):
        if nodes["kind"][k] == 0:
            u[k] = weight_compensating_input(model, int(nodes["mode"][k]))
    return x, u


# ---------------------------------------------------------------------------------------------------------------
# Warm start of a receding-horizon solve from the previous solution (SURVEY.md section 8(f) rank 1, A.5 "Initial guess").
# [OCS2-upstream, recalled] SqpSolver::initializeStateInputTrajectories, multiple_shooting::toPrimalSolution,
# multiple_shooting::initializeIntermediateNode(PrimalSolution&, ...), LinearController::computeInput,
# LinearInterpolation::timeSegment.  Used with mpc.coldStart false (task.info:173) and sqp.useFeedbackPolicy true
# (task.info:80).
# ---------------------------------------------------------------------------------------------------------------
def primal_solution_arrays(prev_nodes, x, u, K):
    """[OCS2-upstream] multiple_shooting::toPrimalSolution: one entry per node time; the input (and gain) of a pre-event
    node and of the terminal node repeat the previous one; uff = u - K x (LinearController bias)."""
    N = int(prev_nodes["N"])
    t = np.asarray(prev_nodes["times"], float)
    nu, nx = u.shape[1], x.shape[1]
    uu = np.zeros((N + 1, nu))
    KK = np.zeros((N + 1, nu, nx))
    for j in range(N + 1):
        repeat = (j == N) or (prev_nodes["kind"][j] == 1 and j > 0)
        if repeat and j > 0:
            uu[j], KK[j] = uu[j - 1], KK[j - 1]
        elif j < N:
            uu[j], KK[j] = u[j], K[j]
    uff = np.array([uu[j] - KK[j] @ x[j] for j in range(N + 1)])
    return t, np.asarray(x, float), uff, KK


def time_segment(times, t):
    """[OCS2-upstream] LinearInterpolation::timeSegment: (index, alpha) with value = alpha v[i] + (1 - alpha) v[i + 1]."""
    n = len(times)
    if t <= times[0]:
        return 0, 1.0
    if t >= times[-1]:
        return n - 2, 0.0
    idx = int(np.searchsorted(times, t, side="left"))      # lower_bound
    i = min(max(idx - 1, 0), n - 2)
    return i, (times[i + 1] - t) / (times[i + 1] - times[i])


def warm_start_from_previous(model, nodes, x_measured, prev_nodes, prev_x, prev_u, prev_K, feedback=True):
    """Initial iterate (x[N+1], u[N]) of a new solve on `nodes` given the previous solve's result.  feedback = sqp.useFeedbackPolicy
    (task.info:80): the previous solution is a LinearController (u = uff(t) + K(t) x) or, false, a FeedforwardController (u = u(t):
    [OCS2-upstream, recalled] multiple_shooting::toPrimalSolution builds it from the same time / input arrays)."""
    N = int(nodes["N"])
    nx, nu = model["nx"], model["nu"]
    tp, xp, uff, KK = primal_solution_arrays(prev_nodes, prev_x, prev_u, prev_K)
    state_till = tp[-1]
    input_till = tp[-2] if len(tp) >= 2 else tp[0]
    x = np.zeros((N + 1, nx))
    u = np.zeros((N, nu))
    x[0] = x_measured
    for i in range(N):
        if nodes["kind"][i] == 1:            # event node: identity jump map as the guess
            x[i + 1] = x[i]
            continue
        t = nodes["ti"][i]
        t_next = t + nodes["dt"][i]
        if t > input_till or t_next > state_till:
            u[i] = weight_compensating_input(model, int(nodes["mode"][i]))      # BipedalRobotInitializer::compute
            x[i + 1] = x[i]
        else:
            j, a = time_segment(tp, t)
            if feedback:
                u[i] = a * uff[j] + (1.0 - a) * uff[j + 1] + (a * KK[j] + (1.0 - a) * KK[j + 1]) @ x[i]
            else:
                uin = uff + np.einsum("kij,kj->ki", KK, xp)      # the input trajectory of the primal solution (pre-event / terminal entries repeated)
                u[i] = a * uin[j] + (1.0 - a) * uin[j + 1]
            j2, a2 = time_segment(tp, t_next)
            x[i + 1] = a2 * xp[j2] + (1.0 - a2) * xp[j2 + 1]
    return x, u


# ---------------------------------------------------------------------------------------------------------------
# RK2 sensitivity discretisation of a flow map, generic ([OCS2-upstream] ocs2_oc SensitivityIntegrator rk2
# as ocs2_sqp uses it, task.info:81 integratorType RK2): x+ = x + dt/2 (f(x, u) + f(x + dt f(x, u), u)) and its Jacobians by the chain rule.
# The C++ oracle has the same composition built in for the robot's flow map; tests/test_third_party_pins.py pins THIS function to a
# symbolic differentiation (sympy) on a toy system and the C++ oracle to this function.
# ---------------------------------------------------------------------------------------------------------------
def rk2_discretize(flow, x, u, dt):
    """flow(x, u) -> (f, df/dx, df/du).  Returns x+, A = dx+/dx, B = dx+/du."""
    f1, A1, B1 = flow(x, u)
    x2 = x + dt * f1
    f2, A2, B2 = flow(x2, u)
    n = len(x)
    xn = x + 0.5 * dt * f1 + 0.5 * dt * f2
    A = np.eye(n) + 0.5 * dt * (A1 + A2 + dt * A2 @ A1)
    B = 0.5 * dt * (B1 + B2 + dt * A2 @ B1)
    return xn, A, B


# ---------------------------------------------------------------------------------------------------------------
# MRT side: MRT_BASE::rolloutPolicy = TimeTriggeredRollout::run under the LinearController of
# the last PrimalSolution, as used by MRT_ROS_Dummy_Loop (ocs2_bipedal_robot_ros/src/BipedalRobotDummyNode.cpp:61,72-86) and
# BipedalController (bipedal_controllers/src/BipedalController.cpp:322).  rollout settings: task.info:158-167 (ODE45,
# AbsTolODE 1e-5, RelTolODE 1e-3, timeStep 0.015).  [OCS2-upstream] RolloutBase::findActiveModesTimeInterval,
# TimeTriggeredRollout::runImpl, LinearController::computeInput; [boost::numeric::odeint, un-vendored] integrate_adaptive with
# make_controlled<runge_kutta_dopri5<...>>(abs, rel): controlled_runge_kutta (FSAL) try_step, default_error_checker,
# default_step_adjuster.  Restated from the published sources as recalled; parity UNPINNED.
# ---------------------------------------------------------------------------------------------------------------
def linear_controller_input(tp, uff, KK, t, x):
    j, a = time_segment(tp, t)
    return a * uff[j] + (1.0 - a) * uff[j + 1] + (a * KK[j] + (1.0 - a) * KK[j + 1]) @ x


DOPRI5_A = ((1.0 / 5.0,),
            (3.0 / 40.0, 9.0 / 40.0),
            (44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0),
            (19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0),
            (9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0))
DOPRI5_C = (1.0 / 5.0, 3.0 / 10.0, 4.0 / 5.0, 8.0 / 9.0, 1.0)
DOPRI5_B = (35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0)
DOPRI5_DB = (35.0 / 384.0 - 5179.0 / 57600.0, 0.0, 500.0 / 1113.0 - 7571.0 / 16695.0, 125.0 / 192.0 - 393.0 / 640.0,
             -2187.0 / 6784.0 - (-92097.0 / 339200.0), 11.0 / 84.0 - 187.0 / 2100.0, -1.0 / 40.0)


def dopri5_step(f, x, dxdt, t, dt):
    """runge_kutta_dopri5::do_step_impl (FSAL): returns x_new, dxdt_new, x_err."""
    k = [dxdt]
    for s in range(5):
        xs = x + dt * sum(a * kk for a, kk in zip(DOPRI5_A[s], k))
        k.append(f(t + dt * DOPRI5_C[s], xs))
    x_new = x + dt * sum(b * kk for b, kk in zip(DOPRI5_B, k))
    k.append(f(t + dt, x_new))
    x_err = dt * sum(d * kk for d, kk in zip(DOPRI5_DB, k))
    return x_new, k[6], x_err


def integrate_adaptive_dopri5(f, x, t0, t1, dt, abs_tol, rel_tol, observer, max_steps=10 ** 6):
    """odeint integrate_adaptive(controlled_stepper_tag) with a fresh controlled dopri5 stepper.  Returns (x, t, accepted, rejected)."""
    eps = np.finfo(float).eps
    t = t0
    dxdt = None
    accepted = rejected = 0
    while t1 - t > eps:                                   # less_with_sign(start_time, end_time, dt)
        observer(x, t)
        if (t + dt) - t1 > eps:                           # less_with_sign(end_time, start_time + dt, dt)
            dt = t1 - t
        while True:
            if dxdt is None:                              # m_first_call: initialize the FSAL derivative
                dxdt = f(t, x)
            x_new, dxdt_new, x_err = dopri5_step(f, x, dxdt, t, dt)
            err = float(np.max(np.abs(x_err) / (abs_tol + rel_tol * (np.abs(x) + abs(dt) * np.abs(dxdt)))))
            if err > 1.0:                                 # decrease_step, error_order 4
                dt *= max(0.9 * err ** (-1.0 / 3.0), 0.2)
                rejected += 1
                if rejected > 500:                        # failed_step_checker (500 consecutive failures) is never reached in the tests
                    raise RuntimeError("Max number of iterations exceeded (500). A new step size was not found.")
                continue
            t = t + dt
            x, dxdt = x_new, dxdt_new
            if err < 0.5:                                 # increase_step, stepper_order 5
                e = max(5.0 ** -5, err)
                dt *= 0.9 * e ** (-1.0 / 5.0)
            accepted += 1
            break
        if accepted > max_steps:
            raise RuntimeError("integration terminated: max number of steps reached")
    observer(x, t)
    return x, t, accepted, rejected


def find_active_modes_time_interval(t0, tf, event_times):
    """[OCS2-upstream] RolloutBase::findActiveModesTimeInterval: split at the events in (t0, tf]; every begin time is nudged by
    weakEpsilon (never past its end)."""
    first = bisect.bisect_right(event_times, t0)
    last = bisect.bisect_right(event_times, tf)
    sw = [t0] + list(event_times[first:last]) + [tf]
    return [(min(sw[i] + WEAK_EPS, sw[i + 1]), sw[i + 1]) for i in range(len(sw) - 1)]


def time_triggered_rollout(flow_map, controller, t0, x0, tf, event_times, settings):
    """[OCS2-upstream] TimeTriggeredRollout::runImpl.  flow_map(x, u) -> dx/dt, controller(t, x) -> u.
    Returns dict(times, states, inputs, post_event_indices, accepted, rejected)."""
    intervals = find_active_modes_time_interval(t0, tf, event_times)
    max_steps = int(settings["maxNumStepsPerSecond"] * max(1.0, intervals[-1][1] - intervals[0][0]))
    times, states, inputs, post = [], [], [], []

    def observer(x, t):
        states.append(np.array(x, float))
        times.append(float(t))

    def f(t, x):
        return flow_map(x, controller(t, x))

    x = np.array(x0, float)
    acc = rej = 0
    for i, (tb, te) in enumerate(intervals):
        x, _, a, r = integrate_adaptive_dopri5(f, x, tb, te, settings["timeStep"], settings["AbsTolODE"], settings["RelTolODE"], observer, max_steps)
        acc += a
        rej += r
        while len(inputs) < len(times):
            inputs.append(controller(times[len(inputs)], states[len(inputs)]))
        if i < len(intervals) - 1:
            post.append(len(states))                      # identity jump map: the next segment starts from the same state
    return dict(times=np.array(times), states=np.array(states), inputs=np.array(inputs), post_event_indices=post, accepted=acc, rejected=rej)
