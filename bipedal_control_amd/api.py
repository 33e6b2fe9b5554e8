"""Host-side mirror of the reference's operator interface for the MPC hot path, over the C ABI (include/bpmpc.h).

Names follow the reference so that code reads like its call sites:
  BipedalRobotInterface(taskFile, urdfFile, referenceFile)   ocs2_bipedal_robot/include/ocs2_bipedal_robot/BipedalRobotInterface.h:56-127
  GaitSchedule.insertModeSequenceTemplate / getModeSchedule  ocs2_bipedal_robot/src/gait/GaitSchedule.cpp:46-102
  loadModeSequenceTemplate                                   ocs2_bipedal_robot/src/gait/ModeSequenceTemplate.cpp:50-71
  cmdVelToTargetTrajectories / goalToTargetTrajectories      bipedal_controllers/src/TargetTrajectoriesPublisher.cpp:60-99
  BatchedSqpMpc(interface, ...)                              the SqpMpc construction sites bipedal_controllers/src/BipedalController.cpp:303-306
                                                             and ocs2_bipedal_robot_ros/src/BipedalRobotSqpMpcNode.cpp:70, for a batch of problems
Errors: the reference throws std::runtime_error / std::invalid_argument; here every non-zero status of the C ABI
raises BpmpcError carrying bpmpc_last_error().
"""
import ctypes as C
import os
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

ModeSchedule = namedtuple("ModeSchedule", ["eventTimes", "modeSequence"])
ModeSequenceTemplate = namedtuple("ModeSequenceTemplate", ["switchingTimes", "modeSequence"])
TargetTrajectories = namedtuple("TargetTrajectories", ["timeTrajectory", "stateTrajectory"])

MODE_NUMBER = {"FLY": 0, "LF": 1, "RF": 2, "STANCE": 3}


class BpmpcError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("bpmpc status %d: %s" % (status, message))
        self.status = status


class _Settings(C.Structure):
    _fields_ = [("device", C.c_int), ("max_batch", C.c_int), ("max_nodes", C.c_int), ("sqp_iterations", C.c_int), ("dt", C.c_double),
                ("return_gains", C.c_int), ("profile", C.c_int), ("stream", C.c_void_p), ("reference_kernels", C.c_int),
                ("pipeline_chunks", C.c_int), ("materialize_lq", C.c_int), ("reg_prim", C.c_double), ("solver", C.c_int), ("feedback_policy", C.c_int)]


class _Schedule(C.Structure):
    _fields_ = [("n_events", C.c_int), ("event_times", _dp), ("modes", _ip)]


class _GaitTemplate(C.Structure):
    _fields_ = [("n_modes", C.c_int), ("switching_times", _dp), ("modes", _ip)]


class _Target(C.Structure):
    _fields_ = [("n_points", C.c_int), ("times", _dp), ("states", _dp)]


class Stats(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("iterations", C.c_int), ("status", C.c_int), ("reserved", C.c_int),
                ("merit_before", C.c_double), ("dynamics_sse_before", C.c_double), ("equality_sse_before", C.c_double),
                ("merit_after", C.c_double), ("dynamics_sse_after", C.c_double), ("equality_sse_after", C.c_double),
                ("step_size", C.c_double), ("armijo_descent", C.c_double), ("dx_norm", C.c_double), ("du_norm", C.c_double)]


def library_path():
    return os.path.join(_HERE, "libbpmpc.so")


def load_library():
    """Loads the in-tree HIP library; raises if it has not been built (no fallback of any kind)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise BpmpcError(-4, "libbpmpc.so is missing - build it with `python -m bipedal_control_amd.build` (hipcc, gfx950)")
        lib = C.CDLL(path)
        lib.bpmpc_last_error.restype = C.c_char_p
        lib.bpmpc_version.restype = C.c_char_p
        _LIB = lib
    return _LIB


def _check(rc):
    if rc < 0:
        raise BpmpcError(rc, load_library().bpmpc_last_error().decode())
    return rc


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class BipedalRobotInterface:
    """Problem definition: model constants + settings (BipedalRobotInterface.cpp:67-204)."""

    def __init__(self, taskFile, urdfFile, referenceFile, useHardFrictionConeConstraint=False):
        """Fourth argument as in the reference (BipedalRobotInterface.h:66-69): friction cones as inequality constraints, which the SQP
        solver penalises with sqp.inequalityConstraintMu / Delta (include/bpmpc.h bpmpc_model_create_ex)."""
        lib = load_library()
        self._h = C.c_void_p()
        self.useHardFrictionConeConstraint = bool(useHardFrictionConeConstraint)
        _check(lib.bpmpc_model_create_ex(str(urdfFile).encode(), str(taskFile).encode(), str(referenceFile).encode(),
                                         1 if useHardFrictionConeConstraint else 0, C.byref(self._h)))
        nx, nu, nc, nj = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(lib.bpmpc_model_dims(self._h, C.byref(nx), C.byref(nu), C.byref(nc), C.byref(nj)))
        self.stateDim, self.inputDim, self.numThreeDofContacts, self.actuatedDofNum = nx.value, nu.value, nc.value, nj.value
        self.taskFile, self.urdfFile, self.referenceFile = str(taskFile), str(urdfFile), str(referenceFile)

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.bpmpc_model_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def get(self, name, capacity=4096):
        out = np.zeros(capacity)
        n = _check(load_library().bpmpc_model_get(self._h, name.encode(), _d(out), capacity))
        return out[:n].copy()

    def getInitialState(self):
        return self.get("initial_state")

    def jointNames(self):
        buf = C.create_string_buffer(256)
        names = []
        for j in range(self.actuatedDofNum):
            _check(load_library().bpmpc_model_joint_name(self._h, j, buf, 256))
            names.append(buf.value.decode())
        return names

    def robotMass(self):
        return float(self.get("robot_mass")[0])

    def sqpSettings(self):
        v = self.get("sqp")
        return dict(dt=v[0], sqpIteration=int(v[1]), deltaTol=v[2], g_max=v[3], g_min=v[4], useFeedbackPolicy=bool(v[5]),
                    projectStateInputEqualityConstraints=bool(v[6]), integratorType="RK2")

    _IPM_FIELDS = ("dt", "ipmIteration", "deltaTol", "g_max", "g_min", "computeLagrangeMultipliers", "useFeedbackPolicy", "initialBarrierParameter",
                   "targetBarrierParameter", "barrierLinearDecreaseFactor", "barrierSuperlinearDecreasePower", "barrierReductionCostTol",
                   "barrierReductionConstraintTol", "fractionToBoundaryMargin", "usePrimalStepSizeForDual", "initialSlackLowerBound",
                   "initialDualLowerBound", "initialSlackMarginRate", "initialDualMarginRate", "nThreads", "threadPriority")
    _DDP_FIELDS = ("algorithm", "maxNumIterations", "minRelCost", "constraintTolerance", "AbsTolODE", "RelTolODE", "timeStep", "maxNumStepsPerSecond",
                   "backwardPassIntegratorType", "constraintPenaltyInitialValue", "constraintPenaltyIncreaseRate", "preComputeRiccatiTerms",
                   "useFeedbackPolicy", "strategy", "lineSearch.minStepLength", "lineSearch.maxStepLength", "lineSearch.hessianCorrectionStrategy",
                   "lineSearch.hessianCorrectionMultiple", "nThreads", "threadPriority")
    _DDP_ENUMS = {"algorithm": ("SLQ", "ILQR"), "strategy": ("LINE_SEARCH", "LEVENBERG_MARQUARDT"),
                  "backwardPassIntegratorType": ("ODE45", "EULER", "ODE45_OCS2", "ADAMS_BASHFORTH", "BULIRSCH_STOER", "MODIFIED_MIDPOINT", "RK4", "RK5_VARIABLE",
                                                 "ADAMS_BASHFORTH_MOULTON"),
                  "lineSearch.hessianCorrectionStrategy": ("DIAGONAL_SHIFT", "CHOLESKY_MODIFICATION", "EIGENVALUE_MODIFICATION", "GERSHGORIN_MODIFICATION")}
    _INT_FIELDS = {"ipmIteration", "nThreads", "threadPriority", "maxNumIterations", "maxNumStepsPerSecond"}
    _BOOL_FIELDS = {"computeLagrangeMultipliers", "useFeedbackPolicy", "usePrimalStepSizeForDual", "preComputeRiccatiTerms"}

    def _settings(self, block, fields, enums=()):
        v = self.get(block)
        out = {}
        for name, x in zip(fields, v):
            if name in enums:
                out[name] = enums[name][int(x)]
            elif name in self._BOOL_FIELDS:
                out[name] = bool(x)
            elif name in self._INT_FIELDS:
                out[name] = int(x)
            else:
                out[name] = float(x)
        return out

    def ipmSettings(self):
        """The `ipm` block of task.info as the reference loads it (BipedalRobotInterface.cpp:100, accessor BipedalRobotInterface.h:80).
        The reference constructs no IPM solver; neither does this engine: the settings are loaded and exposed, nothing consumes them."""
        return self._settings("ipm", self._IPM_FIELDS)

    def ddpSettings(self):
        """The `ddp` block of task.info (BipedalRobotInterface.cpp:98); consumed in the reference by the stand-alone DDP node
        (BipedalRobotDdpMpcNode.cpp:70-74), a solver this engine does not have."""
        return self._settings("ddp", self._DDP_FIELDS, self._DDP_ENUMS)

    def rolloutSettings(self):
        v = self.get("rollout")
        return dict(AbsTolODE=v[0], RelTolODE=v[1], timeStep=v[2], maxNumStepsPerSecond=int(v[3]))

    def mpcSettings(self):
        return dict(timeHorizon=float(self.get("time_horizon")[0]))

    def costMatrices(self):
        nx, nu = self.stateDim, self.inputDim
        return self.get("Q").reshape(nx, nx), self.get("R").reshape(nu, nu)

    # --- target trajectories (TargetTrajectoriesPublisher.cpp:60-99)
    def cmdVelToTargetTrajectories(self, cmdVel, time, state, timeToTarget=None):
        if timeToTarget is None:
            timeToTarget = self.mpcSettings()["timeHorizon"]
        cmd, x = _f64(cmdVel), _f64(state)
        times, states = np.zeros(2), np.zeros((2, self.stateDim))
        _check(load_library().bpmpc_cmd_vel_to_targets(self._h, _d(cmd), C.c_double(time), _d(x), C.c_double(timeToTarget), _d(times), _d(states)))
        return TargetTrajectories(times, states)

    def goalToTargetTrajectories(self, goal, time, state):
        g, x = _f64(goal), _f64(state)
        times, states = np.zeros(2), np.zeros((2, self.stateDim))
        _check(load_library().bpmpc_goal_to_targets(self._h, _d(g), C.c_double(time), _d(x), _d(times), _d(states)))
        return TargetTrajectories(times, states)


def loadModeSequenceTemplate(filename, topicName):
    times, modes, n = np.zeros(65), np.zeros(64, np.int32), C.c_int()
    _check(load_library().bpmpc_gait_load_template(str(filename).encode(), topicName.encode(), _d(times), _i(modes), 64, C.byref(n)))
    return ModeSequenceTemplate(times[:n.value + 1].copy(), modes[:n.value].copy())


class GaitSchedule:
    """GaitSchedule(initModeSchedule, defaultModeSequenceTemplate, phaseTransitionStanceTime) as loaded by
    BipedalRobotInterface::loadGaitSchedule (BipedalRobotInterface.cpp:209-234)."""

    def __init__(self, interface):
        self._h = C.c_void_p()
        _check(load_library().bpmpc_gait_create(interface.handle, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.bpmpc_gait_destroy(self._h)
            self._h = None

    def insertModeSequenceTemplate(self, modeSequenceTemplate, startTime, finalTime):
        t, m = _f64(modeSequenceTemplate.switchingTimes), np.ascontiguousarray(modeSequenceTemplate.modeSequence, np.int32)
        _check(load_library().bpmpc_gait_insert_template(self._h, _d(t), _i(m), len(m), C.c_double(startTime), C.c_double(finalTime)))

    def getModeSchedule(self, lowerBoundTime, upperBoundTime, capacity=4096):
        ev, ms, n = np.zeros(capacity), np.zeros(capacity, np.int32), C.c_int()
        _check(load_library().bpmpc_gait_mode_schedule(self._h, C.c_double(lowerBoundTime), C.c_double(upperBoundTime), _d(ev), _i(ms), capacity, C.byref(n)))
        return ModeSchedule(ev[:n.value].copy(), ms[:n.value + 1].copy())


def swing_reference(interface, modeSchedule, times):
    """SwingTrajectoryPlanner::update + getZpositionConstraint / getZvelocityConstraint (SwingTrajectoryPlanner.cpp:50-118)."""
    ev, ms, t = _f64(modeSchedule.eventTimes), np.ascontiguousarray(modeSchedule.modeSequence, np.int32), _f64(times)
    z, zd = np.zeros((len(t), 4)), np.zeros((len(t), 4))
    _check(load_library().bpmpc_swing_reference(interface.handle, _d(ev), _i(ms), len(ev), _d(t), len(t), _d(z), _d(zd)))
    return z, zd


def time_discretization_with_events(initTime, finalTime, dt, eventTimes, capacity=8192):
    ev = _f64(eventTimes)
    t, e, n = np.zeros(capacity), np.zeros(capacity, np.int32), C.c_int()
    _check(load_library().bpmpc_time_grid(C.c_double(initTime), C.c_double(finalTime), C.c_double(dt), _d(ev), len(ev), _d(t), _i(e), capacity, C.byref(n)))
    return t[:n.value].copy(), e[:n.value].copy()


class BatchedSqpMpc:
    # (the DDP variant is the same handle with solver="ddp": BatchedDdpMpc below)
    """A batch of independent SqpMpc instances on one MI355X.  `run` plays the role of MPC_BASE::run(t, x) /
    MPC_MRT_Interface::advanceMpc() (BipedalController.cpp:339) for every problem of the batch at once."""

    def __init__(self, interface, max_batch, max_nodes, sqp_iterations=0, dt=0.0, return_gains=False, profile=False, device=0, stream=None,
                 reference_kernels=False, pipeline_chunks=0, materialize_lq=False, reg_prim=0.0, solver="sqp", feedback_policy=None):
        """solver: "sqp" (SqpMpc, default) or "ddp" (GaussNewtonDDP_MPC of BipedalRobotDdpMpcNode.cpp:70-71: one ILQR iteration per run, see
        bpmpc_settings.solver in include/bpmpc.h).  feedback_policy: None = sqp.useFeedbackPolicy of task.info, True / False overrides it for the warm
        start, the policy rollout and the controller built from the solution alike."""
        lib = load_library()
        self.interface = interface
        self.max_batch, self.max_nodes = int(max_batch), int(max_nodes)
        self.nx, self.nu = interface.stateDim, interface.inputDim
        self.return_gains = bool(return_gains)
        if stream is not None and int(stream) == 0:
            # the C ABI reads a NULL stream as "create your own": the legacy default stream cannot be passed.  Work that must be
            # ordered against torch has to run on an explicit stream (torch.cuda.Stream().cuda_stream), see bench.py.
            raise ValueError("stream=0 (the default stream) cannot be handed over; pass an explicit stream handle or None for a solver-owned stream")
        st = _Settings(int(device), self.max_batch, self.max_nodes, int(sqp_iterations), float(dt), int(bool(return_gains)), int(profile),
                       C.c_void_p(int(stream)) if stream is not None else None, int(bool(reference_kernels)), int(pipeline_chunks), int(bool(materialize_lq)), float(reg_prim),
                       {"sqp": 0, "ddp": 1}[solver], 0 if feedback_policy is None else (1 if feedback_policy else 2))
        self.solver = solver
        self._h = C.c_void_p()
        _check(lib.bpmpc_solver_create(interface.handle, C.byref(st), C.byref(self._h)))
        self._keep = None
        self.batch = 0

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.bpmpc_solver_destroy(self._h)
            self._h = None

    # ---- argument marshalling
    def _marshal(self, t0, x0, modeSchedules, targetTrajectories, warm_x, warm_u):
        x0 = _f64(x0).reshape(-1, self.nx)
        B = x0.shape[0]
        t0 = _f64(np.broadcast_to(np.asarray(t0, float), (B,)))
        if isinstance(modeSchedules, ModeSchedule):
            modeSchedules = [modeSchedules]
        if isinstance(targetTrajectories, TargetTrajectories):
            targetTrajectories = [targetTrajectories] * B
        if len(targetTrajectories) != B:
            raise ValueError("%d target trajectories for %d problems (pass one per problem, or a single TargetTrajectories to share)" % (len(targetTrajectories), B))
        if len(modeSchedules) not in (1, B):
            raise ValueError("%d mode schedules for %d problems (pass one to share, or one per problem)" % (len(modeSchedules), B))
        keep = [t0, x0]
        sched = (_Schedule * len(modeSchedules))()
        for i, ms in enumerate(modeSchedules):
            ev, mo = _f64(ms.eventTimes), np.ascontiguousarray(ms.modeSequence, np.int32)
            keep += [ev, mo]
            sched[i] = _Schedule(len(ev), _d(ev), _i(mo))
        tg = (_Target * B)()
        for i, tt in enumerate(targetTrajectories):
            ts, xs = _f64(tt.timeTrajectory), _f64(tt.stateTrajectory)
            keep += [ts, xs]
            tg[i] = _Target(len(ts), _d(ts), _d(xs))
        wx = wu = None
        if warm_x is not None:
            if warm_u is None:
                raise ValueError("warm_x and warm_u must be given together")
            wx, wu = _f64(warm_x), _f64(warm_u)
            if wx.size != B * (self.max_nodes + 1) * self.nx or wu.size != B * self.max_nodes * self.nu:
                raise ValueError("warm start arrays must have the solver's strides: x [%d, %d, %d], u [%d, %d, %d]"
                                 % (B, self.max_nodes + 1, self.nx, B, self.max_nodes, self.nu))
            keep += [wx, wu]
        return B, t0, x0, sched, len(modeSchedules), tg, wx, wu, keep

    def setup(self, t0, x0, modeSchedules, targetTrajectories, horizon=None, warm_x=None, warm_u=None):
        if horizon is None:
            horizon = self.interface.mpcSettings()["timeHorizon"]
        B, t0, x0, sched, ns, tg, wx, wu, keep = self._marshal(t0, x0, modeSchedules, targetTrajectories, warm_x, warm_u)
        _check(load_library().bpmpc_solver_setup(self._h, B, C.c_double(horizon), _d(t0), _d(x0), sched, ns, tg, _d(wx), _d(wu)))
        self.batch = B
        return self.layout()

    def setup_from_previous(self, t0, x0, modeSchedules, targetTrajectories, horizon=None):
        """Receding-horizon step (MPC_BASE::run with mpc.coldStart false): new measured states / schedules / targets, initial
        iterate shifted on the device from the previous solve of this handle (bpmpc_solver_setup_from_previous)."""
        if horizon is None:
            horizon = self.interface.mpcSettings()["timeHorizon"]
        B, t0, x0, sched, ns, tg, _, _, keep = self._marshal(t0, x0, modeSchedules, targetTrajectories, None, None)
        _check(load_library().bpmpc_solver_setup_from_previous(self._h, B, C.c_double(horizon), _d(t0), _d(x0), sched, ns, tg))
        self.batch = B
        return self.layout()

    def setup_commands(self, t0, x0, gaits, gait_of_problem, gait_start, cmd_vel, horizon=None, time_to_target=0.0, from_previous=False, goal=False):
        """The whole pre-pass on the device (bpmpc_solver_setup_commands): `gaits` is a list of ModeSequenceTemplate, problem b
        follows gaits[gait_of_problem[b]] inserted at gait_start[b] (index < 0: initial schedule only) and tracks the velocity
        command cmd_vel[b] = (vx, vy, vz, yaw rate), or with goal=True moves to the pose (x, y, -, yaw) like goalToTargetTrajectories."""
        if horizon is None:
            horizon = self.interface.mpcSettings()["timeHorizon"]
        if x0 is None:                      # continue from the end states of the last rollout (device-resident)
            B = self.batch
        else:
            x0 = _f64(x0).reshape(-1, self.nx)
            B = x0.shape[0]
        t0 = _f64(np.broadcast_to(np.asarray(t0, float), (B,)))
        gop = np.ascontiguousarray(np.broadcast_to(np.asarray(gait_of_problem, np.int32), (B,)))
        gst = _f64(np.broadcast_to(np.asarray(gait_start, float), (B,)))
        cmd = _f64(np.broadcast_to(np.asarray(cmd_vel, float), (B, 4)))
        keep, tm = [], (_GaitTemplate * max(1, len(gaits)))()
        for i, g in enumerate(gaits):
            sw, mo = _f64(g.switchingTimes), np.ascontiguousarray(g.modeSequence, np.int32)
            keep += [sw, mo]
            tm[i] = _GaitTemplate(len(mo), _d(sw), _i(mo))
        _check(load_library().bpmpc_solver_setup_commands(self._h, B, C.c_double(horizon), _d(t0), _d(x0), tm, len(gaits), _i(gop), _d(gst), _d(cmd),
                                                          int(bool(goal)), C.c_double(time_to_target), int(bool(from_previous))))
        self.batch = B
        return self.layout()

    def rollout(self, duration, t_start=None, x_start=None, fetch=True):
        """MRT_BASE::rolloutPolicy for the batch (bpmpc_solver_rollout): returns (x_end, u_end, steps[batch, 2]); fetch=False only
        enqueues (the end states stay on the device for setup_commands(x0=None))."""
        B = self.batch
        ts = None if t_start is None else _f64(np.broadcast_to(np.asarray(t_start, float), (B,)))
        xs = None if x_start is None else _f64(x_start).reshape(B, self.nx)
        if not fetch:
            _check(load_library().bpmpc_solver_rollout(self._h, _d(ts), _d(xs), C.c_double(duration), None, None, None))
            return None
        x_end, u_end, steps = np.zeros((B, self.nx)), np.zeros((B, self.nu)), np.zeros((B, 2), np.int32)
        _check(load_library().bpmpc_solver_rollout(self._h, _d(ts), _d(xs), C.c_double(duration), _d(x_end), _d(u_end), _i(steps)))
        return x_end, u_end, steps

    def advance(self, t0, x0, modeSchedules, targetTrajectories, horizon=None, gains=False):
        """One MPC tick for the whole batch: warm start from the previous solution, solve, fetch."""
        self.setup_from_previous(t0, x0, modeSchedules, targetTrajectories, horizon)
        self.enqueue()
        return self.fetch(gains=gains)

    def layout(self):
        b, n, g, nx, nu = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(load_library().bpmpc_solver_layout(self._h, C.byref(b), C.byref(n), C.byref(g), C.byref(nx), C.byref(nu)))
        return dict(batch=b.value, n_nodes_max=n.value, n_grids=g.value, nx=nx.value, nu=nu.value)

    def reset(self):
        _check(load_library().bpmpc_solver_reset(self._h))

    def enqueue(self):
        _check(load_library().bpmpc_solver_run(self._h))

    def synchronize(self):
        _check(load_library().bpmpc_solver_sync(self._h))

    def stage(self, name):
        _check(load_library().bpmpc_solver_stage(self._h, name.encode()))

    def fetch(self, gains=False):
        B, N = self.batch, self.max_nodes
        t = np.zeros((B, N + 1))
        x = np.zeros((B, N + 1, self.nx))
        u = np.zeros((B, N, self.nu))
        K = np.zeros((B, N, self.nu, self.nx)) if gains else None
        stats = (Stats * B)()
        _check(load_library().bpmpc_solver_fetch(self._h, _d(t), _d(x), _d(u), _d(K), stats))
        return t, x, u, K, list(stats)

    def run(self, t0, x0, modeSchedules, targetTrajectories, horizon=None, warm_x=None, warm_u=None, gains=False):
        """One MPC solve for every problem: returns (t, x, u, K, stats) with strides max_nodes (see stats[b].n_nodes)."""
        self.setup(t0, x0, modeSchedules, targetTrajectories, horizon, warm_x, warm_u)
        self.enqueue()
        self.synchronize()
        return self.fetch(gains)

    def solve_batch(self, t0, x0, modeSchedules, targetTrajectories, horizon=None, warm_x=None, warm_u=None, gains=False):
        """The single-call entry point bpmpc_solve_batch (host buffers in, host buffers out)."""
        if horizon is None:
            horizon = self.interface.mpcSettings()["timeHorizon"]
        B, t0, x0, sched, ns, tg, wx, wu, keep = self._marshal(t0, x0, modeSchedules, targetTrajectories, warm_x, warm_u)
        N = self.max_nodes
        t = np.zeros((B, N + 1))
        x = np.zeros((B, N + 1, self.nx))
        u = np.zeros((B, N, self.nu))
        K = np.zeros((B, N, self.nu, self.nx)) if gains else None
        stats = (Stats * B)()
        _check(load_library().bpmpc_solve_batch(self._h, B, C.c_double(horizon), _d(t0), _d(x0), sched, ns, tg, _d(wx), _d(wu), _d(t), _d(x),
                                                _d(u), _d(K), stats))
        self.batch = B
        return t, x, u, K, list(stats)

    def read(self, name):
        """Named device buffer as a flat float64 array (tests / debugging)."""
        lib = load_library()
        cap = _check(lib.bpmpc_solver_read(self._h, name.encode(), None, C.c_long(0)))
        out = np.zeros(cap)
        n = _check(lib.bpmpc_solver_read(self._h, name.encode(), _d(out), C.c_long(cap)))
        return out[:n]

    def set_materialize(self, materialize_lq):
        """True: the lineariser writes the complete per-node LQ model (parity stages, roofline pass); False: the fused solve mode."""
        _check(load_library().bpmpc_solver_set_materialize(self._h, int(bool(materialize_lq))))

    def set_profile(self, level):
        """0 off, 1 every kernel class, 2 the linearisation kernel only."""
        _check(load_library().bpmpc_solver_set_profile(self._h, int(level)))

    def kernel_time(self, kernel, reset=True):
        ms, n = C.c_double(), C.c_int()
        _check(load_library().bpmpc_solver_kernel_time(self._h, kernel.encode(), int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def device_trajectories(self):
        xp, up = _dp(), _dp()
        _check(load_library().bpmpc_solver_device_trajectories(self._h, C.byref(xp), C.byref(up)))
        return C.cast(xp, C.c_void_p).value, C.cast(up, C.c_void_p).value

    def constraint_values(self):
        """Values of the active equality rows at the current iterate (after a solve: the solution): (values [B, max_nodes, 16],
        rows [B, max_nodes], modes [B, max_nodes]) - include/bpmpc.h bpmpc_solver_constraint_values."""
        lay = self.layout()
        B = lay["batch"]
        v = np.zeros((B, self.max_nodes, 16))
        rows = np.zeros((B, self.max_nodes), np.int32)
        modes = np.zeros((B, self.max_nodes), np.int32)
        _check(load_library().bpmpc_solver_constraint_values(self._h, _d(v), _i(rows), _i(modes)))
        return v, rows, modes

    def export_trajectories(self, x_dst_ptr, u_dst_ptr):
        """Async D2D copy of the iterate into device buffers given by raw pointers (e.g. torch tensors' data_ptr())."""
        _check(load_library().bpmpc_solver_export_trajectories(self._h, C.c_void_p(x_dst_ptr), C.c_void_p(u_dst_ptr)))


class WeightedWbc:
    """A batch of WeightedWbc instances on one MI355X (bipedal_wbc/include/bipedal_wbc/WeightedWbc.h; construction + loadTasksSetting as
    in bipedal_controllers/src/BipedalController.cpp:97-100).  `update` mirrors WeightedWbc::update(stateDesired, inputDesired,
    rbdStateMeasured, mode, period) (BipedalController.cpp:229) with a leading batch dimension; it returns (x, status) where
    x[b] = [generalised accelerations, contact forces, joint torques] and status[b] = 1 when that robot's QP was not solved and its
    previous solution was returned instead (lastQpSol_)."""

    def __init__(self, interface, taskFile=None, max_batch=1, device=0):
        lib = load_library()
        self.interface = interface
        self._h = C.c_void_p()
        _check(lib.bpmpc_wbc_create(interface.handle, str(taskFile or interface.taskFile).encode(), int(device), int(max_batch), C.byref(self._h)))
        n, nv = C.c_int(), C.c_int()
        _check(lib.bpmpc_wbc_dims(self._h, C.byref(n), C.byref(nv)))
        self.numDecisionVars, self.generalizedCoordinatesNum, self.max_batch = n.value, nv.value, int(max_batch)

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.bpmpc_wbc_destroy(self._h)
            self._h = None

    def update(self, stateDesired, inputDesired, rbdStateMeasured, mode, period=0.002, debug=False):
        x = _f64(stateDesired).reshape(-1, self.interface.stateDim)
        B = x.shape[0]
        u = _f64(inputDesired).reshape(B, self.interface.inputDim)
        rbd = _f64(rbdStateMeasured).reshape(B, 2 * self.generalizedCoordinatesNum)
        md = np.ascontiguousarray(np.broadcast_to(np.asarray(mode, np.int32), (B,)))
        sol = np.zeros((B, self.numDecisionVars))
        status = np.zeros(B, np.int32)
        dbg = np.zeros((B, 1024)) if debug else None
        _check(load_library().bpmpc_wbc_update(self._h, B, _d(x), _d(u), _d(rbd), _i(md), C.c_double(period), _d(sol), _i(status), _d(dbg)))
        return (sol, status, dbg) if debug else (sol, status)

    def reset(self):
        _check(load_library().bpmpc_wbc_reset(self._h))


class BatchedDdpMpc(BatchedSqpMpc):
    """A batch of GaussNewtonDDP_MPC instances (ocs2_bipedal_robot_ros/src/BipedalRobotDdpMpcNode.cpp:70-71; ddp block of task.info): the same
    handle as BatchedSqpMpc with bpmpc_settings.solver = BPMPC_SOLVER_DDP.  `run` returns the accepted roll-out on its own time points:
    t[b, :stats[b].n_nodes + 1], x likewise, u[b, :stats[b].n_nodes]; stats[b].step_size is the accepted step length."""

    def __init__(self, interface, max_batch, max_nodes, **kw):
        kw.pop("solver", None)
        super().__init__(interface, max_batch, max_nodes, solver="ddp", return_gains=True, **kw)
