"""Synthetic problem sets = the configurations of BASELINE.json, built through the product's own
reference-manager API so that bench.py, smoke() and the GPU tests all see the same inputs."""
import os

import numpy as np

from .api import BipedalRobotInterface, GaitSchedule, loadModeSequenceTemplate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H1 = dict(task=os.path.join(ROOT, "assets/h1/task.info"), urdf=os.path.join(ROOT, "assets/h1/h1_mpc.urdf"),
          reference=os.path.join(ROOT, "assets/h1/reference.info"), gait=os.path.join(ROOT, "assets/h1/gait.info"))
# the reference's own 12-leg-joint robot (nx = nu = 24): stands in for BASELINE.json configs[3] ("G1 ... different DoF"),
# for which the reference ships no OCS2 configuration
OPENLOONG = dict(task=os.path.join(ROOT, "assets/openloong/task.info"), urdf=os.path.join(ROOT, "assets/openloong/openloong_mpc.urdf"),
                 reference=os.path.join(ROOT, "assets/openloong/reference.info"), gait=os.path.join(ROOT, "assets/openloong/gait.info"))
ROBOTS = {"h1": H1, "openloong": OPENLOONG}
DT = 0.015
SEED = 20241008
# phase offset of the steady-state gait: the template starts 3.5 periods-halves before t = 0 so that t0 = 0 is mid-swing
GAIT_START = -1.225


def h1_interface():
    return BipedalRobotInterface(H1["task"], H1["urdf"], H1["reference"])


def interface(robot="h1"):
    r = ROBOTS[robot]
    itf = BipedalRobotInterface(r["task"], r["urdf"], r["reference"])
    itf.gaitFile = r["gait"]
    return itf


def perturbed_initial_states(itf, batch, seed=SEED):
    """x0_b = initialState + U(-a, a) per block."""
    rng = np.random.default_rng(seed)
    nx, nj = itf.stateDim, itf.actuatedDofNum
    amp = np.concatenate([np.full(3, 0.1), np.full(3, 0.05), [0.05, 0.05, 0.02], np.full(3, 0.05), np.full(nj, 0.05)])
    return itf.getInitialState()[None, :] + rng.uniform(-1.0, 1.0, size=(batch, nx)) * amp[None, :]


def gait_schedule(itf, gait_name, t0, horizon, gait_file=None, start=GAIT_START):
    """Steady-state schedule of a named gait as the reference's GaitSchedule would hold it at solve time."""
    gs = GaitSchedule(itf)
    if gait_name != "stance":
        tmpl = loadModeSequenceTemplate(gait_file or getattr(itf, "gaitFile", H1["gait"]), gait_name)
        gs.insertModeSequenceTemplate(tmpl, start, t0 + 2 * horizon)
    return gs.getModeSchedule(t0 - horizon, t0 + 2 * horizon)


def stance_problem(itf, n_intervals=20):
    """Config 1: H1 stance, horizon N*dt, single problem, constant target."""
    horizon = n_intervals * DT
    x0 = itf.getInitialState()[None, :]
    sched = gait_schedule(itf, "stance", 0.0, horizon)
    xs = np.zeros((2, itf.stateDim))
    xs[:, 8] = float(itf.get("com_height")[0])
    xs[:, 12:] = itf.get("default_joint_state")
    from .api import TargetTrajectories
    return dict(t0=0.0, x0=x0, schedule=sched, targets=[TargetTrajectories(np.array([0.0, horizon]), xs)], horizon=horizon)


def trot_problem(itf, batch, n_intervals=100, cmd_vel=(0.3, 0.0, 0.0, 0.0), gait="trot", seed=SEED, offset=0):
    """Config 2/3: H1 trot, horizon N*dt, `batch` perturbed initial states starting at problem index `offset`."""
    horizon = n_intervals * DT
    x0 = perturbed_initial_states(itf, offset + batch, seed)[offset:]
    sched = gait_schedule(itf, gait, 0.0, horizon)
    targets = [itf.cmdVelToTargetTrajectories(cmd_vel, 0.0, x0[b], horizon) for b in range(batch)]
    return dict(t0=0.0, x0=x0, schedule=sched, targets=targets, horizon=horizon)


def max_nodes_for(n_intervals, horizon, period_min=0.03):
    """Upper bound of grid intervals: every event adds at most two nodes."""
    return int(n_intervals + 2 * (horizon / period_min) + 4)


def gait_sweep_problem(itf, gaits, commands, n_intervals=150, seed=SEED):
    """Config 5 in miniature: one problem per (gait, velocity command); every problem has its OWN mode schedule.
    `commands` is a list of (v_x, omega_z)."""
    horizon = n_intervals * DT
    nb = len(gaits) * len(commands)
    x0 = perturbed_initial_states(itf, nb, seed)
    schedules, targets = [], []
    for g in gaits:
        sched = gait_schedule(itf, g, 0.0, horizon)
        for (vx, wz) in commands:
            b = len(schedules)
            schedules.append(sched)
            targets.append(itf.cmdVelToTargetTrajectories((vx, 0.0, 0.0, wz), 0.0, x0[b], horizon))
    return dict(t0=np.zeros(nb), x0=x0, schedule=schedules, targets=targets, horizon=horizon)
