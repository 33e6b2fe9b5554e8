"""Synthetic problem sets = the configurations of BASELINE.json, built through the product's own
reference-manager API so that bench.py, smoke() and the GPU tests all see the same inputs."""
import os

import numpy as np

from .api import BipedalRobotInterface, GaitSchedule, loadModeSequenceTemplate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H1 = dict(task=os.path.join(ROOT, "assets/h1/task.info"), urdf=os.path.join(ROOT, "assets/h1/h1_mpc.urdf"),
          reference=os.path.join(ROOT, "assets/h1/reference.info"), gait=os.path.join(ROOT, "assets/h1/gait.info"))
# the reference's own 12-leg-joint robot (nx = nu = 24)
OPENLOONG = dict(task=os.path.join(ROOT, "assets/openloong/task.info"), urdf=os.path.join(ROOT, "assets/openloong/openloong_mpc.urdf"),
                 reference=os.path.join(ROOT, "assets/openloong/reference.info"), gait=os.path.join(ROOT, "assets/openloong/gait.info"))
# Unitree G1 (BASELINE.json configs[3]): the reference ships only g1.urdf - sole frames and the INFO files are authored by
# tools/make_assets.py (see assets/ATTRIBUTION.md); results on it are self-defined, not reference parity
G1 = dict(task=os.path.join(ROOT, "assets/g1/task.info"), urdf=os.path.join(ROOT, "assets/g1/g1_mpc.urdf"),
          reference=os.path.join(ROOT, "assets/g1/reference.info"), gait=os.path.join(ROOT, "assets/g1/gait.info"))
# Hunter: the reference's third complete configuration (10 leg joints) and the only one with model_settings.positionErrorGain != 0
HUNTER = dict(task=os.path.join(ROOT, "assets/hunter/task.info"), urdf=os.path.join(ROOT, "assets/hunter/hunter_mpc.urdf"),
              reference=os.path.join(ROOT, "assets/hunter/reference.info"), gait=os.path.join(ROOT, "assets/hunter/gait.info"))
ROBOTS = {"h1": H1, "openloong": OPENLOONG, "g1": G1, "hunter": HUNTER}
DT = 0.015
SEED = 20241008
# phase offset of the steady-state gait: the template starts 3.5 periods-halves before t = 0 so that t0 = 0 is mid-swing
GAIT_START = -1.225


def h1_interface():
    return BipedalRobotInterface(H1["task"], H1["urdf"], H1["reference"])


def interface(robot="h1"):
    """robot name, optionally with the suffix ":hard" = useHardFrictionConeConstraint (friction cones as inequality constraints)."""
    name, _, variant = robot.partition(":")
    r = ROBOTS[name]
    itf = BipedalRobotInterface(r["task"], r["urdf"], r["reference"], useHardFrictionConeConstraint=(variant == "hard"))
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


def trot_problem(itf, batch, n_intervals=100, cmd_vel=(0.3, 0.0, 0.0, 0.0), gait="trot", seed=SEED, offset=0, gait_start=GAIT_START):
    """Config 2/3: H1 trot, horizon N*dt, `batch` perturbed initial states starting at problem index `offset`.
    gait_start: time at which the gait template is inserted.  The default puts t0 = 0 in the middle of a swing phase (steady state);
    gait_start = 0.0 is SURVEY.md section 8(d) config 2 to the letter ("template tiled from t = 0": the solve starts on a mode switch)."""
    horizon = n_intervals * DT
    x0 = perturbed_initial_states(itf, offset + batch, seed)[offset:]
    sched = gait_schedule(itf, gait, 0.0, horizon, start=gait_start)
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


# ---- BASELINE.json configs[4]: gait-library sweep (SURVEY.md section 8d "Config 5")
SYNTHETIC_TROT_PERIODS = (0.5, 0.6, 0.9, 1.0)


def gait_library(itf, gait_file=None):
    """The 8 gait modes of the sweep: the reference's four templates (gait.info:1-7: stance, trot, standing_trot, flying_trot) and four
    synthetic variants - the trot template scaled to periods 0.5 / 0.6 / 0.9 / 1.0 s (the reference defines only four gaits).
    Returns (names, templates)."""
    from .api import ModeSequenceTemplate
    gf = gait_file or getattr(itf, "gaitFile", H1["gait"])
    names = ["stance", "trot", "standing_trot", "flying_trot"]
    lib = [loadModeSequenceTemplate(gf, n) for n in names]
    trot = lib[1]
    period = float(trot.switchingTimes[-1])
    for T in SYNTHETIC_TROT_PERIODS:
        names.append("trot_%.1fs" % T)
        lib.append(ModeSequenceTemplate(np.asarray(trot.switchingTimes) * (T / period), np.asarray(trot.modeSequence)))
    return names, lib


def command_grid(n_vx=32, n_wz=16, vx_max=0.5, wz_max=0.3):
    """512 velocity commands (v_x, 0, 0, omega_z): v_x in [-0.5, 0.5] (32) x omega_z in [-0.3, 0.3] (16), limits from reference.info:1-2."""
    return np.array([(vx, 0.0, 0.0, wz) for vx in np.linspace(-vx_max, vx_max, n_vx) for wz in np.linspace(-wz_max, wz_max, n_wz)])


def gait_sweep_commands(itf, gait_indices, n_intervals=150, n_vx=32, n_wz=16, seed=SEED):
    """Inputs of bpmpc_solver_setup_commands for the gaits `gait_indices` of the library (one block of n_vx * n_wz problems per gait,
    problem index = gait * 512 + command, so that any sharding by gait reproduces the same problems)."""
    names, lib = gait_library(itf)
    cmds = command_grid(n_vx, n_wz)
    per = len(cmds)
    x0_all = perturbed_initial_states(itf, len(lib) * per, seed)
    gop = np.repeat(np.asarray(gait_indices, np.int32), per)
    x0 = np.concatenate([x0_all[g * per:(g + 1) * per] for g in gait_indices]) if len(gait_indices) else x0_all[:0]
    cmd = np.tile(cmds, (len(gait_indices), 1))
    return dict(t0=0.0, x0=x0, gaits=lib, gait_names=names, gait_of_problem=gop, gait_start=GAIT_START, cmd_vel=cmd, horizon=n_intervals * DT)


def commands_problem_on_host(itf, cp, b):
    """Problem b of a gait_sweep_commands() set as a host-side scenario (schedule from GaitSchedule, target from cmdVelToTargetTrajectories):
    what the oracle-side checker of bench.py / the tests solves."""
    gs = GaitSchedule(itf)
    g = int(cp["gait_of_problem"][b])
    gs.insertModeSequenceTemplate(cp["gaits"][g], cp["gait_start"], cp["t0"] + 2 * cp["horizon"])
    sched = gs.getModeSchedule(cp["t0"] - cp["horizon"], cp["t0"] + 2 * cp["horizon"])
    tg = itf.cmdVelToTargetTrajectories(tuple(cp["cmd_vel"][b]), cp["t0"], cp["x0"][b], cp["horizon"])
    return dict(t0=cp["t0"], x0=cp["x0"][b:b + 1], schedule=sched, targets=[tg], horizon=cp["horizon"])
