#!/usr/bin/env python3
"""Compare a PrimalSolution dumped from the reference's own SqpMpc (tools/ocs2_dump_primal.cpp, run where OCS2 exists) with this
repository's oracle and, when a GPU is present, with the HIP path.

usage: compare_ocs2_dump.py <dump.csv> [--robot h1|hunter|openloong] [--no-gpu]
The three commands a maintainer runs (reference workspace built; H1 shown, see INTEGRATION.md for the Hunter case):
    rosrun ocs2_bipedal_robot_ros ocs2_dump_primal task.info h1_with_sole.urdf reference.info stance.csv 20
    rosrun ocs2_bipedal_robot_ros ocs2_dump_primal task.info h1_with_sole.urdf reference.info trot.csv 67 gait.info trot 0.02
    python tools/compare_ocs2_dump.py stance.csv && python tools/compare_ocs2_dump.py trot.csv && python tools/compare_ocs2_dump.py trot.csv.2
Tolerances: 1e-6 abs on x, 1e-4 relative on u (different CppAD / HPIPM rounding, HPIPM's reg_prim, LU vs
other null-space bases are all below that).  Exit code 0 = within tolerance.  `write_dump` produces the same format from any
(t, x, u) so that the tool chain can be tested without OCS2 (tests/test_ocs2_dump.py)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOL_X_ABS, TOL_U_REL = 1e-6, 1e-4


def read_dump(path):
    with open(path) as f:
        head = f.readline().strip()
        if not (head.startswith("# bpmpc-ocs2-dump v1") or head.startswith("# bpmpc-ocs2-dump v2")):
            raise ValueError("not a bpmpc-ocs2-dump v1 / v2 file: " + path)
        fields = head.split(",")
        nx, nu, nodes, intervals, gait = int(fields[1]), int(fields[2]), int(fields[3]), int(fields[4]), fields[5]
        t0, solve = (float(fields[6]), int(fields[7])) if len(fields) >= 8 else (0.0, 0)
        rows = np.array([[float(v) for v in line.split(",")] for line in f if line.strip()])
    if rows.shape != (nodes, 2 + nx + nu):
        raise ValueError("dump has shape %s, header says %d nodes of %d + %d" % (rows.shape, nodes, nx, nu))
    return dict(nx=nx, nu=nu, intervals=intervals, gait=gait, t0=t0, solve=solve, t=rows[:, 1], x=rows[:, 2:2 + nx], u=rows[:, 2 + nx:])


def primal_inputs(kind, u):
    """Inputs as multiple_shooting::toPrimalSolution [OCS2-upstream, recalled] lays them out: one per node time, the terminal node and
    every pre-event node repeat the previous entry (the engine and the oracle keep u = 0 at event nodes)."""
    n = len(u)
    uu = np.zeros((n + 1, u.shape[1]))
    for j in range(n + 1):
        repeat = j > 0 and (j == n or int(kind[j]) == 1)
        uu[j] = uu[j - 1] if repeat else u[min(j, n - 1)]
    return uu


def write_dump(path, t, x, u, kind, intervals, gait, t0=0.0, solve=0):
    """x: [nodes, nx], u: [nodes - 1, nu] as the solver keeps it; written in PrimalSolution's arrangement (primal_inputs)."""
    nodes = len(t)
    uu = primal_inputs(kind, u)
    with open(path, "w") as f:
        f.write("# bpmpc-ocs2-dump v2,%d,%d,%d,%d,%s,%r,%d\n" % (x.shape[1], uu.shape[1], nodes, intervals, gait, float(t0), solve))
        for k in range(nodes):
            f.write(",".join([str(k), repr(float(t[k]))] + [repr(float(v)) for v in x[k]] + [repr(float(v)) for v in uu[k]]) + "\n")


def problems_of(dump, robot):
    """The problem(s) tools/ocs2_dump_primal.cpp solved, rebuilt through this repository's reference-manager mirror: a list with the
    first solve (t = 0, cold start) and, for a dump of the second solve, that solve (t = period, warm start from the first) - both from
    ONE GaitSchedule object, as the reference's SwitchedModelReferenceManager holds one (getModeSchedule erases passed events)."""
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios as sc
    itf = sc.interface(robot)
    horizon = dump["intervals"] * sc.DT
    if dump["gait"] == "stance":
        if dump["solve"] != 0:
            raise ValueError("the stance case has no second solve")
        return itf, [sc.stance_problem(itf, dump["intervals"])]
    x0 = itf.getInitialState()[None, :]
    gs = bp.GaitSchedule(itf)
    gs.insertModeSequenceTemplate(bp.loadModeSequenceTemplate(sc.ROBOTS[robot]["gait"], dump["gait"]), sc.GAIT_START, 2 * horizon)
    probs = []
    for t0 in ([0.0] if dump["solve"] == 0 else [0.0, dump["t0"]]):
        probs.append(dict(t0=t0, x0=x0, schedule=gs.getModeSchedule(t0 - horizon, t0 + 2 * horizon),
                          targets=[itf.cmdVelToTargetTrajectories((0.3, 0.0, 0.0, 0.0), t0, x0[0], horizon)], horizon=horizon))
    return itf, probs


def compare(dump, robot="h1", gpu=True):
    from oracle import reference_py as rp
    from tests import oracle_bridge as ob
    itf, probs = problems_of(dump, robot)
    m, om = ob.model(robot), ob.oracle(robot)
    sq = m["sqp"]
    report = {}

    def diff(name, t, x, u, kind):
        n = len(t)
        if n != len(dump["t"]):
            report[name] = dict(ok=False, why="node count %d vs %d in the dump" % (n, len(dump["t"])))
            return
        uu = primal_inputs(kind, u)
        ex = float(np.abs(x - dump["x"]).max())
        # relative per block: contact forces (N) and joint velocities (rad/s) have different scales
        ef = float(np.abs(uu[:, :12] - dump["u"][:, :12]).max() / max(1.0, np.abs(dump["u"][:, :12]).max()))
        ev = float(np.abs(uu[:, 12:] - dump["u"][:, 12:]).max() / max(1.0, np.abs(dump["u"][:, 12:]).max()))
        et = float(np.abs(t - dump["t"]).max())
        report[name] = dict(ok=bool(ex < TOL_X_ABS and max(ef, ev) < TOL_U_REL and et < 1e-9), max_abs_x=ex, max_rel_u=max(ef, ev), max_rel_force=ef,
                            max_rel_joint_velocity=ev, max_abs_t=et)

    prev = None
    for prob in probs:
        nodes = ob.oracle_nodes(prob, 0, robot=robot)
        x0 = prob["x0"][0]
        xi, ui = rp.cold_start(m, nodes, x0) if prev is None else rp.warm_start_from_previous(m, nodes, x0, *prev)
        xo, uo, Ko, _ = om.solve(nodes, x0, xi, ui, iterations=1, g_max=sq["g_max"], g_min=sq["g_min"], delta_tol=sq["deltaTol"])
        prev = (nodes, xo, uo, Ko)
    diff("oracle", np.asarray(nodes["times"], float), xo, uo, nodes["kind"])
    if gpu:
        try:
            import bipedal_control_amd as bp
            mpc = bp.BatchedSqpMpc(itf, max_batch=1, max_nodes=len(dump["t"]) + 8, return_gains=True)
            for i, prob in enumerate(probs):
                run = mpc.run if i == 0 else mpc.advance
                t, x, u, _, st = run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
            n = st[0].n_nodes
            diff("hip", t[0, :n + 1], x[0, :n + 1], u[0, :n], nodes["kind"])
        except Exception as e:  # no GPU on this box
            report["hip"] = dict(ok=None, why=str(e))
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dump")
    ap.add_argument("--robot", default="h1")
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    rep = compare(read_dump(a.dump), a.robot, gpu=not a.no_gpu)
    for k, v in rep.items():
        print(k, v)
    sys.exit(0 if all(v["ok"] is not False for v in rep.values()) else 1)


if __name__ == "__main__":
    main()
