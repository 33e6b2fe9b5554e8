#!/usr/bin/env python3
"""Compare a PrimalSolution dumped from the reference's own SqpMpc (tools/ocs2_dump_primal.cpp, run where OCS2 exists) with this
repository's oracle and, when a GPU is present, with the HIP path.

usage: compare_ocs2_dump.py <dump.csv> [--robot h1] [--no-gpu]
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
        if not head.startswith("# bpmpc-ocs2-dump v1"):
            raise ValueError("not a bpmpc-ocs2-dump v1 file: " + path)
        fields = head.split(",")
        nx, nu, nodes, intervals, gait = int(fields[1]), int(fields[2]), int(fields[3]), int(fields[4]), fields[5]
        rows = np.array([[float(v) for v in line.split(",")] for line in f if line.strip()])
    if rows.shape != (nodes, 2 + nx + nu):
        raise ValueError("dump has shape %s, header says %d nodes of %d + %d" % (rows.shape, nodes, nx, nu))
    return dict(nx=nx, nu=nu, intervals=intervals, gait=gait, t=rows[:, 1], x=rows[:, 2:2 + nx], u=rows[:, 2 + nx:])


def write_dump(path, t, x, u, intervals, gait):
    """x: [nodes, nx], u: [nodes - 1, nu] (the terminal node repeats the last input, like PrimalSolution)."""
    nodes = len(t)
    uu = np.vstack([u, u[-1:]])
    with open(path, "w") as f:
        f.write("# bpmpc-ocs2-dump v1,%d,%d,%d,%d,%s\n" % (x.shape[1], uu.shape[1], nodes, intervals, gait))
        for k in range(nodes):
            f.write(",".join([str(k), repr(float(t[k]))] + [repr(float(v)) for v in x[k]] + [repr(float(v)) for v in uu[k]]) + "\n")


def problem_of(dump, robot):
    """The problem tools/ocs2_dump_primal.cpp solved, rebuilt through this repository's reference-manager mirror."""
    from bipedal_control_amd import scenarios as sc
    itf = sc.interface(robot)
    if dump["gait"] == "stance":
        prob = sc.stance_problem(itf, dump["intervals"])
    else:
        horizon = dump["intervals"] * sc.DT
        x0 = itf.getInitialState()[None, :]
        prob = dict(t0=0.0, x0=x0, schedule=sc.gait_schedule(itf, dump["gait"], 0.0, horizon),
                    targets=[itf.cmdVelToTargetTrajectories((0.3, 0.0, 0.0, 0.0), 0.0, x0[0], horizon)], horizon=horizon)
    return itf, prob


def compare(dump, robot="h1", gpu=True):
    from tests import oracle_bridge as ob
    itf, prob = problem_of(dump, robot)
    xo, uo, _, _ = ob.oracle_solve_like(prob, 0, robot=robot)
    report = {}

    def diff(name, t, x, u):
        n = len(t)
        if n != len(dump["t"]):
            report[name] = dict(ok=False, why="node count %d vs %d in the dump" % (n, len(dump["t"])))
            return
        ex = float(np.abs(x - dump["x"]).max())
        eu = float(np.abs(u - dump["u"][:n - 1]).max() / max(1.0, np.abs(dump["u"]).max()))
        et = float(np.abs(t - dump["t"]).max())
        report[name] = dict(ok=bool(ex < TOL_X_ABS and eu < TOL_U_REL and et < 1e-9), max_abs_x=ex, max_rel_u=eu, max_abs_t=et)

    nodes = ob.oracle_nodes(prob, 0, robot=robot)
    diff("oracle", np.asarray(nodes["times"], float), xo, uo)
    if gpu:
        try:
            import bipedal_control_amd as bp
            mpc = bp.BatchedSqpMpc(itf, max_batch=1, max_nodes=len(dump["t"]) + 8)
            t, x, u, _, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
            n = st[0].n_nodes
            diff("hip", t[0, :n + 1], x[0, :n + 1], u[0, :n])
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
