"""The OCS2 adaptor EXECUTED: integration/mock_run.cpp links integration/HipSqpMpc.h / HipSqpSolver.h against libbpmpc.so with the labelled
stand-ins of integration/mock_ocs2 in place of OCS2 and runs three MPC iterations (cold start through bpmpc_solve_batch, then two
receding-horizon runs through setup_from_previous / run / fetch).  What it hands back through SolverBase::getPrimalSolution - time,
state and input trajectories, LinearController bias uff_k = u_k - K_k x_k and gains, with the input / gain of the terminal node and
of every pre-event node repeating the previous entry - must equal the same solves through the Python mirror, arranged by the ORACLE's
restatement of multiple_shooting::toPrimalSolution (oracle/reference_py.py primal_solution_arrays; the round-2 version of this test
repeated the adaptor's own indexing and so could not see that event nodes exported u = 0, K = 0).  This checks the adaptor's logic; it pins nothing about OCS2 (mock_ocs2/README.md)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("robot,urdf", [("h1", "h1_mpc.urdf"), ("hunter", "hunter_mpc.urdf")])
def test_adaptor_runs_and_matches_the_python_mirror(tmp_path, robot, urdf):
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios as sc
    from oracle import reference_py as rp
    from tests import oracle_bridge as ob
    exe = str(tmp_path / "mock_run")
    lib = os.path.join(ROOT, "bipedal_control_amd")
    inc = [os.path.join(ROOT, d) for d in ("include", "integration", os.path.join("integration", "mock_ocs2"))]
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror"] + [a for i in inc for a in ("-I", i)] + [os.path.join(ROOT, "integration", "mock_run.cpp"),
                   "-L", lib, "-lbpmpc", "-Wl,-rpath," + lib, "-o", exe], check=True)
    out = subprocess.run([exe, os.path.join(ROOT, "assets", robot), urdf], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == 3

    def fields(line):
        w = line.split()
        return {w[i]: float(w[i + 1]) for i in range(0, len(w) - 1, 2)}

    itf = sc.interface(robot)
    nx = nu = itf.stateDim
    horizon, period = 1.005, 0.02
    gs = bp.GaitSchedule(itf)
    gs.insertModeSequenceTemplate(bp.loadModeSequenceTemplate(sc.ROBOTS[robot]["gait"], "trot"), -1.225, 3 * horizon)
    sched = gs.getModeSchedule(-horizon, 3 * horizon)
    x0 = itf.getInitialState()
    mpc = bp.BatchedSqpMpc(itf, max_batch=1, max_nodes=96, return_gains=True)
    iterations = 0
    for k in range(3):
        t0 = k * period
        target = [itf.cmdVelToTargetTrajectories((0.3, 0.0, 0.0, 0.0), t0, x0, horizon)]
        if k == 0:
            t, x, u, K, st = mpc.run(t0, x0.reshape(1, nx), sched, target, horizon=horizon, gains=True)
        else:
            t, x, u, K, st = mpc.advance(t0, x0.reshape(1, nx), sched, target, horizon=horizon, gains=True)
        n = st[0].n_nodes
        iterations += st[0].iterations
        # multiple_shooting::toPrimalSolution as the oracle restates it, on the oracle's own node table of this problem
        nodes = ob.oracle_nodes({"schedule": sched, "targets": target, "t0": t0, "x0": x0.reshape(1, nx), "horizon": horizon}, 0, robot=robot)
        assert int(nodes["N"]) == n and (np.asarray(nodes["kind"])[:n] == 1).sum() >= 2        # gait events inside the horizon: the case that matters
        tt, xx, bias, KK = rp.primal_solution_arrays(nodes, x[0, :n + 1], u[0, :n], K[0, :n])
        assert np.array_equal(tt, t[0, :n + 1])
        uu = bias + np.einsum("kij,kj->ki", KK, xx)
        for j in np.nonzero(np.asarray(nodes["kind"])[:n] == 1)[0]:                             # pre-event entries repeat, they are not zero
            assert np.array_equal(KK[j], KK[j - 1]) and np.abs(KK[j]).max() > 0
        i = np.arange(n + 1)[:, None]
        st_ = float(np.sum(tt * (1 + np.arange(n + 1) % 3)))
        sx = float(np.sum(xx * (1 + (i + np.arange(nx)[None, :]) % 7)))
        su = float(np.sum(uu * (1 + (i + np.arange(nu)[None, :]) % 5)))
        sb = float(np.sum(bias * (1 + (i + np.arange(nu)[None, :]) % 4)))
        a, b = np.meshgrid(np.arange(nu), np.arange(nx), indexing="ij")
        sk = float(np.sum(KK * (1 + (a + 2 * b) % 3)[None]))
        got = fields(out[k])
        assert got["feedforward"] == 0
        assert got["run"] == k and got["points"] == n + 1 and got["iterations"] == iterations
        assert got["merit"] == st[0].merit_after and got["dyn"] == st[0].dynamics_sse_after and got["final"] == tt[-1]
        for name, ref in (("st", st_), ("sx", sx), ("su", su), ("sb", sb), ("sk", sk)):       # sums in another order: rounding only
            assert abs(got[name] - ref) <= 1e-11 * max(1.0, abs(ref)), (k, name, got[name], ref)
        # ProblemMetrics: the "<foot>_zeroVelocity" terms of every intermediate node at the solution, from the ORACLE's constraint rows there
        om = ob.oracle(robot)
        kinds, modes = np.asarray(nodes["kind"])[:n], np.asarray(nodes["mode"])[:n]
        szv, nzv, i_inter = 0.0, 0, 0
        for j in range(n):
            if kinds[j] != 0:
                continue
            lq = om.node_lq(0, nodes["dt"][j], x[0, j], u[0, j], x[0, j + 1], nodes["xref"][j], modes[j], nodes["zref"][j], nodes["zdref"][j])
            e, row = np.asarray(lq["e"]), 0
            for c in range(4):
                stance = (modes[j] & 1) != 0 if c < 2 else (modes[j] & 2) != 0
                if stance:
                    for a in range(3):
                        szv += e[row + a] * (1 + (i_inter + c + a) % 3)
                    nzv += 3
                    row += 3
                else:
                    row += 4
            assert row == lq["nc"]
            i_inter += 1
        assert got["nzv"] == nzv and got["prejumps"] == int((kinds != 0).sum()) and abs(got["szv"] - szv) <= 1e-9 * max(1.0, abs(szv)), (got["szv"], szv)


def test_adaptor_returns_a_feedforward_controller_when_the_task_file_says_so(tmp_path):
    """sqp.useFeedbackPolicy false (task.info:80) reaches the adaptor through the model (HipSqpSolver::Settings::useFeedbackPolicy = -1:
    the file decides): getPrimalSolution carries a FeedforwardController over the input trajectory (pre-event / terminal entries repeated),
    and the receding-horizon warm start on the device evaluates the previous solution as such - the same solves through the Python mirror."""
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios as sc
    from oracle import reference_py as rp
    from tests import oracle_bridge as ob
    exe = str(tmp_path / "mock_run")
    lib = os.path.join(ROOT, "bipedal_control_amd")
    inc = [os.path.join(ROOT, d) for d in ("include", "integration", os.path.join("integration", "mock_ocs2"))]
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror"] + [a for i in inc for a in ("-I", i)] + [os.path.join(ROOT, "integration", "mock_run.cpp"),
                   "-L", lib, "-lbpmpc", "-Wl,-rpath," + lib, "-o", exe], check=True)
    text = open(sc.H1["task"]).read()
    task = tmp_path / "task_feedforward.info"
    task.write_text(text.replace("useFeedbackPolicy true", "useFeedbackPolicy false", 1))
    out = subprocess.run([exe, os.path.join(ROOT, "assets", "h1"), "h1_mpc.urdf", str(task)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == 3
    itf = bp.BipedalRobotInterface(str(task), sc.H1["urdf"], sc.H1["reference"])
    itf.gaitFile = sc.H1["gait"]
    nx = nu = itf.stateDim
    horizon, period = 1.005, 0.02
    gs = bp.GaitSchedule(itf)
    gs.insertModeSequenceTemplate(bp.loadModeSequenceTemplate(sc.H1["gait"], "trot"), -1.225, 3 * horizon)
    sched = gs.getModeSchedule(-horizon, 3 * horizon)
    x0 = itf.getInitialState()
    mpc = bp.BatchedSqpMpc(itf, max_batch=1, max_nodes=96, return_gains=True)
    for k in range(3):
        t0 = k * period
        target = [itf.cmdVelToTargetTrajectories((0.3, 0.0, 0.0, 0.0), t0, x0, horizon)]
        t, x, u, K, st = (mpc.run if k == 0 else mpc.advance)(t0, x0.reshape(1, nx), sched, target, horizon=horizon, gains=True)
        n = st[0].n_nodes
        nodes = ob.oracle_nodes({"schedule": sched, "targets": target, "t0": t0, "x0": x0.reshape(1, nx), "horizon": horizon}, 0)
        tt, xx, bias, KK = rp.primal_solution_arrays(nodes, x[0, :n + 1], u[0, :n], K[0, :n])
        uu = bias + np.einsum("kij,kj->ki", KK, xx)          # the input trajectory with the repeated entries
        i = np.arange(n + 1)[:, None]
        su = float(np.sum(uu * (1 + (i + np.arange(nu)[None, :]) % 5)))
        sb = float(np.sum(uu * (1 + (i + np.arange(nu)[None, :]) % 4)))
        w = out[k].split()
        got = {w[j]: float(w[j + 1]) for j in range(0, len(w) - 1, 2)}
        assert got["feedforward"] == 1 and got["sk"] == 0.0 and got["points"] == n + 1
        assert got["merit"] == st[0].merit_after and got["dyn"] == st[0].dynamics_sse_after
        for name, ref in (("su", su), ("sb", sb)):
            assert abs(got[name] - ref) <= 1e-11 * max(1.0, abs(ref)), (k, name, got[name], ref)


def test_ddp_adaptor_runs_and_matches_the_python_mirror(tmp_path):
    """integration/HipDdpMpc.h (the counterpart of GaussNewtonDDP_MPC at BipedalRobotDdpMpcNode.cpp:70-71) executed: three MPC runs - a cold
    start and two receding-horizon runs - hand back the accepted roll-out on its own time points with a FeedforwardController
    (ddp.useFeedbackPolicy false, task.info:146); the same runs through the Python mirror (BatchedDdpMpc)."""
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios as sc
    exe = str(tmp_path / "mock_run")
    lib = os.path.join(ROOT, "bipedal_control_amd")
    inc = [os.path.join(ROOT, d) for d in ("include", "integration", os.path.join("integration", "mock_ocs2"))]
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror"] + [a for i in inc for a in ("-I", i)] + [os.path.join(ROOT, "integration", "mock_run.cpp"),
                   "-L", lib, "-lbpmpc", "-Wl,-rpath," + lib, "-o", exe], check=True)
    out = subprocess.run([exe, os.path.join(ROOT, "assets", "h1"), "h1_mpc.urdf", "-", "ddp"], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == 3
    itf = sc.interface("h1")
    nx = nu = itf.stateDim
    horizon, period = 1.005, 0.02
    gs = bp.GaitSchedule(itf)
    gs.insertModeSequenceTemplate(bp.loadModeSequenceTemplate(sc.H1["gait"], "trot"), -1.225, 3 * horizon)
    sched = gs.getModeSchedule(-horizon, 3 * horizon)
    x0 = itf.getInitialState()
    mpc = bp.BatchedDdpMpc(itf, max_batch=1, max_nodes=96)
    iterations = 0
    for k in range(3):
        t0 = k * period
        target = [itf.cmdVelToTargetTrajectories((0.3, 0.0, 0.0, 0.0), t0, x0, horizon)]
        t, x, u, K, st = (mpc.run if k == 0 else mpc.advance)(t0, x0.reshape(1, nx), sched, target, horizon=horizon)
        n = st[0].n_nodes
        iterations += st[0].iterations
        tt, xx = t[0, :n + 1], x[0, :n + 1]
        uu = np.vstack([u[0, :n], u[0, n - 1:n]])                      # the input of the terminal point repeats the previous one
        i = np.arange(n + 1)[:, None]
        ref = dict(st=float(np.sum(tt * (1 + np.arange(n + 1) % 3))), sx=float(np.sum(xx * (1 + (i + np.arange(nx)[None, :]) % 7))),
                   su=float(np.sum(uu * (1 + (i + np.arange(nu)[None, :]) % 5))), sb=float(np.sum(uu * (1 + (i + np.arange(nu)[None, :]) % 4))))
        w = out[k].split()
        got = {w[j]: float(w[j + 1]) for j in range(0, len(w) - 1, 2)}
        assert got["feedforward"] == 1 and got["sk"] == 0.0 and got["points"] == n + 1 and got["iterations"] == iterations
        assert got["merit"] == st[0].merit_after and got["final"] == tt[-1] and st[0].step_size > 0
        for name in ("st", "sx", "su", "sb"):
            assert abs(got[name] - ref[name]) <= 1e-11 * max(1.0, abs(ref[name])), (k, name, got[name], ref[name])
