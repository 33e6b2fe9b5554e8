"""Hunter on the GPU: the reference's configuration with positionErrorGain = 20 (bipedal_robot_example/hunter/legged_hunter_config/config/
task/task.info:12), its own gait templates and cost weights, through the C ABI against the oracle - LQ model 1e-11, solves 1e-8 like H1 -
including the device-side reference generation from velocity commands and the eight-wave / four-wave sweeps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.tolerances import rel_K, rel_u, rel_x  # noqa: E402  (per physical block: forces vs joint velocities, ...)

ROBOT = "hunter"


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.fixture(scope="module")
def ctx():
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios as sc
    from tests import oracle_bridge as ob
    return dict(bp=bp, sc=sc, ob=ob, itf=sc.interface(ROBOT))


@pytest.mark.parametrize("gait", ["trot", "standing_trot", "flying_trot", "stance"])
def test_hunter_solve_matches_oracle(ctx, gait):
    bp, sc, ob, itf = ctx["bp"], ctx["sc"], ctx["ob"], ctx["itf"]
    assert itf.stateDim == 22 and itf.get("position_error_gain")[0] == 20.0
    B, NN = 3, 72
    prob = sc.trot_problem(itf, batch=B, n_intervals=45, gait=gait)
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=NN, return_gains=True, sqp_iterations=2, materialize_lq=True)
    t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    assert all(s.status == 0 for s in st)
    for b in range(B):
        xo, uo, Ko, so = ob.oracle_solve_like(prob, b, iterations=2, robot=ROBOT)
        n = st[b].n_nodes
        assert st[b].step_size == so[st[b].iterations - 1][3]
        assert rel_x(x[b, :n + 1], xo) < 1e-11 and rel_u(u[b, :n], uo) < 1e-11 and rel_K(K[b, :n], Ko) < 1e-10
    # the LQ model at the solution: every quantity of problem 0 against the oracle's node_lq
    mpc.stage("linearize"); mpc.synchronize()
    shapes = dict(A=(22, 22), B=(22, 22), b=(22,), q=(22,), r=(22,), C=(16, 22), D=(16, 22), e=(16,), perf=(3,))
    dev = {k: mpc.read(k).reshape(B, NN, *sh) for k, sh in shapes.items()}
    nodes = ob.oracle_nodes(prob, 0, robot=ROBOT)
    om = ob.oracle(ROBOT)
    worst = {}
    for k in range(nodes["N"]):
        o = om.node_lq(nodes["kind"][k], nodes["dt"][k], x[0, k], u[0, k], x[0, k + 1], nodes["xref"][k], nodes["mode"][k], nodes["zref"][k], nodes["zdref"][k])
        for name in shapes:
            worst[name] = max(worst.get(name, 0.0), _rel(dev[name][0, k], o[name]))
    assert max(worst.values()) < 1e-11, worst
    # the fused mode and the reference kernel bodies give the same solution
    fused = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=NN, return_gains=True, sqp_iterations=2)
    t2, x2, u2, K2, _ = fused.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    assert np.array_equal(x2, x) and np.array_equal(u2, u) and np.array_equal(K2, K)
    ref = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=NN, sqp_iterations=2, reference_kernels=True)
    t3, x3, u3, _, _ = ref.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    assert rel_x(x3, x) < 1e-9 and rel_u(u3, u) < 1e-9


def test_hunter_large_batch_and_commands(ctx, monkeypatch):
    """Batch 300 (four-wave sweep, two workgroups per CU: BPMPC_R8_ROUNDS=1) agrees with batch 3 (eight-wave sweep) to rounding; the device-side
    reference generation (gait template + velocity command) reproduces the host pre-pass on this robot's gait files (same time grid,
    solutions to 1e-9)."""
    bp, sc, ob, itf = ctx["bp"], ctx["sc"], ctx["ob"], ctx["itf"]
    monkeypatch.setenv("BPMPC_R8_ROUNDS", "1")
    prob = sc.trot_problem(itf, batch=300, n_intervals=40)
    big = bp.BatchedSqpMpc(itf, max_batch=300, max_nodes=64, sqp_iterations=2)
    t, x, u, _, st = big.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    assert all(s.status == 0 for s in st)
    sub = [299, 0, 123]
    prob2 = dict(prob, x0=prob["x0"][sub], targets=[prob["targets"][i] for i in sub])
    small = bp.BatchedSqpMpc(itf, max_batch=3, max_nodes=64, sqp_iterations=2)
    t2, x2, u2, _, _ = small.run(prob2["t0"], prob2["x0"], prob2["schedule"], prob2["targets"], horizon=prob2["horizon"])
    for j, i in enumerate(sub):
        assert rel_x(x2[j], x[i]) < 1e-10 and rel_u(u2[j], u[i]) < 1e-10
    xo, uo, _, _ = ob.oracle_solve_like(prob, 123, iterations=2, robot=ROBOT)
    n = st[123].n_nodes
    assert rel_x(x[123, :n + 1], xo) < 1e-11 and rel_u(u[123, :n], uo) < 1e-11
    # commands path: trot template of hunter's gait.info, 0.3 m/s forward
    horizon = 40 * sc.DT
    tm = [bp.loadModeSequenceTemplate(sc.ROBOTS[ROBOT]["gait"], "trot")]
    x0 = prob["x0"][:4]
    cmd = np.tile([0.3, 0.0, 0.0, 0.0], (4, 1))
    dev = bp.BatchedSqpMpc(itf, max_batch=4, max_nodes=64)
    dev.setup_commands(0.0, x0, tm, np.zeros(4, np.int32), sc.GAIT_START, cmd, horizon=horizon); dev.enqueue()
    td, xd, ud, _, _ = dev.fetch()
    sched = sc.gait_schedule(itf, "trot", 0.0, horizon)
    targets = [itf.cmdVelToTargetTrajectories(tuple(cmd[b]), 0.0, x0[b], horizon) for b in range(4)]
    host = bp.BatchedSqpMpc(itf, max_batch=4, max_nodes=64)
    th, xh, uh, _, _ = host.run(0.0, x0, sched, targets, horizon=horizon)
    assert np.array_equal(td, th) and _rel(xd, xh) < 1e-9 and _rel(ud, uh) < 1e-9
