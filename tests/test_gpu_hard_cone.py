"""GPU tier of the hard friction cone (tests/test_hard_friction_cone.py has the definition): HIP path against the oracle on the LQ model
(1e-11), the QP step and the solve (1e-11 per physical block), on gaits with single support, double support and flight."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.tolerances import rel_K, rel_u, rel_x  # noqa: E402


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("robot,gait", [("h1:hard", "trot"), ("h1:hard", "flying_trot"), ("h1:hard", "standing_trot"), ("hunter:hard", "trot")])
def test_hard_cone_lq_model_and_solve_match_oracle(robot, gait):
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios as sc
    from tests import oracle_bridge as ob
    itf = sc.interface(robot)
    nx = nu = itf.stateDim
    B, N = 3, 56
    prob = sc.trot_problem(itf, batch=B, n_intervals=40, gait=gait)
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=N, materialize_lq=True, return_gains=True)
    t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    n = st[0].n_nodes
    worst = 0.0
    for b in range(B):
        xo, uo, Ko, so = ob.oracle_solve_like(prob, b, robot=robot)
        assert st[b].step_size == so[0][3]
        assert rel_x(x[b, :n + 1], xo) < 1e-11 and rel_u(u[b, :n], uo) < 1e-11 and rel_K(K[b, :n], Ko) < 1e-10
        worst = max(worst, rel_u(u[b, :n], uo))
    # the LQ model at the accepted iterate
    mpc.stage("linearize"); mpc.synchronize()
    xs = mpc.read("x").reshape(B, N + 1, nx); us = mpc.read("u").reshape(B, N, nu)
    shapes = dict(Q=(nx, nx), R=(nu, nu), q=(nx,), r=(nu,), c=(), perf=(3,))
    dev = {k: mpc.read(k).reshape(B, N, *s) for k, s in shapes.items()}
    om = ob.oracle(robot)
    for b in range(B):
        nodes = ob.oracle_nodes(prob, b, robot=robot)
        for k in range(n):
            o = om.node_lq(nodes["kind"][k], nodes["dt"][k], xs[b, k], us[b, k], xs[b, k + 1], nodes["xref"][k], nodes["mode"][k], nodes["zref"][k], nodes["zdref"][k])
            for name in shapes:
                assert _rel(dev[name][b, k], o[name]) < 1e-11, (name, b, k)
    # it is a different problem from the soft-cone one (same barrier parameters in task.info: the difference is the constraint's curvature and shift)
    soft = bp.BatchedSqpMpc(sc.interface(robot.split(":")[0]), max_batch=B, max_nodes=N)
    _, x2, u2, _, _ = soft.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    assert np.abs(u2[:, :n] - u[:, :n]).max() > 1e-6


def test_hard_cone_fused_and_materialised_modes_agree_bitwise():
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios as sc
    itf = sc.interface("h1:hard")
    prob = sc.trot_problem(itf, batch=4, n_intervals=45, gait="trot")
    out = []
    for mat in (True, False):
        mpc = bp.BatchedSqpMpc(itf, max_batch=4, max_nodes=64, sqp_iterations=2, return_gains=True, materialize_lq=mat)
        out.append(mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True))
    assert all(np.array_equal(out[0][i], out[1][i]) for i in (1, 2, 3))
