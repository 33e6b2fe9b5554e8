"""Whole-body controller (first slice): the reference's WeightedWbc QP for a batch of robots.
CPU tier: the oracle (oracle/wbc_py.py - Lagrangian mechanics with complex-step derivatives, generic active-set QP) against invariants and
hand-checkable answers.  GPU tier: the HIP path (recursive rigid-body algorithms, structurally reduced QP) against the oracle - rigid-body
quantities 1e-10, decision vector 1e-8 relative, KKT residual of the returned vector in the ORACLE's QP 1e-8, same set of tight inequalities,
same fallback (status 1 + previous solution) when the constraints are inconsistent.  Parity status: unpinned (no reference vectors exist)."""
import os

import numpy as np
import pytest

from oracle import reference_py as rp, wbc_py as wp
from tests import oracle_bridge as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _task(robot):
    return os.path.join(ROOT, "assets", robot, "task.info")


def _case(m, mode, rng, speed=0.3, consistent=True):
    nv = 6 + m["nj"]
    x = m["initial_state"].copy()
    x[:6] = [0.2, 0.05, 0.0, 0.0, 0.02, 0.0]
    x[6:] += 0.02 * rng.standard_normal(nv)
    u = rp.weight_compensating_input(m, mode)
    u[12:] = 0.2 * rng.standard_normal(m["nj"])
    q = x[6:] + 0.03 * rng.standard_normal(nv)
    v = speed * rng.standard_normal(nv)
    if consistent:
        v = wp.consistent_measured_state(m, q, v, mode)
    return x, u, wp.rbd_from(m, q, v), q, v


def test_oracle_rigid_body_quantities():
    m = ob.model("h1")
    rng = np.random.default_rng(0)
    x, u, rbd, q, v = _case(m, 3, rng)
    M = wp.mass_matrix(m, q)
    assert np.abs(M - M.T).max() < 1e-12 and np.linalg.eigvalsh(M).min() > 0
    assert np.allclose(M[:3, :3], m["robot_mass"] * np.eye(3), atol=1e-12)          # translating the base moves the whole mass
    # kinetic energy through M equals the sum over bodies
    R, o, axes = wp.fk(m, q)
    T = 0.0
    for b in range(m["nj"] + 1):
        c = o[b] + R[b] @ m["com"][b]
        Jv, Jw = wp.point_jacobian(m, R, o, axes, b, c)
        T += 0.5 * m["mass"][b] * (Jv @ v) @ (Jv @ v) + 0.5 * (Jw @ v) @ (R[b] @ m["inertia"][b] @ R[b].T) @ (Jw @ v)
    assert abs(0.5 * v @ M @ v - T) < 1e-12 * max(1.0, T)
    # at rest the nonlinear effects are the gravity torques: the base feels the whole weight, and energy conservation d(T + V)/dt = 0
    # along the unforced motion  M vdot = -nle  checks the velocity-product terms
    g0 = wp.nonlinear_effects(m, q, np.zeros_like(v))
    assert abs(g0[2] - m["robot_mass"] * 9.81) < 1e-9 and np.abs(g0[:2]).max() < 1e-9
    nle = wp.nonlinear_effects(m, q, v)
    vdot = np.linalg.solve(M, -nle)
    h = 1e-6
    e = lambda qq, vv: 0.5 * vv @ wp.mass_matrix(m, qq) @ vv + wp.potential(m, qq)     # noqa: E731
    de = (e(q + h * v, v + h * vdot) - e(q - h * v, v - h * vdot)) / (2 * h)
    assert abs(de) < 1e-5 * max(1.0, abs(e(q, v)))
    # Jdot v against a finite difference of the contact velocity along the motion (vdot = 0)
    J = wp.contact_jacobian(m, q)
    djv = wp.jdot_v(wp.contact_jacobian, m, q, v)
    fd = (wp.contact_jacobian(m, q + h * v) @ v - wp.contact_jacobian(m, q - h * v) @ v) / (2 * h)
    assert np.abs(djv - fd).max() < 1e-6 and np.abs(J @ v - (np.array(wp.contact_points(m, *wp.fk(m, q + h * v)[:2])) - np.array(wp.contact_points(m, *wp.fk(m, q - h * v)[:2]))).ravel() / (2 * h)).max() < 1e-7
    # centroidal momentum matrix of the WBC restatement = the MPC oracle's
    A, com = wp.centroidal_momentum_matrix(m, q)
    A2, com2 = ob.oracle("h1").cmm(q)
    assert np.abs(A - A2).max() < 1e-11 and np.abs(com - com2).max() < 1e-13


@pytest.mark.parametrize("mode", [3, 1, 2, 0])
def test_oracle_qp_satisfies_kkt(mode):
    m = ob.model("h1")
    st = wp.load_settings(_task("h1"), 10)
    assert st["friction"] == 0.3 and st["contact_tolerance"] == 5.0 and st["w_swing"] == 100.0 and list(st["torque_limits"]) == [500.0] * 5
    x, u, rbd, q, v = _case(m, mode, np.random.default_rng(mode))
    sol, p = wp.update(m, st, x, u, rbd, mode)
    assert p["status"] == 0
    nst = sum(wp.mode_flags(mode))
    assert p["Aeq"].shape == (16 + 3 * (4 - nst), 38) and p["D"].shape == (20 + 5 * nst + 3 * (4 - nst) + 6 * nst, 38)    # incl. the zero rows the reference allocates for swing contacts
    G = np.vstack([p["Aeq"], p["D"]])
    assert np.abs(p["H"] @ sol + p["g"] + G.T @ p["mult"]).max() < 1e-8                     # stationarity
    assert np.abs(p["Aeq"] @ sol - p["beq"]).max() < 1e-9 and (p["D"] @ sol - p["f"]).max() < 1e-9
    mu = p["mult"][len(p["Aeq"]):]
    assert mu.min() >= 0.0 and np.abs(mu * (p["D"] @ sol - p["f"])).max() < 1e-6          # dual feasibility, complementarity
    # the reference's no-contact-motion rows amount to J a + Jdot v = tolerance for every stance contact; swing forces vanish
    acc = p["J"] @ sol[:16] + p["djv"]
    for i, fl in enumerate(wp.mode_flags(mode)):
        if fl:
            assert np.abs(acc[3 * i:3 * i + 3] - 5.0).max() < 1e-8
        else:
            assert np.abs(sol[16 + 3 * i:19 + 3 * i]).max() < 1e-12
    # torques are what the equations of motion give
    tau = p["M"][6:] @ sol[:16] - p["J"][:, 6:].T @ sol[16:28] + p["nle"][6:]
    assert np.abs(tau - sol[28:]).max() < 1e-8


def test_oracle_falls_back_when_the_contact_equalities_are_inconsistent():
    m = ob.model("h1")
    st = wp.load_settings(_task("h1"), 10)
    x, u, rbd, q, v = _case(m, 3, np.random.default_rng(5), consistent=False)      # a stance foot that rotates: its two points cannot both
    last = np.arange(38.0)                                                            # accelerate by (5, 5, 5)
    sol, p = wp.update(m, st, x, u, rbd, 3, last=last)
    assert p["status"] == 1 and np.array_equal(sol, last)


# ------------------------------------------------------------------------------------------------------------------- GPU tier
def _tight(p, sol, tol=1e-7):
    return set(np.nonzero(np.abs(p["D"] @ sol - p["f"]) < tol)[0].tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["h1", "g1", "hunter", "openloong"])
def test_hip_wbc_matches_oracle(robot):
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios as sc
    m = ob.model(robot)
    nj = m["nj"]
    nv, n = 6 + nj, 6 + nj + 12 + nj
    st = wp.load_settings(_task(robot), nj)
    itf = sc.interface(robot)
    rng = np.random.default_rng(11)
    modes = [3, 1, 2, 0, 3, 1, 2, 3]
    cases = [_case(m, md, rng, speed=0.4) for md in modes]
    wbc = bp.WeightedWbc(itf, max_batch=len(modes))
    assert wbc.numDecisionVars == n
    sol, status, dbg = wbc.update([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], modes, debug=True)
    active_seen = 0
    for b, (x, u, rbd, q, v) in enumerate(cases):
        so, p = wp.update(m, st, x, u, rbd, modes[b])
        assert p["status"] == 0 and status[b] == 0
        d = dbg[b]
        M = d[:nv * nv].reshape(nv, nv); nle = d[nv * nv:nv * nv + nv]; J = d[nv * nv + nv:nv * nv + nv + 12 * nv].reshape(12, nv)
        djv = d[nv * nv + nv + 12 * nv:nv * nv + nv + 12 * nv + 12]
        rel = lambda a_, b_: float(np.abs(a_ - b_).max() / max(1.0, np.abs(b_).max()))        # noqa: E731
        assert rel(M, p["M"]) < 1e-10 and rel(nle, p["nle"]) < 1e-10 and rel(J, p["J"]) < 1e-12 and rel(djv, p["djv"]) < 1e-10
        # the returned vector satisfies the ORACLE's full QP: feasible, stationary on its tight set, same objective value
        assert np.abs(p["Aeq"] @ sol[b] - p["beq"]).max() < 1e-7 and (p["D"] @ sol[b] - p["f"]).max() < 1e-7
        tight = sorted(_tight(p, sol[b]))
        Ga = np.vstack([p["Aeq"], p["D"][tight]])
        grad = p["H"] @ sol[b] + p["g"]
        lam = np.linalg.lstsq(Ga.T, -grad, rcond=None)[0]
        assert np.abs(grad + Ga.T @ lam).max() < 1e-6 * max(1.0, np.abs(p["g"]).max())
        obj = lambda xx: 0.5 * xx @ p["H"] @ xx + p["g"] @ xx                                  # noqa: E731
        assert abs(obj(sol[b]) - obj(so)) < 1e-9 * max(1.0, abs(obj(so)))
        assert _tight(p, sol[b]) == _tight(p, so)
        if robot in ("h1", "hunter"):
            # five joints per leg: the minimiser is unique.  With six (G1, OpenLoong) a swing leg keeps one direction - the rotation
            # about the line through its two sole points - that neither a task nor a constraint sees, the QP has a line of minimisers
            # (qpOASES picks one through its regularisation) and only the gauge-free statements above are comparable
            assert rel(sol[b], so) < 1e-8, (robot, b, modes[b], np.abs(sol[b] - so).max())
        nst = sum(wp.mode_flags(modes[b]))
        active_seen += len(_tight(p, so)) > 6 * nst + 3 * (4 - nst)      # more than the contact pairs and the all-zero rows: a pyramid / torque row is tight
    assert active_seen >= 1
    # the handle keeps the last solutions: an unsolvable QP (rotating stance foot) returns them and says so
    bad = [_case(m, 3, rng, consistent=False) for _ in modes]
    sol2, status2 = wbc.update([c[0] for c in bad], [c[1] for c in bad], [c[2] for c in bad], 3)
    assert np.all(status2 == 1) and np.array_equal(sol2, sol)
    wbc.reset()
    sol3, status3 = wbc.update([c[0] for c in bad], [c[1] for c in bad], [c[2] for c in bad], 3)
    assert np.all(status3 == 1) and np.abs(sol3).max() == 0.0


@pytest.mark.gpu
def test_hip_wbc_batch_independence_and_torque_limits():
    """A robot's result does not depend on its neighbours (bitwise); with the torque limit lowered to 20 N m the limit rows become active
    and the torques respect them."""
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios as sc
    m = ob.model("h1")
    itf = sc.interface("h1")
    rng = np.random.default_rng(3)
    modes = [1, 2, 3, 1] * 16
    cases = [_case(m, md, rng, speed=0.2) for md in modes]
    big = bp.WeightedWbc(itf, max_batch=64)
    sol, status = big.update([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], modes)
    assert np.all(status == 0)
    one = bp.WeightedWbc(itf, max_batch=1)
    for b in (0, 17, 63):
        s1, st1 = one.update(cases[b][0], cases[b][1], cases[b][2], modes[b])
        assert st1[0] == 0 and np.array_equal(s1[0], sol[b])
    text = open(_task("h1")).read()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "task.info")
        open(path, "w").write(text.replace("(3,0) 500\n  (4,0) 500\n}\nfrictionConeTask", "(3,0) 20\n  (4,0) 500\n}\nfrictionConeTask"))
        st = wp.load_settings(path, 10)
        assert list(st["torque_limits"]) == [500.0, 500.0, 500.0, 20.0, 500.0]
        lim = bp.WeightedWbc(itf, taskFile=path, max_batch=4)
        s, stt = lim.update([c[0] for c in cases[:4]], [c[1] for c in cases[:4]], [c[2] for c in cases[:4]], modes[:4])
        hit = 0
        for b in range(4):
            so, p = wp.update(m, st, cases[b][0], cases[b][1], cases[b][2], modes[b])
            assert stt[b] == p["status"] == 0 and np.abs(s[b] - so).max() < 1e-7 * max(1.0, np.abs(so).max())
            assert np.abs(s[b][28:][[3, 8]]).max() <= 20.0 + 1e-7
            hit += np.abs(s[b][28:][[3, 8]]).max() > 20.0 - 1e-6
        assert hit >= 1
