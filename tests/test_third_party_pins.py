"""Pins pieces of the ORACLE to third-party implementations that ship in this image (scipy), not to the builder's own numpy: the oracle is a
restatement of un-vendored dependencies (the reference holds no vectors for this path), and wherever an independent,
published implementation of the same mathematics is at hand it is used as the witness.

* Dormand-Prince 5(4): boost::numeric::odeint's runge_kutta_dopri5 (the reference's rollout, rollout.timeStep / AbsTolODE / RelTolODE of
  task.info) and scipy.integrate.RK45 implement the same published tableau; one step and its embedded error estimate must agree.
* ZYX Euler angles: the reference's state holds base orientation as (yaw, pitch, roll) ([OCS2-upstream] getRotationMatrixFromZyxEulerAngles,
  getMappingFromEulerAnglesZyxDerivativeToGlobalAngularVelocity); scipy.spatial.transform.Rotation is the witness for the rotation matrix, the
  rate map and the rotation vector of the WBC's orientation error.
* URDF rpy: scipy's extrinsic 'xyz' is the URDF convention (fixed-axis roll, pitch, yaw).
* The swing-leg cubic (ocs2_bipedal_robot/src/foot_planner/CubicSpline.cpp) against scipy.interpolate.CubicHermiteSpline.
"""
import numpy as np
import pytest
from scipy.integrate._ivp import rk as scipy_rk
from scipy.interpolate import CubicHermiteSpline
from scipy.spatial.transform import Rotation

from oracle import ingest, reference_py as rp, wbc_py


def test_dopri5_step_is_scipy_rk45_step():
    rng = np.random.default_rng(7)
    M = rng.standard_normal((6, 6))
    f = lambda t, x: np.tanh(M @ x) + np.sin(3.0 * t) * x[::-1]                              # noqa: E731
    x = rng.standard_normal(6)
    t, dt = 0.3, 0.015
    k0 = f(t, x)
    x_new, dxdt_new, x_err = rp.dopri5_step(f, x, k0, t, dt)
    K = np.empty((7, 6))
    y_new, f_new = scipy_rk.rk_step(f, t, x, k0, dt, scipy_rk.RK45.A, scipy_rk.RK45.B, scipy_rk.RK45.C, K)
    err = dt * (K.T @ scipy_rk.RK45.E)
    assert np.abs(x_new - y_new).max() < 1e-15 and np.abs(dxdt_new - f_new).max() < 1e-14
    # error estimate: odeint reports 5th-order minus embedded 4th-order solution, scipy the opposite sign (only |err| enters either controller)
    assert np.abs(x_err + err).max() < 1e-16
    # the tableau coefficient by coefficient
    for s in range(5):
        assert np.allclose(rp.DOPRI5_A[s], scipy_rk.RK45.A[s + 1][:s + 1], rtol=0, atol=1e-16)
    assert np.allclose(rp.DOPRI5_C, scipy_rk.RK45.C[1:], rtol=0, atol=1e-16)
    assert np.allclose(rp.DOPRI5_B, scipy_rk.RK45.B, rtol=0, atol=1e-16)
    assert np.allclose(rp.DOPRI5_DB, -scipy_rk.RK45.E, rtol=0, atol=1e-16)


def test_zyx_rotation_and_rate_map_are_scipys():
    rng = np.random.default_rng(11)
    for _ in range(20):
        zyx = rng.uniform(-1.2, 1.2, 3)
        R = Rotation.from_euler("ZYX", zyx).as_matrix()                                     # intrinsic z, y', x''
        assert np.abs(rp.rot_zyx(zyx) - R).max() < 1e-15
        assert np.abs(wbc_py.rot_zyx(zyx) - R).max() < 1e-15
        assert np.abs(ingest.rot_z(zyx[0]) @ ingest.rot_y(zyx[1]) @ ingest.rot_x(zyx[2]) - R).max() < 1e-15
        # world angular velocity = E(theta) thetadot: skew(w) = Rdot R' with Rdot by central differences of scipy's matrix
        rates = rng.standard_normal(3)
        h = 1e-6
        Rd = (Rotation.from_euler("ZYX", zyx + h * rates).as_matrix() - Rotation.from_euler("ZYX", zyx - h * rates).as_matrix()) / (2 * h)
        W = Rd @ R.T
        w = np.array([W[2, 1], W[0, 2], W[1, 0]])
        assert np.abs(wbc_py.euler_rate_map(zyx) @ rates - w).max() < 1e-8
        # orientation error of the WBC's base task: rotation vector of Rl Rr'
        other = Rotation.from_euler("ZYX", zyx + 0.3 * rng.standard_normal(3))
        e = wbc_py.rotation_error_in_world(R, other.as_matrix())
        assert np.abs(e - (Rotation.from_matrix(R) * other.inv()).as_rotvec()).max() < 1e-12
    # small-angle branch
    tiny = Rotation.from_rotvec([1e-6, -2e-6, 5e-7])
    assert np.abs(wbc_py.rotation_error_in_world(tiny.as_matrix(), np.eye(3)) - tiny.as_rotvec()).max() < 1e-15


def test_urdf_rpy_is_scipys_extrinsic_xyz():
    rng = np.random.default_rng(13)
    for _ in range(10):
        rpy = rng.uniform(-3.0, 3.0, 3)
        assert np.abs(ingest.rpy_to_rot(rpy) - Rotation.from_euler("xyz", rpy).as_matrix()).max() < 1e-15


def test_swing_cubic_is_a_cubic_hermite_spline():
    """CubicSpline.cpp: the cubic through (t0, p0, v0) and (t1, p1, v1); position, velocity and the value at the knots."""
    rng = np.random.default_rng(17)
    for _ in range(5):
        t0 = rng.uniform(-1.0, 1.0); t1 = t0 + rng.uniform(0.05, 0.6)
        p0, p1, v0, v1 = rng.standard_normal(4)
        sp = rp.CubicSpline((t0, p0, v0), (t1, p1, v1))
        w = CubicHermiteSpline([t0, t1], [p0, p1], [v0, v1])
        for t in np.linspace(t0, t1, 9):
            assert abs(sp.position(t) - w(t)) < 1e-13 and abs(sp.velocity(t) - w(t, 1)) < 1e-12
