"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's weighted whole-body controller QP
(SURVEY.md section 8(f) rank 4): bipedal_wbc/src/WbcBase.cpp (task builders) + bipedal_wbc/src/WeightedWbc.cpp:20-84 (QP), numpy.

Independent of the product's recursive rigid-body algorithms on purpose:
  * mass matrix from its definition  M = sum_b m_b Jv_b' Jv_b + Jw_b' I_b Jw_b  (body Jacobians = d(com) / dq, world angular velocity map),
  * nonlinear effects from the Lagrangian,  nle = Mdot v - 1/2 d(v' M v)/dq + dV/dq,  every derivative by the complex-step method
    (exact to rounding: the kinematics below are analytic functions of q),
  * Jdot v and Adot v as complex-step directional derivatives along v,
  * the QP by a generic dense active-set method on the full 38-variable problem (no structural elimination).
[Pinocchio / OCS2 upstream] semantics restated from their published definitions (crba, nonLinearEffects, getFrameJacobian
LOCAL_WORLD_ALIGNED, getFrameJacobianTimeVariation, dccrba, CentroidalModelRbdConversions::computeBaseKinematicsFromCentroidalModel,
rotationErrorInWorld) - recalled, not read; parity status UNPINNED like the rest of the oracle.

Generalised coordinates (centroidal_model::createPinocchioInterface: translation + ZYX Euler composite base joint):
  q = [p (3), yaw, pitch, roll, joints (nj)],  v = dq/dt (base linear velocity in the world frame, Euler-angle rates, joint rates).
Decision variables of the QP (WbcBase.cpp:40): x = [vdot (nv), F (3 per contact, 12), tau (nj)].

Reference behaviour preserved although it looks unintended (results must be the reference's):
  * formulateNoContactMotionTask (WbcBase.cpp:171-198): the two one-sided rows of a stance contact read  J a + Jdot v <= tol  and
    J a + Jdot v >= tol, i.e. together the EQUALITY  J a + Jdot v = tol  (tolerance 5 in task.info:332-335);
  * formulateBaseAccelPDTask (WbcBase.cpp:274): the "angular velocity error" is  desiredBaseVelocity.head<3>(3) - measured.head<3>(3),
    which is the LINEAR velocity error (head<3>(3) = the first three entries);
  * the joint acceleration of the desired base motion is zero (WbcBase.cpp:243) and inputLast_ is unused.
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module."""
import numpy as np

from . import ingest

GRAVITY = 9.81


# ---------------------------------------------------------------------------------------------------------------------
# settings (WbcBase::loadTasksSetting, WbcBase.cpp:405-447; WeightedWbc::loadTasksSetting, WeightedWbc.cpp:100-116)
# ---------------------------------------------------------------------------------------------------------------------
def load_settings(task_path, nj):
    t = ingest.parse_info(task_path)
    per_leg = nj // 2
    return dict(torque_limits=ingest.load_matrix(t, "torqueLimitsTask", per_leg, 1)[:, 0],
                friction=float(ingest.info_get(t, "frictionConeTask.frictionCoefficient")),
                swing_kp=float(ingest.info_get(t, "swingLegTask.kp")), swing_kd=float(ingest.info_get(t, "swingLegTask.kd")),
                base_kp=ingest.load_matrix(t, "baseAccelPDTask.baseKp", 6, 1)[:, 0], base_kd=ingest.load_matrix(t, "baseAccelPDTask.baseKd", 6, 1)[:, 0],
                # absent in the Hunter configuration: loadPtreeValue keeps noContactMotionTolerance_{} = 0 (WbcBase.h:131, WbcBase.cpp:446)
                contact_tolerance=float(ingest.info_get(t, "noContactMotionTask.tolerance") or 0.0),
                w_swing=float(ingest.info_get(t, "weight.swingLeg")), w_base=float(ingest.info_get(t, "weight.baseAccel")),
                w_force=float(ingest.info_get(t, "weight.contactForce")))


# ---------------------------------------------------------------------------------------------------------------------
# kinematics, valid for complex q (complex-step differentiation)
# ---------------------------------------------------------------------------------------------------------------------
def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def _ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _axis_angle(a, th):
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) * np.cos(th) + (1 - np.cos(th)) * np.outer(a, a) + np.sin(th) * K


def fk(m, q):
    """Per-body world rotation / origin and the world axis / origin of every generalised coordinate."""
    nj = m["nj"]
    Rz, Ry, Rx = _rz(q[3]), _ry(q[4]), _rx(q[5])
    R = [Rz @ Ry @ Rx]
    o = [np.array(q[0:3])]
    axes = [np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 0, 1.0]),
            np.array([0, 0, 1.0]), Rz @ np.array([0, 1.0, 0]), Rz @ Ry @ np.array([1.0, 0, 0])]
    for j in range(nj):
        lam = m["parent"][j]
        Rj = R[lam] @ m["Rfix"][j]
        oj = o[lam] + R[lam] @ m["pfix"][j]
        axes.append(Rj @ m["axis"][j])
        R.append(Rj @ _axis_angle(m["axis"][j], q[6 + j]))
        o.append(oj)
    return R, o, axes


def _on_path(m, g, body):
    """Does generalised coordinate g move `body`?  (base coordinates move everything)"""
    if g < 6:
        return True
    return ingest.is_ancestor_or_self(m, g - 5, body)


def point_jacobian(m, R, o, axes, body, p):
    """d(world point p fixed to `body`) / dq (3 x nv) and the angular velocity map of the body (3 x nv)."""
    nv = 6 + m["nj"]
    Jv = np.zeros((3, nv), dtype=p.dtype if np.iscomplexobj(p) else R[0].dtype)
    Jw = np.zeros((3, nv), dtype=Jv.dtype)
    for g in range(nv):
        if not _on_path(m, g, body):
            continue
        if g < 3:
            Jv[:, g] = axes[g]
        else:
            og = o[0] if g < 6 else o[g - 5]
            Jv[:, g] = np.cross(axes[g], p - og)
            Jw[:, g] = axes[g]
    return Jv, Jw


def contact_points(m, R, o):
    return [o[b] + R[b] @ m["contact_off"][i] for i, b in enumerate(m["contact_body"])]


def contact_jacobian(m, q):
    """Stacked d(contact position) / dq, 12 x nv (pinocchio::getFrameJacobian, LOCAL_WORLD_ALIGNED, linear rows)."""
    R, o, axes = fk(m, q)
    pts = contact_points(m, R, o)
    return np.vstack([point_jacobian(m, R, o, axes, b, pts[i])[0] for i, b in enumerate(m["contact_body"])])


def base_angular_jacobian(m, q):
    R, o, axes = fk(m, q)
    return point_jacobian(m, R, o, axes, 0, o[0])[1]


def mass_matrix(m, q):
    R, o, axes = fk(m, q)
    nv = 6 + m["nj"]
    M = np.zeros((nv, nv), dtype=R[0].dtype)
    for b in range(m["nj"] + 1):
        c = o[b] + R[b] @ m["com"][b]
        Jv, Jw = point_jacobian(m, R, o, axes, b, c)
        Iw = R[b] @ m["inertia"][b] @ R[b].T
        M = M + m["mass"][b] * (Jv.T @ Jv) + Jw.T @ Iw @ Jw
    return M


def potential(m, q):
    R, o, _ = fk(m, q)
    return sum(m["mass"][b] * GRAVITY * (o[b] + R[b] @ m["com"][b])[2] for b in range(m["nj"] + 1))


def centroidal_momentum_matrix(m, q):
    """A(q) with h = [m v_com; sum_b I_b w_b + (c_b - c) x m_b v_cb] = A v, and the centre of mass."""
    R, o, axes = fk(m, q)
    nb = m["nj"] + 1
    cs = [o[b] + R[b] @ m["com"][b] for b in range(nb)]
    mt = m["mass"].sum()
    com = sum(m["mass"][b] * cs[b] for b in range(nb)) / mt
    nv = 6 + m["nj"]
    A = np.zeros((6, nv), dtype=R[0].dtype)
    for b in range(nb):
        Jv, Jw = point_jacobian(m, R, o, axes, b, cs[b])
        Iw = R[b] @ m["inertia"][b] @ R[b].T
        d = cs[b] - com
        K = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
        A[:3] += m["mass"][b] * Jv
        A[3:] += Iw @ Jw + m["mass"][b] * (K @ Jv)
    return A, com


_H = 1e-30


def nonlinear_effects(m, q, v):
    """C(q, v) v + g(q) (pinocchio::nonLinearEffects) from the Lagrangian with complex-step derivatives."""
    q = np.asarray(q, float)
    nv = len(q)
    Mdot = np.imag(mass_matrix(m, q + 1j * _H * v)) / _H                 # dM/dt along qdot = v
    out = Mdot @ v
    for k in range(nv):
        e = np.zeros(nv); e[k] = 1.0
        dM = np.imag(mass_matrix(m, q + 1j * _H * e)) / _H
        dV = np.imag(potential(m, q + 1j * _H * e)) / _H
        out[k] += -0.5 * v @ dM @ v + dV
    return out


def jdot_v(jac_fn, m, q, v):
    """(d/dt J(q(t))) v along qdot = v."""
    return (np.imag(jac_fn(m, np.asarray(q, float) + 1j * _H * np.asarray(v, float))) / _H) @ v


# ---------------------------------------------------------------------------------------------------------------------
# Euler ZYX helpers ([OCS2-upstream] ocs2_robotic_tools RotationTransforms.h / RotationDerivativesTransforms.h, recalled)
# ---------------------------------------------------------------------------------------------------------------------
def euler_rate_map(zyx):
    """E(theta): world angular velocity = E thetadot  (columns: world axes of the z, y', x'' rotations)."""
    z, y = zyx[0], zyx[1]
    return np.array([[0.0, -np.sin(z), np.cos(z) * np.cos(y)], [0.0, np.cos(z), np.sin(z) * np.cos(y)], [1.0, 0.0, -np.sin(y)]])


def euler_rates_from_angular_velocity(zyx, w):
    return np.linalg.solve(euler_rate_map(zyx), w)


def angular_acceleration_from_euler(zyx, rates, accel):
    """getGlobalAngularAccelerationFromEulerAnglesZyxDerivatives: d/dt (E thetadot) = E thetaddot + Edot thetadot."""
    Edot = np.imag(euler_rate_map(np.asarray(zyx, float) + 1j * _H * np.asarray(rates, float))) / _H
    return euler_rate_map(zyx) @ accel + Edot @ rates


def rotation_error_in_world(Rl, Rr):
    """rotationErrorInWorld(lhs, rhs) = rotation vector of lhs rhs' ([OCS2-upstream] rotationMatrixToRotationVector: small-angle
    branch (0.5 - tmp / 6) skew with tmp = (trace - 3) / 2 when -tmp < 1e-8, otherwise theta / (2 sin theta) skew)."""
    E = Rl @ Rr.T
    tr = np.trace(E)
    skew = np.array([E[2, 1] - E[1, 2], E[0, 2] - E[2, 0], E[1, 0] - E[0, 1]])
    tmp = 0.5 * (tr - 3.0)
    if -tmp < 1e-8:
        return (0.5 - tmp / 6.0) * skew
    th = np.arccos(0.5 * (tr - 1.0))
    return th / (2.0 * np.sin(th)) * skew


def rot_zyx(zyx):
    return _rz(zyx[0]) @ _ry(zyx[1]) @ _rx(zyx[2])


# ---------------------------------------------------------------------------------------------------------------------
# the controller (WbcBase::update, WeightedWbc::update)
# ---------------------------------------------------------------------------------------------------------------------
def mode_flags(mode):
    return {0: [0, 0, 0, 0], 1: [1, 1, 0, 0], 2: [0, 0, 1, 1], 3: [1, 1, 1, 1]}[int(mode)]


def measured_state(m, rbd):
    """WbcBase::updateMeasured (WbcBase.cpp:58-77): rbd = [zyx, pos, joints, angular vel (world), linear vel, joint vel]."""
    nj = m["nj"]
    nv = 6 + nj
    q = np.concatenate([rbd[3:6], rbd[0:3], rbd[6:6 + nj]])
    v = np.concatenate([rbd[nv + 3:nv + 6], euler_rates_from_angular_velocity(q[3:6], rbd[nv:nv + 3]), rbd[nv + 6:nv + 6 + nj]])
    return q, v


def desired_state(m, x, u):
    """CentroidalModelPinocchioMapping::getPinocchioJointPosition / Velocity ([OCS2-upstream]): q = x[6:], v_base = Ab^-1 (m hbar - Aj vj)."""
    nj = m["nj"]
    q = np.array(x[6:], float)
    A, com = centroidal_momentum_matrix(m, q)
    vj = np.array(u[12:12 + nj], float)
    vb = np.linalg.solve(A[:, :6], m["robot_mass"] * np.asarray(x[:6], float) - A[:, 6:] @ vj)
    return q, np.concatenate([vb, vj]), A, com


def base_kinematics_from_centroidal(m, x, u):
    """CentroidalModelRbdConversions::computeBaseKinematicsFromCentroidalModel ([OCS2-upstream], recalled) with zero joint acceleration."""
    q, v, A, com = desired_state(m, x, u)
    R, o, _ = fk(m, q)
    pts = contact_points(m, R, o)
    F = np.asarray(u[:12], float).reshape(4, 3)
    hdot = np.concatenate([F.sum(axis=0) + np.array([0, 0, -GRAVITY * m["robot_mass"]]), sum(np.cross(pts[i] - com, F[i]) for i in range(4))])
    adot_v = (np.imag(centroidal_momentum_matrix(m, q + 1j * _H * v)[0]) / _H) @ v
    qbdd = np.linalg.solve(A[:, :6], hdot - adot_v)
    pose = q[:6].copy()
    vel = np.concatenate([v[:3], euler_rate_map(q[3:6]) @ v[3:6]])
    acc = np.concatenate([qbdd[:3], angular_acceleration_from_euler(q[3:6], v[3:6], qbdd[3:])])
    return pose, vel, acc


def formulate(m, st, x_des, u_des, rbd, mode):
    """Constraint task (a x = b, d x <= f) and weighted cost task (a x ~ b) of WeightedWbc::update, in the reference's row order."""
    nj = m["nj"]
    nv = 6 + nj
    n = nv + 12 + nj
    flags = mode_flags(mode)
    nst = sum(flags)
    q, v = measured_state(m, np.asarray(rbd, float))
    M = mass_matrix(m, q)
    nle = nonlinear_effects(m, q, v)
    J = contact_jacobian(m, q)
    djv = jdot_v(contact_jacobian, m, q, v)
    Jb = base_angular_jacobian(m, q)
    dJb_v = jdot_v(base_angular_jacobian, m, q, v)
    # ---- constraints: EoM + torque limits + friction cone + no contact motion (WeightedWbc.cpp:86-92)
    S = np.hstack([np.zeros((nj, 6)), np.eye(nj)])
    a_eom = np.hstack([M, -J.T, -S.T])
    b_eom = -nle
    d_tau = np.zeros((2 * nj, n)); d_tau[:nj, nv + 12:] = np.eye(nj); d_tau[nj:, nv + 12:] = -np.eye(nj)
    f_tau = np.tile(st["torque_limits"], 4)
    a_fc = np.zeros((3 * (4 - nst), n)); j = 0
    for i in range(4):
        if not flags[i]:
            a_fc[3 * j:3 * j + 3, nv + 3 * i:nv + 3 * i + 3] = np.eye(3); j += 1
    mu = st["friction"]
    pyr = np.array([[0, 0, -1], [1, 0, -mu], [-1, 0, -mu], [0, 1, -mu], [0, -1, -mu]], float)
    d_fc = np.zeros((5 * nst + 3 * (4 - nst), n)); j = 0
    for i in range(4):
        if flags[i]:
            d_fc[5 * j:5 * j + 5, nv + 3 * i:nv + 3 * i + 3] = pyr; j += 1
    d_nc = np.zeros((6 * nst, n)); f_nc = np.zeros(6 * nst); j = 0
    tol = st["contact_tolerance"]
    for i in range(4):
        if flags[i]:
            d_nc[6 * j:6 * j + 3, :nv] = J[3 * i:3 * i + 3]
            d_nc[6 * j + 3:6 * j + 6, :nv] = -J[3 * i:3 * i + 3]
            f_nc[6 * j:6 * j + 3] = -djv[3 * i:3 * i + 3] + tol
            f_nc[6 * j + 3:6 * j + 6] = djv[3 * i:3 * i + 3] - tol
            j += 1
    Aeq = np.vstack([a_eom, a_fc]); beq = np.concatenate([b_eom, np.zeros(len(a_fc))])
    D = np.vstack([d_tau, d_fc, d_nc]); f = np.concatenate([f_tau, np.zeros(len(d_fc)), f_nc])
    # ---- weighted tasks: swing leg, base acceleration PD, contact force (WeightedWbc.cpp:94-101)
    R, o, axes = fk(m, q)
    pos_m = contact_points(m, R, o)
    vel_m = (J @ v).reshape(4, 3)
    qd, vd, _, _ = desired_state(m, x_des, u_des)
    Rd, od, _ = fk(m, qd)
    pos_d = contact_points(m, Rd, od)
    vel_d = (contact_jacobian(m, qd) @ vd).reshape(4, 3)
    a_sw = np.zeros((3 * (4 - nst), n)); b_sw = np.zeros(3 * (4 - nst)); j = 0
    for i in range(4):
        if not flags[i]:
            acc = st["swing_kp"] * (pos_d[i] - pos_m[i]) + st["swing_kd"] * (vel_d[i] - vel_m[i])
            a_sw[3 * j:3 * j + 3, :nv] = J[3 * i:3 * i + 3]
            b_sw[3 * j:3 * j + 3] = acc - djv[3 * i:3 * i + 3]
            j += 1
    a_ba = np.zeros((6, n)); a_ba[:6, :6] = np.eye(6); a_ba[3:6, :nv] = Jb
    pose_d, vel_d6, acc_d = base_kinematics_from_centroidal(m, x_des, u_des)
    vel_meas = np.concatenate([v[:3], euler_rate_map(q[3:6]) @ v[3:6]])
    e_pos = pose_d[:3] - q[:3]
    e_lin = vel_d6[:3] - vel_meas[:3]
    e_rot = rotation_error_in_world(rot_zyx(pose_d[3:6]), rot_zyx(q[3:6]))
    e_ang = vel_d6[:3] - vel_meas[:3]          # reference: .head<3>(3) is the first three entries (linear velocity), WbcBase.cpp:274
    b_ba = np.concatenate([acc_d[:3] + st["base_kp"][:3] * e_pos + st["base_kd"][:3] * e_lin,
                           acc_d[3:] + st["base_kp"][3:] * e_rot + st["base_kd"][3:] * e_ang - dJb_v])
    a_cf = np.zeros((12, n)); a_cf[:, nv:nv + 12] = np.eye(12)
    b_cf = np.asarray(u_des[:12], float)
    Aw = np.vstack([st["w_swing"] * a_sw, st["w_base"] * a_ba, st["w_force"] * a_cf])
    bw = np.concatenate([st["w_swing"] * b_sw, st["w_base"] * b_ba, st["w_force"] * b_cf])
    return dict(Aeq=Aeq, beq=beq, D=D, f=f, Aw=Aw, bw=bw, M=M, nle=nle, J=J, djv=djv, q=q, v=v, n=n)


class Infeasible(Exception):
    pass


FEAS_TOL = 1e-8      # relative to max(1, |rhs|): equality rows (explicit ones and opposite pairs) that cannot all hold


def solve_qp(H, g, Aeq, beq, D, f, max_iter=400, tol=1e-9):
    """min 1/2 x'Hx + g'x  s.t. Aeq x = beq, D x <= f  by a primal-dual active-set iteration on dense KKT systems (null-space form, rank
    deficient working sets allowed).  H may be singular as long as it is positive definite on the null space of the working set.
    Returns x, multipliers of [Aeq; D] (zero for inactive rows), the working set, iterations."""
    n = len(g)
    # a pair of opposite one-sided rows with opposite bounds (d x <= f and -d x <= -f) is the equality d x = f: taken out of the
    # inequality set up front (the reference's no-contact-motion task is made of such pairs; kept as inequalities they are linearly
    # dependent members of every working set and their multipliers are not unique)
    n_eq0 = len(Aeq)
    pair_of = {}
    for i in range(len(D)):
        for j in range(i + 1, len(D)):
            if i not in pair_of and j not in pair_of and np.array_equal(D[i], -D[j]) and f[i] == -f[j]:
                pair_of[i] = j; pair_of[j] = i
    firsts = sorted(i for i in pair_of if i < pair_of[i])
    if firsts:
        keep = [i for i in range(len(D)) if i not in pair_of]
        x, mult, work, iters = solve_qp(H, g, np.vstack([Aeq, D[firsts]]), np.concatenate([beq, f[firsts]]), D[keep], f[keep], max_iter, tol)
        full = np.zeros(n_eq0 + len(D))
        full[:n_eq0] = mult[:n_eq0]
        for k, i in enumerate(firsts):       # multiplier of the equality split by sign between the two one-sided rows
            lam = mult[n_eq0 + k]
            full[n_eq0 + i] = max(lam, 0.0); full[n_eq0 + pair_of[i]] = max(-lam, 0.0)
        for k, i in enumerate(keep):
            full[n_eq0 + i] = mult[n_eq0 + len(firsts) + k]
        return x, full, sorted(list(pair_of) + [keep[w] for w in work]), iters
    work = []
    for it in range(max_iter):
        C = np.vstack([Aeq] + [D[i:i + 1] for i in work]) if (len(Aeq) or work) else np.zeros((0, n))
        d = np.concatenate([beq] + [f[i:i + 1] for i in work]) if len(C) else np.zeros(0)
        U, s, Vt = np.linalg.svd(C, full_matrices=True)
        r = int((s > 1e-11 * max(1.0, s[0] if len(s) else 1.0)).sum())
        xp = Vt[:r].T @ ((U[:, :r].T @ d) / s[:r])
        if len(C) and np.abs(C @ xp - d).max() > FEAS_TOL * max(1.0, np.abs(d).max()):
            raise Infeasible("working-set equations are inconsistent")
        Z = Vt[r:].T
        if Z.shape[1]:
            Hz = Z.T @ H @ Z
            z = np.linalg.solve(Hz, -Z.T @ (H @ xp + g))
            x = xp + Z @ z
        else:
            x = xp
        lam = np.linalg.lstsq(C.T, -(H @ x + g), rcond=None)[0] if len(C) else np.zeros(0)
        viol = D @ x - f
        viol[work] = -np.inf
        worst = int(np.argmax(viol)) if len(viol) else -1
        if worst >= 0 and viol[worst] > tol:
            work.append(worst)
            continue
        mu = lam[len(Aeq):]
        if len(mu) and mu.min() < -tol:
            work.pop(int(np.argmin(mu)))
            continue
        mult = np.zeros(len(Aeq) + len(D))
        mult[:len(Aeq)] = lam[:len(Aeq)]
        for k, i in enumerate(work):
            mult[len(Aeq) + i] = mu[k]
        return x, mult, sorted(work), it + 1
    raise RuntimeError("active-set iteration did not terminate")


def update(m, st, x_des, u_des, rbd, mode, last=None):
    """WeightedWbc::update (WeightedWbc.cpp:20-84): returns the decision vector and a dict with the QP and its KKT data.
    A QP that is not solved returns `last` (lastQpSol_, WeightedWbc.cpp:68-81); here "not solved" = infeasible constraints - e.g. the
    no-contact-motion equalities of the two points of a foot that rotates (their required accelerations are then incompatible with a
    rigid body).  qpOASES' other failure (more than nWSR = 20 working-set recalculations) has no counterpart in this exact method."""
    p = formulate(m, st, x_des, u_des, rbd, mode)
    H = p["Aw"].T @ p["Aw"]
    g = -p["Aw"].T @ p["bw"]
    p.update(H=H, g=g)
    try:
        x, mult, work, iters = solve_qp(H, g, p["Aeq"], p["beq"], p["D"], p["f"])
    except Infeasible:
        p.update(status=1, x=None)
        return (np.zeros(p["n"]) if last is None else np.array(last, float)), p
    p.update(x=x, mult=mult, work=work, iterations=iters, status=0)
    return x, p


def consistent_measured_state(m, q, v, mode):
    """A generalised velocity near v with the stance contact points at rest (projection onto the null space of their Jacobian): what a
    measured state looks like when the planned contacts hold - and what makes the reference's contact-acceleration equalities solvable."""
    flags = mode_flags(mode)
    J = contact_jacobian(m, np.asarray(q, float))
    rows = [r for i in range(4) if flags[i] for r in (3 * i, 3 * i + 1, 3 * i + 2)]
    if not rows:
        return np.array(v, float)
    Js = J[rows]
    return np.array(v, float) - np.linalg.pinv(Js) @ (Js @ v)


def rbd_from(m, q, v):
    """[zyx, pos, joints, angular velocity (world), linear velocity, joint velocities] (CentroidalModelRbdConversions layout)."""
    return np.concatenate([q[3:6], q[0:3], q[6:], euler_rate_map(q[3:6]) @ v[3:6], v[0:3], v[6:]])
