"""Unitree G1 (BASELINE.json configs[3]) on the CPU tier: both ingests agree on the authored assets, the reduced URDF equals the
reference's g1.urdf (when the reference tree is present), hand-derived known answers, and the oracle's dynamics on this model pass
the same invariants / finite-difference checks that pin it for H1.  G1 exercises what H1 and OpenLoong do not: joint frames with a
fixed pitch (hip pitch -0.34907, knee +0.5096, ankle -0.16053), a hip order pitch-roll-yaw, massless contact frames.
The configuration is authored here (tools/make_assets.py): self-defined, not reference parity."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from oracle import ingest, reference_py as rp
from tests import oracle_bridge as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = os.path.join(ROOT, "assets", "g1")
REF_URDF = "/root/reference/bipedal_robot_example/unitree_g1/g1_description/g1.urdf"
LEG = ["hip_pitch", "hip_roll", "hip_yaw", "knee", "ankle_pitch", "ankle_roll"]


@pytest.fixture(scope="module")
def both():
    import bipedal_control_amd as bp
    itf = bp.BipedalRobotInterface(os.path.join(A, "task.info"), os.path.join(A, "g1_mpc.urdf"), os.path.join(A, "reference.info"))
    return itf, ob.model("g1"), ob.oracle("g1")


def test_g1_dimensions_order_and_known_answers(both):
    itf, m, _ = both
    assert (itf.stateDim, itf.inputDim, itf.numThreeDofContacts, itf.actuatedDofNum) == (24, 24, 4, 12)
    names = ["%s_%s_joint" % (s, j) for s in ("left", "right") for j in LEG]
    assert itf.jointNames() == m["joint_names"] == names
    # sum of the <mass> entries of g1.urdf (every link, welded or not); sole frames are massless
    masses = [float(l.find("inertial/mass").get("value")) for l in ET.parse(os.path.join(A, "g1_mpc.urdf")).getroot().iter("link") if l.find("inertial") is not None]
    assert abs(itf.robotMass() - sum(masses)) < 1e-12 and abs(itf.get("body_mass").sum() - sum(masses)) < 1e-12 and 30.0 < sum(masses) < 36.0
    assert list(itf.get("joint_parent")) == [0, 1, 2, 3, 4, 5, 0, 7, 8, 9, 10, 11]
    ax = itf.get("joint_axis").reshape(12, 3)
    assert np.array_equal(ax[:6], [[0, 1, 0], [1, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 0], [1, 0, 0]]) and np.array_equal(ax[6:], ax[:6])
    assert list(itf.get("contact_body")) == [6, 6, 12, 12]
    assert np.array_equal(itf.get("contact_offset").reshape(4, 3), [[0.13, 0, -0.03], [-0.06, 0, -0.03]] * 2)
    # fixed pitch of the hip-pitch joint frame: Rfix = Ry(-0.34907)
    Rf = itf.get("joint_rotation").reshape(12, 3, 3)
    c, s = np.cos(-0.34907), np.sin(-0.34907)
    assert np.allclose(Rf[0], [[c, 0, s], [0, 1, 0], [-s, 0, c]], atol=1e-16) and np.allclose(Rf[1], np.eye(3), atol=0)
    # the three fixed pitch offsets of a leg cancel: at zero joint angles (and in the default pose, whose pitch angles cancel too)
    # the sole is flat - both sole points of a foot at the same height
    for q in (np.zeros(12), m["default_joint_state"]):
        p = ingest.contact_positions(m, np.r_[np.zeros(6), q])
        assert abs(p[0, 2] - p[1, 2]) < 2e-6 and abs(p[2, 2] - p[3, 2]) < 2e-6 and abs(p[0, 2] - p[2, 2]) < 1e-15
    # comHeight = base height with the soles on the ground in the default pose (two decimals)
    p = ingest.contact_positions(m, np.r_[np.zeros(6), m["default_joint_state"]])
    assert abs(-p[:, 2].min() - m["com_height"]) < 5e-3
    # joint weights of Q follow the joint (OpenLoong roll 1000, yaw 800, pitch 20)
    Q, R = itf.costMatrices()
    assert list(np.diag(Q)[12:18]) == [20.0, 1000.0, 800.0, 20.0, 20.0, 800.0] and list(np.diag(Q)[18:]) == list(np.diag(Q)[12:18])
    # joint block J^T (2 I) J: two contact points per rigid foot give 6 rows of rank 5 per leg, so with six joints per leg the block
    # is positive SEMI-definite with one null direction per leg (rotation about the line through the two sole points) - as for the
    # reference's own 12-joint OpenLoong configuration; H1's five joints per leg make it definite
    Rj = R[12:, 12:]
    ev = np.linalg.eigvalsh(Rj)
    assert np.allclose(Rj, Rj.T, atol=1e-15) and ev.min() > -1e-15 and (ev > 1e-12).sum() == 10 and np.abs(Rj[:6, 6:]).max() == 0


def test_g1_product_matches_oracle_ingest(both):
    itf, m, _ = both
    pairs = [("body_mass", m["mass"]), ("body_com", m["com"]), ("body_inertia", m["inertia"]), ("joint_rotation", m["Rfix"]),
             ("joint_offset", m["pfix"]), ("joint_axis", m["axis"]), ("contact_offset", m["contact_off"]), ("Q", m["Q"]), ("R", m["R"]),
             ("initial_state", m["initial_state"]), ("default_joint_state", m["default_joint_state"])]
    for name, ref in pairs:
        got = itf.get(name)
        ref = np.asarray(ref, float).reshape(-1)
        assert got.shape == ref.shape, name
        assert np.abs(got - ref).max() <= 1e-15 * max(1.0, np.abs(ref).max()), name


@pytest.mark.skipif(not os.path.isfile(REF_URDF), reason="reference tree not present (GPU box)")
def test_g1_reduced_urdf_equals_reference_file(both):
    """Kinematic tree and inertias parsed from the reference's g1.urdf equal those of the reduced asset (the sole frames are the
    only addition and carry no mass)."""
    _, m, _ = both
    links, joints = ingest.parse_urdf(REF_URDF)
    names = set(m["joint_names"])
    # the reference file has no sole frames: attach the contact points to the feet for the comparison of everything else
    tree = ingest.build_tree(links, joints, names, ["left_ankle_roll_link", "left_ankle_roll_link", "right_ankle_roll_link", "right_ankle_roll_link"])
    for k in ("parent", "Rfix", "pfix", "axis", "mass", "com", "inertia", "contact_body"):
        assert np.array_equal(np.asarray(tree[k]), np.asarray(m[k])), k


def test_g1_oracle_invariants_and_finite_differences(both):
    _, m, om = both
    mass = m["robot_mass"]
    x0 = m["initial_state"]
    A, com = om.cmm(x0[6:])
    assert np.allclose(A[:3, :3], mass * np.eye(3), atol=1e-12) and np.abs(A[3:, :3]).max() < 1e-12
    f, Ax, Bu = om.flow_map(x0, rp.weight_compensating_input(m, 3), lin=True)
    assert np.abs(f[:3]).max() < 1e-13 and np.abs(f[6:]).max() < 1e-13
    rng = np.random.default_rng(11)
    x = x0 + 0.2 * rng.standard_normal(24)
    u = rp.weight_compensating_input(m, 3) + rng.standard_normal(24) * np.r_[np.full(12, 20.0), np.full(12, 0.5)]
    # independent numpy kinematics: contact positions, centre of mass, momentum consistency A qdot = m hbar
    pos, vel = om.ee_kinematics(x, u)
    assert np.abs(pos - ingest.contact_positions(m, x[6:])).max() < 1e-14
    R, o = ingest.fk(m, x[6:])
    c = sum(m["mass"][b] * (o[b] + R[b] @ m["com"][b]) for b in range(13)) / m["mass"].sum()
    assert np.abs(om.cmm(x[6:])[1] - c).max() < 1e-14
    f, Ax, Bu = om.flow_map(x, u, lin=True)
    assert np.abs(om.cmm(x[6:])[0] @ f[6:] - mass * x[:6]).max() < 1e-11
    pos, vel, dpdx, dvdx, dvdu = om.ee_kinematics(x, u, lin=True)
    eps = 1e-6
    for i in range(24):
        d = np.zeros(24); d[i] = eps
        assert np.abs((om.flow_map(x + d, u) - om.flow_map(x - d, u)) / (2 * eps) - Ax[:, i]).max() < 1e-6 * max(1, np.abs(Ax[:, i]).max())
        assert np.abs((om.flow_map(x, u + d) - om.flow_map(x, u - d)) / (2 * eps) - Bu[:, i]).max() < 1e-6
        pp, vp = om.ee_kinematics(x + d, u); pm, vm = om.ee_kinematics(x - d, u)
        assert np.abs(((vp - vm) / (2 * eps)).ravel() - dvdx[:, i]).max() < 1e-6 * max(1, np.abs(dvdx[:, i]).max())
        assert np.abs(((pp - pm) / (2 * eps)).ravel() - dpdx[:, i]).max() < 1e-7


def test_g1_walk_solve_converges_on_cpu(both):
    """The oracle's SQP on the G1 walk (standing_trot) brings the constraint violation down: the authored configuration is a
    well-posed problem (known-answer free sanity check of the self-defined config)."""
    from bipedal_control_amd import scenarios as sc
    itf = sc.interface("g1")
    prob = sc.trot_problem(itf, batch=1, n_intervals=20, gait="standing_trot")
    xo, uo, _, st = ob.oracle_solve_like(prob, 0, iterations=4, robot="g1")
    viol = [np.sqrt(r[1] + r[2]) for r in st if r[10] > 0]
    assert all(r[3] > 0 for r in st if r[10] > 0) and viol[-1] < 0.05 * viol[0]
    assert 0.5 < xo[-1, 8] < 0.9 and np.abs(xo[:, 9:12]).max() < 0.5
