"""Hunter (bipedal_robot_example/hunter): the reference's third complete MPC configuration - 10 leg joints like H1, two contact points
per foot, and the only configuration with model_settings.positionErrorGain != 0 (task.info:12: 20) - through both ingests, the host
lane emulation of the kernel bodies and the oracle.  CPU tier; the GPU parity of the same robot is tests/test_gpu_hunter.py."""
import os

import numpy as np
import pytest

from oracle import ingest
from tests import oracle_bridge as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = os.path.join(ROOT, "assets", "hunter")
REF = "/root/reference/bipedal_robot_example/hunter"


@pytest.fixture(scope="module")
def both():
    import bipedal_control_amd as bp
    itf = bp.BipedalRobotInterface(os.path.join(A, "task.info"), os.path.join(A, "hunter_mpc.urdf"), os.path.join(A, "reference.info"))
    return itf, ob.model("hunter")


def test_dimensions_joint_order_and_gain(both):
    itf, m = both
    assert (itf.stateDim, itf.inputDim, itf.numThreeDofContacts, itf.actuatedDofNum) == (22, 22, 4, 10)
    names = ["leg_%s%d_joint" % (s, i) for s in "lr" for i in range(1, 6)]          # task.info:18-30
    assert itf.jointNames() == m["joint_names"] == names
    assert m["position_error_gain"] == 20.0                                           # task.info:12
    # contact points 0, 1 on the left foot link, 2, 3 on the right one (task.info:35-41)
    assert list(itf.get("contact_body")) == [5, 5, 10, 10]
    x0 = itf.getInitialState()
    assert x0[8] > 0.5 and x0.shape == (22,)


def test_product_matches_oracle_ingest(both):
    itf, m = both
    pairs = [("body_mass", m["mass"]), ("body_com", m["com"]), ("body_inertia", m["inertia"]), ("joint_rotation", m["Rfix"]),
             ("joint_offset", m["pfix"]), ("joint_axis", m["axis"]), ("contact_offset", m["contact_off"]), ("Q", m["Q"]), ("R", m["R"]),
             ("initial_state", m["initial_state"]), ("default_joint_state", m["default_joint_state"])]
    for name, ref in pairs:
        got = itf.get(name)
        ref = np.asarray(ref, float).reshape(-1)
        assert got.shape == ref.shape, name
        assert np.abs(got - ref).max() <= 1e-15 * max(1.0, np.abs(ref).max()), name
    assert abs(itf.robotMass() - m["robot_mass"]) < 1e-12


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reduced_assets_equal_reference_files():
    urdf = os.path.join(REF, "legged_hunter_description/urdf/hunter.urdf")
    task = os.path.join(REF, "legged_hunter_config/config/task/task.info")
    refi = os.path.join(REF, "legged_hunter_config/config/command/reference.info")
    full = ingest.model_blob(ingest.build_model(urdf, task, refi))
    red = ingest.model_blob(ob.model("hunter"))
    assert np.array_equal(full, red)


def test_position_gain_enters_the_contact_rows():
    """With positionErrorGain = 20 the z rows of the zero-velocity constraint (stance) and the normal-velocity constraint (swing) carry
    gain * p_z resp. gain * (p_z - z_ref) and the linearisation gain * dp_z/dx (BipedalRobotInterface.cpp:350-359,
    BipedalRobotPreComputation.cpp:65-87, EndEffectorLinearConstraint.cpp:92-111): the oracle's rows of the Hunter model against that
    definition, mode by mode, and against the same model with the gain removed."""
    from oracle import oracle_py
    m, om = ob.model("hunter"), ob.oracle("hunter")
    gain = m["position_error_gain"]
    m0 = dict(m); m0["position_error_gain"] = 0.0
    om0 = oracle_py.OracleModel(ingest.model_blob(m0))
    rng = np.random.default_rng(7)
    x = np.asarray(m["initial_state"], float) + 0.1 * rng.standard_normal(22)
    u = rng.standard_normal(22) * np.r_[np.full(12, 20.0), np.full(10, 0.5)]
    zref, zdref = np.array([0.01, 0.02, 0.03, 0.04]), np.array([0.1, -0.2, 0.3, -0.4])
    pos, vel, dpdx, dvdx, dvdu = om.ee_kinematics(x, u, lin=True)
    assert np.abs(pos - ingest.contact_positions(m, x[6:])).max() < 1e-14
    for mode, flags in ((3, [1, 1, 1, 1]), (1, [1, 1, 0, 0]), (2, [0, 0, 1, 1]), (0, [0, 0, 0, 0])):
        o = om.node_lq(0, 0.015, x, u, x, x, mode, zref, zdref)
        o0 = om0.node_lq(0, 0.015, x, u, x, x, mode, zref, zdref)
        assert o["nc"] == o0["nc"] and np.array_equal(o["D"], o0["D"])     # the position term has no input dependence
        row = 0
        for i, st in enumerate(flags):
            if st:      # three zero-velocity rows
                for a in range(3):
                    g = gain if a == 2 else 0.0
                    assert abs(o["e"][row] - (vel[i, a] + g * pos[i, a])) < 1e-13
                    assert np.abs(o["C"][row] - (dvdx[3 * i + a] + g * dpdx[3 * i + a])).max() < 1e-12
                    row += 1
            else:       # three zero-force rows, one normal-velocity row
                row += 3
                assert abs(o["e"][row] - (vel[i, 2] - zdref[i] + gain * (pos[i, 2] - zref[i]))) < 1e-13
                assert np.abs(o["C"][row] - (dvdx[3 * i + 2] + gain * dpdx[3 * i + 2])).max() < 1e-12
                row += 1
        assert row == o["nc"]
        assert np.abs(o["C"] - o0["C"]).max() > 1.0 and np.abs(o["e"] - o0["e"]).max() > 1.0
