"""External parity hook: tools/ocs2_dump_primal.cpp dumps the PrimalSolution of the reference's own SqpMpc
where OCS2 exists; tools/compare_ocs2_dump.py diffs it against the oracle (and the HIP path on a GPU box).

* `BPMPC_OCS2_DUMP=<file>` set: the real comparison, tolerances 1e-6 abs on x, 1e-4 rel on u - this is the test that would lift the
  oracle's "parity unpinned" status.  Not available in this container (the reference cannot be built here), so it is skipped.
* Always: the tool chain itself on a dump written by the oracle (format round trip, node times, both example problems), and that a
  perturbed dump is rejected - so the day a real dump arrives the tooling is known to work."""
import os

import numpy as np
import pytest

from tools import compare_ocs2_dump as cd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_dump(tmp_path, robot, gait, intervals, second=0.0):
    """What ocs2_dump_primal would write if the reference's solver were the oracle: file of the first solve (and of the second)."""
    from oracle import reference_py as rp
    from tests import oracle_bridge as ob
    _, probs = cd.problems_of(dict(gait=gait, intervals=intervals, solve=1 if second else 0, t0=second), robot)
    m, om = ob.model(robot), ob.oracle(robot)
    sq = m["sqp"]
    prev, paths = None, []
    for i, prob in enumerate(probs):
        nodes = ob.oracle_nodes(prob, 0, robot=robot)
        x0 = prob["x0"][0]
        xi, ui = rp.cold_start(m, nodes, x0) if prev is None else rp.warm_start_from_previous(m, nodes, x0, *prev)
        xo, uo, Ko, _ = om.solve(nodes, x0, xi, ui, iterations=1, g_max=sq["g_max"], g_min=sq["g_min"], delta_tol=sq["deltaTol"])
        prev = (nodes, xo, uo, Ko)
        path = str(tmp_path / ("dump_%s_%s.csv%s" % (robot, gait, ".2" if i else "")))
        cd.write_dump(path, np.asarray(nodes["times"], float), xo, uo, nodes["kind"], intervals, gait, t0=prob["t0"], solve=i)
        paths.append(path)
    return paths


@pytest.mark.parametrize("robot,gait,intervals,second", [("h1", "stance", 20, 0.0), ("h1", "trot", 30, 0.0), ("h1", "trot", 30, 0.02),
                                                          ("hunter", "trot", 30, 0.02)])
def test_dump_tool_chain_round_trip(tmp_path, robot, gait, intervals, second):
    """The cases a maintainer is asked to dump: stance (configs[0]), trot, a warm-started second solve, Hunter (positionErrorGain 20)."""
    for i, path in enumerate(_oracle_dump(tmp_path, robot, gait, intervals, second)):
        d = cd.read_dump(path)
        assert d["nx"] == 22 and d["nu"] == 22 and d["gait"] == gait and d["solve"] == i and len(d["t"]) >= intervals + 1
        rep = cd.compare(d, robot, gpu=False)
        assert rep["oracle"]["ok"] and rep["oracle"]["max_abs_x"] == 0.0 and rep["oracle"]["max_rel_u"] == 0.0, rep
        if gait != "stance":
            # PrimalSolution's arrangement: pre-event nodes repeat the previous input (they are NOT the zeros the solver keeps there)
            ev = [k for k in range(1, len(d["t"]) - 1) if d["t"][k] == d["t"][k + 1]]
            assert ev and all(np.array_equal(d["u"][k], d["u"][k - 1]) and np.abs(d["u"][k]).max() > 0 for k in ev)
        # a dump that differs beyond the tolerance is rejected: states, forces and joint velocities each on their own scale
        d["x"][3, 8] += 5e-6
        assert not cd.compare(d, robot, gpu=False)["oracle"]["ok"]
        d["x"][3, 8] -= 5e-6
        d["u"][2, 15] += 2e-4 * max(1.0, np.abs(d["u"][:, 12:]).max())      # a joint velocity: invisible on the scale of the forces
        assert not cd.compare(d, robot, gpu=False)["oracle"]["ok"]


_DRIVER = r"""
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ocs2_dump_target.h"
// argv: kind(0 cmd, 1 goal) nj com_height rot_vel disp_vel t_now T c0 c1 c2 c3 then nj default joints then 12 + nj state entries (hex floats)
int main(int argc, char** argv) {
  int a = 1;
  const int kind = std::atoi(argv[a++]), nj = std::atoi(argv[a++]);
  auto next = [&]() { return std::strtod(argv[a++], nullptr); };
  const double com = next(), rv = next(), dv = next(), t = next(), T = next();
  double c[4]; for (double& v : c) v = next();
  std::vector<double> dj(nj), x(12 + nj), xs(2 * (12 + nj));
  for (double& v : dj) v = next();
  for (double& v : x) v = next();
  if (a != argc) return 3;
  const bpmpc_dump::TargetSettings s{nj, com, dj.data(), rv, dv};
  double times[2];
  if (kind == 0) bpmpc_dump::cmd_vel_to_targets(s, c, t, x.data(), T, times, xs.data());
  else bpmpc_dump::goal_to_targets(s, c, t, x.data(), times, xs.data());
  std::printf("%a %a", times[0], times[1]);
  for (double v : xs) std::printf(" %a", v);
  std::printf("\n");
  return 0;
}
"""


@pytest.mark.parametrize("robot", ["h1", "hunter", "openloong"])
def test_dump_target_header_equals_the_library_bit_for_bit(tmp_path, robot):
    """tools/ocs2_dump_target.h (what the dump program hands to the reference's solver) against bpmpc_cmd_vel_to_targets /
    bpmpc_goal_to_targets (what the comparer hands to the oracle and the engine): identical bits, for rotated bases, all four command
    components, negative times - the tool and the comparer cannot drift apart again (the hand-written target of the dump
    program lacked the momentum reference and the z / pitch / roll of the first point)."""
    import subprocess
    from bipedal_control_amd import scenarios as sc
    src, exe = tmp_path / "driver.cpp", tmp_path / "driver"
    src.write_text(_DRIVER)
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tools"), str(src), "-o", str(exe)], check=True)
    itf = sc.interface(robot)
    nj, nx = itf.actuatedDofNum, itf.stateDim
    com = float(itf.get("com_height")[0])
    dj = itf.get("default_joint_state")
    # targetRotationVelocity / targetDisplacementVelocity of reference.info, read by the oracle's own INFO parser
    from tests import oracle_bridge as ob
    m = ob.model(robot)
    rv, dv = float(m["target_rotation_velocity"]), float(m["target_displacement_velocity"])
    rng = np.random.default_rng(7)
    for trial in range(12):
        x = itf.getInitialState() + rng.uniform(-0.3, 0.3, nx)
        x[9:12] = rng.uniform(-1.2, 1.2, 3)                               # yaw, pitch, roll well away from zero
        cmd = rng.uniform(-0.6, 0.6, 4)
        t, T = float(rng.uniform(-1.0, 3.0)), float(rng.uniform(0.3, 2.5))
        for kind in (0, 1):
            args = [str(kind), str(nj), com.hex(), rv.hex(), dv.hex(), t.hex(), T.hex()] + [float(v).hex() for v in cmd] + [float(v).hex() for v in dj] + [float(v).hex() for v in x]
            out = subprocess.run([str(exe)] + args, check=True, capture_output=True, text=True).stdout.split()
            got = np.array([float.fromhex(v) for v in out])
            tt = itf.cmdVelToTargetTrajectories(cmd, t, x, T) if kind == 0 else itf.goalToTargetTrajectories(cmd, t, x)
            want = np.concatenate([np.asarray(tt.timeTrajectory, float), np.asarray(tt.stateTrajectory, float).ravel()])
            assert got.shape == want.shape and np.array_equal(got, want), (robot, trial, kind, np.abs(got - want).max())
            if kind == 0:      # the entries round 2's tool forgot
                assert np.abs(got[2:5]).max() > 0 and np.array_equal(got[2:5], got[2 + nx:5 + nx]) and got[2 + 8] == com and got[2 + 10] == 0.0 and got[2 + 11] == 0.0


def test_dump_program_builds_its_target_through_the_header():
    src = open(os.path.join(ROOT, "tools", "ocs2_dump_primal.cpp")).read()
    assert '#include "ocs2_dump_target.h"' in src and "bpmpc_dump::cmd_vel_to_targets" in src and "bpmpc_dump::target_pose_to_targets" in src
    assert "std::cos(yaw)" not in src      # no second, hand-written copy of the arithmetic


def test_dump_program_cites_existing_reference_interfaces():
    """The C++ hook only uses interfaces that exist in the reference tree (checked when the tree is present)."""
    src = open(os.path.join(ROOT, "tools", "ocs2_dump_primal.cpp")).read()
    ref = "/root/reference/ocs2_bipedal_robot/include/ocs2_bipedal_robot/BipedalRobotInterface.h"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    hdr = open(ref).read()
    for name in ("mpcSettings", "sqpSettings", "getOptimalControlProblem", "getInitializer", "getReferenceManagerPtr", "getInitialState",
                 "getCentroidalModelInfo", "getSwitchedModelReferenceManagerPtr"):
        assert name in src and name in hdr, name
    assert "loadModeSequenceTemplate" in open("/root/reference/ocs2_bipedal_robot/include/ocs2_bipedal_robot/gait/ModeSequenceTemplate.h").read()
    assert "insertModeSequenceTemplate" in open("/root/reference/ocs2_bipedal_robot/include/ocs2_bipedal_robot/gait/GaitSchedule.h").read()


@pytest.mark.skipif(not os.environ.get("BPMPC_OCS2_DUMP"), reason="no dump of the reference's own solver available (set BPMPC_OCS2_DUMP)")
def test_reference_dump_matches_oracle():
    d = cd.read_dump(os.environ["BPMPC_OCS2_DUMP"])
    rep = cd.compare(d, os.environ.get("BPMPC_OCS2_DUMP_ROBOT", "h1"), gpu=False)
    assert rep["oracle"]["ok"], rep


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("BPMPC_OCS2_DUMP"), reason="no dump of the reference's own solver available (set BPMPC_OCS2_DUMP)")
def test_reference_dump_matches_hip_path():
    d = cd.read_dump(os.environ["BPMPC_OCS2_DUMP"])
    rep = cd.compare(d, os.environ.get("BPMPC_OCS2_DUMP_ROBOT", "h1"), gpu=True)
    assert rep["hip"]["ok"], rep
