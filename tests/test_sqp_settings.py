"""The sqp block of task.info that the reference hands to SqpMpc as sqp::Settings (bipedal_controllers/src/BipedalController.cpp:303-306,
ocs2_bipedal_robot/src/BipedalRobotInterface.cpp:99; task.info:66-83).  Keys that select the ARITHMETIC of the solver are either
implemented or refused: a drop-in must not run a different optimiser than the file asks for without saying so."""
import re

import pytest

import bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc


def _interface(robot, tmp_path, edit=None):
    files = sc.ROBOTS[robot]
    task = files["task"]
    if edit is not None:
        text = open(task).read()
        key, value = edit
        new, n = re.subn(r"(\n\s*%s\s+)\S+" % re.escape(key), r"\g<1>%s" % value, text, count=1)
        assert n == 1, key
        task = str(tmp_path / ("task_%s_%s.info" % (key, value)))
        open(task, "w").write(new)
    return bp.BipedalRobotInterface(task, files["urdf"], files["reference"])


@pytest.mark.parametrize("robot", ["h1", "hunter", "openloong", "g1"])
def test_the_shipped_task_files_are_accepted(robot, tmp_path):
    s = _interface(robot, tmp_path).sqpSettings()
    assert s["integratorType"] == "RK2" and s["projectStateInputEqualityConstraints"] is True and s["useFeedbackPolicy"] is True
    assert s["dt"] == 0.015 and s["sqpIteration"] == 1


@pytest.mark.parametrize("key,value", [("integratorType", "EULER"), ("integratorType", "RK4"), ("projectStateInputEqualityConstraints", "false")])
def test_what_is_not_implemented_is_refused(key, value, tmp_path):
    with pytest.raises(bp.BpmpcError) as e:
        _interface("h1", tmp_path, (key, value))
    assert e.value.status == -3                      # BPMPC_ERR_UNSUPPORTED (include/bpmpc.h)
    assert "sqp." + key in str(e.value) and value in str(e.value)


def test_feedforward_policy_is_a_setting_of_the_model(tmp_path):
    s = _interface("h1", tmp_path, ("useFeedbackPolicy", "false")).sqpSettings()
    assert s["useFeedbackPolicy"] is False and s["integratorType"] == "RK2"
