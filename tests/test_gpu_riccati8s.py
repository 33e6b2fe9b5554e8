"""The eight-wave Riccati sweep that keeps the value function in registers between two stages (kernels/riccati_mfma8s.h, opt-in with
BPMPC_RICCATI8_S=1) against the default eight-wave sweep (kernels/riccati_mfma8.h): the same operations on the same values in another
schedule, so dx, du, K and the summaries must agree BIT FOR BIT - on three gaits (single support, double support: a third block column,
event nodes) and with the horizon swept in chunks (the value function is handed from launch to launch)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios
itf = scenarios.h1_interface()
out = {}
for gait in ("trot", "stance", "standing_trot"):
    prob = scenarios.trot_problem(itf, batch=8, n_intervals=60, gait=gait)
    mpc = bp.BatchedSqpMpc(itf, 8, 80)
    mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    for st in ("linearize", "project", "riccati"):
        mpc.stage(st)
    mpc.synchronize()
    for q in ("dx", "du", "K", "summary"):
        out[gait + "_" + q] = mpc.read(q).copy()
    t, x, u, K, stats = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)      # the whole solve: chunked sweeps, line search
    out[gait + "_x"], out[gait + "_u"], out[gait + "_Kfull"] = x, u, K
np.savez(sys.argv[1], **out)
"""


@pytest.mark.gpu
def test_sweep_with_the_value_function_in_registers_is_bit_identical(tmp_path):
    res = {}
    for tag, flag in (("default", "0"), ("registers", "1")):
        path = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, BPMPC_RICCATI8_S=flag, PYTHONPATH=ROOT)
        subprocess.check_call([sys.executable, "-c", CHILD, path], env=env, cwd=ROOT)
        res[tag] = np.load(path)
    for key in res["default"].files:
        a, b = res["default"][key], res["registers"][key]
        assert not np.isnan(b).any(), key
        assert np.array_equal(a, b), (key, float(np.nanmax(np.abs(a - b))))
