"""The C-ABI library loads and exports every symbol include/bpmpc.h declares; without a GPU the solver refuses to exist."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "bpmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bpmpc_[a-z_]+)\s*\(", text)))


def test_exports_every_declared_symbol():
    import bipedal_control_amd as bp
    lib = bp.load_library()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libbpmpc.so does not export " + n
    assert b"gfx950" in lib.bpmpc_version()


def test_product_library_has_no_oracle_or_emulation():
    """The shipped library must not link the checker or the CPU emulation of the kernels."""
    import subprocess
    import bipedal_control_amd.api as api
    out = subprocess.run(["nm", "-D", "--defined-only", api.library_path()], capture_output=True, text=True).stdout
    assert "oracle_" not in out and "emu_" not in out
    deps = subprocess.run(["ldd", api.library_path()], capture_output=True, text=True).stdout
    assert "liboracle" not in deps and "hostemu" not in deps and "libamdhip64" in deps


def test_solver_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import bipedal_control_amd as bp
    from bipedal_control_amd import scenarios
    itf = scenarios.h1_interface()
    with pytest.raises(bp.BpmpcError) as ei:
        bp.BatchedSqpMpc(itf, 2, 16)
    assert ei.value.status == -4 and "no CPU path" in str(ei.value)


def test_argument_validation():
    import bipedal_control_amd as bp
    lib = bp.load_library()
    assert lib.bpmpc_model_create(None, None, None, None) == -1
    out = C.c_void_p()
    assert lib.bpmpc_model_create(b"/nope.urdf", b"/nope.info", b"/nope.info", C.byref(out)) == -2 and not out.value
    assert lib.bpmpc_solver_run(None) == -1 and lib.bpmpc_solver_create(None, None, None) == -1
