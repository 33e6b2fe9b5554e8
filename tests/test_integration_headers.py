"""The OCS2 adaptor shipped as files (integration/HipSqpSolver.h, HipSqpMpc.h) meets a compiler: syntax check against
integration/mock_ocs2 (stand-ins for the OCS2 / Eigen declarations it touches - pins nothing about OCS2) and against the real
include/bpmpc.h (so the adaptor's calls match the C ABI's signatures)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adaptor_headers_compile_against_mock_and_c_abi():
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "integration"),
           "-I", os.path.join(ROOT, "integration", "mock_ocs2"), os.path.join(ROOT, "integration", "syntax_check.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_adaptor_overrides_every_recalled_virtual():
    """Every SolverBase virtual SURVEY.md section 8(b) lists appears as an override in the adaptor."""
    text = open(os.path.join(ROOT, "integration", "HipSqpSolver.h")).read()
    for name in ("reset", "getFinalTime", "getPrimalSolution", "getSolutionMetrics", "getNumIterations", "getOptimalControlProblem", "getPerformanceIndeces",
                 "getIterationsLog", "getValueFunction", "getHamiltonian", "getStateInputEqualityConstraintLagrangian", "getIntermediateDualSolution", "runImpl"):
        assert any(name in line and "override" in line for line in text.splitlines()), name
    assert text.count("void runImpl(") == 3      # plain, external controller, primal solution
