"""Build libbpmpc.so (HIP, gfx950) in-tree.  `python -m bipedal_control_amd.build`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbpmpc.so")
SOURCES = ["solver.hip", "wbc.hip", "capi.cpp", "info_tree.cpp", "urdf_tree.cpp", "robot_model.cpp", "reference_gen.cpp", "device_model.cpp"]


def _newest_source():
    newest = 0.0
    for root, _, files in os.walk(CSRC):
        for f in files:
            newest = max(newest, os.path.getmtime(os.path.join(root, f)))
    newest = max(newest, os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "bpmpc.h")))
    return newest


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-o", LIB]
    cmd += os.environ.get("BPMPC_EXTRA_FLAGS", "").split()      # e.g. -DBPMPC_PROJECT_PROFILE for tools/*_phase_profile.py
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
