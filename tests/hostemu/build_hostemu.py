"""TEST-ONLY build of the CPU lane-emulation of the kernel bodies (see hostemu.cpp)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libbpmpc_hostemu.so")


def build(force=False, optimised=False, outdir=None):
    """optimised: -O3 -march=native into a second library, built on the box that runs it (bench.py's cpu_baseline_analytic leg, which
    passes a scratch `outdir` so that a benchmark never writes next to the libraries of a test run in progress)."""
    csrc = os.path.join(ROOT, "bipedal_control_amd", "csrc")
    srcs = [os.path.join(HERE, "hostemu.cpp")] + [os.path.join(csrc, f) for f in ("info_tree.cpp", "urdf_tree.cpp", "robot_model.cpp", "device_model.cpp")]
    newest = max(os.path.getmtime(p) for p in srcs)
    for root, _, files in os.walk(os.path.join(csrc, "kernels")):
        newest = max([newest] + [os.path.getmtime(os.path.join(root, f)) for f in files])
    lib = LIB.replace(".so", "_native.so") if optimised else LIB
    if outdir:
        lib = os.path.join(outdir, "bpmpc_" + os.path.basename(lib))
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= newest:
        return lib
    flags = ["-O3", "-march=native"] if optimised else ["-O2"]
    subprocess.check_call(["g++"] + flags + ["-std=c++17", "-fPIC", "-shared", "-DBPMPC_HOST_EMULATION", "-o", lib] + srcs)
    return lib


if __name__ == "__main__":
    print(build(True))
