"""Build libbpmpc.so (HIP, gfx950) in-tree.  `python -m bipedal_control_amd.build [--force]`

One translation unit per kernel family (csrc/k_*.hip) plus the host code; they compile in parallel into csrc/build/*.o and are
re-compiled only when a source they depend on (conservatively: any header, or the unit itself) is newer than the object.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libbpmpc.so")
SOURCES = ["k_node.hip", "k_project.hip", "k_riccati.hip", "k_riccati_wave.hip", "k_ddp.hip", "solver.hip", "wbc.hip", "capi.cpp", "info_tree.cpp", "urdf_tree.cpp",
           "robot_model.cpp", "reference_gen.cpp", "device_model.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def csrc_hash(csrc=CSRC):
    """sha1 over the names and contents of every source under csrc/ (objects excluded): what the library is built from.  tools/summarize_pmc.py
    stores it beside the counter traffic of a profiled run; bench.py reports that traffic only while the hash still matches (a kernel
    change without a re-collected profile would otherwise carry stale bytes into the bench line)."""
    h = hashlib.sha1()
    for root, dirs, files in os.walk(csrc):           # (dirs pruned and ordered in place: the walk itself is deterministic)
        dirs[:] = sorted(d for d in dirs if d != "build")
        for f in sorted(files):
            if f.endswith((".h", ".hip", ".cpp")):
                path = os.path.join(root, f)
                h.update(os.path.relpath(path, csrc).encode())
                with open(path, "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()


def _newest_header():
    newest = os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "bpmpc.h"))
    for root, dirs, files in os.walk(CSRC):
        if os.path.basename(root) == "build":
            continue
        for f in files:
            if f.endswith(".h"):
                newest = max(newest, os.path.getmtime(os.path.join(root, f)))
    return newest


def _newest_source():
    return max([_newest_header()] + [os.path.getmtime(os.path.join(CSRC, s)) for s in SOURCES])


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("BPMPC_EXTRA_FLAGS", "").split()      # e.g. -DBPMPC_LINFAST_PROFILE for tools/*_phase_profile.py
    tag = hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8] if extra else "default"
    # the library is up to date only if it was linked from THIS tag's objects (objects are cached per flag set; the stamp names the last link)
    stamp = os.path.join(OBJ, "linked_tag")
    last_tag = open(stamp).read().strip() if os.path.exists(stamp) else None
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source() and last_tag == tag:
        return LIB
    objdir = os.path.join(OBJ, tag)
    os.makedirs(objdir, exist_ok=True)
    headers = _newest_header()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(headers, os.path.getmtime(path)):
            return obj
        cmd = [hipcc] + FLAGS + extra + ["-x", "hip", "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    for stale in os.listdir(objdir):            # objects of units that are no longer in SOURCES
        if stale.endswith(".o") and stale not in {os.path.splitext(x)[0] + ".o" for x in SOURCES}:
            os.remove(os.path.join(objdir, stale))
    with open(stamp, "w") as f:
        f.write(tag)
    return LIB


def build_probes(force=False, verbose=False):
    """Measurement helpers outside the product library: tools/probes/libwrite_roof.so, the write-only stream with the lineariser's store pattern
    that bench.py runs beside the timed region to calibrate `roofline.frac` (roofline.write_roof)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "probes", "write_roof.hip")
    lib = os.path.join(root, "tools", "probes", "libwrite_roof.so")
    if not os.path.exists(src):
        return None
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-DWRITE_ROOF_LIB", src, "-o", lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_probes(force="--force" in sys.argv, verbose=True))
