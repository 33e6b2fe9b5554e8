"""Resource hygiene of the shipped code objects (no GPU needed: the metadata notes of libbpmpc.so are read with the LLVM tools of the ROCm image).
VERDICT r04 item 4: no product kernel may use scratch memory - a scratch reload waits for every memory request in flight, which is how 52 B in the
value-only kernel at nx = 24 cost 6 % of its line search.  The reference kernels (lane-emulation bodies, used for cross-checks, the DDP slice and the
WBC) are listed with what they are allowed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
# kernel-name prefix -> scratch bytes it may use (reference bodies written as phases over an LDS workspace: not on the measured path)
ALLOWED = {"k_linearize<": 64, "k_trial<": 64, "k_riccati<": 256, "k_project<": 64, "k_wbc<": 320, "k_ddp_cost<": 64,
           "k_ls_tail<12, false>": 64}      # nx = 24 on a tree that is not two serial legs: no robot of the reference


def _kernels():
    lib = os.path.join(ROOT, "bipedal_control_amd", "libbpmpc.so")
    bundler, readelf, objcopy = (os.path.join(LLVM, t) for t in ("clang-offload-bundler", "llvm-readelf", "llvm-objcopy"))
    if not (os.path.exists(lib) and all(os.path.exists(t) for t in (bundler, readelf, objcopy)) and shutil.which("c++filt")):
        pytest.skip("library or LLVM tools not available")
    import tempfile
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "copy.so")], check=True, capture_output=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)] + [len(data)]
        for n, (a, b) in enumerate(zip(starts[:-1], starts[1:])):
            part, co = os.path.join(tmp, "b%d.bin" % n), os.path.join(tmp, "b%d.co" % n)
            open(part, "wb").write(data[a:b])
            subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co], check=True, capture_output=True)
            notes = subprocess.run([readelf, "--notes", co], check=True, capture_output=True, text=True).stdout
            for k in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                get = lambda key: re.search(r"\." + key + r":\s+(\S+)", k).group(1)
                out.append((get("name"), int(get("private_segment_fixed_size")), int(get("vgpr_count")), int(get("group_segment_fixed_size"))))
    names = subprocess.run(["c++filt"], input="\n".join(k[0] for k in out), capture_output=True, text=True).stdout.split("\n")
    return [(n.replace("void bpmpc::", "").replace("bpmpc::", ""),) + k[1:] for n, k in zip(names, out)]


def test_no_product_kernel_uses_scratch_memory():
    kernels = _kernels()
    assert len(kernels) > 60          # every translation unit was found
    bad = []
    for name, scratch, vgpr, lds in kernels:
        limit = max([v for p, v in ALLOWED.items() if name.startswith(p)] or [0])
        if scratch > limit:
            bad.append((name, scratch, limit))
    assert not bad, bad
    # the kernels the bench line runs exist under the names the profiles carry
    for must in ("k_linearize_fast<10, true, true, false>", "k_linearize_fast<10, false, true, true>", "k_project_lu_s<10, 8, true>", "k_project_fast<10, true, false>", "k_riccati_fast8<10, false>",
                 "k_trial_fast<12, true>", "k_ls_tail<12, true>"):
        assert any(n.startswith(must) for n, *_ in kernels), must


def test_lds_of_every_kernel_fits_a_compute_unit():
    for name, scratch, vgpr, lds in _kernels():
        assert lds <= 160 * 1024, (name, lds)
