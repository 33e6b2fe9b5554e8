"""Measurement hygiene of bench.py: the counter traffic in the bench line comes from a committed profile of an
EARLIER run; it may only be reported while the kernels are still the ones that were profiled."""
import importlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_csrc_hash_follows_the_kernel_sources(tmp_path):
    from bipedal_control_amd.build import CSRC, csrc_hash
    copy = tmp_path / "csrc"
    shutil.copytree(CSRC, copy, ignore=shutil.ignore_patterns("build", "*.o"))
    assert csrc_hash(str(copy)) == csrc_hash()
    header = copy / "kernels" / "riccati_mfma8.h"
    header.write_text(header.read_text() + "\n// edited\n")
    assert csrc_hash(str(copy)) != csrc_hash()
    os.utime(header, None)                      # the time stamp alone changes nothing: contents are hashed
    (copy / "build").mkdir()
    (copy / "build" / "x.o").write_text("object files are not sources")
    h = csrc_hash(str(copy))
    header.write_text(header.read_text()[:-len("\n// edited\n")])
    assert csrc_hash(str(copy)) == csrc_hash() and h != csrc_hash()


def test_traffic_of_other_kernels_is_not_reported(tmp_path, monkeypatch):
    from bipedal_control_amd.build import csrc_hash
    bench = importlib.import_module("bench")
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "traffic_index.json").write_text(json.dumps({"h1:trot:trot:256:100": "t.json"}))
    kernels = {"k_linearize_fast<10, true, true>": {"hbm_bytes_per_launch": 654000000}, "k_riccati_fast8<10, false>": {"hbm_bytes_per_launch": 844500000}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "t.json").write_text(json.dumps({"csrc_hash": csrc_hash(), "all_kernels": kernels, "fused_hbm_bytes_per_step": 1}))
    ok = bench.committed_traffic("h1", "trot", False, 256, 100)
    assert ok["stale"] is False and ok["kernels"]["linearize_materialised"] == 654000000 and ok["kernels"]["riccati"] == 844500000
    (prof / "t.json").write_text(json.dumps({"csrc_hash": "0" * 40, "all_kernels": kernels, "fused_hbm_bytes_per_step": 1}))
    stale = bench.committed_traffic("h1", "trot", False, 256, 100)
    assert stale["stale"] is True and stale["kernels"].get("linearize_materialised") is None and stale["fused_hbm_bytes_per_step"] is None
    (prof / "t.json").write_text(json.dumps({"all_kernels": kernels}))           # a profile from before the hash existed: stale too
    assert bench.committed_traffic("h1", "trot", False, 256, 100)["stale"] is True
    assert bench.committed_traffic("h1", "trot", False, 512, 100) is None       # no committed pass for this workload


def test_default_bench_scenario_is_config_2_to_the_letter(monkeypatch):
    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    args = bench.parse_args()
    assert args.gait_start == 0.0 and args.batch == 256 and args.gpus == 1     # the trot template tiled from t = 0
