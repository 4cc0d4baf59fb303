"""Host logic of bench.py that needs no GPU: the counter profiles under profiles/ are quoted only
while they were measured on the kernel source in the tree, and what `roofline.limiter` says."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_kernel_source_stamp_is_gits_blob_hash():
    data = open(os.path.join(ROOT, "ilqr_amd", "csrc", "kernels.hpp"), "rb").read()
    assert bench.kernel_source_stamp() == hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def test_profile_of_another_kernel_source_is_not_quoted(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_stamp", lambda: "abc")
    (prof / "hbm_traffic.json").write_text(json.dumps({"backward": {"bytes_per_launch": 1.0}, "_source": {"kernels_hpp": "abc"}}))
    assert bench.stamped_profile("hbm_traffic.json")["backward"]["bytes_per_launch"] == 1.0
    (prof / "hbm_traffic.json").write_text(json.dumps({"backward": {"bytes_per_launch": 1.0}, "_source": {"kernels_hpp": "old"}}))
    assert bench.stamped_profile("hbm_traffic.json") is None          # stale: measured on other kernels
    (prof / "hbm_traffic.json").write_text(json.dumps({"backward": {"bytes_per_launch": 1.0}}))
    assert bench.stamped_profile("hbm_traffic.json") is None          # unstamped
    assert bench.stamped_profile("missing.json") is None


def test_committed_profiles_belong_to_the_kernels_in_the_tree():
    """The files the driver's bench line will quote: regenerate them (tools/round_profile.sh) after a kernel change."""
    for name in ("hbm_traffic.json", "kernel_limits.json"):
        assert bench.stamped_profile(name) is not None, name


def test_limiter_names_both_limits_when_both_apply():
    vec = {"limiter": "fp64 vector issue", "valu_busy": 0.66}
    mem = {"limiter": "HBM bandwidth", "valu_busy": 0.3}
    assert bench.limiter_of(None, 1e9, 0.3) is None
    assert bench.limiter_of(mem, 1.4e9, 0.28) == "HBM bandwidth"
    # 1.35 GB in 0.285 ms = 4.7 TB/s >= 55 % of the 8 TB/s peak
    assert bench.limiter_of(vec, 1.35e9, 0.285) == "fp64 vector issue and HBM bandwidth"
    assert bench.limiter_of(vec, 0.5e9, 0.285) == "fp64 vector issue"
    assert bench.limiter_of(vec, None, 0.285) == "fp64 vector issue"
