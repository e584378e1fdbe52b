"""bench.py host-side helpers that decide what the JSON line may claim (CPU)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_traffic_is_trusted_only_for_the_profiled_build_or_a_documented_successor(tmp_path, monkeypatch):
    b = _bench()
    rec = {"lib_sha16": "aaaa", "source": "src", "workloads": {"lokr/sdxl/linear": {"lokr_kron3": {"bytes_per_launch": 123}},
                                                                "lokr/sdxl/conv": {"lokr_conv": {"bytes_per_pass": 9}}},
           "also_valid_for": {"bbbb": {"workloads": ["lokr/sdxl/linear"], "difference": "conv kernels only"}}}
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps(rec))
    monkeypatch.setattr(b, "PMC_FILE", str(f))
    monkeypatch.setattr(b, "lib_sha", lambda: "aaaa")
    fam, src = b.pmc_traffic("lokr_kron3", "lokr/sdxl/linear")
    assert fam == {"bytes_per_launch": 123} and src == "src"
    monkeypatch.setattr(b, "lib_sha", lambda: "bbbb")  # a later build whose difference is written down, for THAT workload only
    fam, src = b.pmc_traffic("lokr_kron3", "lokr/sdxl/linear")
    assert fam == {"bytes_per_launch": 123} and "conv kernels only" in src and "aaaa" in src and "bbbb" in src
    fam, src = b.pmc_traffic("lokr_conv", "lokr/sdxl/conv")
    assert fam is None and "collected on build aaaa" in src
    monkeypatch.setattr(b, "lib_sha", lambda: "cccc")  # an unknown build: nothing is claimed
    assert b.pmc_traffic("lokr_kron3", "lokr/sdxl/linear")[0] is None
    monkeypatch.setattr(b, "lib_sha", lambda: "aaaa")
    assert b.pmc_traffic("lokr_kron3", "lokr/sd15/linear")[0] is None  # no pass for that workload


def test_the_committed_traffic_file_names_the_library_it_was_collected_on():
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert len(rec["lib_sha16"]) == 16 and "lokr/sdxl/linear" in rec["workloads"]
    for sha, also in rec.get("also_valid_for", {}).items():
        assert len(sha) == 16 and also["difference"] and set(also["workloads"]) <= set(rec["workloads"])
