"""bench.py prints ONE JSON line with the keys the driver reads (task contract) plus `roofline` and `cpu_baseline`.
Rehearsed on the CPU with the oracle library standing in for the device (`--lib`): the numbers mean nothing here,
the shape of the line does."""
import json
import os
import subprocess
import sys

from conftest import ROOT, _build_oracle


def test_bench_line_has_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--lib", _build_oracle(), "--steps", "1",
                        "--warmup", "0", "--cpu-iters", "4", "--cpu-cores", "2", "--secondary", "c2,c2_x2,tiny", "--c2-frames", "30000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(out[k], t), k
    assert out["vs_baseline"] is None and out["n_gpus"] == 1 and out["steps"] == 1 and out["scaling"] == "n/a"
    assert out["unit"] == "frames/s" and out["data"] == "synthetic" and out["dtype"] == "f64" and "workload" in out["config"]
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and out["roofline"]["bound"] == "latency"
    assert "k_step_assign" in out["roofline"]["kernel"] and set(out["roofline_assign"]) >= {"bound", "achieved", "peak", "frac"}
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and out["cpu_baseline"]["kind"] == "port"
    assert out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["all_cores"]["cores"] == 2
    assert "solve" in out and "full_pipeline" in out
    # round 2: the dominant kernels are on the line, the product-default flux mode is what is timed, both modes are reported,
    # the exchange path is named, PMC bytes are labelled as committed (not measured in the run), the ingest leg ran
    stages = {e["stage"] for e in out["roofline_by_kernel"]}
    assert {"training steps", "assign (2 launches)", "k-means++ seeding", "flux matrix"} <= stages
    assert out["config"]["flux_mode"] == "deterministic" and set(out["flux_ms"]) == {"atomic", "deterministic"}
    assert out["config"]["exchange"].startswith("none") and "traffic_source" in out["roofline_assign"] and out["labels_to_host_ms"] >= 0
    assert set(out["ingest"]) == {"npz_files", "raw_files", "in_memory"} and out["ingest"]["npz_files"]["GBps"] > 0
    assert out["ingest"]["raw_files"]["direct_read_blocks"] == out["ingest"]["raw_files"]["blocks"]   # file -> slab by the library
    # round 5: the reference's host-side post-condition is on the line (labels to the host; the step including them), the ingest
    # leg verified what it staged, and the other single-GPU BASELINE configs follow under `secondary` with their own objects
    assert out["labels_to_host"]["bytes_dtrajs"] == 8 * out["config"]["frames_per_gpu"] and out["step_incl_host_labels"]["value"] > 0
    assert all("byte for byte" in v["verified"] for v in out["ingest"].values())
    sec = out["secondary"]
    assert list(sec) == ["c2", "c2_x2", "tiny"]
    for k in ("c2", "c2_x2"):
        assert sec[k]["roofline"]["bound"] == "hbm" and sec[k]["roofline"]["unit"] == "GB/s" and sec[k]["value"] > 0
        assert sec[k]["roofline"]["algorithmic_bytes_per_launch"] == 52 * sec[k]["config"]["frames"]
    assert sec["c2"]["cpu_baseline"]["labels_equal_gpu"] and "cpu_baseline" not in sec["c2_x2"]
    assert sec["tiny"]["unit"] == "frames/s" and "roofline" in sec["tiny"] and "cpu_baseline" in sec["tiny"] and "full_pipeline" in sec["tiny"]
