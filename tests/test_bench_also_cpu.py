"""bench.py's "also" block (the other BASELINE configs as child processes): the bookkeeping around the children, exercised without a GPU.  Here every
child exits with "bench.py needs an MI355X", which is exactly the path that must not take the headline line down: each entry reports its error and wall time."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_also_specs_name_every_baseline_config_and_failures_are_contained():
    sys.path.insert(0, REPO)
    import bench
    names = [s[0] for s in bench.ALSO_SPECS]
    assert names == ["flame512", "infer256", "train64_f16x3", "train64_f16x1", "full_shard128_collective"]
    wl = {s[0]: s[1] for s in bench.ALSO_SPECS}
    assert wl["full_shard128_collective"] == ["--workload", "full", "--global-batch", "128", "--force-collective"]
    assert wl["train64_f16x1"][-1] == "f16x1" and wl["train64_f16x3"][-1] == "f16x3"
    import torch
    if torch.cuda.is_available():
        return                                                         # on a GPU box the children really run: covered by the bench line itself
    args = bench.parse_args([])
    out = bench.run_also(args)
    assert set(names) <= set(out) and "_note" in out
    for n in names:
        assert "error" in out[n] and "MI355X" in out[n]["error"] and out[n]["wall_s"] >= 0, out[n]


def test_child_command_line_is_one_a_builder_can_run_by_hand():
    """every child argv parses with bench.py's own parser and switches off what only the parent does (also, counter passes); every entry has a CPU baseline
    of its own (a bounded sample timed by the child) or names the line that holds the same CPU path"""
    sys.path.insert(0, REPO)
    import bench
    for name, argv, steps, warmup, cpu_faces in bench.ALSO_SPECS:
        a = bench.parse_args(argv + ["--steps", str(steps), "--warmup", str(warmup), "--traffic", "off", "--cpu-faces", str(cpu_faces), "--cpu-passes", "3", "--no-also"])
        assert a.no_also and a.traffic == "off" and a.cpu_faces == cpu_faces and a.cpu_passes == 3 and a.steps == steps and a.gpus == 1, name
        assert cpu_faces > 0 or name in bench.ALSO_CPU_SAME_AS, name
    assert {n for n, *_ in bench.ALSO_SPECS if _[-1] > 0} == {"flame512", "infer256", "train64_f16x3"}


def test_rocprofv3_per_launch_figure_is_quoted_only_for_the_same_kernel_sources(tmp_path, monkeypatch):
    """roofline.rocprofv3: bench.py quotes rocprofv3's own per-launch average of its dominant kernel from profiles/rocprofv3_kernel_avg_<workload>.json — and withholds it
    (with the reason) when that summary was taken on other kernel sources, for another kernel, or does not exist"""
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "kernel_sources_sha", lambda: "abc")
    r = bench.rocprofv3_avg("full", "conv_halo_kernel<5,0>[256x128,8w,halo]")
    assert r["avg_launch_us"] is None and "no profiles/" in r["why"]
    f = tmp_path / "profiles" / "rocprofv3_kernel_avg_full.json"
    f.write_text(json.dumps({"kernel_sources_sha": "OTHER", "command": "c", "summary_file": "s.txt", "kernels": {"conv_halo_kernel<5,0>": {"calls": 96, "avg_us": 2600.0}}}))
    r = bench.rocprofv3_avg("full", "conv_halo_kernel<5,0>[256x128,8w,halo]")
    assert r["avg_launch_us"] is None and "other kernel sources" in r["why"]
    f.write_text(json.dumps({"kernel_sources_sha": "abc", "command": "c", "summary_file": "s.txt", "kernels": {"conv_halo_kernel<5,0>": {"calls": 96, "avg_us": 2600.0}}}))
    r = bench.rocprofv3_avg("full", "conv_halo_kernel<5,0>[256x128,8w,halo]")
    assert r["avg_launch_us"] == 2600.0 and r["calls"] == 96 and r["file"] == "profiles/s.txt"
    assert bench.rocprofv3_avg("full", "raster_tile")["avg_launch_us"] is None


def test_committed_evidence_belongs_to_the_committed_kernel_sources():
    """the PMC traffic table and the rocprofv3 per-launch averages under profiles/ that bench.py reads were taken on the kernel sources in the tree"""
    sys.path.insert(0, REPO)
    import bench
    sha = bench.kernel_sources_sha()
    for name in ("pmc_traffic_full.json", "rocprofv3_kernel_avg_full.json", "rocprofv3_kernel_avg_train64.json"):
        j = json.load(open(os.path.join(REPO, "profiles", name)))
        assert j["kernel_sources_sha"] == sha, name
