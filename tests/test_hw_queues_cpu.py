"""The package asks the HIP runtime for one hardware queue per stream: `import smirk_amd` (and bench.py) default $GPU_MAX_HW_QUEUES to 16 before the
runtime can have read it, never override a value the user exported, and the trace tools that found the queue sharing keep working on a synthetic trace."""
import csv
import gzip
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, env_extra=None, drop=("GPU_MAX_HW_QUEUES",)):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=REPO, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


def test_import_sets_the_queue_default_and_respects_an_exported_value():
    assert _run("import os, smirk_amd; print(os.environ['GPU_MAX_HW_QUEUES'], smirk_amd.HW_QUEUES_TOO_LATE)") == "16 False"
    assert _run("import os, smirk_amd; print(os.environ['GPU_MAX_HW_QUEUES'], smirk_amd.HW_QUEUES_TOO_LATE)", {"GPU_MAX_HW_QUEUES": "6"}) == "6 False"
    # torch imported first but the device untouched: the runtime has not read the variable yet, the default still lands in time
    assert _run("import torch, os, smirk_amd; print(os.environ['GPU_MAX_HW_QUEUES'], smirk_amd.HW_QUEUES_TOO_LATE)") == "16 False"


def test_bench_sets_the_queue_default_before_torch():
    src = open(os.path.join(REPO, "bench.py")).read()
    assert src.index('os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")') < src.index("    import torch")     # torch is imported inside main() / the workloads only
    assert _run("import sys, os; sys.argv = ['bench.py']; import bench; print(os.environ['GPU_MAX_HW_QUEUES'], 'torch' in sys.modules)") == "16 False"


def test_step_timeline_names_the_stream_a_backbone_waited_for(tmp_path, capsys):
    """synthetic trace: streams 1 (generator) and 2 (backbone) share queue 7, stream 3 has queue 8; the backbone's first kernel starts right after a generator kernel"""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import step_timeline
    rows, t = [], 0
    for step in range(3):
        base = step * 10_000_000
        rows.append(("raster_tile", 3, 8, base, base + 900_000))
        rows.append(("conv_halo_kernel<5; 0>", 1, 7, base + 1_000_000, base + 5_000_000))
        rows.append(("mbconv_fused_kernel<1; true; 2>", 2, 7, base + 5_010_000, base + 5_400_000))
        rows.append(("conv_halo_kernel<5; 0>", 1, 7, base + 5_400_000, base + 9_000_000))
    p = tmp_path / "t.csv.gz"
    with gzip.open(p, "wt") as f:
        w = csv.writer(f)
        w.writerow(["name", "stream", "queue", "start", "end", "grid", "wg"])
        for n, s, q, a, b in rows:
            w.writerow([n, s, q, a, b, 256, 256])
    step_timeline.main(str(p))
    out = capsys.readouterr().out
    assert "queue 7: stream 1" in out and "stream 2" in out.split("queue 7:")[1].splitlines()[0]
    line = next(ln for ln in out.splitlines() if ln.strip().startswith("stream   2"))
    assert "+10.0 us after conv_halo_kernel" in line and "(stream 1, same queue)" in line
    assert "= 10.000 ms" in out.splitlines()[0]
