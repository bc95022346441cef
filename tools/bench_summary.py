"""one-screen summary of a bench.py JSON line:  python tools/bench_summary.py gpurun_out/x.json [n_kernels]"""
import json, sys
j = None
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
print(round(j["value"], 1), "faces/s", round(j["ms_per_step"], 2), "ms/step", "host enqueue (idle queue)", j.get("host_enqueue_ms_one_step_idle_queue", j.get("host_enqueue_ms_per_step")), "ms", j["config"].get("frames_per_gpu_per_step"))
r = j.get("roofline")
if r:
    print("dominant:", r["kernel"], "frac", round(r["frac"], 4), "achieved", round(r["achieved"], 1), r["unit"], "traffic", r.get("traffic"))
    for k, v in list(r["kernels"].items())[:n]:
        print(f"  {k:58s} {v['ms_per_pass']:8.3f} ms  x{v['launches']:<3d} {v.get('tflops', '')} {v.get('gbps', '')}")
    print("  kernel ms per pass", round(r["profiled_kernel_ms_per_pass"], 2))
if j.get("cpu_baseline"):
    print("cpu_baseline", j["cpu_baseline"].get("value"), j["cpu_baseline"].get("unit"), "cores", j["cpu_baseline"].get("cores"))
for k, v in (j.get("also") or {}).items():
    if isinstance(v, dict):
        print("also", k, v.get("error") or (round(v["value"], 1), "faces/s", round(v["ms_per_step"], 3), "ms x", v["steps"], "launches", v.get("launches_per_step"),
                                            "frac", round((v.get("roofline") or {}).get("frac") or 0, 4), (v.get("roofline") or {}).get("kernel"), "wall", v.get("wall_s")))
