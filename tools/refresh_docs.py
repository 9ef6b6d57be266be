"""Write the numbers of the last full bench run (gpurun_out/bench_full.log) and of profiles/ into DESIGN.md / README.md."""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
full = json.loads([l for l in open(os.path.join(ROOT, "gpurun_out", "bench_full.log")) if l.startswith("{")][-1])
tr = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
rms = [float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_kernel_stats.csv")))
       if "mjrl_step_kernel" in r["Name"]][0]
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
rf, cb = full["roofline"], full["cpu_baseline"]
s = re.sub(r"\| env-steps/s, 1 GPU, device-resident \(the `value`\) \|[^\n]*\n",
           f"| env-steps/s, 1 GPU, device-resident (the `value`) | **{full['value'] / 1e6:.2f} M** ({full['ms_per_step']:.3f} ms per step of 4096 copies) | `python bench.py` |\n", s)
s = re.sub(r"\| kernel average duration \|[^\n]*\n",
           f"| kernel average duration | {rf['kernel_ms']:.4f} ms (HIP events on the launch stream) vs {rms:.4f} ms (`rocprofv3 --stats`) | `profiles/r01_kernel_stats.csv` |\n", s)
s = re.sub(r"\| `roofline.achieved` \|[^\n]*\n",
           f"| `roofline.achieved` | 2524 B × 4096 ÷ {rf['kernel_ms']:.4f} ms = {rf['achieved']:.1f} GB/s of 8000 GB/s → `frac` {rf['frac']:.4f} | bench line |\n", s)
s = re.sub(r"(\| `roofline.traffic` \| )[0-9.]+ MB per launch = [0-9.]+ ×",
           lambda m: f"{m.group(1)}{tr['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch = {tr['ratio_to_algorithmic']:.2f} ×", s)
s = re.sub(r"\| CPU baseline, kind \"port\"[^\n]*\n",
           f"| CPU baseline, kind \"port\" (the oracle; mujoco is not on the box) | {cb['single_thread'] / 1e3:.1f} k env-steps/s on 1 core; {cb['value'] / 1e6:.2f} M on all {cb['cores']} host cores (8 s samples) | bench line `cpu_baseline` |\n", s)
open(p, "w").write(s)
p = os.path.join(ROOT, "README.md")
s = open(p).read()
s = re.sub(r"fp64\): [0-9.]+ M env-steps/s", f"fp64): {full['value'] / 1e6:.1f} M env-steps/s", s)
open(p, "w").write(s)
print(f"{full['value'] / 1e6:.2f} M env-steps/s, kernel {rf['kernel_ms']:.4f} ms (rocprof {rms:.4f}), traffic {tr['hbm_bytes_per_launch'] / 1e6:.1f} MB")
