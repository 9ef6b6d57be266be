"""Turn the rocprofv3 output of tools/profile_run.sh (gpurun_out/prof_<round>_<level>) into the committed summaries:
profiles/<round>_kernel_stats_<level>.csv (the --stats table of the exact driver command), profiles/<round>_timed_window_<level>.json
(the step kernel's launches of the TIMED region, taken from the kernel trace: the last `steps` launches -- pre-roll and
warm-up launches come before them) and profiles/<round>_hbm_traffic_<level>.json (HBM bytes per launch from the FETCH_SIZE /
WRITE_SIZE passes, corrected with the calibration copy as /opt/skills/guides/MI355X_MICROARCH.md prescribes).
Usage: profile_summary.py [level] [steps]"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
level = sys.argv[1] if len(sys.argv) > 1 else "two_agent"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ROUND = os.environ.get("ROUND", "r04")
src = os.path.join(ROOT, "gpurun_out", f"prof_{ROUND}_{level}")
dst = os.environ.get("PROFILE_OUT", os.path.join(ROOT, "profiles"))
os.makedirs(dst, exist_ok=True)
STEP = "mjrl_step_kernel"          # matches the generic kernel and the specialised one (mjrl_step_kernel_spec)
COMMAND = f"python3 bench.py --gpus 1 --steps {steps} --warmup 5" + ("" if level == "two_agent" else f" --level {level}")

bench_line = None
for line in open(os.path.join(src, "stats.log")):
    if line.startswith("{"):
        bench_line = json.loads(line)
envs = bench_line["config"]["envs_per_gpu"]
lead = bench_line["config"]["preroll_steps"] + bench_line["warmup"]     # untimed launches in front of the timed ones
SPEC = "mjrl_step_kernel_spec"     # the headline's launches (the one generic launch per handle is the reset image's)
algo = bench_line["roofline"]["algorithmic_bytes_per_env_step"]


def newest(pattern):
    """gpurun merges every call's files into the same directories: earlier runs stay behind, the last one counts"""
    paths = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not paths:
        raise SystemExit(f"nothing matches {pattern}")
    return paths[-1]


def counter(run, name, kernel, last=None):
    vals = []
    for path in [newest(os.path.join(src, run, "*", "*counter_collection.csv"))]:
        rows = [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals += [float(r["Counter_Value"]) for r in rows]
    if not vals:
        raise SystemExit(f"no {name} samples for {kernel} under {run}")
    if last:
        # the timed launches by position: the default command goes on to measure the other configs after them
        vals = vals[lead:lead + last] if len(vals) >= lead + last else vals[-last:]
    return sum(vals) / len(vals), len(vals)


stats = [newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))]
shutil.copy(stats[0], os.path.join(dst, f"{ROUND}_kernel_stats_{level}.csv"))
row = [r for r in csv.DictReader(open(stats[0])) if SPEC in r["Name"]][0]
trace = newest(os.path.join(src, "stats", "*", "*kernel_trace.csv"))
launches = [r for r in csv.DictReader(open(trace)) if SPEC in r["Kernel_Name"]]
launches.sort(key=lambda r: int(r["Start_Timestamp"]))
all_dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in launches]
launches = launches[:lead + steps]          # the headline's: what follows belongs to the `configs` array of the line
dur = all_dur[:lead + steps]
timed = dur[-steps:]
window = {
    "command": f"rocprofv3 --kernel-trace --stats -- {COMMAND}   [tools/profile_run.sh, tools/profile_summary.py]",
    "kernel": row["Name"], "launches_total": len(dur), "all_launches_average_ns": sum(dur) / len(dur),
    "stats_table_average_ns": float(row["AverageNs"]),
    "timed_launches": steps, "timed_average_ns": sum(timed) / len(timed), "timed_min_ns": min(timed), "timed_max_ns": max(timed),
    "first_launch_of_the_timed_region_start_ns": int(launches[-steps]["Start_Timestamp"]),
    "timed_region_span_ns_per_step": (int(launches[-1]["End_Timestamp"]) - int(launches[-steps]["Start_Timestamp"])) / steps,
    "bench_line": {k: bench_line[k] for k in ("value", "ms_per_step", "steps", "warmup")},
    "bench_roofline": bench_line["roofline"],
    "roofline_from_this_profile": {"algorithmic_bytes_per_launch": envs * algo,
                                   "achieved_GBps": envs * algo / (sum(timed) / len(timed)),
                                   "frac_of_8000_GBps": envs * algo / (sum(timed) / len(timed)) / 8000.0},
    "note": "pre-roll (1024) and warm-up launches precede the timed ones in the trace; the first ~170 pre-roll launches run "
            "with every copy still airborne and are lighter, so the all-launch average sits slightly below the timed one",
}
json.dump(window, open(os.path.join(dst, f"{ROUND}_timed_window_{level}.json"), "w"), indent=1)
print(json.dumps(window, indent=1))

fetch_kb, n = counter("pmc_FETCH_SIZE", "FETCH_SIZE", SPEC, last=steps)
write_kb, _ = counter("pmc_WRITE_SIZE", "WRITE_SIZE", SPEC, last=steps)
cal_fetch, _ = counter("calib_FETCH_SIZE", "FETCH_SIZE", "copy")
cal_write, _ = counter("calib_WRITE_SIZE", "WRITE_SIZE", "copy")
CAL_BYTES = 512 << 20
fc, wc = CAL_BYTES / (cal_fetch * 1024), CAL_BYTES / (cal_write * 1024)
hbm = fetch_kb * 1024 * fc + write_kb * 1024 * wc
out = {
    "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- {COMMAND} --no-cpu-baseline   [tools/profile_run.sh, tools/profile_summary.py]",
    "kernel": row["Name"], "envs_per_launch": envs, "launches_sampled": n, "which": "the launches of the timed region",
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "calibration": {"bytes_read": CAL_BYTES, "bytes_written": CAL_BYTES, "FETCH_SIZE_KB": cal_fetch, "WRITE_SIZE_KB": cal_write,
                    "note": "tools/calib_copy.hip: 8 B/lane coalesced copy of 512 MiB each way; FETCH_SIZE reports half of the bytes "
                            "read (as MI355X_MICROARCH.md documents), WRITE_SIZE is exact",
                    "fetch_correction": fc, "write_correction": wc},
    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": envs * algo,
    "ratio_to_algorithmic": hbm / (envs * algo),
}
json.dump(out, open(os.path.join(dst, f"{ROUND}_hbm_traffic_{level}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
