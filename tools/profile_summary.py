"""Turn the rocprofv3 output of tools/profile_run.sh (gpurun_out/prof_r01) into the committed summaries:
profiles/r01_kernel_stats.csv (the --stats table of the default bench run) and profiles/r01_hbm_traffic.json (HBM
bytes per launch of the step kernel from the FETCH_SIZE / WRITE_SIZE passes, corrected with the calibration copy as
/opt/skills/guides/MI355X_MICROARCH.md prescribes)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r01")
dst = os.path.join(ROOT, "profiles")
STEP = "mjrl_step_kernel"          # matches the generic kernel and the specialised one (mjrl_step_kernel_spec)
ENVS, ALGO_BYTES = 4096, 2524      # bench.py defaults: env copies per launch, algorithmic bytes per env-step


def counter(run, name, kernel):
    vals = []
    for path in glob.glob(os.path.join(src, run, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(path)):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {name} samples for {kernel} under {run}")
    return sum(vals) / len(vals), len(vals)


stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "r01_kernel_stats.csv"))
    name = [r["Name"] for r in csv.DictReader(open(stats[0])) if STEP in r["Name"]][0]
else:
    name = STEP
fetch_kb, n = counter("pmc_FETCH_SIZE", "FETCH_SIZE", STEP)
write_kb, _ = counter("pmc_WRITE_SIZE", "WRITE_SIZE", STEP)
cal_fetch, _ = counter("calib_FETCH_SIZE", "FETCH_SIZE", "copy")
cal_write, _ = counter("calib_WRITE_SIZE", "WRITE_SIZE", "copy")
CAL_BYTES = 512 << 20
fc, wc = CAL_BYTES / (cal_fetch * 1024), CAL_BYTES / (cal_write * 1024)
hbm = fetch_kb * 1024 * fc + write_kb * 1024 * wc
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --no-double-buffer --steps 256   [tools/profile_run.sh, tools/profile_summary.py]",
    "kernel": name, "envs_per_launch": ENVS, "launches_sampled": n,
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "calibration": {"bytes_read": CAL_BYTES, "bytes_written": CAL_BYTES, "FETCH_SIZE_KB": cal_fetch, "WRITE_SIZE_KB": cal_write,
                    "note": "tools/calib_copy.hip: 8 B/lane coalesced copy of 512 MiB each way; FETCH_SIZE reports half of the bytes "
                            "read (as MI355X_MICROARCH.md documents), WRITE_SIZE is exact",
                    "fetch_correction": fc, "write_correction": wc},
    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": ENVS * ALGO_BYTES,
    "ratio_to_algorithmic": hbm / (ENVS * ALGO_BYTES),
}
json.dump(out, open(os.path.join(dst, "r01_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
