"""Differential fuzz of the HIP step kernel against the oracle on many random scenes (tests/test_fuzz_scenes.random_scene:
one to four trees of spheres / capsules / boxes, free / hinge / slide joints, a floor, mostly a wall): every step's contact,
row and sweep counts and the final states, through the C-ABI, with the generic kernels in both forms (full-batch and
few-copies solver forms).  The GPU test suite runs sixteen such scenes; this runs hundreds and prints a summary for
profiles/.  Usage: parity_fuzz.py [n_scenes] [steps] [sensors | articulated]
sensors: every body also carries a site with one to three sensors -- rangefinder, touch, accelerometer, frame axes -- and
every step's sensordata is compared too.  articulated: random_articulated_scene instead -- trees of two to seven bodies
with joint limits, springs, damping, armature and motors, under random controls (beyond their clamp) redrawn every ten
steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import _capi, blob, mjcf
from oracle.oracle import OracleEnv
from tests.test_fuzz_scenes import random_articulated_scene, random_scene

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
with_sensors = len(sys.argv) > 3 and sys.argv[3] == "sensors"
articulated = len(sys.argv) > 3 and sys.argv[3] == "articulated"
sens_worst = 0.0
t0 = time.time()
worst, checked, mismatched, paths, capped = 0.0, 0, [], {}, 0
for few in ("0", "1"):
    os.environ["MJRL_FEW"] = few
    for seed in range(7000, 7000 + n_scenes):
        rng = np.random.default_rng(seed)
        if articulated:
            xml, nu = random_articulated_scene(rng)
            model = mjcf.compile_mjcf_string(xml, nconmax=32, njmax=160)
            crng = np.random.default_rng(seed + 1)
        else:
            model = mjcf.compile_mjcf_string(random_scene(rng, sensors=with_sensors), nconmax=24, njmax=120)
        packed = blob.pack(model)
        h = _capi.Handle(packed, 2, specialize=False)
        h.reset()
        ora = OracleEnv(packed)
        qvel = h.get_field("qvel")
        for j in range(0 if articulated else model.njnt):
            if model.jnt_type[j] == mjcf.JNT_FREE:
                qa, da = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
                ora.qvel[da:da + 2] = -2.0 * ora.qpos[qa:qa + 2]
                qvel[:, da:da + 2] = -2.0 * model.qpos0[qa:qa + 2]
        h.set_field("qvel", qvel)
        ok = True
        for step in range(steps):
            if articulated and step % 10 == 0 and model.nu:
                ctrl = crng.uniform(-1.3, 1.3, model.nu)
                ora.ctrl[:model.nu] = ctrl
                h.set_field("ctrl", np.tile(ctrl, (2, 1)))
            h.step_host(None, 1)
            ora.step()
            stats = h.get_field("solver_stats")
            checked += 1
            if not ((stats[:, 0] == ora.ncon).all() and (stats[:, 1] == ora.nefc).all() and (stats[:, 2] == ora.niter).all()):
                mismatched.append((few, seed, step, stats[0, :3].tolist(), [ora.ncon, ora.nefc, ora.niter]))
                ok = False
                break
            if with_sensors:
                sd = h.get_field("sensordata")
                err = float(np.abs(sd - ora.sensordata).max())
                sens_worst = max(sens_worst, err)
                if err > 1e-7:
                    mismatched.append((few, seed, step, ["sensordata", err], []))
                    ok = False
                    break
            key = (model.ntree, int(model.rowmap), min(ora.nefc // 17, 3))
            if ora.niter >= 100:
                capped += 1
            paths[key] = paths.get(key, 0) + 1
        if ok:
            q = h.get_field("qpos")
            err = float(np.abs(q - ora.qpos).max() / max(1.0, np.abs(ora.qpos).max()))
            worst = max(worst, err)
        h.close(); ora.close()
print(f"{2 * n_scenes} runs ({n_scenes} scenes x 2 kernel forms) x {steps} steps: {checked} steps compared, "
      f"{len(mismatched)} runs with a count mismatch, worst final |qpos - oracle| (relative) {worst:.2e}, {time.time() - t0:.0f} s")
print(f"steps whose solve ran to the 100-sweep cap: {capped}")
if with_sensors:
    print(f"sensors on: worst |sensordata - oracle| over every step {sens_worst:.2e}")
print("solver paths met (trees, lane map, rows // 17): " + ", ".join(f"{k}: {v}" for k, v in sorted(paths.items())))
for m in mismatched[:10]:
    print("MISMATCH few=%s seed=%d step=%d kernel %s oracle %s" % m)
sys.exit(1 if mismatched or worst > 1e-7 else 0)
