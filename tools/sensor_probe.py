"""One fuzz scene with sensors on the GPU against the oracle, step by step (tools/parity_fuzz.py found the readings apart)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import _capi, blob, mjcf
from oracle.oracle import OracleEnv
from tests.test_fuzz_scenes import random_scene
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7020
text = random_scene(np.random.default_rng(seed), sensors=True)
model = mjcf.compile_mjcf_string(text, nconmax=24, njmax=120)
packed = blob.pack(model)
print("sensor types", list(model.sensor_type), "cutoffs", list(model.sensor_cutoff), "site body", list(model.site_bodyid), "ntree", model.ntree)
for spec in (False, True):
    h = _capi.Handle(packed, 2, specialize=spec)
    h.reset()
    ora = OracleEnv(packed)
    qvel = h.get_field("qvel")
    for j in range(model.njnt):
        if model.jnt_type[j] == mjcf.JNT_FREE:
            qa, da = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
            ora.qvel[da:da + 2] = -2.0 * ora.qpos[qa:qa + 2]
            qvel[:, da:da + 2] = -2.0 * model.qpos0[qa:qa + 2]
    h.set_field("qvel", qvel)
    shown = 0
    for step in range(300):
        h.step_host(None, 1)
        ora.step()
        sd = h.get_field("sensordata")
        if np.abs(sd - ora.sensordata).max() > 1e-7 and shown < 3:
            print("spec" if spec else "generic", "step", step, "gpu", sd[0], "oracle", ora.sensordata, "ncon", ora.ncon, "stats", h.get_field("solver_stats")[0])
            shown += 1
    print("spec" if spec else "generic", "mismatching steps shown:", shown, "final qpos err", np.abs(h.get_field("qpos") - ora.qpos).max())
    h.close()

# the diagnostic kernel's LDS image after the forward pass, at the first step whose reading differs
h = _capi.Handle(packed, 1, specialize=False)
h.reset()
ora = OracleEnv(packed)
qvel = h.get_field("qvel")
for j in range(model.njnt):
    if model.jnt_type[j] == mjcf.JNT_FREE:
        qa, da = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
        ora.qvel[da:da + 2] = -2.0 * ora.qpos[qa:qa + 2]
        qvel[:, da:da + 2] = -2.0 * model.qpos0[qa:qa + 2]
h.set_field("qvel", qvel)
off = {k: h.lds_offset(k) for k in ("row", "sens", "con", "ints", "i_conadr", "i_cong1", "i_cong2", "xpos", "xquat")}
print("lds offsets", off)
for step in range(300):
    img = h.step_debug(None, 0, 1, stage=0)[0]
    ora.step()
    sd = h.get_field("sensordata")[0]
    if np.abs(sd - ora.sensordata).max() > 1e-7:
        ints = img[off["ints"]:].view(np.int32)
        print("diag step", step, "gpu", sd, "oracle", ora.sensordata, "lds sens", img[off["sens"]:off["sens"] + model.nsensordata])
        print("ints head", ints[:8], "conadr", ints[off["i_conadr"]:off["i_conadr"] + 2], "g1", ints[off["i_cong1"]:off["i_cong1"] + 2],
              "g2", ints[off["i_cong2"]:off["i_cong2"] + 2])
        print("rows (R, B, F, ARII)", img[off["row"]:off["row"] + 32].reshape(8, 4))
        print("con0", img[off["con"]:off["con"] + 15])
        print("oracle efc_force", getattr(ora, "efc_force", None))
        break
else:
    print("diag kernel: no mismatch in 300 steps")
h.close()
