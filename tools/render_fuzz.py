"""Differential fuzz of the ray kernel against the oracle's ray caster on random scenes (tests/test_fuzz_scenes.random_scene
with cameras: coloured spheres / capsules / boxes after they have fallen and come to rest against each other, a wall, a
camera on every body plus two fixed ones, a spot or directional light with or without shadows, sometimes a second light
on a body).  Per image: the share of pixels that differ by more than one level from the oracle's (fp64, every geom tested
against every ray; the kernel: fp32, candidates culled per block), and -- exact -- the same images with the kernel's tight
culls switched off (MJRL_RENDER_LOOSE=1) byte for byte.  Usage: render_fuzz.py [n_scenes] [first_seed]"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import _capi, blob, mjcf
from oracle.oracle import OracleEnv
from tests.test_fuzz_scenes import random_scene, unexplained

variant = os.environ.get("RENDER_FUZZ_VARIANT", "")
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
t0 = time.time()
images, worst_share, worst_any, cull_diffs, bad, dumps = 0, 0.0, 0.0, 0, [], {}
n_unexplained, worst_out, skipped = 0, 0, 0
for seed in range(first, first + n_scenes):
    xml = random_scene(np.random.default_rng(seed), cameras=True)
    if "nobodylight" in variant:
        xml = re.sub(r'<light pos="0 0 0.3"[^>]*>', "", xml)
    if "onlybodylight" in variant:
        xml = re.sub(r'<light pos="(?!0 0 0.3)[^>]*>', "", xml)
    if "noshadow" in variant:
        xml = xml.replace(' castshadow="false"', "").replace("<light ", '<light castshadow="false" ')
    model = mjcf.compile_mjcf_string(xml, nconmax=24, njmax=120)
    packed = blob.pack(model)
    h = _capi.Handle(packed, 2, specialize=False)
    h.set_scene_cache(True)       # the frames of the step's forward pass, as MuJoCoRL sets it (what mjv_updateScene reads)
    h.reset()
    ora = OracleEnv(packed)
    steps = 40 + 60 * (seed % 5)            # in the air, landing, at rest
    for _ in range(steps):
        h.step_host(None, 1)
    ora.step(steps)
    assert np.abs(h.get_field("qpos") - ora.qpos).max() < 1e-9, seed
    for w, hh in ((64, 64), (72, 40)):
        os.environ.pop("MJRL_RENDER_LOOSE", None)
        got = h.render(w, hh)
        os.environ["MJRL_RENDER_LOOSE"] = "1"
        loose = h.render(w, hh)
        os.environ.pop("MJRL_RENDER_LOOSE", None)
        if not np.array_equal(got, loose):
            cull_diffs += 1
            bad.append((seed, w, hh, "tight culls change pixels", int((got != loose).any(axis=-1).sum())))
        for cam in range(model.ncam):
            # (a camera that ended up under the floor -- its body lies on its side -- looks at the wall's bottom face, which
            # is coplanar with the floor: whether the floor shadows that face is a coin toss of the last bit)
            if ora.cam_xpos[cam][2] < 0.01:
                skipped += 1
                continue
            ref = ora.render(cam, w, hh).reshape(hh, w, 3).astype(int)
            differ = np.abs(got[0, cam].astype(int) - ref).max(axis=-1)
            share, share_any = float((differ > 1).mean()), float((differ > 0).mean())
            n_out = int(unexplained(got[0, cam].astype(int), ref).sum())
            n_unexplained += n_out
            worst_out = max(worst_out, n_out)
            worst_share, worst_any = max(worst_share, share), max(worst_any, share_any)
            images += 1
            if share > 0.02 or n_out > 8:
                bad.append((seed, w, hh, f"camera {cam}", share))
                if os.environ.get("RENDER_FUZZ_DUMP") and len(dumps) < 40:
                    dumps[f"{seed}_{w}_{cam}_gpu"] = got[0, cam].copy()
                    dumps[f"{seed}_{w}_{cam}_ref"] = ref.astype(np.uint8)
    h.close(); ora.close()
print(f"{n_scenes} scenes, {images} images against the oracle: worst share of pixels off by more than one level {worst_share:.4f}, "
      f"by any amount {worst_any:.4f}; images changed by the tight culls: {cull_diffs}; {time.time() - t0:.0f} s")
if dumps:
    np.savez_compressed(os.environ["RENDER_FUZZ_DUMP"], **dumps)
print(f"views from under the floor left out: {skipped}")
print(f"pixels with a channel outside the range of the oracle's 3 x 3 pixels around them by more than two levels (not an edge or "
      f"a steep gradient shifted by one pixel): {n_unexplained} in all, {worst_out} in the worst image" + (f"  [variant {variant}]" if variant else ""))
for b in bad[:20]:
    print("BAD", b)
sys.exit(1 if bad else 0)
