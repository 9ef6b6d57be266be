"""The camera encoder (/root/reference vision/autoencoder.py:12-18) on the matrix cores, through the
C-ABI, against a plain PyTorch fp32 reference of the same network: Conv2D(32, 3x3, relu, stride 2, same) ->
Conv2D(64, 3x3, relu, stride 2, same) -> Flatten (h, w, c) -> Dense(latent, relu) on images scaled by 1/255, TensorFlow
"same" padding (one row / column at the END for stride 2 on an even size).

Tolerances (floating point, stated here as the tier rule asks): the kernel computes in bf16 operands with fp32
accumulation and rounds the activations to bf16 between the layers.  Against a reference that applies the same roundings
(weights and activations to bf16, fp32 sums) it agrees to 2e-3 of the largest latent (summation order only); against the
all-fp32 network to 3e-2 of the largest latent."""
import numpy as np
import pytest

from mjrl_amd import _capi, blob, levels, mjcf
from mjrl_amd.mujoco_rl import MuJoCoRL

pytestmark = pytest.mark.gpu
AGENTS = ["sender", "receiver"]


def make_weights(latent, seed=0):
    rng = np.random.default_rng(seed)
    he = lambda *shape, fan: (rng.standard_normal(shape) * np.sqrt(2.0 / fan)).astype(np.float32)
    return {"w1": he(3, 3, 3, 32, fan=27), "b1": (0.1 * rng.standard_normal(32)).astype(np.float32),
            "w2": he(3, 3, 32, 64, fan=288), "b2": (0.1 * rng.standard_normal(64)).astype(np.float32),
            "wd": he(16384, latent, fan=16384), "bd": (0.1 * rng.standard_normal(latent)).astype(np.float32)}


def reference(images, w, relu=True, bf16=False):
    """[n, 64, 64, 3] uint8 -> [n, latent]; ``bf16``: round weights and inter-layer activations like the kernel does."""
    import torch
    import torch.nn.functional as F
    r = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if bf16 else (lambda t: t)
    x = torch.from_numpy(images.astype(np.float32))
    w1 = torch.from_numpy(w["w1"])
    if bf16:
        w1, x = r(w1 / 255.0), x            # the kernel folds 1/255 into the first layer's weights
    else:
        x = x / 255.0
    x = x.permute(0, 3, 1, 2)                                         # NCHW
    x = F.pad(x, (0, 1, 0, 1))                                        # TF "same", stride 2: pad at the end only
    x = r(F.relu(F.conv2d(x, w1.permute(3, 2, 0, 1), torch.from_numpy(w["b1"]), stride=2)))
    x = F.pad(x, (0, 1, 0, 1))
    x = r(F.relu(F.conv2d(x, r(torch.from_numpy(w["w2"])).permute(3, 2, 0, 1), torch.from_numpy(w["b2"]), stride=2)))
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                 # Keras Flatten of NHWC
    y = x @ r(torch.from_numpy(w["wd"])) + torch.from_numpy(w["bd"])
    return (F.relu(y) if relu else y).numpy()


@pytest.mark.parametrize("latent,relu,n_img", [(100, True, 37), (32, False, 16), (7, True, 1)])
def test_encoder_against_the_torch_reference(latent, relu, n_img):
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    h = _capi.Handle(blob.pack(model), 2)
    w = make_weights(latent, seed=latent)
    h.encoder_load(w, relu=relu)
    rng = np.random.default_rng(1)
    images = rng.integers(0, 256, (n_img, 64, 64, 3), dtype=np.uint8)
    images[0] = 0
    if n_img > 2:
        images[1] = 255
        images[2, :, :, :] = np.arange(64, dtype=np.uint8)[None, :, None] * 4      # a ramp: catches transposed indexing
    got = h.encode(images)
    assert got.shape == (n_img, latent) and np.isfinite(got).all()
    same = reference(images, w, relu, bf16=True)
    full = reference(images, w, relu, bf16=False)
    scale = np.abs(full).max()
    assert np.abs(got - same).max() < 2e-3 * scale
    assert np.abs(got - full).max() < 3e-2 * scale
    if relu:
        assert (got >= 0).all() and (got == 0).any() and (got > 0).any()
    h.close()


def test_layer_indexing_with_exact_integer_weights():
    """Asymmetric integer data that every layer carries exactly in bf16: any swapped row / column / tap / channel index
    shows as a large error, not as rounding noise."""
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    h = _capi.Handle(blob.pack(model), 2)
    latent = 16
    w = {k: np.zeros_like(v) for k, v in make_weights(latent).items()}
    # conv1: channel n copies input channel n % 3 at tap (n % 9) scaled by 255 (so activations are the pixel values)
    for n in range(32):
        w["w1"][(n % 9) // 3, (n % 9) % 3, n % 3, n] = 255.0 / (1 + (n >= 16))
    # conv2: channel m copies a1 channel (m * 5) % 32 at tap (m % 9), halved so that sums stay small integers
    for m in range(64):
        w["w2"][(m % 9) // 3, (m % 9) % 3, (m * 5) % 32, m] = 0.5
    # dense: latent j sums a sparse, position-dependent selection with weights +-1
    rng = np.random.default_rng(0)
    idx = rng.choice(16384, (latent, 24), replace=False)
    for j in range(latent):
        w["wd"][idx[j], j] = rng.choice([-1.0, 1.0], 24)
    w["bd"][:] = np.arange(latent)
    h.encoder_load(w, relu=False)
    images = (np.random.default_rng(3).integers(0, 64, (5, 64, 64, 3)) * 4).astype(np.uint8)     # multiples of 4: halves stay exact
    got = h.encode(images)
    ref = reference(images, w, relu=False, bf16=False)
    assert np.abs(got - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())
    h.close()


def test_camera_latents_in_the_observation():
    """mjrl_set_camera_obs: every step's observation ends with the encoding of the agent camera's image at the new
    state -- equal to rendering (mjrl_render_*) and encoding (mjrl_encode_*) by hand -- on the host plugin path and with
    a fused program; config 5's batch size."""
    import torch
    from mjrl_amd.dynamics import Language
    latent = 24
    w = make_weights(latent, seed=5)
    cfg = {"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 6, "agentCameras": True,
           "cameraEncoder": {"weights": w, "relu": True}}
    env = MuJoCoRL(cfg)
    assert env.observation_space("sender").shape == (59 + latent,)
    obs0, _ = env.reset()
    assert obs0["sender"].shape == (6, 59 + latent) and not obs0["sender"][:, 59:].any()
    rng = np.random.default_rng(2)
    for _ in range(30):
        obs, rew, term, trunc, info = env.step({a: rng.uniform(-1, 1, (6, 8)) for a in AGENTS})
    images = env._handle.render(64, 64)                       # [6, 2, 64, 64, 3] at the state the step left
    lat = env._handle.encode(images.reshape(-1, 64, 64, 3)).reshape(6, 2, latent)
    for k, a in enumerate(AGENTS):
        assert obs[a].shape == (6, 59 + latent)
        assert np.array_equal(obs[a][:, 59:].astype(np.float32), lat[:, k])
        assert np.abs(obs[a][:, 59:]).max() > 0
    ref = reference(images.reshape(-1, 64, 64, 3), w, True, bf16=True).reshape(6, 2, latent)
    assert np.abs(lat - ref).max() < 2e-3 * np.abs(ref).max()
    env.close()
    # fused program + camera latents, 512 copies (config 5), device-resident
    fused = MuJoCoRL(dict(cfg, numEnvs=512, environmentDynamics=[Language]))
    assert fused._program is not None and fused._handle.size("obs_dim") == 60 + latent
    fused.reset_batched()
    g = torch.Generator(device="cuda").manual_seed(0)
    out = None
    for _ in range(5):
        act = torch.rand((512, 2, 9), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
        act[..., 8] = (act[..., 8] + 1) * 1.5
        out = fused.step_batched(act)
    torch.cuda.synchronize()
    o = out[0].cpu().numpy()
    images = fused._handle.render(64, 64)
    lat = fused._handle.encode(images.reshape(-1, 64, 64, 3)).reshape(512, 2, latent)
    assert np.array_equal(o[:, :, 60:].astype(np.float32), lat)
    assert np.array_equal(o[:, 1, 59], np.trunc(act[:, 0, 8].cpu().numpy()))          # the Language slot is still in place
    fused.close()
