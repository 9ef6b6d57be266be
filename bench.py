"""Headline benchmark: env-steps/sec of the batched multi-agent env step on N MI355X of one node.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  Typed plainly with N > 1 it starts the N rank
processes itself (fresh children under ``python -m torch.distributed.run``, before this process touches a GPU); the
driver's own ``torch.distributed.run`` launch (WORLD_SIZE set) is used as it comes.  A "step" is one step() of every env
copy: action scatter -> skip_frames=1 physics step -> per-agent observation gather (+ the fused plugin ops), one kernel
launch per rank.  The env batch shards across ranks (weak scaling: 4096 copies per GPU) with no collective on the step
path; torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the timed region.

Workload: episodes of maxSteps = 1024 steps with every copy at its own point of its episode (copy e is ``phase(e)``
steps ahead of copy 0, phases uniform over 0..1023), so every launch sees the stationary mix of airborne, landed and
resetting copies that an autoreset sampler sees -- and a window of K steps measures the same thing whatever K is.  The
mix is set up by an untimed pre-roll of one episode length; a copy whose episode is over is reset inside the step
launch (mjrl_set_step_reset_mask), as in the reference's loop (benchmarking/different_env_configs/fps_benchmark.py:
33-38: ``env.reset()`` then 1024 x ``env.step``).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ENVS_PER_GPU = 4096
ACT_RING = 64
EPISODE = 1024                   # maxSteps of the reference's benchmark configs (fps_benchmark.py:21)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, MI355X_MICROARCH.md chip-level table
WORKLOADS = {
    # stand-in for the unshipped MultiEnvs.xml = benchmarking/levels/MultiAgentModel.xml
    "two_agent": dict(level="two_agent.xml", agents=["sender", "receiver"],
                      what="2-agent ant arena (two_agent.xml = benchmarking/levels/MultiAgentModel.xml, stand-in for the "
                           "unshipped MultiEnvs.xml)"),
    # BASELINE config 4: no such level ships; the 2-agent arena with four ants
    "four_agent": dict(level="four_agent.xml", agents=["sender", "receiver", "agent_3", "agent_4"],
                       what="4-agent contact-heavy arena (four_agent.xml: the 2-agent arena with the ant instantiated four "
                            "times, SURVEY 8d config 4; no such level ships with the reference)"),
    # BASELINE config 5: every step also ray-casts both agent cameras (64x64x3 uint8 each) into HBM; 512 copies
    "camera": dict(level="two_agent.xml", agents=["sender", "receiver"], cameras=True, envs=512,
                   what="2-agent ant arena + both agent cameras (64x64x3 uint8 each, ray cast) per step, BASELINE config 5"),
    # BASELINE config 5 as a vision policy consumes it: the images are encoded on the matrix cores (vision/autoencoder.py:
    # 12-18, latent 100 as vision/train.py:70, random-init weights) and the latents end the observation rows
    "camera_latents": dict(level="two_agent.xml", agents=["sender", "receiver"], cameras=True, encoder=True, envs=512,
                           what="2-agent ant arena + both agent cameras (64x64x3 uint8, ray cast) + encoder latents (100 per "
                                "agent, bf16 MFMA) in the observation per step, BASELINE config 5 feeding vision obs"),
}
# what the default run measures after the headline, ~50 steps each (bench line field `configs`): BASELINE.json configs[1],
# [3] per GPU, [4] with and without the encoder.  (level, language channel, env copies)
EXTRA_CONFIGS = [("config 2", "two_agent", False, 1024), ("config 4", "four_agent", False, 4096),
                 ("config 5", "camera", False, 512), ("config 5 + encoder", "camera_latents", False, 512)]
ENCODER_LATENT = 100


def algorithmic_bytes_per_env_step(nq, nv, n_act_total, obs_total, n_agent, n_slot=0, s=8):
    """state in/out + warm start in/out + actions + observations + data-store read/write +
    per-agent reward/term/trunc.  fp64 (s = 8): 2-agent level without plugins 2460 B, with the Language channel
    (one action, one observation and one data-store slot more per agent) 2524 B; 4-agent arena 6800 B + 24."""
    return s * (2 * nq + 2 * nv + 2 * nv + n_act_total + obs_total) + n_agent * n_slot * 2 * s + n_agent * (4 + 1 + 1)


def action_stream(seed, first_env, n_env, n_steps, n_agent, act_dim, n_phys):
    """uniform(-1, 1) keyed on the GLOBAL env id (Philox counter), so a copy's trajectory does not depend on the GPU
    count -- and the CPU baseline steps env id w on exactly the stream the GPU feeds that copy.  Slots past the motors
    are utterances of the Language channel: uniform(0, 3)."""
    out = np.empty((n_steps, n_env, n_agent, act_dim), np.float64)
    for e in range(n_env):
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, first_env + e]))
        out[:, e] = rng.uniform(-1.0, 1.0, (n_steps, n_agent, act_dim))
    if act_dim > n_phys:
        out[..., n_phys:] = 1.5 * (out[..., n_phys:] + 1.0)
    return out


def phase_of(global_env, envs_per_gpu):
    """Episode phase of a copy: uniform over 0..EPISODE-1 inside every rank's shard, a function of the global id."""
    return (np.asarray(global_env, dtype=np.int64) * EPISODE // max(envs_per_gpu, 1)) % EPISODE


def cpu_share(limit=None):
    """Worker processes for the CPU baseline: the cores this job may actually use -- the scheduler affinity, cut by the
    cgroup's CPU quota where one is set -- and at most `limit` (default 16: a one-GPU box's share of its host; the other
    cores of a 256-thread host belong to the other GPUs' jobs, and 256 workers there measured the neighbours' load:
    1.44 M env-steps/s one day, 0.66 M the next, at an unchanged 50 k per core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, min(n, limit or 16))


def cpu_baseline(level_file, blob_bytes, scatter, n_agent, act_dim, n_phys, seconds=8.0, language=False, workers=None,
                 single_only=False):
    """The step on the host cores, on a bounded sample of the same workload: worker w steps GLOBAL env id w with the
    action stream the GPU feeds that copy (same Philox key and counter), episodes of 1024 steps from reset, for
    `seconds` of wall time; once on one core, once with one process per core of this job's CPU share (cpu_share).  kind "reference": the box has the
    reference's own physics library (`mujoco`), run as the reference runs it (mj_step + numpy gather, solver set to PGS
    like the kernel).  kind "port": it has not (the case in this image) and the repo's fp64 restatement
    (oracle/ora_step.c) stands in."""
    import multiprocessing as mp
    have_mujoco = importlib.util.find_spec("mujoco") is not None
    flat = scatter.reshape(-1)
    routed = flat >= 0

    def make_env():
        if have_mujoco:
            import mujoco
            model = mujoco.MjModel.from_xml_path(level_file)
            model.opt.solver = mujoco.mjtSolver.mjSOL_PGS
            data = mujoco.MjData(model)

            class Ref:
                qpos, qvel, ctrl, sensordata = data.qpos, data.qvel, data.ctrl, data.sensordata
                def step(self): mujoco.mj_step(model, data)
                def reset(self):
                    mujoco.mj_resetData(model, data)
                    mujoco.mj_forward(model, data)
            return Ref()
        from oracle.oracle import OracleEnv
        return OracleEnv(blob_bytes)

    def run_one(q, worker, duration):
        env = make_env()
        ring = action_stream(0, worker, 1, ACT_RING, n_agent, act_dim, n_phys)[:, 0]
        n = 0
        store = [{} for _ in range(n_agent)]
        env.reset()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < duration:
            for _ in range(64):
                act = ring[n % ACT_RING]
                env.ctrl[flat[routed]] = act.reshape(-1)[routed]
                env.step()
                # the reference's observation gather: sensordata | qpos | qvel per agent (+ the language channel)
                for a in range(n_agent):
                    obs = np.concatenate([env.sensordata[[a]], env.qpos, env.qvel])
                    if language:
                        store[a]["utterance"] = int(act[a, n_phys])
                        obs = np.concatenate((obs, np.array([store[(a + 1) % n_agent].get("utterance", 0)])))
                n += 1
                if n % EPISODE == 0:
                    env.reset()
                    store = [{} for _ in range(n_agent)]
        q.put((n, time.perf_counter() - t0))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    error = None
    try:
        run_one(q, 0, seconds)
    except Exception as exc:            # a broken mujoco install must not cost the bench line
        if not have_mujoco:
            raise
        error, have_mujoco = repr(exc), False
        run_one(q, 0, seconds)
    n1, t1 = q.get()
    if single_only:              # config 1: the reference's own shape -- one env, one thread
        found = importlib.util.find_spec("mujoco") is not None
        return {"value": n1 / t1, "unit": "env-steps/s", "cores": 1, "kind": "reference" if have_mujoco else "port",
                "sample": f"{'mujoco' if have_mujoco else 'CPU oracle (oracle/ora_step.c)'}; 1 process x 1 env copy x "
                          f"{seconds:.0f} s of {os.path.basename(level_file)}, episodes of {EPISODE} steps from reset, "
                          "step + numpy obs gather", "mujoco_found": found}
    cores = cpu_share(workers)
    procs = [ctx.Process(target=run_one, args=(q, i, seconds)) for i in range(cores)]
    for p in procs:
        p.start()
    results = [q.get() for _ in procs]
    for p in procs:
        p.join()
    total = sum(n / t for n, t in results)
    found = importlib.util.find_spec("mujoco") is not None
    engine = "mujoco (the reference's physics, solver PGS)" if have_mujoco else "CPU oracle (oracle/ora_step.c)"
    out = {"value": total, "unit": "env-steps/s", "cores": cores, "kind": "reference" if have_mujoco else "port",
           "sample": f"{engine}; importlib.util.find_spec('mujoco') -> {'found' if found else 'None'}; {cores} processes x "
                     f"1 env copy x {seconds:.0f} s of {os.path.basename(level_file)}, worker w = global env id w on the GPU "
                     f"run's Philox action stream, episodes of {EPISODE} steps from reset, step + numpy obs gather",
           "mujoco_found": found, "single_thread": n1 / t1, "host_threads": os.cpu_count()}
    if error:
        out["mujoco_error"] = error
    return out


class DeviceBatch:
    """The product path: ``MuJoCoRL`` over libmjrl_hip.so, tensors resident in HBM."""

    def __init__(self, torch, dev, level_file, agents, n_env, plugins, args, stream, cameras=False, encoder=False,
                 shares_device=False):
        from mjrl_amd.mujoco_rl import MuJoCoRL
        self.torch, self.dev = torch, dev
        cfg = {"xmlPath": level_file, "agents": agents, "numEnvs": n_env, "deviceId": dev.index,
               "sharesDevice": shares_device,
               "skipFrames": 1, "maxSteps": EPISODE, "environmentDynamics": plugins,
               "nconmax": args.nconmax, "njmax": args.njmax, "agentCameras": cameras}
        if encoder:          # random-init weights of the reference's architecture (no checkpoint ships, no network)
            rng = np.random.default_rng(0)
            he = lambda *shape, fan: (rng.standard_normal(shape) * np.sqrt(2.0 / fan)).astype(np.float32)
            cfg["cameraEncoder"] = {"relu": True, "weights": {
                "w1": he(3, 3, 3, 32, fan=27), "b1": np.zeros(32, np.float32), "w2": he(3, 3, 32, 64, fan=288),
                "b2": np.zeros(64, np.float32), "wd": he(16384, ENCODER_LATENT, fan=16384), "bd": np.zeros(ENCODER_LATENT, np.float32)}}
        self.env = MuJoCoRL(cfg)
        self.rgb = None
        if cameras and not encoder:   # the images of every step land here (get_camera_data's content, kept in HBM);
                                      # with the encoder the step itself renders and encodes them (mjrl_set_camera_obs)
            self.rgb = torch.empty((n_env, self.env._handle.size("ncam"), 64, 64, 3), dtype=torch.uint8, device=dev)
        self.stream = stream
        self.env.set_stream(stream.cuda_stream)
        self.env.reset_batched()
        self.obs_dim = self.env._handle.size("obs_dim")
        self.model = self.env._compiled
        self.kernel = "mjrl_step_kernel_spec" if self.env._handle.kernel == "specialised" else "mjrl_step_kernel"
        self.agents_action_index = self.env.agents_action_index
        self.blob = self.env._blob

    def buffers(self, n_env, n_agent):
        t, dev = self.torch, self.dev
        return (t.empty((n_env, n_agent, self.obs_dim), dtype=t.float64, device=dev),
                t.empty((n_env, n_agent), dtype=t.float64, device=dev),
                t.empty((n_env, n_agent), dtype=t.uint8, device=dev),
                t.empty((n_env, n_agent), dtype=t.uint8, device=dev))

    def to_device(self, array):
        return self.torch.from_numpy(np.ascontiguousarray(array)).to(self.dev)

    def set_step_reset_mask(self, row):
        self.env._handle.set_step_reset_mask(None if row is None else row.data_ptr())

    def step_batched(self, actions, obs, reward, term, trunc):
        with self.torch.cuda.stream(self.stream):
            self.env.step_batched(actions, obs, reward, term, trunc)
            if self.rgb is not None:
                self.env._handle.render(64, 64, d_rgb=self.rgb.data_ptr())

    def solver_stats(self):
        return self.env._handle.get_field("solver_stats")

    def timesteps(self):
        return self.env._handle.get_field("timestep").astype(np.int64)

    def cap_overflows(self):
        return self.env._handle.cap_overflows()

    def finite(self, obs):
        return bool(self.torch.isfinite(obs).all().item())

    def close(self):
        self.env.close()


class RehearsalBatch:
    """MJRL_BENCH_REHEARSAL=cpu: the same control flow without a GPU -- the env copies are stepped by the CPU
    lane-emulation of the device source (tests/emu, test infrastructure).  Its line is marked as a rehearsal and is not
    a measurement."""

    def __init__(self, level_file, agents, n_env, language):
        from tests.emu.batch import EmuBatch
        self.b = EmuBatch(level_file, agents, n_env, language=language, max_steps=EPISODE)
        self.obs_dim, self.model, self.kernel = self.b.obs_dim, self.b.model, self.b.kernel
        self.agents_action_index, self.blob = self.b.agents_action_index, self.b.blob

    def buffers(self, n_env, n_agent):
        return (np.zeros((n_env, n_agent, self.obs_dim)), np.zeros((n_env, n_agent)),
                np.zeros((n_env, n_agent), np.uint8), np.zeros((n_env, n_agent), np.uint8))

    def to_device(self, array):
        return np.ascontiguousarray(array)

    def set_step_reset_mask(self, row):
        self.b.set_step_reset_mask(row)

    def step_batched(self, actions, obs, reward, term, trunc):
        self.b.step_batched(actions, obs, reward, term, trunc)

    def solver_stats(self):
        return self.b.solver_stats()

    def timesteps(self):
        return self.b.timesteps()

    def cap_overflows(self):
        return self.b.cap_overflows()

    def finite(self, obs):
        return bool(np.isfinite(obs).all())

    def close(self):
        self.b.close()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)     # one reference episode (maxSteps = 1024)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preroll", type=int, default=EPISODE,
                    help="untimed steps before the warm-up that spread the copies over their episodes (default: one episode)")
    ap.add_argument("--level", choices=sorted(WORKLOADS), default="two_agent")
    ap.add_argument("--nconmax", type=int, default=None, help="contact cap per env copy (default: the compiler's)")
    ap.add_argument("--njmax", type=int, default=None, help="constraint-row cap per env copy (default: the compiler's)")
    ap.add_argument("--envs-per-gpu", type=int, default=None, help="env copies per GPU (default 4096; 512 for --level camera)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-workers", type=int, default=None,
                    help="processes of the CPU baseline (default: this job's CPU share, at most 16)")
    ap.add_argument("--groups", type=int, default=1,
                    help="step the batch as this many independent groups on their own streams (1 = lockstep, the headline)")
    ap.add_argument("--double-buffer", action="store_true",
                    help="also measure the batch as two independent half-batches on two streams (reported beside the value)")
    ap.add_argument("--no-language", action="store_true", help="config 2: physics + gather only, no Language channel")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the BASELINE configs 2, 4, 5 that the default one-GPU run measures after the headline")
    ap.add_argument("--extra-steps", type=int, default=50, help="timed steps of each of those configs")
    return ap.parse_args(argv)


def launch_ranks(n):
    """`python bench.py --gpus N` typed plainly: start the N ranks as fresh children BEFORE this process imports torch
    or touches a GPU (a process that has initialised the GPU must never be replaced or forked into ranks), pass their
    output through and exit with their status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the CPU baseline of the line: measured HERE, in the parent that never touches a GPU, before the ranks exist (so that
    # it has the host cores to itself), and handed to rank 0 through the environment
    args = parse_args()
    if not args.no_cpu_baseline and os.environ.get("MJRL_BENCH_REHEARSAL", "") != "cpu":
        entry.load_package()
        env["MJRL_BENCH_CPU_BASELINE"] = json.dumps(host_cpu_baseline(args))
    return subprocess.call(cmd, env=env)


def host_tables(level_file, agents):
    """What the CPU baseline needs of a level, built without touching a GPU: the packed model and every agent's ctrl
    indices (mujoco_parent.py:274-314)."""
    from mjrl_amd import blob as blob_mod, mjcf
    from mjrl_amd.mujoco_parent import MuJoCoParent
    tables = MuJoCoParent.tables_only(level_file)
    index = {}
    for a in agents:
        tables.get_action_space_mujoco(a)
        index[a] = list(tables.agents_action_index[a])
    return blob_mod.pack(mjcf.compile_mjcf(level_file)), index


def host_cpu_baseline(args):
    """The `cpu_baseline` object of the bench line for the workload `args` names.  Touches no GPU."""
    from mjrl_amd import levels
    work = WORKLOADS[args.level]
    agents, level_file = work["agents"], levels.level_path(work["level"])
    n_agent = len(agents)
    language = n_agent == 2 and not args.no_language and not work.get("cameras")
    packed, index = host_tables(level_file, agents)
    n_phys0 = max(len(index[a]) for a in agents)
    act_dim0 = n_phys0 + (1 if language else 0)
    scatter = np.full((n_agent, act_dim0), -1, np.int64)
    for k, a in enumerate(agents):
        scatter[k, :len(index[a])] = index[a]
    return cpu_baseline(level_file, packed, scatter, n_agent, act_dim0, n_phys0, language=language, workers=args.cpu_workers)


def config_one_cpu(seconds=4.0):
    """BASELINE configs[0], CPU side: single_agent.xml, 1 env copy, one thread."""
    from mjrl_amd import levels
    level_file = levels.level_path("single_agent.xml")
    packed, index = host_tables(level_file, ["sender"])
    n_phys = len(index["sender"])
    scatter = np.array([index["sender"]], np.int64)
    return cpu_baseline(level_file, packed, scatter, 1, n_phys, n_phys, seconds=seconds, single_only=True)


def config_one_product(episodes=3):
    """BASELINE configs[0], product side: the drop-in at numEnvs = 1 through the reference's own loop
    (benchmarking/different_env_configs/fps_benchmark.py:52-62: reset, then 1024 x step({agent: action_space.sample()}),
    FPS = 1024 / elapsed; skipFrames = 1 so that the physics runs): dict API, one launch and one synchronous
    copy back per step."""
    from mjrl_amd import levels
    from mjrl_amd.mujoco_rl import MuJoCoRL
    env = MuJoCoRL({"xmlPath": levels.level_path("single_agent.xml"), "agents": ["sender"], "rewardFunctions": [],
                    "doneFunctions": [], "skipFrames": 1, "environmentDynamics": [], "freeJoint": False,
                    "renderMode": False, "maxSteps": EPISODE})
    fps = []
    for _ in range(episodes):
        env.reset()
        t0 = time.perf_counter()
        for _ in range(EPISODE):
            env.step({"sender": env.action_space("sender").sample()})
        fps.append(EPISODE / (time.perf_counter() - t0))
    kernel = env._handle.kernel
    few = env._handle.size("few")
    env.close()
    return {"value": float(np.median(fps)), "unit": "env-steps/s", "episodes": episodes, "api": "MuJoCoRL.step (dict of agents)",
            "kernel": kernel, "few_copies_build": bool(few)}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsals of the multi-rank control flow: "1" = every rank on device 0 of a one-GPU box, gloo instead of RCCL;
    # "cpu" = no GPU at all, the copies stepped by the CPU emulation of the device source (tests/emu)
    rehearsal = os.environ.get("MJRL_BENCH_REHEARSAL", "")
    on_cpu = rehearsal == "cpu"
    if rehearsal:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    entry.load_package()
    from mjrl_amd import levels
    from mjrl_amd.dynamics import Language

    work = WORKLOADS[args.level]
    agents, level_file = work["agents"], levels.level_path(work["level"])
    n_agent = len(agents)
    # the Language channel is a 2-agent dynamic (README.md:109-136)
    language = n_agent == 2 and not args.no_language and not work.get("cameras")

    # The CPU baseline runs FIRST, before this process makes its first GPU call: its workers are forked from a process
    # without HIP state (round 2 forked them from one whose runtime was live).
    # (N > 1: the launcher parent of `python bench.py --gpus N` has measured it before it started the ranks; under an
    # external launcher rank 0 measures it here, before it joins the rendezvous, while the other ranks wait there)
    cpu_line, config_one = None, None
    if rank == 0 and not args.no_cpu_baseline and not on_cpu:
        if os.environ.get("MJRL_BENCH_CPU_BASELINE"):
            cpu_line = json.loads(os.environ["MJRL_BENCH_CPU_BASELINE"])
        else:
            cpu_line = host_cpu_baseline(args)
        if world == 1 and args.level == "two_agent" and not args.no_extra_configs:
            config_one = {"cpu": config_one_cpu()}

    import torch
    dev = None
    if not on_cpu:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_run(work, language, n_env, steps, warmup, preroll, groups):
        """The batch as `groups` env objects of n_env / groups copies, each on its own HIP stream (1: the whole batch
        in lockstep on the current stream).  Returns (batches, wall seconds, ms per step by HIP events, stats, n_phys)."""
        agents, level_file = work["agents"], levels.level_path(work["level"])
        n_agent = len(agents)
        plugins = [Language] if language else []
        cameras, encoder = bool(work.get("cameras")), bool(work.get("encoder"))
        per = n_env // groups
        batches, bufs, acts, masks = [], [], [], []
        for g in range(groups):
            if on_cpu:
                batch = RehearsalBatch(level_file, agents, per, bool(plugins))
            else:
                stream = torch.cuda.current_stream(dev) if groups == 1 else torch.cuda.Stream(dev)
                batch = DeviceBatch(torch, dev, level_file, agents, per, plugins, args, stream, cameras=cameras, encoder=encoder,
                                    shares_device=groups > 1)
            n_phys = max(len(batch.agents_action_index[a]) for a in agents)
            act_dim = n_phys + len(plugins)
            first = rank * n_env + g * per
            # a cyclic buffer of ACT_RING steps of actions and the reset masks of every episode phase, resident in HBM
            # before the timed region
            acts.append(batch.to_device(action_stream(0, first, per, ACT_RING, n_agent, act_dim, n_phys)))
            phase = phase_of(first + np.arange(per), n_env)
            masks.append(batch.to_device((phase[None, :] == np.arange(EPISODE)[:, None]).astype(np.uint8)))
            bufs.append(batch.buffers(per, n_agent))
            batches.append(batch)
        if not on_cpu:
            torch.cuda.synchronize(dev)

        def one_step(i):
            # copy e starts a new episode at the steps i with (i - phase(e)) % 1024 == 0: it is reset inside the launch
            for batch, a, m, (obs, rew, term, trunc) in zip(batches, acts, masks, bufs):
                batch.set_step_reset_mask(m[i % EPISODE] if i else None)
                batch.step_batched(a[i % ACT_RING], obs, rew, term, trunc)

        lead = preroll + warmup
        for i in range(lead):
            one_step(i)
        if not on_cpu:
            torch.cuda.synchronize(dev)
        barrier()
        counters_before = [b.timesteps() for b in batches]
        kernel_ms = None
        if not on_cpu:
            # HIP events on the stream the step kernel is launched on
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if not on_cpu:
            ev0.record(batches[0].stream)
        for i in range(lead, lead + steps):
            one_step(i)
        if not on_cpu:
            ev1.record(batches[0].stream)
            torch.cuda.synchronize(dev)
        barrier()
        wall = time.perf_counter() - t0
        if not on_cpu and groups == 1:
            kernel_ms = ev0.elapsed_time(ev1) / steps
        if world > 1:
            t = torch.tensor([wall], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        for batch, (obs, _, _, _) in zip(batches, bufs):
            if not batch.finite(obs):
                raise SystemExit("non-finite observations after the timed region")
        # Every copy must have BEEN stepped: its step counter runs 1..EPISODE (an in-launch reset sets it to 0, the step
        # of that launch to 1), so over the timed region it advanced by exactly `steps` modulo the episode length -- a
        # kernel that returned early, read its arguments in the wrong place or skipped copies is an error, not a rate.
        for batch, before in zip(batches, counters_before):
            after = batch.timesteps()
            moved = (after - before - steps) % EPISODE
            if moved.any():
                bad = int(np.flatnonzero(moved != 0)[0])
                raise SystemExit(f"bench: {int((moved != 0).sum())} of {len(after)} copies were not stepped exactly {steps} "
                                 f"times in the timed region (copy {bad}: counter {int(before[bad])} -> {int(after[bad])})")
        # what the copies did in the last timed step: with staggered episodes one step is a sample of the stationary mix
        stats = np.concatenate([np.asarray(b.solver_stats()) for b in batches]).astype(np.float64)
        n_phys = max(len(batches[0].agents_action_index[a]) for a in agents)
        return batches, wall, kernel_ms, stats, n_phys

    def roofline_of(work, level_name, batch, language, n_env, steps, wall, kernel_ms):
        """The bench line's `roofline` object for one measured workload (algorithmic bytes per env-step x the
        env-steps of a launch / the launch's duration by HIP events, against the HBM peak)."""
        m, n_agent = batch.model, len(work["agents"])
        n_phys = max(len(batch.agents_action_index[a]) for a in work["agents"])
        act_dim = n_phys + (1 if language else 0)
        bytes_per = algorithmic_bytes_per_env_step(m.nq, m.nv, n_agent * act_dim, n_agent * batch.obs_dim, n_agent,
                                                   n_slot=1 if language else 0)
        cameras = bool(work.get("cameras"))
        if cameras:                             # + the pixels written per env-step
            bytes_per += m.ncam * 64 * 64 * 3
        if kernel_ms is None:                 # several groups / no GPU: the step time is the wall time's
            kernel_ms = wall / steps * 1e3
        achieved = bytes_per * n_env / (kernel_ms * 1e-3) / 1e9
        traffic, source = None, None
        # HBM bytes per launch: NOT measured in this run -- read from the committed rocprofv3 --pmc passes of this same
        # command (profiles/, newest round first); `traffic_source` says so in the line
        for rnd in ("r04", "r03", "r02"):
            pmc = os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic_{level_name}.json")
            if os.path.exists(pmc) and n_env == work.get("envs", ENVS_PER_GPU) and not on_cpu and language == (level_name == "two_agent"):
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                source = f"profiles/{os.path.basename(pmc)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not measured in this run"
                break
        kernels = batch.kernel
        if cameras:
            kernels += " + mjrl_render_kernel"       # (the scene rows come from the step kernel: mjrl_set_scene_cache)
        if work.get("encoder"):
            kernels += " + mjrl_encoder_conv_kernel + mjrl_encoder_dense_kernel"
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": source, "kernel": kernels, "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_env_step": bytes_per}

    if args.envs_per_gpu is None:
        args.envs_per_gpu = work.get("envs", ENVS_PER_GPU)
    n_env = args.envs_per_gpu
    batches, wall, kernel_ms, stats, n_phys = timed_run(work, language, n_env, args.steps, args.warmup, args.preroll, args.groups)
    batch = batches[0]
    obs_dim = batch.obs_dim
    act_dim = n_phys + (1 if language else 0)
    # frames (pre-roll and warm-up included) in which a copy ran into its contact / row cap, i.e. dropped work: must be zero
    overflows = [sum(c) for c in zip(*(b.cap_overflows() for b in batches))]
    if any(overflows):
        print(f"bench: rank {rank}: {overflows[0]} frames hit nconmax, {overflows[1]} hit njmax -- raise the caps", file=sys.stderr)
    for b in batches[1:]:
        b.close()
    # Not the headline: the same batch as two half-batches that step independently on two streams (what a
    # double-buffered sampler does: the policy works on one half while the other half steps).
    double_buffered = None
    if world == 1 and args.groups == 1 and args.double_buffer and not on_cpu:
        b2, wall2, _, _, _ = timed_run(work, language, n_env, args.steps, args.warmup, args.preroll, 2)
        double_buffered = {"groups": 2, "value": n_env * args.steps / wall2, "unit": "env-steps/s",
                           "ms_per_step": wall2 / args.steps * 1e3}
        for b in b2:
            b.close()

    line = None
    if rank == 0:
        m = batch.model
        line = {
            "metric": "env-steps/sec", "value": n_env * world * args.steps / wall, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": ("REHEARSAL on the CPU emulation of the device source -- not a measurement" if on_cpu else
                     "REHEARSAL: every rank on device 0, gloo in place of RCCL -- not a measurement" if rehearsal else "synthetic"),
            "config": {"workload": f"{work['what']}, {n_env} env copies per GPU, skipFrames=1, PGS solver, episodes of "
                                   f"{EPISODE} steps with the copies' episode phases uniform over 0..{EPISODE - 1} (in-launch "
                                   f"reset of the copies whose episode is over); action scatter + physics step + per-agent "
                                   f"obs gather{' + Language channel' if language else ''} fused in one launch",
                       "level": args.level, "envs_per_gpu": n_env, "agents": n_agent, "nq": m.nq, "nv": m.nv,
                       "obs_dim": obs_dim, "act_dim": act_dim, "nconmax": m.nconmax, "njmax": m.njmax,
                       "preroll_steps": args.preroll, "episode_steps": EPISODE,
                       "mean_ncon": float(stats[:, 0].mean()), "mean_nefc": float(stats[:, 1].mean()),
                       "mean_solver_sweeps": float(stats[:, 2].mean()),
                       "max_ncon": int(stats[:, 0].max()), "max_nefc": int(stats[:, 1].max()),
                       "max_solver_sweeps": int(stats[:, 2].max()),
                       "cap_overflow_frames": list(overflows), "groups": args.groups},
            "roofline": roofline_of(work, args.level, batch, language, n_env, args.steps, wall, kernel_ms),
        }
        if double_buffered:
            line["double_buffered"] = double_buffered
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
    batch.close()

    # The other BASELINE configs, measured by the same command after the headline (one GPU, default workload only): the
    # driver runs nothing but the default command, so this is where their numbers come from.  ~50 timed steps each after
    # their own pre-roll (every copy at its own point of its episode, as above); the headline `value` / `config` are
    # untouched by them.
    if world == 1 and not on_cpu and args.level == "two_agent" and not args.no_extra_configs and args.groups == 1:
        extra = []
        for name, level_name, lang, envs in EXTRA_CONFIGS:
            w = WORKLOADS[level_name]
            bs, wall_x, kms_x, stats_x, _ = timed_run(w, lang, envs, args.extra_steps, 5, args.preroll, 1)
            b = bs[0]
            over = list(b.cap_overflows())
            extra.append({
                "name": name, "workload": f"{w['what']}, {envs} env copies, skipFrames=1, no plugins, staggered episodes",
                "level": level_name, "envs_per_gpu": envs, "steps": args.extra_steps,
                "value": envs * args.extra_steps / wall_x, "unit": "env-steps/s", "ms_per_step": wall_x / args.extra_steps * 1e3,
                "obs_dim": b.obs_dim, "mean_ncon": float(stats_x[:, 0].mean()), "mean_nefc": float(stats_x[:, 1].mean()),
                "mean_solver_sweeps": float(stats_x[:, 2].mean()), "cap_overflow_frames": over,
                "roofline": roofline_of(w, level_name, b, lang, envs, args.extra_steps, wall_x, kms_x)})
            b.close()
        if config_one is not None:
            # BASELINE configs[0]: what the drop-in costs at numEnvs = 1 (one lone wave per step, a synchronous launch, the
            # dict API) next to one CPU core running the same level
            config_one["product"] = config_one_product()
            extra.insert(0, {"name": "config 1", "workload": "single_agent.xml (benchmarking/levels/SingleAgentModel.xml), "
                             "agents [sender], 1 env copy, skipFrames=1, the loop of fps_benchmark.py:52-62",
                             "level": "single_agent", "envs_per_gpu": 1, **config_one,
                             "value": config_one["product"]["value"], "unit": "env-steps/s"})
        line["configs"] = extra
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
