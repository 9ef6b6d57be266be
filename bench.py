"""Headline benchmark: env-steps/sec of the batched 2-agent env step on N MI355X of one node.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched under
``python -m torch.distributed.run`` with one rank per GPU.  A "step" is one step() of every env copy:
action scatter -> skip_frames=1 physics step -> per-agent observation gather, one kernel launch per rank.
The env batch shards across ranks (weak scaling: 4096 copies per GPU) with no collective on the step path;
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the timed region.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ENVS_PER_GPU = 4096
ACT_RING = 64
AGENTS = ["sender", "receiver"]
LEVEL = "two_agent.xml"          # stand-in for the unshipped MultiEnvs.xml
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, MI355X_MICROARCH.md chip-level table


def algorithmic_bytes_per_env_step(nq, nv, n_act_total, obs_total, n_agent, n_slot=0, s=8):
    """state in/out + warm start in/out + actions + observations + data-store read/write +
    per-agent reward/term/trunc.  fp64 (s = 8): 2-agent level without plugins 2460 B, with the Language channel
    (one action, one observation and one data-store slot more per agent) 2524 B."""
    return s * (2 * nq + 2 * nv + 2 * nv + n_act_total + obs_total) + n_agent * n_slot * 2 * s + n_agent * (4 + 1 + 1)


def action_stream(seed, first_env, n_env, n_steps, n_agent, act_dim):
    """uniform(-1, 1) keyed on the GLOBAL env id, so a copy's trajectory does not depend on the GPU count.  Slots past
    the eight motors are utterances of the Language channel: uniform(0, 3)."""
    out = np.empty((n_steps, n_env, n_agent, act_dim), np.float64)
    for e in range(n_env):
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, first_env + e]))
        out[:, e] = rng.uniform(-1.0, 1.0, (n_steps, n_agent, act_dim))
    if act_dim > 8:
        out[..., 8:] = 1.5 * (out[..., 8:] + 1.0)
    return out


def cpu_baseline(blob_bytes, scatter, seconds=8.0, language=False):
    """The CPU oracle (kind "port": the repo's own fp64 restatement of the step; mujoco is not installed on the
    box) timed on the host cores on a bounded sample of the same workload: one env copy per worker, the same
    action distribution, for `seconds` of wall time; once single-threaded, once with one process per core."""
    import multiprocessing as mp
    from oracle.oracle import OracleEnv

    def run_one(q, seed, duration):
        env = OracleEnv(blob_bytes)
        rng = np.random.default_rng(seed)
        n = 0
        store = [{} for _ in range(scatter.shape[0])]
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < duration:
            for _ in range(64):
                act = rng.uniform(-1, 1, scatter.shape)
                env.ctrl[scatter.reshape(-1)] = act.reshape(-1)
                env.step()
                # the reference's observation gather: sensordata | qpos | qvel per agent (+ the language channel)
                for a in range(scatter.shape[0]):
                    obs = np.concatenate([env.sensordata[[a]], env.qpos, env.qvel])
                    if language:
                        store[a]["utterance"] = int(rng.uniform(0, 3))
                        obs = np.concatenate((obs, np.array([store[1 - a].get("utterance", 0)])))
            n += 64
            if env.time > 2.0:          # episode of 1024 steps ~ 2 s of sim time
                env.reset()
        q.put((n, time.perf_counter() - t0))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    run_one(q, 0, seconds)
    n1, t1 = q.get()
    cores = os.cpu_count() or 1
    procs = [ctx.Process(target=run_one, args=(q, 100 + i, seconds)) for i in range(cores)]
    for p in procs:
        p.start()
    results = [q.get() for _ in procs]
    for p in procs:
        p.join()
    total = sum(n / t for n, t in results)
    return {"value": total, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (oracle/ora_step.c), {cores} processes x 1 env copy x {seconds:.0f} s of "
                      f"{LEVEL} with uniform(-1,1) actions, step + numpy obs gather",
            "single_thread": n1 / t1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)     # one reference episode (maxSteps = 1024)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nconmax", type=int, default=None, help="contact cap per env copy (default: the compiler's)")
    ap.add_argument("--njmax", type=int, default=None, help="constraint-row cap per env copy (default: the compiler's)")
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--groups", type=int, default=1,
                    help="step the batch as this many independent groups on their own streams (1 = lockstep, the headline)")
    ap.add_argument("--no-double-buffer", action="store_true", help="skip the secondary two-group measurement")
    ap.add_argument("--no-language", action="store_true", help="config 2: physics + gather only, no Language channel")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank control flow on a box with one GPU: every rank on device 0, gloo instead of RCCL
    rehearsal = os.environ.get("MJRL_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    entry.load_package()
    from mjrl_amd import levels
    from mjrl_amd.mujoco_rl import MuJoCoRL

    n_env = args.envs_per_gpu
    from mjrl_amd.dynamics import Language
    plugins = [] if args.no_language else [Language]
    n_agent = len(AGENTS)
    act_dim = 8 + len(plugins)
    total_steps = args.warmup + args.steps
    # a cyclic buffer of ACT_RING steps of actions, resident in HBM before the timed region
    acts_host = action_stream(0, rank * n_env, n_env, ACT_RING, n_agent, act_dim)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_run(groups):
        """The batch as `groups` env objects of n_env / groups copies, each on its own HIP stream (1: the whole batch
        in lockstep on the current stream).  Returns (env objects, wall seconds, ms per step by HIP events, obs)."""
        per = n_env // groups
        envs, streams, bufs = [], [], []
        for g in range(groups):
            env = MuJoCoRL({"xmlPath": levels.level_path(LEVEL), "agents": AGENTS, "numEnvs": per, "deviceId": local,
                            "skipFrames": 1, "maxSteps": 1024, "environmentDynamics": plugins,
                            "nconmax": args.nconmax, "njmax": args.njmax})
            stream = torch.cuda.current_stream(dev) if groups == 1 else torch.cuda.Stream(dev)
            env._handle.set_stream(stream.cuda_stream)
            env.reset_batched()
            obs_dim = env._handle.size("obs_dim")
            acts = torch.from_numpy(np.ascontiguousarray(acts_host[:, g * per:(g + 1) * per])).to(dev)
            bufs.append((acts,
                         torch.empty((per, n_agent, obs_dim), dtype=torch.float64, device=dev),
                         torch.empty((per, n_agent), dtype=torch.float64, device=dev),
                         torch.empty((per, n_agent), dtype=torch.uint8, device=dev),
                         torch.empty((per, n_agent), dtype=torch.uint8, device=dev)))
            envs.append(env)
            streams.append(stream)
        torch.cuda.synchronize(dev)

        def one_step(i):
            # episodes are maxSteps = 1024 steps long, like the reference's benchmark loop (fps_benchmark.py:33-41):
            # every copy is reset when its episode is over; the reset launch is part of the timed workload
            for env, stream, (acts, obs, rew, term, trunc) in zip(envs, streams, bufs):
                with torch.cuda.stream(stream):
                    if i and i % 1024 == 0:
                        env.reset_batched()
                    env.step_batched(acts[i % ACT_RING], obs, rew, term, trunc)

        for i in range(args.warmup):
            one_step(i)
        torch.cuda.synchronize(dev)
        barrier()
        # HIP events on the stream the step kernel is launched on
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(streams[0])
        for i in range(args.warmup, total_steps):
            one_step(i)
        ev1.record(streams[0])
        torch.cuda.synchronize(dev)
        barrier()
        wall = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / args.steps if groups == 1 else None
        if world > 1:
            t = torch.tensor([wall], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        for _, obs, _, _, _ in bufs:
            if not torch.isfinite(obs).all().item():
                raise SystemExit("non-finite observations after the timed region")
        return envs, wall, kernel_ms

    envs, wall, kernel_ms = timed_run(args.groups)
    env = envs[0]
    obs_dim = env._handle.size("obs_dim")
    # frames (warm-up included) in which a copy ran into its contact / row cap, i.e. dropped work: must be zero
    overflows = [sum(c) for c in zip(*(e._handle.cap_overflows() for e in envs))]
    if any(overflows):
        print(f"bench: rank {rank}: {overflows[0]} frames hit nconmax, {overflows[1]} hit njmax -- raise the caps", file=sys.stderr)
    for e in envs[1:]:
        e.close()
    # Not the headline: the same batch as two half-batches that step independently on two streams (what a
    # double-buffered sampler does: the policy works on one half while the other half steps).  The tail of one half's
    # launch -- a few long solves -- then overlaps the body of the other's.
    double_buffered = None
    if world == 1 and args.groups == 1 and not args.no_double_buffer:
        envs2, wall2, _ = timed_run(2)
        double_buffered = {"groups": 2, "value": n_env * args.steps / wall2, "unit": "env-steps/s",
                           "ms_per_step": wall2 / args.steps * 1e3}
        for e in envs2:
            e.close()

    if rank == 0:
        m = env._compiled
        bytes_per = algorithmic_bytes_per_env_step(m.nq, m.nv, n_agent * act_dim, n_agent * obs_dim, n_agent,
                                                   n_slot=len(plugins))
        if kernel_ms is None:                 # several groups: launches overlap, the step time is the wall time's
            kernel_ms = wall / args.steps * 1e3
        achieved = bytes_per * n_env / (kernel_ms * 1e-3) / 1e9
        traffic = None
        # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (profiles/)
        pmc = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(pmc) and n_env == ENVS_PER_GPU:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        line = {
            "metric": "env-steps/sec", "value": n_env * world * args.steps / wall, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"2-agent ant arena ({LEVEL} = benchmarking/levels/MultiAgentModel.xml, stand-in for the "
                                   f"unshipped MultiEnvs.xml), {n_env} env copies per GPU, skipFrames=1, PGS solver, "
                                   f"action scatter + physics step + per-agent obs gather fused in one launch",
                       "envs_per_gpu": n_env, "agents": n_agent, "nq": m.nq, "nv": m.nv, "obs_dim": obs_dim,
                       "nconmax": m.nconmax, "njmax": m.njmax, "cap_overflow_frames": list(overflows),
                       "groups": args.groups},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mjrl_step_kernel_spec" if env._handle.kernel == "specialised" else "mjrl_step_kernel",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_env_step": bytes_per},
        }
        if double_buffered:
            line["double_buffered"] = double_buffered
        if world == 1 and not args.no_cpu_baseline:
            scatter = np.array([env.agents_action_index[a] for a in AGENTS])
            line["cpu_baseline"] = cpu_baseline(env._blob, scatter, language=bool(plugins))
        print(json.dumps(line), flush=True)
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
