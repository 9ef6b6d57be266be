"""The N>1 path of bench.py on CPU: world_size-2 gloo ranks shard the env batch with no data-path collective; the
only collectives are the barrier and the max-reduce of the timed region.  The check that matters for the physics:
a copy's trajectory is a function of its GLOBAL env id only, so the union of the two shards equals the
single-process batch bit for bit.  The copies are stepped by the product's own arithmetic: the DEVICE step source
(csrc/mjrl_step.h) in its CPU lane-emulation build (tests/emu/batch.py), with the episode-phase reset masks of bench.py;
on the GPU box the same property is tested through the C-ABI in test_gpu_parity.py::test_full_batch_properties.
test_bench_py_typed_plainly_starts_its_own_ranks runs bench.py's own control flow (rank children, rendezvous, barrier,
max over ranks, rank-0 line) from the bare command."""
import json
import subprocess
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ENV, STEPS = 4, 12


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def shard_run(first_env, n_env):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    entry.load_package()
    import bench
    from mjrl_amd import levels
    from tests.emu.batch import EmuBatch
    batch = EmuBatch(levels.level_path("two_agent.xml"), ["sender", "receiver"], n_env, language=True)
    acts = bench.action_stream(0, first_env, n_env, STEPS, 2, 9, 8)
    # episodes shortened to 8 steps so that the in-launch resets of bench.py's phase schedule are part of the run
    phase = (first_env + np.arange(n_env)) % 8
    obs, rew = np.zeros((n_env, 2, batch.obs_dim)), np.zeros((n_env, 2))
    term, trunc = np.zeros((n_env, 2), np.uint8), np.zeros((n_env, 2), np.uint8)
    for t in range(STEPS):
        batch.set_step_reset_mask((phase == t % 8).astype(np.uint8) if t else None)
        batch.step_batched(acts[t], obs, rew, term, trunc)
    return np.concatenate([np.stack([e.qpos for e in batch.envs]), obs.reshape(n_env, -1)], axis=1)


def worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = N_ENV // world
    dist.barrier()
    qpos = shard_run(rank * per, per)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # the timing reduction bench.py does
    gathered = [torch.zeros((per, qpos.shape[1]), dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(qpos))   # off the step path: only to compare here
    if rank == 0:
        result["qpos"] = torch.cat(gathered).numpy()
        result["max"] = float(t.item())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_reproduce_the_single_process_batch():
    manager = mp.Manager()
    result = manager.dict()
    mp.spawn(worker, args=(2, free_port(), result), nprocs=2, join=True)
    assert result["max"] == 2.0
    single = shard_run(0, N_ENV)
    assert np.array_equal(result["qpos"], single)


def test_action_stream_is_keyed_on_global_env_id():
    sys.path.insert(0, ROOT)
    import bench
    whole = bench.action_stream(3, 0, 8, 5, 2, 8, 8)
    assert np.array_equal(bench.action_stream(3, 4, 4, 5, 2, 8, 8), whole[:, 4:])
    assert whole.min() >= -1 and whole.max() <= 1 and not np.array_equal(whole[:, 0], whole[:, 1])
    # the CPU baseline's worker w steps global env id w on the same stream
    assert np.array_equal(bench.action_stream(3, 5, 1, 5, 2, 8, 8)[:, 0], whole[:, 5])
    # episode phases: a function of the global id, uniform inside every rank's shard
    phase = bench.phase_of(np.arange(8192), 4096)
    assert np.array_equal(phase[:4096], phase[4096:]) and np.array_equal(np.bincount(phase[:4096]), np.full(1024, 4))


def test_bench_py_typed_plainly_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: bench.py starts the two ranks itself, as fresh children, before
    it touches a GPU, and relays rank 0's line (here the ranks step the CPU emulation of the device source)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MJRL_BENCH_REHEARSAL"] = "cpu"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--preroll", "3",
           "--envs-per-gpu", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert "REHEARSAL" in line["data"] and line["config"]["preroll_steps"] == 3
    # whole-job rate: all ranks' copies over the slowest rank's time
    assert abs(line["value"] - 2 * 2 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    # a mismatch between --gpus and an inherited WORLD_SIZE is refused loudly
    env["WORLD_SIZE"] = "3"
    bad = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def test_config_one_cpu_leg_of_the_bench_line():
    """BASELINE configs[0] (single_agent.xml, one copy): the CPU leg of the bench's `configs[0]` entry runs here -- one
    thread of the oracle through the reference's loop shape -- and reports what it measured on."""
    import bench
    bench.entry.load_package()
    leg = bench.config_one_cpu(seconds=0.4)
    assert leg["cores"] == 1 and leg["kind"] in ("port", "reference") and leg["unit"] == "env-steps/s"
    assert leg["value"] > 1000 and "single_agent.xml" in leg["sample"]
