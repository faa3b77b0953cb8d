"""The N > 1 path on CPU: world_size-2 gloo process group, batch sharding, table broadcast from
rank 0, per-item results gathered back in batch order (audiotools_amd/dist.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.dont_write_bytecode = True
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import audiotools_amd as A
    from audiotools_amd import dist as adist, tables
    from oracle import restate
    from tests import synth

    r, w, device = adist.init(backend="gloo")
    assert (r, w) == (rank, world) and device.type == "cpu"
    B = 5
    x = synth.audio_batch(B, 2, 16000, seed=21, gaps=False, sample_rate=16000)
    lo, hi = adist.shard_range(B, rank, world)
    # shared table: only rank 0 builds it (other ranks would produce garbage on purpose)
    key = ("window", "hann", 512)
    built = []

    def builder():
        built.append(rank)
        return tables.window_np("hann", 512) if rank == 0 else np.full(512, np.nan, dtype=np.float32)

    win = adist.broadcast_table(key, builder, device)
    assert built == ([0] if rank == 0 else [])
    assert torch.equal(win, torch.from_numpy(tables.window_np("hann", 512)))
    # a tuple-valued table
    tup = adist.broadcast_table(("t", 1), lambda: (np.arange(6, dtype=np.int32).reshape(3, 2), np.ones(4, np.float32)), device)
    assert tup[0].shape == (3, 2) and tup[0].dtype == torch.int32 and float(tup[1].sum()) == 4.0
    # a whole configuration's tables in ONE packed broadcast (mixed dtypes, odd sizes, a tuple): only rank 0 builds them
    items = [(("p", 0), lambda: np.arange(7, dtype=np.float32) if rank == 0 else None),
             (("p", 1), lambda: (np.arange(5, dtype=np.int32), np.arange(3, dtype=np.uint32) + 7) if rank == 0 else None),
             (("p", 2), lambda: np.linspace(0, 1, 301).astype(np.float64).reshape(7, 43) if rank == 0 else None)]
    calls = {"n": 0}
    real = dist.broadcast

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    dist.broadcast = counting
    try:
        got = adist.broadcast_tables(items, device)
    finally:
        dist.broadcast = real
    assert calls["n"] == 1
    assert torch.equal(got[0], torch.arange(7, dtype=torch.float32))
    assert torch.equal(got[1][0], torch.arange(5, dtype=torch.int32)) and got[1][1].dtype == torch.uint32
    assert got[2].shape == (7, 43) and got[2].dtype == torch.float64 and float(got[2][-1, -1]) == 1.0
    assert tables.device_table(("p", 0), device, lambda: None) is got[0]          # installed in the cache
    # the data path: each rank processes its slab, no collective
    sig = A.AudioSignal(x[lo:hi].clone(), 16000)
    lufs_local = sig.loudness()
    mel_local = sig.mel_spectrogram(40)
    full = adist.gather_items(lufs_local, B)
    adist.barrier()
    if rank == 0:
        ref = restate.loudness(x, 16000)
        q.put(("lufs", float((full - ref).abs().max())))
        q.put(("mel_shape", tuple(mel_local.shape)))
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(2))
    assert got["lufs"] < 1e-4
    assert got["mel_shape"] == (3, 2, 40, 126)


def test_shard_range_partitions_batch():
    from audiotools_amd import dist as adist

    for B in (1, 5, 8, 512, 513):
        for W in (1, 2, 3, 8):
            spans = [adist.shard_range(B, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("config", ["north_star", "cfg4", "cfg5"])
def test_bench_spawns_its_own_ranks(config):
    """`python bench.py --gpus 2` without a torchrun environment starts 2 ranks itself, checks the
    world size, shards the batch and broadcasts the configuration's tables (gloo here; the same code
    runs RCCL on a GPU node).  --dry-run stops before the kernels."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--config", config],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"] == config and d["items_rank0"] * 2 == d["global_batch"]
    # a world size that does not match --gpus is an error, not a warning
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env2,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "--gpus 2" in (bad.stderr + bad.stdout)


def test_forced_single_rank_collectives_dry_run():
    """bench.py --force-nccl: ONE rank initialises the process group (gloo here, RCCL on a GPU box) and runs
    the multi-GPU set-up path on itself -- the packed table broadcast, a 100 MB bank broadcast, barriers -- so that the first
    8-GPU run is not the first time the collective library is loaded.  The tables installed through the collective path are
    the ones the local path builds."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MASTER_PORT"] = "29631"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run", "--force-nccl"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = d["rccl_contact"]
    assert d["n_gpus"] == 1 and d["backend"] == "gloo" and c["world_size"] == 1 and c["ir_bank_checksum_ok"] is True
    assert c["ir_bank_100MB_broadcast_ms"] > 0 and c["stft_mel_tables_broadcast_ms"] > 0 and c["barrier_ms"] >= 0
