"""N>1 path on CPU: world_size-2 gloo.  Covers the one collective of the hot path (voice broadcast) and the
round-robin request sharding that bench.py / the serving loop use across GPUs."""
import os
import sys
import numpy as np
import pytest

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "qwen3-tts-rust_amd", "python"))
    from q3tts.dist import broadcast_voice, shard_requests
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)
    emb = rng.standard_normal(2048).astype(np.float32)
    codes = rng.integers(0, 2048, 62 * 16)
    ids = rng.integers(0, 151000, 24)
    if rank == 0:
        e, c, t = broadcast_voice(emb, codes, ids, src=0)
        e2, c2, t2 = broadcast_voice(emb, None, None, src=0)           # preset voice: embedding only
    else:
        e, c, t = broadcast_voice(None, None, None, src=0)
        e2, c2, t2 = broadcast_voice(None, None, None, src=0)
    mine = shard_requests(7, rank, world)
    # weak-scaling bookkeeping the bench does: max over ranks of the elapsed time
    tt = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    q.put((rank, np.array_equal(e, emb), np.array_equal(c, codes), np.array_equal(t, ids), np.array_equal(e2, emb), c2.size, t2.size,
           mine, float(tt.item())))
    dist.destroy_process_group()


def test_voice_broadcast_and_sharding_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_e, ok_c, ok_t, ok_e2, nc2, nt2, mine, tmax in res:
        assert ok_e and ok_c and ok_t and ok_e2 and nc2 == 0 and nt2 == 0 and tmax == 2.0
    assert res[0][7] == [0, 2, 4, 6] and res[1][7] == [1, 3, 5]


def test_single_process_passthrough():
    sys.path.insert(0, os.path.join(ROOT, "qwen3-tts-rust_amd", "python"))
    from q3tts.dist import broadcast_voice, shard_requests
    e, c, t = broadcast_voice(np.ones(2048, np.float32), [1, 2], None)
    assert e.shape == (2048,) and c.tolist() == [1, 2] and t.size == 0
    assert shard_requests(5, 0, 1) == [0, 1, 2, 3, 4]


def test_bench_launcher_and_rank_aggregation_world2():
    """bench.py --gpus 2 with no launcher in the environment must START two ranks (torch.distributed.run), and its whole-job value must be
    SUM(audio seconds over ranks) / MAX(elapsed over ranks).  Runs the real bench.py rank logic over gloo with a stub engine (no GPU)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-engine",
                        "--requests", "5", "--batch", "4", "--ragged"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    # every rank ran the same 5 ragged requests of the stub: frames = sum(max(1, int(12 * (0.5 + 0.5 * ((i * 7) % 11) / 10)))) per rank
    frames = sum(max(1, int(12 * (0.5 + 0.5 * ((i * 7) % 11) / 10.0))) for i in range(5))
    assert abs(out["audio_s_total"] - 2 * frames * 0.08) < 1e-9
    assert abs(out["value"] - out["audio_s_total"] / out["elapsed_s"]) < 1e-9
    assert out["elapsed_s"] >= 0.04          # rank 1's stub sleeps 40 ms: the MAX over ranks, not rank 0's 20 ms
    # the line describes its own ranks: what every rank generated and took, the world size the collective itself saw, backend, devices
    rk = out["ranks"]
    assert rk["rccl_world"] == 2 and rk["backend"] == "gloo" and len(rk["devices"]) == 2
    assert len(rk["per_rank_audio_s"]) == 2 and all(abs(a - frames * 0.08) < 1e-9 for a in rk["per_rank_audio_s"])
    assert abs(sum(rk["per_rank_audio_s"]) - out["audio_s_total"]) < 1e-9 and abs(max(rk["per_rank_elapsed_s"]) - out["elapsed_s"]) < 1e-9
    assert rk["per_rank_elapsed_s"][1] > rk["per_rank_elapsed_s"][0] * 0.9   # rank 1's stub is the slower one (40 vs 20 ms of sleep)


def test_bench_rejects_mismatched_world():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-engine"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)
