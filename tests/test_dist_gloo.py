"""N > 1 without GPUs: the replica launcher logic of bench.py on 2 gloo ranks (CPU).

The path shards by sequence (one block pool + one block manager per GPU, no exchange step), so what has to be right
for N > 1 is (a) the partition of sequences over ranks, (b) independent per-rank block managers, and (c) the
barrier / max-over-ranks timing and whole-job aggregation bench.py prints."""
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
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_seqs, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _load_pkg
    pkv = _load_pkg.load()
    mine = [s for s in range(n_seqs) if s % world == rank]             # sequence s -> GPU s mod N
    c = pkv.PagedKVCache(n_blocks=64, block_size=128, n_seq_max=len(mine), n_layer=0)
    tabs = {}
    for i, s in enumerate(mine):                                       # prefill 300 + s tokens, then 5 decode steps
        n = 300 + s
        c.find_slot([i] * n, list(range(n)))
    for step in range(5):
        c.find_slot(list(range(len(mine))), [300 + s + step for s in mine])
    for i, s in enumerate(mine):
        tabs[s] = (c.seq_blocks(i).tolist(), c.seq_len(i))
    # timing protocol of bench.py: barrier, local time, MAX over ranks, whole-job aggregate
    dist.barrier()
    local_dt = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(local_dt, op=dist.ReduceOp.MAX)
    tok = torch.tensor([len(mine) * 5], dtype=torch.float64)
    dist.all_reduce(tok, op=dist.ReduceOp.SUM)
    q.put((rank, tabs, float(local_dt.item()), float(tok.item()), c.get_n_tokens()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas():
    world, n_seqs = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_seqs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = {}
    for rank, tabs, dt, tok, n_tok in res:
        assert dt == pytest.approx(0.020)          # max over ranks
        assert tok == n_seqs * 5                    # whole-job aggregate
        assert n_tok == sum(300 + s + 5 for s in tabs)
        for s, (blocks, ln) in tabs.items():
            assert s % world == rank and ln == 300 + s + 5
            assert len(blocks) == -(-ln // 128) and len(set(blocks)) == len(blocks)
            seen[s] = blocks
        # pools are per rank: block ids only need to be unique inside a rank
        flat = [b for blocks, _ in tabs.values() for b in blocks]
        assert len(flat) == len(set(flat))
    assert sorted(seen) == list(range(n_seqs))


def _run_bench(extra_env, *argv):
    import json
    import subprocess
    env = dict(os.environ, PKV_BENCH_CPU_REHEARSAL="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the process starts 2 ranks itself (children, gloo on CPU in this rehearsal), relays ONE JSON line
    that says n_gpus 2 with the whole-job aggregate — never again a silent one-GPU run"""
    rc, lines, err = _run_bench({}, "--gpus", "2", "--steps", "40", "--warmup", "4", "--n-seq", "3")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["steps"] == 40 and ln["scaling"] == "weak" and "REHEARSAL" in ln["data"]
    assert ln["value"] == pytest.approx(2 * 3 * 40 / (ln["ms_per_step"] * 40 / 1e3), rel=1e-3)       # whole job: both ranks' sequences over the max-over-ranks time


def test_bench_fails_when_a_rank_fails():
    rc, lines, _ = _run_bench({"PKV_BENCH_REHEARSAL_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "5", "--warmup", "1")
    assert rc != 0 and lines == []


def test_bench_refuses_a_mismatched_world():
    """under an outside launcher WORLD_SIZE must equal --gpus (a 1-rank world with --gpus 2 is an error, not a one-GPU result)"""
    import subprocess
    env = dict(os.environ, PKV_BENCH_CPU_REHEARSAL="1", WORLD_SIZE="1", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
