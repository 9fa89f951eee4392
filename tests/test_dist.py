"""N>1 path on CPU: world_size-2 gloo processes exercise the scene sharding and the fused
[payload | count] all-reduce of lidar_vision_vqa_amd.dist; plus the reference's own style of test
(training-test/utils/test_distributed.py:60-79: monkeypatched init_process_group)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from lidar_vision_vqa_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_scenes, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = D.init_dist_if_needed(backend="gloo")
    assert (r, lr, w) == (rank, rank, world) and D.is_dist()
    mine = D.shard_scenes(n_scenes, rank, world)
    d = 8
    # "fused tokens" of scene i: constant i+1 over [n=3, d]
    per_scene = torch.stack([torch.full((3, d), float(i + 1)) for i in mine]) if mine else torch.zeros(0, 3, d)
    buf = torch.zeros(d + 1)
    D.reduce_step(per_scene, buf)
    mean = D.all_reduce_mean(torch.tensor([float(len(mine))]), 1.0)
    mx = D.max_over_ranks(float(rank + 1))
    # head attached (BASELINE configs[3] / [4]): loss of scene i = i + 1, its answer logits constant i + 1 over [n_answer = 2, V = 3]
    hb = torch.zeros(2 + 2 * 3)
    if mine:
        D.reduce_head_step(torch.tensor(sum(i + 1.0 for i in mine) / len(mine)), torch.stack([torch.full((2, 3), float(i + 1)) for i in mine]), hb)
    else:
        D.reduce_head_step(torch.tensor(0.0), torch.zeros(0, 2, 3), hb)
    D.barrier()
    q.put((rank, mine, buf.tolist(), float(mean), mx, hb.tolist()))
    D.finalize()


@pytest.mark.parametrize("world,n_scenes", [(2, 5), (2, 8)])
def test_scene_sharding_and_fused_allreduce_gloo(world, n_scenes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_scenes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(i for _, mine, *_ in res for i in mine)
    assert owned == list(range(n_scenes))                       # every scene exactly once, no exchange needed
    expect_sum = 3.0 * sum(i + 1 for i in range(n_scenes))     # sum over scenes and tokens
    for rank, mine, buf, mean, mx, hb in res:
        tot = float(sum(i + 1 for i in range(n_scenes)))
        assert abs(hb[0] - tot) < 1e-4 and hb[1] == float(n_scenes) and all(abs(v - tot) < 1e-4 for v in hb[2:])     # [loss sum | scenes | logit sums]
        assert buf[-1] == float(n_scenes)                       # fused count
        assert all(abs(v - expect_sum) < 1e-4 for v in buf[:-1])
        assert abs(mean - n_scenes / world) < 1e-6              # all_reduce(sum)/count (commu_utils.py:148-168 average=True)
        assert mx == float(world)


def test_single_process_is_a_noop(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    called = []
    monkeypatch.setattr(torch.distributed, "init_process_group", lambda *a, **k: called.append(1))
    assert D.init_dist_if_needed() == (0, 0, 1)
    assert not called and not D.is_dist()
    buf = torch.zeros(5)
    D.reduce_step(torch.ones(2, 3, 4), buf)
    assert buf.tolist() == [6.0, 6.0, 6.0, 6.0, 2.0]
    assert D.max_over_ranks(1.5) == 1.5


def test_init_called_iff_world_gt_1(monkeypatch):
    """Reference behaviour (distributed.py:15-21): init_process_group('nccl'|'gloo', 'env://') iff WORLD_SIZE > 1."""
    calls = []
    monkeypatch.setenv("WORLD_SIZE", "4"); monkeypatch.setenv("RANK", "2"); monkeypatch.setenv("LOCAL_RANK", "2")
    monkeypatch.setattr(torch.distributed, "is_initialized", lambda: False)
    monkeypatch.setattr(torch.distributed, "init_process_group", lambda **k: calls.append(k))
    assert D.init_dist_if_needed(backend="gloo") == (2, 2, 4)
    assert calls and calls[0]["init_method"] == "env://" and calls[0]["world_size"] == 4 and calls[0]["rank"] == 2
