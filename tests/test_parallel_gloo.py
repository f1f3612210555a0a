"""CPU, world_size 2, gloo: the data-parallel plan (row sharding + bucketed gradient all-reduce driven by the
engine's per-layer callback order).  The HIP kernels are not involved; the same GradAllReducer code runs
with RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rectorch_amd.parallel import GradAllReducer, shard_rows
    # flat gradient layout of a [I,H,L]/[L,H,I] VAE: 4 layers, each W then b, 64-float aligned tensors
    I, H, L = 2000, 96, 32
    shapes = [(H, I), (H,), (2 * L, H), (2 * L,), (H, L), (H,), (I, H), (I,)]
    offs, total = [], 0
    for s in shapes:
        offs.append(total)
        total += (int(np.prod(s)) + 63) // 64 * 64
    ranges = [(offs[2 * l], offs[2 * l + 1] + int(np.prod(shapes[2 * l + 1]))) for l in range(4)]
    flat = torch.zeros(total)
    red = GradAllReducer(flat, ranges, min_bucket_bytes=64 << 10)
    # plan: the two big layers go alone (first the decoder output layer), the small ones are coalesced
    bk = red.buckets()
    assert [b[0] for b in bk] == [3, 0] or [b[0] for b in bk] == [3, 1, 0], bk
    covered = sorted((s, e) for _, s, e in bk)
    assert covered[0][0] == 0 and covered[-1][1] == ranges[-1][1]
    for a, b in zip(covered[:-1], covered[1:]):
        assert a[1] <= b[0]
    # two "steps": per-rank gradients = rank-dependent constants; layers complete last-first (engine order)
    for step in range(2):
        for l in range(4):
            flat[ranges[l][0]:ranges[l][1]] = (rank + 1) * (l + 1) + step
        for l in (3, 2, 1, 0):
            red.on_layer(l)
        red.wait()
        for l in range(4):
            want = sum((r + 1) * (l + 1) + step for r in range(world))
            assert torch.all(flat[ranges[l][0]:ranges[l][1]] == want), (rank, l)
    # bf16 exchange + per-bucket optimizer callback: same sums (small integers are exact in bf16), every layer handed
    # to the optimizer exactly once, bucket by bucket in completion order, each AFTER its own exchange
    flat2 = torch.zeros(total)
    red2 = GradAllReducer(flat2, ranges, min_bucket_bytes=64 << 10, comm_dtype=torch.bfloat16, tensor_offsets=offs)
    calls = []

    def adam(lo, hi):
        for l in range(lo, hi):
            want = sum((r + 1) * (l + 1) for r in range(world))
            assert torch.all(red2.flat16[ranges[l][0]:ranges[l][1]].float() == want), (rank, l)
        calls.append((lo, hi))

    red2.adam = adam
    for l in range(4):
        flat2[ranges[l][0]:ranges[l][1]] = (rank + 1) * (l + 1)
    for l in (3, 2, 1, 0):
        red2.on_layer(l)
    red2.wait()
    assert [c[0] for c in calls] == [b[0] for b in red2.buckets()], calls
    assert sorted(l for lo, hi in calls for l in range(lo, hi)) == [0, 1, 2, 3], calls
    ptrs = red2.grads16_ptrs()
    assert len(ptrs) == 8 and ptrs[0] == red2.flat16.data_ptr() and ptrs[2] - ptrs[0] == 2 * offs[2]
    assert red.global_batch(250 + rank) == sum(250 + r for r in range(world))
    assert abs(red.reduce_scalar(torch.tensor([0.5 * (rank + 1)])) - sum(0.5 * (r + 1) for r in range(world))) < 1e-6
    # row sharding: every rank takes its slice of the same global permutation -> disjoint cover
    np.random.seed(7)
    perm = np.random.permutation(501)
    s, e = shard_rows(len(perm), rank, world)
    mine = torch.zeros(501)
    mine[perm[s:e]] = 1
    dist.all_reduce(mine)
    assert torch.all(mine == 1)
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_dp_plan_world2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))
