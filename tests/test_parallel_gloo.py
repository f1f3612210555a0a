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
    # sharded optimizer plan: the two big layers are reduce-scattered in equal blocks of padded rows, each rank's optimizer
    # callback gets ITS rows (with the summed gradient in place), the compute copy is all-gathered; small layers all-reduce
    prow = {0: (H + 1 + 127) // 128 * 128, 3: (I + 1 + 127) // 128 * 128}
    offs3, total3 = [], 0
    for k, sh in enumerate(shapes):
        offs3.append(total3)
        n = int(np.prod(sh)) if len(sh) == 1 else prow[k // 2] * sh[1] if k // 2 in prow else int(np.prod(sh))
        total3 += (n + 63) // 64 * 64
    ranges3 = [(offs3[2 * l], offs3[2 * l + 1] + int(np.prod(shapes[2 * l + 1]))) for l in range(4)]
    for comm in (torch.float32, torch.bfloat16):
        flat3 = torch.zeros(total3)
        shard_layers = {0: (H, I, prow[0]), 3: (I, H, prow[3])}
        red3 = GradAllReducer(flat3, ranges3, min_bucket_bytes=64 << 10, comm_dtype=comm, tensor_offsets=offs3, shard_layers=shard_layers)
        assert [b[0] for b in red3.buckets()] == [3, 1, 0], red3.buckets()
        shadows = {l: torch.zeros(shard_layers[l][2], shard_layers[l][1] + 7) for l in shard_layers}
        seen_rows, seen_small = [], []

        def adam_rows(layer, lo, hi):
            rows, cols, pr = shard_layers[layer]
            per = pr // world
            assert (lo, hi) == (rank * per, (rank + 1) * per)
            src = red3.flat16 if comm == torch.bfloat16 else red3.flat
            w0 = offs3[2 * layer]
            mine = src[w0 + lo * cols:w0 + min(hi, rows) * cols].float()
            want = sum((r + 1) * (layer + 1) for r in range(world))
            assert torch.all(mine == want), (rank, layer)
            b0 = offs3[2 * layer + 1]
            assert torch.all(src[b0:b0 + rows].float() == want)          # the bias is all-reduced in full
            shadows[layer][lo:hi] = float(rank + 1)                       # "updated rows" of the compute copy
            seen_rows.append(layer)

        def adam_small(lo, hi):
            seen_small.append((lo, hi))

        red3.adam, red3.adam_rows, red3.shadow = adam_small, adam_rows, (lambda layer: shadows[layer])
        for l in range(4):
            flat3[ranges3[l][0]:ranges3[l][1]] = 0.0
            w0, (rows, cols) = offs3[2 * l], shapes[2 * l]
            flat3[w0:w0 + rows * cols] = (rank + 1) * (l + 1)
            b0 = offs3[2 * l + 1]
            flat3[b0:b0 + rows] = (rank + 1) * (l + 1)
        for l in (3, 2, 1, 0):
            red3.on_layer(l)
        red3.wait()
        assert seen_rows == [3, 0] and seen_small == [(1, 3)], (seen_rows, seen_small)
        for l, shd in shadows.items():      # after the all-gather: block r of the compute copy comes from rank r
            per = shard_layers[l][2] // world
            for r in range(world):
                assert torch.all(shd[r * per:(r + 1) * per] == float(r + 1)), (rank, l, r)
        # gather_state: the master rows of every rank's shard end up everywhere
        masters = {l: torch.full((shard_layers[l][0], shard_layers[l][1]), -1.0) for l in shard_layers}
        for l in masters:
            lo, hi = red3.shard_rows_of(l)
            masters[l][lo:min(hi, shard_layers[l][0])] = float(10 + rank)
        red3.gather_state(lambda layer: [masters[layer]])
        for l, mt in masters.items():
            per = shard_layers[l][2] // world
            for r in range(world):
                blk = mt[r * per:min((r + 1) * per, shard_layers[l][0])]
                assert torch.all(blk == float(10 + r)), (rank, l, r)
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
