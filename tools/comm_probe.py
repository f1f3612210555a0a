#!/usr/bin/env python
"""Host and device cost of the C-side RCCL hooks (rtx_comm_*) with ONE rank: per call, same stream vs alternating streams,
with and without group brackets.  Explains what the engine-scheduled data-parallel step pays per collective."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectorch_amd import _lib  # noqa: E402

L = _lib.lib()
torch.cuda.set_device(0)


def comm():
    buf = (C.c_uint8 * 128)()
    _lib.check(L.rtx_comm_unique_id(buf))
    h = C.c_void_p()
    _lib.check(L.rtx_comm_init(buf, 0, 1, C.byref(h)))
    return h


def timeit(label, fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-58s host %.1f us/call, with drain %.1f us/call" % (label, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)


c1, c2 = comm(), comm()
x = torch.zeros(12 << 20, dtype=torch.bfloat16, device="cuda")      # 24 MB: one n_items x 600 bf16 gradient image
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
p = C.c_void_p(x.data_ptr())
n = x.numel()
st = [C.c_void_p(s0.cuda_stream), C.c_void_p(s1.cuda_stream)]
k = [0]


def ar_same():
    _lib.check(L.rtx_comm_allreduce(c1, p, n, _lib.RTX_BF16, st[0]))


def ar_alt():
    k[0] ^= 1
    _lib.check(L.rtx_comm_allreduce(c1, p, n, _lib.RTX_BF16, st[k[0]]))


def ar_alt_two_comms():
    k[0] ^= 1
    _lib.check(L.rtx_comm_allreduce(c1 if k[0] else c2, p, n, _lib.RTX_BF16, st[k[0]]))


def ar_group():
    _lib.check(L.rtx_comm_group_start(c1))
    _lib.check(L.rtx_comm_allreduce(c1, p, n, _lib.RTX_BF16, st[0]))
    _lib.check(L.rtx_comm_group_end(c1))


def rs_ag():
    _lib.check(L.rtx_comm_reduce_scatter(c1, p, n, _lib.RTX_BF16, st[0]))
    _lib.check(L.rtx_comm_allgather(c1, p, n * 2, st[0]))


def small_ar():
    _lib.check(L.rtx_comm_allreduce(c1, p, 1024, _lib.RTX_BF16, st[0]))


def memcpy_d2d():
    torch.cuda.current_stream()
    y[:n // 8 * 7].copy_(x[:n // 8 * 7], non_blocking=True)


y = torch.empty_like(x)
timeit("all_reduce 24 MB, one stream", ar_same)
timeit("all_reduce 24 MB, alternating streams, one comm", ar_alt)
timeit("all_reduce 24 MB, alternating streams, a comm per stream", ar_alt_two_comms)
timeit("group { all_reduce 24 MB }", ar_group)
timeit("reduce_scatter + all_gather 24 MB", rs_ag)
timeit("all_reduce 2 KB", small_ar)
timeit("torch copy_ 21 MB d2d (what the emulation moves)", memcpy_d2d)
import torch.distributed as dist
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29555", rank=0, world_size=1)
timeit("torch.distributed all_reduce 24 MB (nccl backend)", lambda: dist.all_reduce(x))
dist.destroy_process_group()
