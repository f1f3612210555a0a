"""Data parallelism over the users of a batch: one process per MI355X, RCCL over xGMI.

The reference has no distributed code; this is the MI355X-native addition the north star
asks for.  The path shards naturally: users (CSR rows) are independent given the weights and the loss is a
mean over the batch (reference models.py:813-814), so the global gradient is the SUM of per-rank gradients
each already scaled by 1/B_global.  One exchange per step:

  * every rank runs forward+backward on its slice of the global batch (``shard_rows``);
  * as soon as the kernels producing layer l's gradients are enqueued (layers finish last-decoder-first) the
    engine calls back (``rtx_layer_cb``); the reducer records an event and issues ``all_reduce(SUM)`` of that
    layer's contiguous slice of the flat gradient buffer on a side HIP stream, so the big decoder bucket
    (dW4: 48 MB of the 98 MB at the ml-20m shape) is in flight while the remaining backward GEMMs run;
  * the compute stream waits for the side stream, then every rank applies the same fused Adam update
    (replicated weights stay bit-identical: same reduced gradient, same arithmetic).

Small layers are coalesced into one message (``min_bucket_bytes``).  Works with any torch.distributed
backend: "nccl" (= RCCL on ROCm) for device tensors, "gloo" for the CPU tests of the plan itself.
"""
import numpy as np
import torch
import torch.distributed as dist

__all__ = ["shard_rows", "GradAllReducer", "init_from_env", "attach"]


def shard_rows(n_rows, rank, world):
    """Contiguous, near-equal split of ``n_rows`` batch rows over ``world`` ranks: ``(start, end)`` of rank
    ``rank``.  The first ``n_rows % world`` ranks get one extra row; a rank may get an empty range."""
    base, rem = divmod(int(n_rows), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class GradAllReducer:
    """Bucketed all-reduce of a flat gradient buffer, overlapped with the backward pass.

    Parameters
    ----------
    flat : 1-D tensor holding every gradient (layer l occupies ``layer_ranges[l] = (start, end)``).
    layer_ranges : list of (start, end), layer order = parameter order (encoder first).
    min_bucket_bytes : layers are coalesced (in completion order, i.e. last layer first) until a bucket
        reaches this size; the big n_items x hidden layers go out on their own.
    """

    def __init__(self, flat, layer_ranges, group=None, min_bucket_bytes=4 << 20):
        self.flat = flat
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.layer_ranges = list(layer_ranges)
        self.on_device = flat.is_cuda
        self.side = torch.cuda.Stream() if self.on_device else None
        # plan: walking layers in completion order, a bucket closes at layer l when it is big enough or l == 0
        self.close_at = {}
        n = len(self.layer_ranges)
        hi = n - 1
        size = 0
        for l in range(n - 1, -1, -1):
            size += (self.layer_ranges[l][1] - self.layer_ranges[l][0]) * flat.element_size()
            if size >= min_bucket_bytes or l == 0:
                self.close_at[l] = (self.layer_ranges[l][0], self.layer_ranges[hi][1])
                hi = l - 1
                size = 0
        self.launched = []
        self._batch_cache = {}

    def buckets(self):
        """(closing layer, start, end) in launch order -- for tests and the design notes."""
        return [(l, *self.close_at[l]) for l in sorted(self.close_at, reverse=True)]

    def on_layer(self, layer, _user=None):
        """Host callback from rtx_engine_loss_grads: gradients of ``layer`` are enqueued on the compute stream."""
        rng = self.close_at.get(int(layer))
        if rng is None:
            return
        view = self.flat[rng[0]:rng[1]]
        if self.on_device:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        self.launched.append(int(layer))

    def wait(self):
        """Make the compute stream wait for every bucket launched since the last wait()."""
        if self.on_device:
            torch.cuda.current_stream().wait_stream(self.side)
        self.launched = []

    def global_batch(self, local_batch):
        """Sum of the ranks' local batch sizes (cached per local size: equal every step but the ragged last)."""
        if self.world == 1:
            return local_batch
        t = torch.tensor([float(local_batch)], device=self.flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(round(t.item()))

    def reduce_scalar(self, t):
        """Sum a 1-element tensor over the ranks and return it as a float (losses are already scaled by
        1/B_global, so the sum is the global mean)."""
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item())


def init_from_env(backend=None):
    """torch.distributed init from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (what
    ``python -m torch.distributed.run`` exports).  backend None -> "nccl" (RCCL) with a HIP device, else gloo."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def attach(model, group=None, min_bucket_bytes=4 << 20, fixed_global_batch=None):
    """Turn a :class:`rectorch_amd.models.AETrainer` into a data-parallel replica: broadcasts rank 0's
    parameters, then every ``train_batch`` all-reduces the gradients as described above.  Each rank must feed
    ITS slice of the global batch (see ``shard_rows``)."""
    st, params, m, v = model._ensure_train_state()
    for p in params:
        dist.broadcast(p.data, src=0, group=group)
    model.network._rtx_shadow_versions.clear()      # parameters changed under the engines: refresh the shadows
    red = GradAllReducer(st.flat_grads, st.layer_ranges, group, min_bucket_bytes)
    if fixed_global_batch is not None:
        red.global_batch = lambda local, _g=int(fixed_global_batch): _g
    st.reducer = red
    return red
