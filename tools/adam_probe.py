#!/usr/bin/env python
"""k_adam on row shards of the two big matrices (what a rank of the sharded optimizer runs) timed alone with HIP events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectorch_amd.utils import hash_state_dict  # noqa: E402
from rectorch_amd.nets import MultiVAE_net  # noqa: E402
from rectorch_amd.models import MultiVAE  # noqa: E402

I, H, L = 20108, 600, 200
net = MultiVAE_net([L, H, I], dropout=0.5)
net.load_state_dict({k: torch.from_numpy(v) for k, v in hash_state_dict([I, H, L], [L, H, I], "vae", 1).items()})
model = MultiVAE(net, numerics="bf16")
st, params, m, v = model._ensure_train_state()
eng = net.rtx_engine("bf16", 500, train_buffers=(st.grads, m, v))
g16 = torch.zeros(st.flat_grads.numel(), dtype=torch.bfloat16, device="cuda")
step = eng._step(beta=0.1, lam=0.0, inv_batch=1 / 500, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1)


def t(label, fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-60s %.1f us" % (label, e0.elapsed_time(e1) / n * 1e3), flush=True)


base = g16.data_ptr()
offs = st.tensor_offsets
for layer, prow in ((3, 20224), (0, 640)):
    for world in (1, 2, 4, 8):
        per = prow // world
        t("layer %d rows [0, %d) of %d (world %d), bias too" % (layer, per, prow, world),
          lambda: eng.apply_adam_rows(step, layer, 0, per, True, base + 2 * offs[2 * layer], base + 2 * offs[2 * layer + 1]))
        t("layer %d rows [0, %d) of %d (world %d), no bias" % (layer, per, prow, world),
          lambda: eng.apply_adam_rows(step, layer, 0, per, False, base + 2 * offs[2 * layer], None))
t("layers 1..2 (hidden, with transposed copies)", lambda: eng.apply_adam_layers(step, 1, 3, [base + 2 * o for o in offs]))
t("all layers", lambda: eng.apply_adam_layers(step, 0, 4, [base + 2 * o for o in offs]))
