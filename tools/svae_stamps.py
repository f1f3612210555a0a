#!/usr/bin/env python
"""Where a time step of the SVAE recurrences goes (forward k_sv_gru_fwd_ks; RTX_SVAE_GRU_KS=0: k_sv_gru_fwd_rows; backward
k_sv_gru_bwd_ks, in one training step).  Forward: shader-clock stamps of steps 8..11 taken by
thread 0 -- step start | mat-vec done | past barrier 1 | gate phase done (the next step's start closes barrier 2)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectorch_amd import _lib  # noqa: E402
from rectorch_amd.nets import SVAE_net  # noqa: E402
from rectorch_amd.models import SVAE  # noqa: E402

torch.manual_seed(0)
I = 3416
net = SVAE_net(n_items=I, embed_size=256, rnn_size=200, dec_dims=[64, 150, I], enc_dims=[200, 150, 64])
model = SVAE(net.to("cuda"), beta=0.2, anneal_steps=20000)
x = torch.randint(0, I, (1, 160))
stamps = torch.zeros(32, dtype=torch.int64, device="cuda")
lib = C.CDLL(_lib.LIB_PATH)
for it in range(3):
    model.predict(x)
lib.rtxdbg_svae_set_stamps(C.c_void_p(stamps.data_ptr()))
model.predict(x)
torch.cuda.synchronize()
lib.rtxdbg_svae_set_stamps(None)
s = stamps.cpu().numpy()[:16].reshape(4, 4)
print("cycles (shader clock) per phase, steps 8..11:   mat-vec | barrier 1 | gate phase | barrier 2 + loop")
for k in range(3):
    print("  step %d: %6d %6d %6d %6d   = %d cycles" % (8 + k, s[k, 1] - s[k, 0], s[k, 2] - s[k, 1], s[k, 3] - s[k, 2], s[k + 1, 0] - s[k, 3], s[k + 1, 0] - s[k, 0]))

# the backward recurrence (k_sv_gru_bwd_ks) in one training step: gate gradients | barrier 1 | transposed mat-vec + lane reduce-scatter | barrier 2 + waves' sums
y = torch.zeros(1, 160, I)
y[0, torch.arange(160), torch.randint(0, I, (160,))] = 1.0
for it in range(2):
    model.train_batch(x.to("cuda"), y.to("cuda"))
stamps.zero_()
lib.rtxdbg_svae_set_stamps(C.c_void_p(stamps.data_ptr()))
model.train_batch(x.to("cuda"), y.to("cuda"))
torch.cuda.synchronize()
lib.rtxdbg_svae_set_stamps(None)
b = stamps.cpu().numpy()[16:].reshape(4, 4)
if b.any():
    print("backward, cycles per phase:   gate gradients | barrier 1 | mat-vec^T + reduce | barrier 2 + sums + loop")
    for k in range(3):
        print("  step T-%d: %6d %6d %6d %6d   = %d cycles" % (9 + k, b[k, 1] - b[k, 0], b[k, 2] - b[k, 1], b[k, 3] - b[k, 2], b[k + 1, 0] - b[k, 3], b[k + 1, 0] - b[k, 0]))
