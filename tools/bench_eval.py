#!/usr/bin/env python
"""Validation throughput: rectorch-style evaluate() (predict -> D2H of [B, n_items] scores -> host argpartition) vs
evaluate_device() (device top-k + metrics), ml-20m-shaped held-out users.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectorch_amd.utils import synth_interactions, hash_state_dict          # noqa: E402
from rectorch_amd.utils.synth import split_heldout                           # noqa: E402
from rectorch_amd.nets import MultiVAE_net                                   # noqa: E402
from rectorch_amd.models import MultiVAE                                     # noqa: E402
from rectorch_amd.samplers import DataSampler                                # noqa: E402
from rectorch_amd.evaluation import evaluate, evaluate_device                # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
I, H, L = 20108, 600, 200
X = synth_interactions(U, I, seed=7)
tr, te = split_heldout(X, 0.2, seed=1)
net = MultiVAE_net([L, H, I])
net.load_state_dict({k: torch.from_numpy(v) for k, v in hash_state_dict([I, H, L], [L, H, I], "vae", 5, bias_std=0.05).items()})
model = MultiVAE(net)
smp = DataSampler(tr, te, batch_size=500, shuffle=False)
mets = ["ndcg@100", "recall@50"]
evaluate_device(model, smp, mets)           # warm-up (engine creation, shadows)
torch.cuda.synchronize()
t0 = time.time(); d = evaluate_device(model, smp, mets); torch.cuda.synchronize(); t_dev = time.time() - t0
t0 = time.time(); h = evaluate(model, smp, mets); torch.cuda.synchronize(); t_host = time.time() - t0
same = all(np.allclose(d[m], h[m], rtol=1e-12, equal_nan=True) for m in mets)
print(json.dumps({"users": U, "metrics": mets, "evaluate_device_s": t_dev, "evaluate_host_s": t_host,
                  "users_per_s_device": U / t_dev, "users_per_s_host": U / t_host, "identical": bool(same)}))
