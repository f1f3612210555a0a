"""Where evaluate_device()'s wall time goes on the host side (round 6): the sampler's row batches, the engine look-up, ONE rtx_engine_evaluate_topk
call (when it returns vs when the GPU is done), the device -> host copy -- against the whole call.   python tools/eval_host_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from rectorch_amd.utils import synth_interactions, hash_state_dict
from rectorch_amd.utils.synth import split_heldout
from rectorch_amd.nets import MultiVAE_net
from rectorch_amd.models import MultiVAE
from rectorch_amd.samplers import DataSampler
from rectorch_amd.evaluation import evaluate_device
U, B = 10000, 500
I, H, L = 20108, 600, 200
X = synth_interactions(U, I, seed=7)
tr, te = split_heldout(X, 0.2, seed=1)
sd = hash_state_dict([I, H, L], [L, H, I], "vae", 5, bias_std=0.05)
smp = DataSampler(tr, te, batch_size=B, shuffle=False)
net = MultiVAE_net([L, H, I]); net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
model = MultiVAE(net, predict_numerics="bf16")
mets = ["ndcg@100", "recall@50"]
evaluate_device(model, smp, mets)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T(); batches = list(smp.iter_rows()); t1 = T()
    eng = model._predict_engine(B); t2 = T()
    offsets = np.concatenate([[0], np.cumsum([len(rb) for rb in batches])])
    rows = torch.cat([rb.rows for rb in batches]); t3 = T()
    t3a = time.perf_counter()
    dn, dr = eng.evaluate_topk(batches[0].tr, batches[0].te, rows, offsets, [50, 100]); t3b = time.perf_counter()
    t4 = T()
    a, b = dn.cpu().numpy(), dr.cpu().numpy(); t5 = T()
    t6 = T(); evaluate_device(model, smp, mets); t7 = T()
    print("iter_rows %.0f us | engine %.0f | cat %.0f | C call returns after %.0f, GPU done after %.0f | D2H %.0f | whole evaluate_device %.0f" % tuple(x * 1e6 for x in (t1 - t0, t2 - t1, t3 - t2, t3b - t3a, t4 - t3, t5 - t4, t7 - t6)))
