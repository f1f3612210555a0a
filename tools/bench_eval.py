#!/usr/bin/env python
"""Validation throughput (reference rectorch/evaluation.py:100-106 -> metrics.py:136-147, 187-196) on ml-20m-shaped
held-out users: users/s of

  evaluate_device   predict (HIP forward on the resident sparse rows) + device top-k / nDCG / Recall kernel, metrics only cross PCIe
  evaluate          rectorch-style: predict -> D2H of the [B, n_items] scores -> host argpartition (rectorch_amd.metrics)
  cpu_baseline      the reference's own op sequence on this box's host cores (oracle/rectorch_cpu.py forward in float32 +
                    host metrics), bounded sample

for both predict numerics (float32 parity mode = the default of predict(); bf16 = the training numerics).  One JSON line with a
`roofline` for the device path: SURVEY 8d's predict forward = 49.0 MFLOP per user against the dense MFMA peak of the numerics.

    python tools/bench_eval.py [users=10000] [batch=500]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rectorch_amd.utils import synth_interactions, hash_state_dict          # noqa: E402
from rectorch_amd.utils.synth import split_heldout                           # noqa: E402
from rectorch_amd.nets import MultiVAE_net                                   # noqa: E402
from rectorch_amd.models import MultiVAE                                     # noqa: E402
from rectorch_amd.samplers import DataSampler                                # noqa: E402
from rectorch_amd.evaluation import evaluate, evaluate_host, evaluate_device   # noqa: E402
from rectorch_amd.metrics import Metrics                                     # noqa: E402

PEAK_TF = {"fp32": 157.3, "bf16": 2500.0}     # v_mfma_f32_32x32x2_f32 / dense bf16 MFMA (MI355X_MICROARCH.md)


def timed(fn, reps):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r


def main():
    U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    I, H, L = 20108, 600, 200
    X = synth_interactions(U, I, seed=7)
    tr, te = split_heldout(X, 0.2, seed=1)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 5, bias_std=0.05)
    mets = ["ndcg@100", "recall@50"]
    flops_user = 2.0 * (I * H + H * 2 * L + L * H + H * I)      # = 49.0 MFLOP
    smp = DataSampler(tr, te, batch_size=B, shuffle=False)
    out = {"metric": "MultiVAE evaluate() users/sec on ml-20m-shaped held-out users (nDCG@100, Recall@50)", "users": U, "batch": B,
           "metrics": mets, "flops_per_user": flops_user}
    ref = None
    for numerics in ("fp32", "bf16"):
        net = MultiVAE_net([L, H, I])
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        model = MultiVAE(net, predict_numerics=numerics)
        evaluate_device(model, smp, mets)           # warm-up (engine creation, compute copies)
        t_dev, d = timed(lambda: evaluate_device(model, smp, mets), 3)
        t_host, h = timed(lambda: evaluate_host(model, smp, mets), 1)
        same = all(np.allclose(d[m], h[m], rtol=1e-12, equal_nan=True) for m in mets)
        if ref is None:
            ref = d
        if numerics == "bf16":
            out_bf16_metrics = d
        tf = flops_user * U / t_dev / 1e12
        out[numerics] = {"evaluate_device_s": t_dev, "users_per_s_device": U / t_dev, "evaluate_host_metrics_s": t_host,
                         "users_per_s_host_metrics": U / t_host, "device_equals_host_metrics": bool(same),
                         "mean_ndcg@100": float(np.mean(d["ndcg@100"])), "mean_recall@50": float(np.mean(d["recall@50"])),
                         "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TF[numerics], "unit": "TFLOP/s",
                                      "frac": tf / PEAK_TF[numerics],
                                      "what": "predict forward 49.0 MFLOP/user over the whole evaluate_device() time (gather, forward, "
                                              "-inf masking, top-k + metrics kernel, one D2H of the metrics)"}}
        del model, net
    out["value"] = out["bf16"]["users_per_s_device"]
    out["unit"] = "users/s"
    # the same held-out users in batches of 2000 (the sampler's batch size is the caller's choice; per batch the path costs three C
    # calls and a handful of tensor allocations on the host, which a 500-user batch does not amortise)
    smp_big = DataSampler(tr, te, batch_size=2000, shuffle=False)
    net = MultiVAE_net([L, H, I])
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model = MultiVAE(net, predict_numerics="bf16")
    evaluate_device(model, smp_big, mets)
    t_big, d_big = timed(lambda: evaluate_device(model, smp_big, mets), 3)
    out["bf16_batch_2000"] = {"evaluate_device_s": t_big, "users_per_s_device": U / t_big,
                              "same_metrics_as_batch_500": bool(all(np.allclose(d_big[m], out_bf16_metrics[m], rtol=1e-12, equal_nan=True) for m in mets))}
    del model, net
    # the reference's op sequence on the host cores: dense batch -> float32 forward (oracle/rectorch_cpu.py) -> -inf at the train
    # items -> Metrics.compute (argpartition); bounded sample of 2 batches after one warm-up batch
    from oracle.rectorch_cpu import CpuNet
    import psutil
    cores = psutil.cpu_count(logical=False) or os.cpu_count()
    cpu = {}
    for threads in (cores, 8):
        torch.set_num_threads(threads)
        cnet = CpuNet([I, H, L], [L, H, I], "vae", 0.5)
        cnet.eval()
        cnet.load_numpy([sd[k] for k in ("enc_layers.0.weight", "enc_layers.0.bias", "enc_layers.1.weight", "enc_layers.1.bias",
                                         "dec_layers.0.weight", "dec_layers.0.bias", "dec_layers.1.weight", "dec_layers.1.bias")])
        ts = []
        for b in range(3):
            rows = slice(b * B, (b + 1) * B)
            t0 = time.perf_counter()
            x = torch.from_numpy(np.asarray(tr[rows].toarray(), dtype=np.float32))
            with torch.no_grad():
                y = cnet.predict(x) if hasattr(cnet, "predict") else cnet.forward(x)[0]
            y = y.numpy().copy()
            y[x.numpy() != 0] = -np.inf
            Metrics.compute(y, np.asarray(te[rows].toarray(), dtype=np.float32), mets)
            ts.append(time.perf_counter() - t0)
        cpu[threads] = B / float(np.median(ts[1:]))
    best = max(cpu, key=cpu.get)
    out["cpu_baseline"] = {"value": cpu[best], "unit": "users/s", "cores": int(best), "cores_physical": int(cores), "kind": "port",
                           "users_per_s_by_threads": {str(k): v for k, v in cpu.items()},
                           "sample": "2 batches of %d users after 1 warm-up: densify, float32 forward (oracle/rectorch_cpu.py), -inf masking, "
                                     "Metrics.compute; torch %s CPU" % (B, torch.__version__)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
