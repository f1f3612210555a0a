#!/usr/bin/env python
"""Headline benchmark: MultiVAE training users/sec on ml-20m-shaped synthetic data (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one MultiVAE.train_batch on one batch of 500 users per GPU: sparse-row gather -> forward ->
multinomial + beta-KL loss -> backward -> (RCCL all-reduce of the gradients when N > 1) -> fused Adam, through
the same ``_fused_step`` that ``MultiVAE.train_epoch`` drives with a device-resident ``DataSampler``.
Workload = BASELINE.json configs[1]: MultiVAE [20108, 600, 200], B = 500 per GPU, bf16 MFMA operands with f32
accumulation / f32 master weights + Adam, dropout 0.5, beta 0.2 annealed over 100 000 steps, lr 1e-3, synthetic
CSR 116 677 x 20 108, inputs resident in HBM before the timed region.  Weak scaling: per-GPU batch
fixed, global batch = 500 * N.

Rank 0 prints ONE JSON line with `roofline` (the dominant kernel = fused Adam, HBM-bound, timed live with HIP
events on the compute stream) and `cpu_baseline` (the oracle's torch-CPU restatement of the reference trainer,
timed on this box's host cores on a bounded sample of the same workload; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_TBS = 8.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=500, help="users per GPU per step")
    ap.add_argument("--numerics", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--users", type=int, default=116677)
    ap.add_argument("--items", type=int, default=20108)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cond-dim", type=int, default=0,
                    help="measure the conditioned variant (CMultiVAE, SURVEY 8f-4): this many condition columns are "
                         "appended to every input row, the target stays the item row; not the headline workload")
    ap.add_argument("--force-dp", action="store_true",
                    help="exercise the data-parallel code path (RCCL all-reduce + split step) even with one rank")
    return ap.parse_args()


def _cpu_run(X, dims, batch, seconds, threads):
    from oracle.rectorch_cpu import CpuNet, CpuTrainer, densify_batch
    torch.set_num_threads(threads)
    I, H, L = dims
    net = CpuNet([I, H, L], [L, H, I], "vae", 0.5)
    tr = CpuTrainer(net, beta=0.2, anneal_steps=100000, lr=1e-3)
    rng = np.random.default_rng(0)
    perm = rng.permutation(X.shape[0])
    t_all, t_smp, t_start, i = [], [], time.time(), 0
    while True:
        idx = perm[(i * batch) % (len(perm) - batch):][:batch]
        t0 = time.time()
        x = densify_batch(X, list(idx))           # DataSampler.__iter__ work (samplers.py:99-100)
        t1 = time.time()
        tr.train_batch(x)                         # MultiVAE.train_batch (models.py:817-835)
        t2 = time.time()
        if i >= 2:                                # 2 warm-up steps
            t_all.append(t2 - t0)
            t_smp.append(t1 - t0)
        i += 1
        if (time.time() - t_start > seconds and len(t_all) >= 3) or len(t_all) >= 40:
            break
    return float(np.median(t_all)), float(np.median(t_smp)), len(t_all)


def cpu_baseline(X, dims, batch, seconds):
    """the reference trainer's op sequence on this box's host cores (oracle/rectorch_cpu.py), bounded sample:
    all physical cores (the reported baseline) and 8 threads (comparable with the survey container's probe)"""
    import psutil
    cores = psutil.cpu_count(logical=False) or os.cpu_count()
    per, smp, n = _cpu_run(X, dims, batch, seconds, cores)
    per8, smp8, n8 = _cpu_run(X, dims, batch, seconds * 0.6, 8)
    best = min(per, per8)
    return {"value": batch / best, "unit": "users/s", "cores": int(cores if per <= per8 else 8), "kind": "port",
            "sample": "%d steps (all %d physical cores) + %d steps (8 threads) of B=%d, sampler densify + train_batch, "
                      "2 warm-up, median; torch %s CPU; faster of the two reported" % (n, cores, n8, batch, torch.__version__),
            "all_cores": {"threads": int(cores), "users_per_s": batch / per, "ms_sampler_plus_step": per * 1e3,
                          "ms_sampler": smp * 1e3, "ms_step_only": (per - smp) * 1e3},
            "threads_8": {"threads": 8, "users_per_s": batch / per8, "ms_sampler_plus_step": per8 * 1e3,
                          "ms_sampler": smp8 * 1e3, "ms_step_only": (per8 - smp8) * 1e3}}


def _flush_c_stdio():
    """RCCL writes an informational line through C stdio, which (not a tty) sits in libc's buffer until exit and would land
    AFTER the result line: flush libc early (every rank, once the communicator exists) and again before the JSON."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def main():
    args = parse()
    from rectorch_amd import parallel
    rank, world, local = parallel.init_from_env()
    assert world == args.gpus, "launch with --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py measures the HIP path: an MI355X is required"
    import torch.distributed as dist
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.nets import MultiVAE_net
    from rectorch_amd.models import MultiVAE
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.engine import RowBatch

    I, H, L, B = args.items, 600, 200, args.batch
    X = synth_interactions(args.users, I, seed=20240927)          # same matrix on every rank
    Cd = args.cond_dim
    if Cd:
        from scipy.sparse import csr_matrix, hstack
        from rectorch_amd.nets import CMultiVAE_net
        from rectorch_amd.models import CMultiVAE
        net = CMultiVAE_net(Cd, [L, H, I], dropout=0.5)
        model = CMultiVAE(net, beta=0.2, anneal_steps=100000, learning_rate=1e-3, numerics=args.numerics)
        rs = np.random.RandomState(7)
        cu = rs.randint(-1, Cd, size=args.users)              # -1: unconditioned example
        has = cu >= 0
        onehot = csr_matrix((np.ones(int(has.sum()), dtype=X.dtype), (np.nonzero(has)[0], cu[has])), shape=(args.users, Cd))
        Xin, Xtg = hstack([X, onehot], format="csr"), X
    else:
        net = MultiVAE_net([L, H, I], dropout=0.5)
        model = MultiVAE(net, beta=0.2, anneal_steps=100000, learning_rate=1e-3, numerics=args.numerics)
        Xin, Xtg = X, None
    net.load_state_dict({k: torch.from_numpy(v) for k, v in hash_state_dict([I + Cd, H, L], [L, H, I], "vae", 1234).items()})
    if world == 1 and args.force_dp:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
    if world > 1 or args.force_dp:
        parallel.attach(model, fixed_global_batch=B * world)
    # resident sampler over the global batch; each rank takes its slice of every global batch
    np.random.seed(20240927)
    smp = DataSampler(Xin, Xtg, batch_size=B * world, shuffle=True)
    batches = []
    for rb in smp.iter_rows():
        if len(rb) < B * world:
            break
        s, e = parallel.shard_rows(len(rb), rank, world)
        batches.append(RowBatch(rb.tr, rb.te if Cd else None, rb.rows[s:e].contiguous()))
    net.train()
    torch.manual_seed(1000 + rank)

    def run(n, start):
        for i in range(n):
            model._fused_step(batches[(start + i) % len(batches)], None, want_loss=False)

    run(args.warmup, 0)
    torch.cuda.synchronize()
    _flush_c_stdio()
    eng = net._rtx_engines[args.numerics]
    eng.set_timing("adam", True)            # HIP events around the dominant kernel, on the compute stream
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timings = eng.get_timings()
    eng.set_timing(None, False)
    loss_mean = model._read_loss_sum() / (args.steps + args.warmup)

    if rank != 0:
        dist.destroy_process_group()
        return
    ms_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    step_bytes, step_flops = eng.step_cost(B)
    P = sum(p.numel() for p in net.parameters())
    adam_ms, adam_n = timings.get("adam", (0.0, 0))
    dp = world > 1 or args.force_dp
    # single GPU: one launch per step.  Data parallel: one launch per gradient bucket, right behind its all-reduce;
    # the per-step figure is the sum of the step's launches
    adam_us = adam_ms * 1e3 / (args.steps if dp else max(adam_n, 1))
    # Adam reads p,g,m,v (16 B/param) and writes p,m,v (12 B/param); with the bf16 gradient exchange of the
    # data-parallel bf16 mode the reduced gradient is read as bf16 (2 B/param less)
    adam_bytes = (26.0 if dp and args.numerics == "bf16" else 28.0) * P
    achieved = adam_bytes / (adam_us * 1e-6) / 1e9 if adam_us else None
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r1_pmc_adam.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": ("CMultiVAE (cond_dim=%d) " % Cd if Cd else "MultiVAE ") + "train users/sec on ml-20m (synthetic, ml-20m-shaped)",
        "value": value, "unit": "users/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.numerics == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": "MultiVAE [20108,600,200] on ml-20m-shaped synthetic CSR %dx%d, B=%d per GPU, dropout 0.5, "
                               "beta 0.2 anneal 100000, Adam lr 1e-3 (BASELINE.json configs[1])" % (args.users, I, B),
                   "global_batch": B * world, "parallelism": "dp%d" % world,
                   "numerics": "bf16 MFMA operands, f32 accumulate, f32 master weights + Adam" if args.numerics == "bf16"
                               else "f32 MFMA (parity mode)"},
        "roofline": {"kernel": "k_adam (fused multi-tensor Adam + shadow refresh)", "bound": "hbm",
                     "achieved": achieved, "peak": HBM_PEAK_TBS * 1000.0, "unit": "GB/s",
                     "frac": (achieved / (HBM_PEAK_TBS * 1000.0)) if achieved else None,
                     "traffic": traffic, "algorithmic_bytes_per_launch": adam_bytes, "avg_us": adam_us, "launches": adam_n},
        "step_roofline": {"algorithmic_bytes_per_step": step_bytes, "achieved_GBps": step_bytes / (ms_step * 1e-3) / 1e9,
                          "frac_of_hbm_peak": step_bytes / (ms_step * 1e-3) / 1e9 / (HBM_PEAK_TBS * 1000.0),
                          "algorithmic_flops_per_step": step_flops,
                          "achieved_TFLOPs": step_flops / (ms_step * 1e-3) / 1e12},
        "mean_loss": loss_mean,
    }
    if world == 1 and not args.no_cpu_baseline and not Cd:
        out["cpu_baseline"] = cpu_baseline(X, (I, H, L), B, args.cpu_seconds)
    if dist.is_initialized():
        dist.destroy_process_group()      # RCCL may print while it shuts down: keep the JSON line the last one
    _flush_c_stdio()
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
