"""ctypes binding of librectorch_hip.so (the C ABI declared in include/rectorch_hip.h).

There is NO CPU implementation behind these entry points: if the shared library is missing, or no HIP
device is visible, every compute call raises.  ``build()`` compiles the library in-tree with hipcc for
gfx950 (cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# RTX_LIB_PATH: load another build of the library (A/B measurements of one kernel variant against the shipped one in ONE gpurun call)
LIB_PATH = os.environ.get("RTX_LIB_PATH") or os.path.join(CSRC, "librectorch_hip.so")
MAX_LAYERS = 8

RTX_VAE, RTX_DAE = 0, 1
RTX_FP32, RTX_BF16 = 0, 1
RTX_STEP_KEEP_GRADS = 1
RTX_STEP_NO_REG_IN_LOSS = 2
RTX_STEP_GRADS_BF16 = 8
RTX_STEP_DEFER_JOIN = 16
NUMERICS = {"fp32": RTX_FP32, "bf16": RTX_BF16}


class RtxError(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [("n_enc", C.c_int32), ("n_dec", C.c_int32),
                ("enc_dims", C.c_int32 * (MAX_LAYERS + 1)), ("dec_dims", C.c_int32 * (MAX_LAYERS + 1)),
                ("variant", C.c_int32), ("numerics", C.c_int32), ("dropout_p", C.c_float),
                ("max_batch", C.c_int32), ("splitk", C.c_int32), ("cond_dim", C.c_int32)]


class SvaeCfg(C.Structure):
    _fields_ = [("n_items", C.c_int32), ("embed_size", C.c_int32), ("rnn_size", C.c_int32),
                ("n_enc", C.c_int32), ("n_dec", C.c_int32),
                ("enc_dims", C.c_int32 * (MAX_LAYERS + 1)), ("dec_dims", C.c_int32 * (MAX_LAYERS + 1)),
                ("max_len", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("csr", C.c_void_p), ("row_ids", C.c_void_p), ("target_csr", C.c_void_p),
                ("x_dense", C.c_void_p), ("target_dense", C.c_void_p), ("batch", C.c_int32)]


class Step(C.Structure):
    _fields_ = [("beta", C.c_float), ("lam", C.c_float), ("inv_batch", C.c_float),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("step", C.c_int32), ("flags", C.c_int32),
                ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("dropout_mask", C.c_void_p), ("eps_noise", C.c_void_p)]


LAYER_CB = C.CFUNCTYPE(None, C.c_int32, C.c_void_p)

# caller-supplied collectives of the engine-scheduled data-parallel step (rtx_dp_ops)
DP_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)   # all_reduce / reduce_scatter
DP_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)              # all_gather
DP_GROUP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class DpOps(C.Structure):
    _fields_ = [("all_reduce", DP_REDUCE_FN), ("reduce_scatter", DP_REDUCE_FN), ("all_gather", DP_GATHER_FN),
                ("group_start", DP_GROUP_FN), ("group_end", DP_GROUP_FN), ("ctx", C.c_void_p)]


class DpCfg(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("sharded", C.c_int32), ("comm_dtype", C.c_int32),
                ("emulate", C.c_int32), ("comm", C.c_void_p), ("ops", C.POINTER(DpOps)),
                ("comm_side", C.c_void_p), ("ops_side", C.POINTER(DpOps)), ("shard_min_elems", C.c_int32)]


# every symbol include/rectorch_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "rtx_last_error": (C.c_char_p, []),
    "rtx_abi_version": (C.c_int32, []),
    "rtx_csr_upload": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.POINTER(_P)]),
    "rtx_csr_destroy": (C.c_int, [_P]),
    "rtx_csr_shape": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "rtx_csr_gather_dense": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "rtx_engine_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(_P)]),
    "rtx_engine_destroy": (C.c_int, [_P]),
    "rtx_engine_n_tensors": (C.c_int32, [_P]),
    "rtx_engine_tensor_shape": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rtx_engine_bind": (C.c_int, [_P, _P, _P, _P, _P]),
    "rtx_engine_sync_shadows": (C.c_int, [_P, _P]),
    "rtx_engine_forward": (C.c_int, [_P, C.POINTER(Batch), C.c_int32, C.POINTER(Step), C.c_int32, _P, _P, _P, _P]),
    "rtx_engine_encode": (C.c_int, [_P, C.POINTER(Batch), C.c_int32, C.POINTER(Step), _P, _P, _P]),
    "rtx_engine_decode": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "rtx_engine_loss_grads": (C.c_int, [_P, C.POINTER(Batch), C.POINTER(Step), _P, _P, LAYER_CB, _P, _P]),
    "rtx_engine_bind_grads16": (C.c_int, [_P, _P]),
    "rtx_engine_apply_adam": (C.c_int, [_P, C.POINTER(Step), _P]),
    "rtx_engine_apply_adam_layers": (C.c_int, [_P, C.POINTER(Step), C.c_int32, C.c_int32, _P, _P]),
    "rtx_engine_apply_adam_rows": (C.c_int, [_P, C.POINTER(Step), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "rtx_engine_shadow_region": (C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rtx_comm_unique_id": (C.c_int, [_P]),
    "rtx_comm_init": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "rtx_comm_destroy": (C.c_int, [_P]),
    "rtx_comm_rank": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rtx_comm_allreduce": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "rtx_comm_allreduce_many": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P]),
    "rtx_comm_reduce_scatter": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "rtx_comm_allgather": (C.c_int, [_P, _P, C.c_int64, _P]),
    "rtx_comm_group_start": (C.c_int, [_P]),
    "rtx_comm_group_end": (C.c_int, [_P]),
    "rtx_engine_dp_attach": (C.c_int, [_P, C.POINTER(DpCfg)]),
    "rtx_engine_train_step_dp": (C.c_int, [_P, C.POINTER(Batch), C.POINTER(Step), _P, _P, _P]),
    "rtx_engine_dp_owned_rows": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rtx_cast_f32_bf16": (C.c_int, [_P, _P, C.c_int64, _P]),
    "rtx_engine_train_step": (C.c_int, [_P, C.POINTER(Batch), C.POINTER(Step), _P, _P, _P]),
    "rtx_engine_set_next_batch": (C.c_int, [_P, C.POINTER(Batch), C.POINTER(Step)]),
    "rtx_engine_join": (C.c_int, [_P, _P]),
    "rtx_engine_loss_mailbox": (C.c_int, [_P, C.c_int32]),
    "rtx_engine_wait_loss": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_float), C.c_double]),
    "rtx_multinomial_loss": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_float, _P, _P]),
    "rtx_sum_l2_norms": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "rtx_topk_metrics": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, _P, C.c_int32, _P]),
    "rtx_engine_evaluate_topk": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P]),
    "rtx_ease_fit": (C.c_int, [_P, C.c_double, C.POINTER(_P), _P]),
    "rtx_ease_destroy": (C.c_int, [_P]),
    "rtx_ease_weights": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int32)]),
    "rtx_ease_copy_weights": (C.c_int, [_P, _P, _P]),
    "rtx_ease_scores": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "rtx_ease_timings": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rtx_svae_create": (C.c_int, [_P, C.POINTER(_P)]),
    "rtx_svae_destroy": (C.c_int, [_P]),
    "rtx_svae_set_option": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "rtx_svae_loss_mailbox": (C.c_int, [_P, C.c_int32]),
    "rtx_svae_wait_loss": (C.c_int, [_P, C.POINTER(C.c_float), C.c_double]),
    "rtx_svae_n_tensors": (C.c_int32, [_P]),
    "rtx_svae_tensor_shape": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rtx_svae_bind": (C.c_int, [_P, _P, _P, _P, _P]),
    "rtx_svae_forward": (C.c_int, [_P, _P, C.c_int32, _P, C.c_uint64, C.c_uint64, C.c_int32, _P, _P, _P, _P, _P]),
    "rtx_svae_train_step": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.POINTER(Step), _P, _P, _P]),
    "rtx_svae_train_pack": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, C.POINTER(Step), _P, _P, _P]),
    "rtx_engine_set_option": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "rtx_engine_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int32)]),
    "rtx_engine_set_timing": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "rtx_engine_get_timings": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.POINTER(C.c_int32)]),
    "rtx_engine_step_cost": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_LIB = None


def build(verbose=False):
    """Compile librectorch_hip.so for gfx950 with hipcc (in-tree, no GPU needed)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=out)
    assert os.path.exists(LIB_PATH)
    return LIB_PATH


def lib():
    """The loaded library.  Raises RtxError when it has not been built: there is no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RtxError("librectorch_hip.so is not built (%s). Run `python -c \"import __graft_entry__ as g; "
                           "g.build()\"` or `make -C rectorch_amd/csrc`. rectorch_amd has no CPU path." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = l
    return _LIB


def check(rc):
    if rc != 0:
        msg = lib().rtx_last_error()
        raise RtxError("librectorch_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RtxError("rectorch_amd computes only on an AMD MI355X (HIP) device and none is visible; "
                       "there is no CPU implementation of this path (the reference's CPU trainer is rectorch itself).")


def make_cfg(enc_dims, dec_dims, variant, numerics, dropout, max_batch, splitk=0, cond_dim=0):
    cfg = Cfg()
    if len(enc_dims) - 1 > MAX_LAYERS or len(dec_dims) - 1 > MAX_LAYERS:
        raise RtxError("at most %d layers per encoder/decoder are supported" % MAX_LAYERS)
    cfg.n_enc, cfg.n_dec = len(enc_dims) - 1, len(dec_dims) - 1
    for i, d in enumerate(enc_dims):
        cfg.enc_dims[i] = int(d)
    for i, d in enumerate(dec_dims):
        cfg.dec_dims[i] = int(d)
    cfg.variant = RTX_VAE if variant == "vae" else RTX_DAE
    cfg.numerics = NUMERICS[numerics] if isinstance(numerics, str) else int(numerics)
    cfg.dropout_p = float(dropout)
    cfg.max_batch = int(max_batch)
    cfg.splitk = int(splitk)
    cfg.cond_dim = int(cond_dim)
    return cfg


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
