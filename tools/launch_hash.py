"""sha256 over every source file that shapes the dominant kernel's launch (the fused weight-gradient + Adam kernel): its own
source, the engine that picks its tile / grouping / streams, and the headers both include.  tools/rocprof_summary.py writes it
into the counter summary, bench.py compares it before citing that summary as ``roofline.traffic``."""
import hashlib
import os

LAUNCH_SOURCES = ("rectorch_amd/csrc/dw_adam.hip", "rectorch_amd/csrc/engine.hip", "rectorch_amd/csrc/engine_internal.h", "rectorch_amd/csrc/engine_api.hip",
                  "rectorch_amd/csrc/rtx_kernels.h", "rectorch_amd/csrc/rtx_gemm.h", "rectorch_amd/csrc/rtx_common.h")   # (engine_api.hip: the options that pick tiles and streams)


def launch_sources_sha(root):
    h = hashlib.sha256()
    for rel in LAUNCH_SOURCES:
        p = os.path.join(root, rel)
        if not os.path.exists(p):
            return None
        h.update(rel.encode() + b"\0")
        h.update(open(p, "rb").read())
    return h.hexdigest()
