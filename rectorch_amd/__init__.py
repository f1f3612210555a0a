"""rectorch_amd -- MI355X-native Mult-VAE / Mult-DAE training and scoring behind the rectorch API.

Drop-in for the one hot path of makgyver/rectorch (``samplers.DataSampler`` -> ``models.MultiVAE/MultiDAE``
-> ``nets.MultiVAE_net/MultiDAE_net`` -> ``evaluation.evaluate`` / ``metrics.Metrics``): same classes and
signatures, computed by hand-written HIP kernels for gfx950 through the C ABI in include/rectorch_hip.h.
"""
__all__ = ["nets", "models", "samplers", "evaluation", "metrics", "data", "parallel"]
__version__ = "0.1.0"
