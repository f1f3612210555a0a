"""Stand-in for the `bottleneck` package (absent from this image, no network).

Test-harness code only: lets `/root/reference/rectorch/metrics.py:18` import in THIS container so
golden vectors can be generated.  Never shipped to / imported on the GPU box.
"""
import numpy as np

__version__ = "0.0.0"


def argpartition(a, kth, axis=-1):
    return np.argpartition(a, kth, axis=axis)
