from .hashinit import hash_uniform, hash_normal, hash_state_dict
from .synth import synth_interactions

__all__ = ["hash_uniform", "hash_normal", "hash_state_dict", "synth_interactions"]
