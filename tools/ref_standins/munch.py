"""Stand-in for the `munch` package (absent from this image, no network).

Only what `/root/reference/rectorch/configuration.py` uses: DefaultMunch(default, dict).
Test-harness code only.
"""


class DefaultMunch(dict):
    def __init__(self, default=None, mapping=None):
        super().__init__()
        object.__setattr__(self, "_default", default)
        for k, v in (mapping or {}).items():
            self[k] = DefaultMunch(default, v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        return self.get(k, object.__getattribute__(self, "_default"))

    def __setattr__(self, k, v):
        self[k] = v
