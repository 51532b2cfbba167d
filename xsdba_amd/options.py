"""Global / contextual options (xsdba.options, options.py:13-83): ``set_options(extra_output=bool)``."""
from __future__ import annotations

EXTRA_OUTPUT = "extra_output"
AS_DATASET = "as_dataset"

OPTIONS = {EXTRA_OUTPUT: False, AS_DATASET: False}
_VALIDATORS = {EXTRA_OUTPUT: lambda opt: isinstance(opt, bool), AS_DATASET: lambda opt: isinstance(opt, bool)}


class set_options:
    """Set options in a controlled context (or globally when not used as a context manager)."""

    def __init__(self, **kwargs):
        self.old = {}
        for k, v in kwargs.items():
            if k not in OPTIONS:
                raise ValueError(f"Argument name {k!r} is not in the set of valid options {set(OPTIONS)!r}.")
            if k in _VALIDATORS and not _VALIDATORS[k](v):
                raise ValueError(f"option {k!r} given an invalid value: {v!r}")
            self.old[k] = OPTIONS[k]
        OPTIONS.update(kwargs)

    def __enter__(self):
        return

    def __exit__(self, option_type, value, traceback):
        OPTIONS.update(self.old)
