"""Package-wide switches of the drop-in API: the same names and behaviour as ``xsdba.set_options`` (reference
options.py:13-83) so that callers' code keeps working — ``with set_options(extra_output=True): ...`` or the plain call for a
lasting change.

Each switch is described by one ``_Switch`` record (default value + accepted type); ``OPTIONS`` is the live mapping the rest
of the package reads (``OPTIONS[EXTRA_OUTPUT]``).
"""
from __future__ import annotations

from dataclasses import dataclass

EXTRA_OUTPUT = "extra_output"  # adjust() returns a Dataset with the diagnostics next to scen
AS_DATASET = "as_dataset"      # always wrap results in a Dataset


@dataclass(frozen=True)
class _Switch:
    default: object
    kind: type

    def accepts(self, value) -> bool:
        # bool is the only kind in use; a bool subclass check keeps 0 / 1 out, like the reference's isinstance test
        return isinstance(value, self.kind)


_SWITCHES = {EXTRA_OUTPUT: _Switch(False, bool), AS_DATASET: _Switch(False, bool)}

OPTIONS = {name: sw.default for name, sw in _SWITCHES.items()}


class set_options:
    """Change switches now; used as a context manager, put the previous values back on exit."""

    def __init__(self, **changes):
        unknown = [name for name in changes if name not in _SWITCHES]
        if unknown:
            raise ValueError(f"Argument name {unknown[0]!r} is not in the set of valid options {set(OPTIONS)!r}.")
        for name, value in changes.items():
            if not _SWITCHES[name].accepts(value):
                raise ValueError(f"option {name!r} given an invalid value: {value!r}")
        self.old = {name: OPTIONS[name] for name in changes}
        OPTIONS.update(changes)

    def __enter__(self):
        return None

    def __exit__(self, exc_type, exc, tb):
        OPTIONS.update(self.old)
        return False
