"""CPU-side checks of the drop-in boundary: libxsdba_hip.so builds for gfx950, loads without a GPU and
exports exactly the symbols include/xsdba_hip.h declares; without a device the product fails loudly."""
from __future__ import annotations

import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "xsdba_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(xsdba_hip_\w+)\s*\(", text)))


def test_header_symbols_are_exported():
    from xsdba_amd import _lib

    _lib.build()
    lib = _lib.load()
    declared = _declared()
    assert declared, "no declarations parsed from the header"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/xsdba_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\bT (xsdba_hip_\w+)", out)))
    assert exported == declared, "the library exports symbols the header does not declare (or vice versa)"
    assert lib.xsdba_hip_abi_version() == 1


def test_code_object_is_gfx950_only():
    from xsdba_amd import _lib

    _lib.build()
    data = _lib.LIB_PATH.read_bytes()
    assert b"gfx950" in data
    for other in (b"gfx90a", b"gfx942", b"sm_90"):
        assert other not in data


def test_fails_loudly_without_gpu():
    from xsdba_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.XsdbaHipError):
        _lib.Context(0)
    import numpy as np
    from xsdba_amd import hip

    with pytest.raises(_lib.XsdbaHipError):  # no silent CPU path
        hip.quantile_groups(np.zeros((1, 10), np.float32), None, 1, 1, np.array([0.5], np.float32))


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under xsdba_amd/ may reference it"""
    for p in (ROOT / "xsdba_amd").rglob("*"):
        if p.suffix in {".py", ".hip", ".hpp", ".h", ".cpp"}:
            txt = p.read_text()
            assert "import oracle" not in txt and "from oracle" not in txt and "xsdba_oracle" not in txt, p
