"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/xpic_hip.h declares (no compute calls: there is no GPU here and no CPU fallback by design)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "xpic_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(xpic_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_header_declares_the_documented_surface():
    import xpic_amd

    assert sorted(xpic_amd.SYMBOLS) == header_symbols()


def test_library_exports_every_declared_symbol():
    import xpic_amd

    if not os.path.exists(xpic_amd.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(xpic_amd.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_no_cpu_fallback_without_a_device():
    """Without a HIP device xpic_create must fail loudly (never route to a CPU path)."""
    import torch
    import xpic_amd

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(xpic_amd.XpicError):
        xpic_amd.Context("ecsim", (8, 8, 8), (0.5, 0.5, 0.5), 1.0)


def test_product_does_not_reference_the_oracle():
    """Nothing under xpic_amd/ or include/ may import, link or mention the oracle."""
    for base in ("xpic_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)
