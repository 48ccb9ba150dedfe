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


def test_library_is_not_an_experiment_build():
    """xpic_version() carries bit 30 when any object was compiled with -DXPIC_EXPERIMENT (FILL_EXP / ESK_EXP ablations
    and in-kernel stamps, some of which compute garbage on purpose): such a library must not pass as the product."""
    import xpic_amd

    if not os.path.exists(xpic_amd.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(xpic_amd.LIB_PATH)
    v = lib.xpic_version()
    assert v & 0x40000000 == 0, "libxpic_hip.so is an XPIC_EXPERIMENT build: make clean all"
    hdr = open(os.path.join(ROOT, "include", "xpic_hip.h")).read()
    assert v == int(re.search(r"#define XPIC_VERSION (\d+)", hdr).group(1))
    # the ablation switches refuse to compile without the experiment switch
    src = open(os.path.join(ROOT, "xpic_amd", "csrc", "common.h")).read()
    assert "#error" in src and "XPIC_EXPERIMENT" in src


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


def test_host_mirror_selftest():
    """Host logic of the C++ mirror (JSON reader, Builder::parse_value units, TableDiagnostic text format against
    literal lines of the reference's golden tables) -- runs without a GPU."""
    import subprocess

    exe = os.path.join(ROOT, "xpic_amd", "host", "xpic_hip.out")
    if not os.path.exists(exe):
        import __graft_entry__

        __graft_entry__.build()
    out = subprocess.run([exe, "--selftest"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "selftest ok" in out.stdout, out.stdout + out.stderr


def test_host_mirror_fails_loudly_without_gpu(tmp_path):
    import json
    import subprocess

    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "xpic_amd", "host", "xpic_hip.out")
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "ecsim_ex1", "config.json")))
    cfg["OutputDirectory"] = str(tmp_path)
    (tmp_path / "c.json").write_text(json.dumps(cfg))
    out = subprocess.run([exe, str(tmp_path / "c.json")], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "no ROCm-capable device" in (out.stdout + out.stderr)
