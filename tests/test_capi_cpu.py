"""CPU checks of the drop-in boundary: the in-tree library loads and exports every
symbol include/fora_hip.h declares; without a GPU the product path refuses to run
(no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "fora_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fora_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from fora_amd import capi
    lib = ctypes.CDLL(capi.lib_path())
    names = _declared()
    assert len(names) >= 20
    for s in names:
        assert hasattr(lib, s), s
    assert sorted(capi.SYMBOLS) == names


def test_product_does_not_touch_oracle():
    # the product path must not route through the checker
    for base, _, files in os.walk(os.path.join(ROOT, "fora_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(base, f)).read()
                for pat in ("oracle_lib", "-lfora_oracle", '#include "fora_oracle', "include \"../../oracle", "CDLL(os.path.join(ROOT, \"oracle"):
                    assert pat not in text, (f, pat)
                if f != "build.py":  # build.py only compiles the checker, it never loads it
                    assert "libfora_oracle" not in text, f


def test_no_gpu_no_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import fora_amd
    with pytest.raises(fora_amd.ForaError):
        fora_amd.Engine(0)


def test_c_binding_compiles_and_fails_loudly_without_gpu():
    """tests/c_smoke/integration_smoke.c is INTEGRATION.md's reference-side binding as a plain C program (gcc -std=c99,
    -lfora_hip).  Here (no GPU) it must build, link and stop at fora_hip_create with FORA_E_NOGPU -- exit code 77."""
    import subprocess
    from fora_amd import build
    exe = build.build_c_smoke(force=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 77, (r.returncode, r.stderr)
