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


def _build_flag(path, name):
    lib = ctypes.CDLL(path)
    lib.fora_hip_get_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]
    v = ctypes.c_int64(-1)
    assert lib.fora_hip_get_option(None, name.encode(), ctypes.byref(v)) == 0  # a property of the build: no context, no GPU
    return v.value


def test_shipped_library_is_not_a_diagnostic_build():
    """fora_amd/csrc/fora_diag.h: every FORA_PROBE_* / FORA_STAMPS* / FORA_DG_FAKE_* macro turns the library into a
    diagnostic build (some of them compute wrong results on purpose).  The product library must be none of them, and the
    kernel bodies must carry no diagnostic #ifdef of their own -- the hooks live in fora_diag.h."""
    from fora_amd import capi
    assert _build_flag(capi.lib_path(), "diag_build") == 0
    assert _build_flag(capi.lib_path(), "test_paths") == 0
    test_lib = os.path.join(ROOT, "fora_amd", "libfora_hip_test.so")
    assert _build_flag(test_lib, "diag_build") == 0 and _build_flag(test_lib, "test_paths") == 1
    for f in ("fora_kernels.h", "fora_team.h", "fora_hip.hip"):
        text = open(os.path.join(ROOT, "fora_amd", "csrc", f)).read()
        for pat in ("FORA_PROBE_", "FORA_DG_FAKE", "ifdef FORA_STAMPS", "defined(FORA_STAMPS"):
            assert pat not in text, (f, pat)
    # ... and a context-bound option still needs a context
    lib = ctypes.CDLL(capi.lib_path())
    lib.fora_hip_get_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]
    v = ctypes.c_int64(0)
    assert lib.fora_hip_get_option(None, b"team_fallbacks", ctypes.byref(v)) != 0
