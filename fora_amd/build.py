"""Builds every native artefact in-tree (no JIT cache): the gfx950 HIP library
fora_amd/libfora_hip.so, the host CLI fora_amd/bin/fora, and (checker only) the
CPU oracle oracle/libfora_oracle.so.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fora_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libfora_hip.so")
CLI = os.path.join(PKG, "bin", "fora")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-Wall", "-Wno-unused-function"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources(*dirs):
    out = []
    for d in dirs:
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h", ".cpp", ".c", ".hpp")) or f == "Makefile":
                out.append(os.path.join(d, f))
    return out


def build_hip(force=False):
    srcs = _sources(CSRC, os.path.join(ROOT, "include"))
    if not force and _newer(LIB, srcs):
        return LIB
    cmd = [HIPCC, *HIP_FLAGS, "-shared", "-o", LIB, os.path.join(CSRC, "fora_hip.hip")]
    subprocess.run(cmd, check=True)
    return LIB


def build_hip_test(force=False):
    """The same library with the schedule experiments compiled in (-DFORA_TEST_PATHS=1: threshold rounds, bounded
    deferral): loaded only by their twin-equivalence tests (tests/conftest.py `engine_test`), never by the CLI or bench."""
    lib = os.path.join(PKG, "libfora_hip_test.so")
    srcs = _sources(CSRC, os.path.join(ROOT, "include"))
    if not force and _newer(lib, srcs):
        return lib
    subprocess.run([HIPCC, *HIP_FLAGS, "-DFORA_TEST_PATHS=1", "-shared", "-o", lib, os.path.join(CSRC, "fora_hip.hip")], check=True)
    return lib


def build_cli(force=False):
    host = os.path.join(CSRC, "host")
    if not os.path.isdir(host):
        return None
    srcs = _sources(host, os.path.join(ROOT, "include")) + [LIB]
    if not force and _newer(CLI, srcs):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cpps = [s for s in _sources(host) if s.endswith(".cpp")]
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", CLI, *cpps,
           "-L", PKG, "-lfora_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
    subprocess.run(cmd, check=True)
    return CLI


def build_tools(force=False):
    """Test / bench tooling: the HIP R-MAT generator (tools/rmat_gen.hip)."""
    src = os.path.join(ROOT, "tools", "rmat_gen.hip")
    lib = os.path.join(ROOT, "tools", "librmat_gen.so")
    if not force and _newer(lib, [src]):
        return lib
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", lib, src], check=True)
    return lib


def build_c_smoke(force=False):
    """The reference-side binding of INTEGRATION.md as a plain C program (gcc, -lfora_hip)."""
    src = os.path.join(ROOT, "tests", "c_smoke", "integration_smoke.c")
    exe = os.path.join(ROOT, "tests", "c_smoke", "integration_smoke")
    if not force and _newer(exe, [src, LIB, os.path.join(ROOT, "include", "fora_hip.h")]):
        return exe
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-o", exe, src,
                    "-L", PKG, "-lfora_hip", "-lm", "-Wl,-rpath,$ORIGIN/../../fora_amd", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def build_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return os.path.join(ROOT, "oracle", "libfora_oracle.so")


def build_all(force=False):
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(2) as ex:  # the two library builds side by side (hipcc is single-threaded)
        f1, f2 = ex.submit(build_hip, force), ex.submit(build_hip_test, force)
        f1.result(); f2.result()
    build_cli(force)
    build_tools(force)
    build_c_smoke(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force=True)
    print("built", LIB)
