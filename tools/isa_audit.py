#!/usr/bin/env python3
"""ISA audit of the SHIPPED code object: for every kernel of fora_amd/libfora_hip.so (or the library given) the
register / spill / scratch / LDS figures of its code-object notes and static instruction counts of its disassembly
(readlane / writelane = SGPR spill traffic, scratch_ = VGPR spill traffic, by unit).

    python tools/isa_audit.py [lib.so] [--kernel SUBSTR] [--json] [--dump DIR]

Used by tests/test_isa_audit.py (CPU): the hot kernels must not spill.  Needs only the LLVM
binutils of the ROCm image (llvm-objcopy, clang-offload-bundler, llvm-readelf, llvm-objdump) -- no GPU.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("FORA_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

NOTE_KEYS = {
    ".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".vgpr_spill_count": "vgpr_spill",
    ".sgpr_spill_count": "sgpr_spill", ".private_segment_fixed_size": "scratch_bytes",
    ".group_segment_fixed_size": "lds_static", ".kernarg_segment_size": "kernarg", ".max_flat_workgroup_size": "wg_max",
}


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def extract(lib, workdir):
    """The gfx950 code object of a HIP shared library -> path."""
    fat = os.path.join(workdir, "fat.bin")
    co = os.path.join(workdir, "dev.co")
    _run(os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib)
    _run(os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
         "--targets=" + TARGET, "--output=" + co)
    return co


def notes(co):
    """{mangled kernel name: {vgpr, sgpr, vgpr_spill, sgpr_spill, scratch_bytes, lds_static, ...}}"""
    out, cur = {}, None
    txt = _run(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
    # one '- .agpr_count:' ... block per kernel; .name may come anywhere inside the block
    for block in re.split(r"\n\s+- (?=\.\w+:)", txt):
        vals = {}
        name = None
        for line in block.splitlines():
            m = re.match(r"\s*(?:- )?(\.[a-z_]+):\s+(\S+)\s*$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == ".name" and not line.startswith(" " * 8):
                name = v
            elif k in NOTE_KEYS:
                try:
                    vals[NOTE_KEYS[k]] = int(v)
                except ValueError:
                    pass
        if name and ".sgpr_count" in block and "vgpr" in vals:
            out[name] = vals
    return out


def classify(mn):
    if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if mn.startswith("scratch_"):
        return "scratch"
    if mn.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith("s_waitcnt"):
        return "waitcnt"
    if mn.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("v_"):
        return "valu"
    return "other"


def disasm_counts(co, dump=None):
    """{mangled name: {total, valu, salu, smem, vmem, lds, waitcnt, lane, readlane, writelane, scratch}}"""
    txt = _run(os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co)
    out, cur, lines = {}, None, []
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            if cur and dump:
                with open(os.path.join(dump, cur[:120] + ".s"), "w") as f:
                    f.write("\n".join(lines) + "\n")
            cur, lines = m.group(1), []
            out[cur] = dict(total=0, valu=0, salu=0, smem=0, vmem=0, lds=0, waitcnt=0, lane=0, readlane=0,
                            writelane=0, scratch=0, other=0)
            continue
        if cur is None:
            continue
        lines.append(line)
        t = line.strip().split()
        if not t or t[0].endswith(":") or t[0].startswith("//"):
            continue
        mn = t[0]
        c = out[cur]
        c["total"] += 1
        c[classify(mn)] += 1
        if mn.startswith("v_readlane"):
            c["readlane"] += 1
        elif mn.startswith("v_writelane"):
            c["writelane"] += 1
    if cur and dump:
        with open(os.path.join(dump, cur[:120] + ".s"), "w") as f:
            f.write("\n".join(lines) + "\n")
    return out


def demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True,
                                 check=True).stdout.splitlines()
            if len(out) == len(names):
                return dict(zip(names, out))
        except Exception:
            pass
    return {n: n for n in names}


def short(dem):
    """fora::k_walk_dg<false, true, true>(fora::Dev, ...) -> k_walk_dg<false,true,true>"""
    s = re.sub(r"^void ", "", dem)
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return s[:cut].replace("fora::", "").replace(" ", "")


def audit(lib=None, dump=None):
    """[{kernel, mangled, vgpr, sgpr, vgpr_spill, sgpr_spill, scratch_bytes, lds_static, total, valu, ...}] by name."""
    lib = lib or os.path.join(ROOT, "fora_amd", "libfora_hip.so")
    with tempfile.TemporaryDirectory() as wd:
        co = extract(lib, wd)
        nt = notes(co)
        if dump:
            os.makedirs(dump, exist_ok=True)
        dc = disasm_counts(co, dump)
    dm = demangle(list(nt))
    rows = []
    for name, v in nt.items():
        row = {"kernel": short(dm[name]), "mangled": name}
        row.update(v)
        row.update(dc.get(name, {}))
        rows.append(row)
    rows.sort(key=lambda r: r["kernel"])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?")
    ap.add_argument("--kernel", default="", help="only kernels whose short name contains this")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--dump", help="write every kernel's disassembly into this directory")
    a = ap.parse_args()
    rows = [r for r in audit(a.lib, a.dump) if a.kernel in r["kernel"]]
    if a.json:
        print(json.dumps(rows, indent=1))
        return
    cols = ["vgpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_bytes", "lds_static", "total", "valu", "salu", "vmem",
            "lds", "readlane", "writelane", "scratch"]
    hdr = ["vgpr", "sgpr", "vspill", "sspill", "scratchB", "ldsB", "insts", "valu", "salu", "vmem", "lds", "rdlane",
           "wrlane", "scr_ops"]
    w = max(len(r["kernel"]) for r in rows) if rows else 10
    print("kernel".ljust(w), *[h.rjust(8) for h in hdr])
    for r in rows:
        print(r["kernel"].ljust(w), *[str(r.get(c, "")).rjust(8) for c in cols])


if __name__ == "__main__":
    sys.exit(main())
