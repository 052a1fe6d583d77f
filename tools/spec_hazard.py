#!/usr/bin/env python
"""Scan config-specialised code objects for the compiler defect that round 5's unexplained parity failure came down to:
a VGPR spill / reload / copy placed at the top of a control-flow join block IN FRONT OF the instruction that restores `exec`
(`s_or_b64 exec, exec, s[..]`), between the `v_writelane` SGPR spills the block prologue starts with.  Reached through the branch that
skipped the `then` side, the block starts with exec == 0: the spill stores nothing, and whoever reloads the slot later gets stale scratch
memory - in the N = 17 single-wave object built with `-amdgpu-use-amdgpu-trackers` that was the environment index, i.e. a wild address.

The library runs the same scan on every object it builds or loads (csrc/quadswarm_hip.hip: spec_verify; the C ABI exports it as
qs_spec_verify) and falls back to other compiler settings when it fires; this tool is the stand-alone form, for whole caches:

    python tools/spec_hazard.py [cache dir | object ...]      -> one line per object: clean / HAZARD + the instructions
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("QS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
SILENT = re.compile(r"^(v_writelane_b32|v_readlane_b32|s_nop|s_waitcnt|s_mov_b32|s_mov_b64|s_add_[iu]32|s_addk_i32)\b")
EXEC_DEP = re.compile(r"^(scratch_(load|store)_\w+|v_mov_b32(_e32|_e64)?|v_mov_b64(_e32|_e64)?|v_accvgpr_(read|write)_b32)\b")
WIDEN = re.compile(r"^(s_or_b64|s_mov_b64|s_or_saveexec_b64|s_xor_b64)\s+(exec\b|s\[\d+:\d+\],)")


def _objdump(elf):
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--symbolize-operands", "--no-show-raw-insn", elf], capture_output=True, text=True, check=True).stdout


def _unbundle(blob_path, out):
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={blob_path}", f"--output={out}"],
                       capture_output=True)
    return r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0


def disassemble(path):
    """disassembly text of the gfx950 code in `path`: a bundled / plain code object (.hsaco), or a shared library whose .hip_fatbin section
    holds one offload bundle per translation unit"""
    with tempfile.TemporaryDirectory() as d:
        if path.endswith(".so"):
            fat = os.path.join(d, "fatbin")
            subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path, os.path.join(d, "copy.so")], check=True, capture_output=True)
            blob, magic, text = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__", ""
            starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
            for k, st in enumerate(starts):
                part, elf = os.path.join(d, f"b{k}"), os.path.join(d, f"b{k}.elf")
                open(part, "wb").write(blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)])
                if _unbundle(part, elf):
                    text += _objdump(elf)
            return text
        elf = os.path.join(d, "o.elf")
        return _objdump(elf if _unbundle(path, elf) else path)   # (else: already a plain code object)


def scan(text):
    """[(kernel, label, [instructions of the block prologue up to the exec restore])] for every hazard"""
    out, kernel, label, run, dep = [], "?", None, None, False
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            if not re.match(r"^L\d+$", m.group(1)):
                kernel = m.group(1)
            label, run, dep = m.group(1), [], False
            continue
        ins = line.split("//")[0].strip()
        if not ins or run is None:
            continue
        run.append(ins)
        w = WIDEN.match(ins)
        if w and ("exec" in ins.split(",")[0] or ins.startswith("s_or_saveexec")):
            if dep:
                out.append((kernel, label, list(run)))
            run = None
        elif EXEC_DEP.match(ins):
            dep = True
        elif SILENT.match(ins) and "exec" not in ins.split(",")[0]:
            pass
        else:
            run = None
    return out


def main():
    args = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quad-swarm-rl_amd", "csrc", "spec_cache")]
    paths = []
    for a in args:
        paths += sorted(os.path.join(a, f) for f in os.listdir(a) if f.endswith((".hsaco", ".so"))) if os.path.isdir(a) else [a]
    bad = 0
    for p in paths:
        hz = scan(disassemble(p))
        bad += bool(hz)
        print(os.path.basename(p), "clean" if not hz else f"HAZARD x{len(hz)}")
        for kernel, label, run in hz:
            print(f"    {kernel} <{label}>: " + " ; ".join(run))
    print(f"{len(paths)} objects, {bad} with a spill in front of an exec restore")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
