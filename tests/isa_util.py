"""Disassembly helpers for the CPU tests that tie bench.py's instruction-count constants to the shipped code object:
the gfx950 code objects are carved out of libblaze_hip.so's .hip_fatbin section (one clang offload bundle per translation
unit) and disassembled with the ROCm LLVM tools."""
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def tools_available() -> bool:
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "llvm-objdump", "clang-offload-bundler"))


def disassemble_library(lib_path: str) -> str:
    """Concatenated `llvm-objdump -d` text of every gfx950 code object in the library."""
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for k, s in enumerate(starts):
            piece = os.path.join(td, f"b{k}.bin")
            with open(piece, "wb") as f:
                f.write(blob[s: starts[k + 1] if k + 1 < len(starts) else len(blob)])
            co = os.path.join(td, f"b{k}.co")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={piece}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            out.append(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout)
    return "\n".join(out)


def function_instructions(text: str, mangled: str):
    """[(address, opcode, branch target offset or None)] of one function of the disassembly."""
    lines = text.split("\n")
    hits = [i for i, l in enumerate(lines) if l.endswith("<" + mangled + ">:")]
    if not hits:
        raise KeyError(mangled)
    start = hits[0]
    ins = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]+ <", l):
            break
        m = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<[^+>]+\+0x([0-9a-fA-F]+)>)?", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), int(m.group(4), 16) if m.group(4) else None))
    return ins


def count(ins, opcode: str) -> int:
    return sum(1 for _, o, _ in ins if o == opcode)


def loops(ins):
    """[(first offset, last offset, instructions)] for every backward branch of the function"""
    base = ins[0][0]
    out = []
    for a, o, t in ins:
        if o.startswith(("s_cbranch", "s_branch")) and t is not None and base + t < a:
            out.append((t, a - base, [x for x in ins if base + t <= x[0] <= a]))
    return out
