"""Build-time guard for the inline-asm LDS read idiom (csrc/conv_bf16.hip `lds_read16` + hand-counted `s_waitcnt lgkmcnt(N)`).

An `asm volatile("ds_read_b128 %0, %1" : "=v"(v) ...)` tells the compiler that `v` is defined when the statement ends; the data
arrives only when a later `s_waitcnt lgkmcnt` retires the read.  Nothing stops the register allocator from copying, spilling or
re-using such a destination in between (it did once, at 160+ VGPRs, in a build of csrc/gemm_x3.hip's 1x1 instantiation: wrong
results, no fault).  The parity tests catch that only if they happen to run the instantiation that broke, so the build checks the
machine code itself:

    for every kernel in the device assembly of a source that uses the idiom, walk the instruction stream in program order with a
    model of the LGKM counter (LDS / scalar-memory operations retire in issue order; `lgkmcnt(N)` leaves the N youngest in flight)
    and fail if any instruction touches a VGPR that an asm `ds_read` has written and no wait has retired yet.

Linear walk: at a label the state of the fall-through predecessor is kept (the loops that use the idiom are fully unrolled
straight-line code between barriers); scalar loads are counted as queue entries, which only makes the check stricter.
"""
from __future__ import annotations

import os
import re
import subprocess
from typing import Iterable, List, Set, Tuple

_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
_COUNTED = ("ds_", "s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "s_sendmsg", "s_scratch_load", "s_atc_probe")


def _regs(text: str) -> Set[int]:
    out: Set[int] = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _lgkm_of_wait(ops: str):
    """lgkmcnt a `s_waitcnt` leaves outstanding, or None when the instruction does not wait on that counter."""
    m = _LGKM.search(ops)
    if m:
        return int(m.group(1))
    ops = ops.strip()
    if re.fullmatch(r"(0x[0-9a-fA-F]+|\d+)", ops):          # raw simm16 (gfx9 layout: lgkmcnt = bits 11:8)
        v = int(ops, 0)
        n = (v >> 8) & 0xF
        return None if n == 0xF else n
    return None


def check_assembly(lines: Iterable[str]) -> List[Tuple[str, int, str, str]]:
    """-> [(kernel, line number, offending instruction, pending read)]; empty = clean."""
    bad: List[Tuple[str, int, str, str]] = []
    kernel = "?"
    queue: List[Tuple[Set[int], str]] = []       # LGKM operations in issue order: (VGPRs an ASM ds_read will write, its text)
    in_asm = False
    for ln, raw in enumerate(lines, 1):
        line = raw.split("//")[0].strip()
        if not line:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if line.startswith(";"):
            continue
        line = line.split(";")[0].strip()
        if not line:
            continue
        if line.startswith(".type") and "@function" in line:
            kernel = line.split()[1].split(",")[0]
            queue = []
            continue
        if line.startswith(".") or line.endswith(":"):
            continue
        parts = line.split(None, 1)
        mnem, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        if mnem == "s_endpgm":
            queue = []
            continue
        if mnem == "s_waitcnt":
            n = _lgkm_of_wait(ops)
            if n is not None and len(queue) > n:
                queue = queue[len(queue) - n:] if n else []
            continue
        pending = set().union(*(q[0] for q in queue)) if queue else set()
        if pending:
            touched = _regs(ops) & pending
            if touched:
                first = next(q[1] for q in queue if q[0] & touched)
                bad.append((kernel, ln, line, first))
        if mnem.startswith(_COUNTED):
            dest: Set[int] = set()
            if in_asm and mnem.startswith("ds_read"):
                dest = _regs(ops.split(",")[0])
            queue.append((dest, f"{line} (line {ln})"))
    return bad


def device_assembly(hipcc: str, flags: List[str], src: str, out_s: str) -> str:
    cmd = [hipcc, *flags, "--offload-device-only", "-S", "-o", out_s, src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"asm_guard: could not produce device assembly of {os.path.basename(src)}:\n{r.stderr}")
    return out_s


def uses_idiom(src: str) -> bool:
    """the source contains an inline-asm ds_read outside `#ifdef SSD_EXPERIMENTAL`-only files (decided by the caller)"""
    with open(src) as f:
        return re.search(r'asm\s+volatile\(\s*"ds_read', f.read()) is not None


def check_source(hipcc: str, flags: List[str], src: str, workdir: str) -> List[Tuple[str, int, str, str]]:
    out_s = os.path.join(workdir, os.path.basename(src).replace(".hip", ".guard.s"))
    device_assembly(hipcc, flags, src, out_s)
    with open(out_s) as f:
        return check_assembly(f)


if __name__ == "__main__":
    import sys
    with open(sys.argv[1]) as f:
        v = check_assembly(f)
    for k, ln, ins, rd in v[:50]:
        print(f"{k}: line {ln}: `{ins}` touches the destination of `{rd}` before a wait retires it")
    print(f"{len(v)} violation(s)")
    sys.exit(1 if v else 0)
