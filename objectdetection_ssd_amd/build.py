"""Builds libssd_gfx950.so in-tree with hipcc (gfx950 only; cross-compiles without a GPU).

    python -m objectdetection_ssd_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libssd_gfx950.so")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# bit-exact box arithmetic: no fused multiply-add may be formed in these files
PER_FILE = {"loss.hip": ["-ffp-contract=off"], "nms.hip": ["-ffp-contract=off"], "map_eval.hip": ["-ffp-contract=off"],
            "preprocess.hip": ["-ffp-contract=off"], "photometric.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 extension cannot be built")
    return exe


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool, verbose: bool) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, "common.h"),
            os.path.join(os.path.dirname(PKG), "include", "ssd_gfx950.h"), os.path.abspath(__file__)]
    if force or _stale(obj, deps):
        extra = os.environ.get("SSD_HIPCC_FLAGS", "").split()          # experiments only (e.g. -DSSD_IGEMM_SETPRIO=1)
        if os.environ.get("SSD_EXPERIMENTAL", "") not in ("", "0"):
            extra.append("-DSSD_EXPERIMENTAL")                        # + the default-off kernels (gemm_nt.hip, wino4_full_kernel)
        cmd = [_hipcc(), *COMMON, *PER_FILE.get(src, []), *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    _guard_asm_reads(src, obj, [*COMMON, *PER_FILE.get(src, []), *os.environ.get("SSD_HIPCC_FLAGS", "").split(),
                                *(["-DSSD_EXPERIMENTAL"] if os.environ.get("SSD_EXPERIMENTAL", "") not in ("", "0") else [])])
    return obj


def _guard_asm_reads(src: str, obj: str, flags) -> None:
    """Sources that read LDS through inline asm with hand-counted waits (conv_bf16.hip's `lds_read16`): the machine code of THIS build
    is checked for any instruction that touches such a read's destination before a wait retires it (asm_guard.py) -- the register
    allocator is free to do that, and did once.  A violation fails the build; the verdict is cached beside the object."""
    from . import asm_guard
    path = os.path.join(CSRC, src)
    if os.environ.get("SSD_NO_ASM_GUARD", "") not in ("", "0") or not asm_guard.uses_idiom(path):
        return
    stamp = obj + ".guard.ok"
    if os.path.exists(stamp) and os.path.getmtime(stamp) >= os.path.getmtime(obj):
        return
    flags = [f for f in flags if f != "-fPIC"]
    bad = asm_guard.check_source(_hipcc(), flags, path, OBJ)
    try:
        os.remove(os.path.join(OBJ, src.replace(".hip", ".guard.s")))      # tens of MB of text: not needed once checked
    except OSError:
        pass
    if bad:
        msg = "\n".join(f"  {k}: line {ln}: `{ins}` touches the destination of `{rd}` before a wait retires it" for k, ln, ins, rd in bad[:20])
        raise RuntimeError(f"asm_guard: {len(bad)} unsafe use(s) of an inline-asm ds_read destination in {src} (this compiler / flag set "
                           f"moved a fragment register before its s_waitcnt):\n{msg}")
    with open(stamp, "w") as f:
        f.write("ok\n")


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link of libssd_gfx950.so failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
