"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for scratch (register-spill) instructions INSIDE loops.

Why it matters here: the kernels that keep LDS-DMA in flight across barriers order it with counted `s_waitcnt vmcnt(n)`.  A spill
reload inside such a loop is a VMEM operation on the same counter and the compiler follows it with `s_waitcnt vmcnt(0)` -- the
whole prefetched ring is drained once per stage (round 6 found exactly that in the dominant weight-gradient kernel: correct, and
the four-deep ring worth nothing).  Spills outside the loops (prologue / epilogue) are harmless.

    python tools/scan_scratch_in_loops.py file.s [...]        -> one line per kernel that has scratch instructions
    scan(text) -> {mangled kernel name: (scratch instructions in loops, scratch instructions in total)}
"""
from __future__ import annotations

import re
import shutil
import subprocess
import sys


def scan(text: str) -> dict:
    out = {}
    for part in re.split(r"\n(?=_Z\w+:)", text):
        m = re.match(r"(_Z\w+):", part)
        if not m:
            continue
        # LLVM's asm printer annotates every basic block that lies in a loop: the header ("; =>This Inner Loop Header: Depth=1") and the
        # others ("; in Loop: Header=BB3_7 Depth=1"), on label lines (".LBB3_9: ; in Loop ...") and on fall-through block comments
        # ("; %bb.12: ; in Loop ...").  A scratch instruction is in a loop iff its block is.
        in_loop = False
        n_in = tot = 0
        for l in part.split("\n"):
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
                in_loop = "Loop" in l
            elif re.match(r"^\s+scratch_", l):
                tot += 1
                n_in += in_loop
        out[m.group(1)] = (n_in, tot)
    return out


def demangle(name: str) -> str:
    f = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not f:
        return name
    return subprocess.run([f, name], capture_output=True, text=True).stdout.strip() or name


if __name__ == "__main__":
    for path in sys.argv[1:]:
        for name, (n_in, tot) in sorted(scan(open(path).read()).items(), key=lambda kv: -kv[1][0]):
            if tot:
                print(f"{n_in:4d} in loops / {tot:4d} total  {demangle(name)[:120]}")
