#!/usr/bin/env python3
"""Safety check for the hand-issued scalar prefetch in csrc/mt_trace.h.

hipcc does not model the loads inside an `asm` statement: it believes their
destination SGPRs are valid right after the statement.  The code is only
correct if NOTHING touches those registers until the matching
`s_waitcnt lgkmcnt(0)` asm statement.  This script compiles the HIP library to
ISA text and verifies exactly that, for every asm-issued s_load:

  * between the load and the next asm wait (in layout order) no instruction
    reads or writes any destination register of the load (that includes
    v_writelane/v_readlane spill code and s_mov copies);
  * every asm wait is immediately preceded (in its own asm block) by nothing
    else, i.e. the wait statement is intact.

Exit status 0 = safe.  Run by tests/test_host_cpu.py and by hand after any
change to the traversal code or the compiler.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def isa_text() -> str:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "mt.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
               "-Wno-pass-failed", "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"),
               "-I", os.path.join(ROOT, "mythtracer_amd", "csrc"), "-o", out,
               os.path.join(ROOT, "mythtracer_amd", "csrc", "mt_capi.hip")]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        return open(out).read()


def sregs(token: str):
    """'s[52:67]' -> {52..67}; 's5' -> {5}."""
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", token)
    return {int(m.group(1))} if m else set()


def referenced_sgprs(line: str):
    regs = set()
    for tok in re.findall(r"s\[\d+:\d+\]|\bs\d+\b", line):
        regs |= sregs(tok)
    return regs


def check(text: str):
    problems, n_loads = [], 0
    for func in re.split(r"\n(?=_Z\w+:)", text):
        name = func.split(":", 1)[0].strip()
        lines = func.split("\n")
        i = 0
        while i < len(lines):
            if "#ASMSTART" in lines[i]:
                j = i + 1
                block = []
                while j < len(lines) and "#ASMEND" not in lines[j]:
                    block.append(lines[j].strip())
                    j += 1
                loads = [b for b in block if b.startswith("s_load_dword")]
                if loads:
                    n_loads += 1
                    inflight = set()
                    for b in loads:
                        inflight |= sregs(b.split()[1].rstrip(","))
                    # walk to the next asm block that holds the wait
                    k = j + 1
                    found = False
                    while k < len(lines):
                        if "#ASMSTART" in lines[k]:
                            nxt = []
                            kk = k + 1
                            while kk < len(lines) and "#ASMEND" not in lines[kk]:
                                nxt.append(lines[kk].strip())
                                kk += 1
                            if any(x.startswith("s_waitcnt lgkmcnt(0)") for x in nxt):
                                found = True
                                break
                            if any(x.startswith("s_load_dword") for x in nxt):
                                problems.append("%s: a second asm load is issued before the wait (line %d)" % (name, k))
                                break
                            k = kk
                        else:
                            code = lines[k].split(";")[0]
                            hit = referenced_sgprs(code) & inflight
                            if hit and not code.strip().startswith("."):
                                problems.append("%s: '%s' touches in-flight s%s" % (name, code.strip(), sorted(hit)[:4]))
                        k += 1
                    if not found and not problems:
                        problems.append("%s: asm load without a following asm wait" % name)
                i = j
            i += 1
    return n_loads, problems


def main():
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 else isa_text()
    n, problems = check(text)
    print("asm-issued scalar loads checked: %d" % n)
    for p in problems[:40]:
        print("UNSAFE:", p)
    if n == 0:
        print("UNSAFE: no asm loads found (pattern changed?)")
        return 2
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
