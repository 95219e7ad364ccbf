#!/usr/bin/env python3
"""Safety check for the hand-issued scalar prefetch in csrc/mt_trace.h.

hipcc does not model the loads inside an `asm` statement: it believes their
destination SGPRs are valid right after the statement.  The code is only
correct if NOTHING touches those registers until the matching
`s_waitcnt lgkmcnt(0)` asm statement.  This script compiles the HIP library to
ISA text and verifies exactly that, for every asm-issued s_load:

  * on EVERY control-flow path from the load to an asm wait no instruction
    reads or writes any destination register of the load (that includes
    v_writelane/v_readlane spill code and s_mov copies), no second asm load is
    issued, no function is called and the function is not left.

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


def parse_function(func: str):
    """-> list of items in layout order: ('label', name) | ('asm', [lines]) | ('ins', text)."""
    items = []
    lines = func.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if "#ASMSTART" in ln:
            j = i + 1
            block = []
            while j < len(lines) and "#ASMEND" not in lines[j]:
                block.append(lines[j].strip())
                j += 1
            items.append(("asm", block))
            i = j + 1
            continue
        code = ln.split(";")[0].strip()
        m = re.fullmatch(r"(\.LBB\w+):", code)
        if m:
            items.append(("label", m.group(1)))
        elif code and not code.startswith(".") and not code.endswith(":"):
            items.append(("ins", code))
        i += 1
    return items


def check(text: str):
    """Control-flow walk: from every asm-issued load follow EVERY path (both
    arms of each conditional branch) up to the asm wait; nothing on the way may
    touch a destination register, issue another asm load, or leave the function."""
    problems, n_loads = [], 0
    for func in re.split(r"\n(?=_Z\w+:)", text):
        name = func.split(":", 1)[0].strip()
        items = parse_function(func)
        label_at = {it[1]: idx for idx, it in enumerate(items) if it[0] == "label"}
        for idx, it in enumerate(items):
            if it[0] != "asm":
                continue
            loads = [b for b in it[1] if b.startswith("s_load_dword")]
            if not loads:
                continue
            n_loads += 1
            inflight = set()
            for b in loads:
                inflight |= sregs(b.split()[1].rstrip(","))
            seen, work, bad = set(), [idx + 1], []
            while work and not bad:
                k = work.pop()
                while True:
                    if k in seen:
                        break
                    seen.add(k)
                    if k >= len(items):
                        bad.append("falls off the function with a load in flight")
                        break
                    kind, val = items[k]
                    if kind == "asm":
                        if any(x.startswith("s_waitcnt lgkmcnt(0)") for x in val):
                            break  # this path is safe
                        if any(x.startswith("s_load_dword") for x in val):
                            bad.append("a second asm load is issued before the wait")
                            break
                    elif kind == "ins":
                        hit = referenced_sgprs(val) & inflight
                        if hit:
                            bad.append("'%s' touches in-flight s%s" % (val, sorted(hit)[:4]))
                            break
                        op = val.split()[0]
                        if op in ("s_setpc_b64", "s_endpgm"):
                            bad.append("leaves the function with a load in flight")
                            break
                        if op == "s_swappc_b64":
                            bad.append("calls a function with a load in flight")
                            break
                        if op == "s_branch":
                            k = label_at.get(val.split()[1], len(items))
                            continue
                        if op.startswith("s_cbranch"):
                            work.append(label_at.get(val.split()[1], len(items)))
                    k += 1
            for b in bad:
                problems.append("%s: %s" % (name, b))
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
