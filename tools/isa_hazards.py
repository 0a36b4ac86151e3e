#!/usr/bin/env python3
"""Hazard lint for hand-issued memory operations (inline assembly the compiler's wait-count pass does not look into):
in the device assembly of qmcp_kernels.hip, between an ASM-issued load (global_load_* / ds_*_rtn_*) and the next
hand-written s_waitcnt, nothing may read or copy the load's destination register -- a copy made before the data has
landed copies what was in the register before (the compiler inserts such copies for loop-carried values it keeps in
another register).  Prints every offending instruction; exit code 1 if any.
usage: tools/isa_hazards.py [-k] <kernel name substring> [...]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/isa/kernels.s"


def assembly(keep=False):
    if not (keep and os.path.exists(OUT)):
        os.makedirs("/tmp/isa", exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{R}/include",
                        f"-I{R}/genome-downsampler_amd/csrc", "--cuda-device-only", "-S",
                        f"{R}/genome-downsampler_amd/csrc/qmcp_kernels.hip", "-o", OUT], check=True, stderr=subprocess.DEVNULL)
    return open(OUT).read()


RING = ["v%d" % i for i in range(96, 120)]   # k_pm_walk's ring registers: hand-written assembly only


def hazards(asm, name, ring_only_in_asm=False):
    """A read of an ASM-issued load's destination is safe once a wait has passed that the load cannot have survived:
    s_waitcnt vmcnt(N) with N <= the number of vector-memory LOADS issued after it and before the wait (loads complete in
    issue order; stores and atomics in between only make a wait longer) -- lgkmcnt(0) for LDS returns.  The scan follows
    the text and takes every backward branch once (the loop's next iteration).  ring_only_in_asm: v96..v112 must not
    occur outside hand-written assembly at all."""
    found = []
    for m in re.finditer(r'^(_ZN4qmcp\w+):.*?\n(.*?)\n\s*\.amdhsa_kernel \1', asm, re.S | re.M):
        if name not in m.group(1):
            continue
        body = m.group(2).split("\n")
        labels = {}
        in_asm_line = [False] * len(body)
        in_asm = False
        for i, line in enumerate(body):
            t = line.strip()
            lm = re.match(r'(\.LBB\w+):', t)
            if lm: labels[lm.group(1)] = i
            if t == ";;#ASMSTART": in_asm = True; continue
            if t == ";;#ASMEND": in_asm = False; continue
            in_asm_line[i] = in_asm
            if ring_only_in_asm and not in_asm and not t.startswith(";"):
                code = t.split(";")[0]
                for r in RING:
                    if re.search(r'\b' + r + r'\b', code) or re.search(r'v\[\d+:\d+\]', code) and any(
                            int(a) <= int(r[1:]) <= int(b) for a, b in re.findall(r'v\[(\d+):(\d+)\]', code)):
                        found.append((m.group(1)[:40], i, "ring register outside assembly", i, code.strip()))
                        break
        for i, line in enumerate(body):
            t = line.strip()
            mm = re.match(r'(global_load_\w+|ds_\w+_rtn_\w+) (v\d+|v\[\d+:\d+\])', t)
            if not (in_asm_line[i] and mm):
                continue
            reg, is_vm = mm.group(2), mm.group(1).startswith("global")
            younger, j, jumped, steps = 0, i + 1, set(), 0
            while j < len(body) and steps < 20000:
                steps += 1
                u = body[j].strip().split(";")[0].strip()
                j += 1
                if not u or u.startswith("."):
                    continue
                w = re.match(r's_waitcnt (.*)', u)
                if w:
                    if is_vm:
                        v = re.search(r'vmcnt\((\d+)\)', w.group(1))
                        if v and int(v.group(1)) <= younger:
                            break    # the load has landed
                    elif "lgkmcnt(0)" in w.group(1):
                        break
                    continue
                if re.match(r'(global|flat|buffer)_load', u):
                    if re.match(r'\S+ ' + re.escape(reg) + r'\b', u):
                        break        # the register is asked for again: its previous answer was consumed before (a read, above)
                    younger += 1
                    continue
                if re.search(r'\b' + re.escape(reg) + r'\b', u):
                    found.append((m.group(1)[:40], i, t, j - 1, u))
                    break
                b = re.match(r's_c?branch\w* (\.LBB\w+)', u)
                if b and b.group(1) in labels and labels[b.group(1)] < j and b.group(1) not in jumped:
                    jumped.add(b.group(1))
                    j = labels[b.group(1)]      # the loop's next iteration
                    continue
                if u.startswith("s_endpgm"):
                    break
    return found


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a != "-k"]
    asm = assembly("-k" in sys.argv)
    bad = []
    for n in names:
        bad += hazards(asm, n, ring_only_in_asm="k_pm_walk" in n)
    for b in bad:
        print("%s: line %d `%s` -> read at line %d `%s`" % b)
    sys.exit(1 if bad else 0)
