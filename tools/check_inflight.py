#!/usr/bin/env python3
"""Static check of hand-counted `s_waitcnt vmcnt(N)` code: no instruction may touch a VGPR that an outstanding
vector-memory load is still going to write.

The kernels that count their vector-memory queue by hand issue loads from inline asm, so the compiler does not know
the destination registers are "in flight" -- it is free to COPY them (a tied "+v" asm operand, a phi at a loop edge, a
live-range split) before the wait that makes them valid, and then the copy holds stale data on some runs only.  This
walks the ISA of every kernel in program order, keeps the in-order vmcnt queue (loads to VGPRs, LDS-DMA loads,
stores), retires entries at each `s_waitcnt vmcnt(N)`, and reports every instruction that reads or writes a register
of a load that is still in the queue.  Control flow is followed linearly (labels and branches are ignored), which is
exact for the straight-line pipelines checked here and conservative around their wave-uniform skips.

usage: check_inflight.py file.hip [kernel-name-substring]      (compiles with hipcc -S for gfx950)"""
import os
import re
import shutil
import subprocess
import sys

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels_of(asm):
    """[(name, [lines: instructions and `label:`])] of an AMDGPU assembly listing"""
    out = []
    for m in re.finditer(r"^([A-Za-z_]\S*):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", asm, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if "s_endpgm" not in body:
            continue
        lines = []
        for raw in body.splitlines():
            line = raw.split(";")[0].strip()
            if not line or (line.startswith(".") and not line.endswith(":")):
                continue
            lines.append(line)
        out.append((name, lines))
    return out


def _blocks(lines):
    """basic blocks [(instructions [(index, text)], successors)] of a kernel body"""
    starts, label_at = [0], {}
    for i, line in enumerate(lines):
        if line.endswith(":"):
            label_at[line[:-1]] = i
            starts.append(i)
        elif line.split()[0].startswith(("s_branch", "s_cbranch", "s_endpgm")):
            starts.append(i + 1)
    starts = sorted(set(x for x in starts if x < len(lines)))
    block_of = {st: b for b, st in enumerate(starts)}
    blocks = []
    for b, st in enumerate(starts):
        en = starts[b + 1] if b + 1 < len(starts) else len(lines)
        ins = [(i, lines[i]) for i in range(st, en) if not lines[i].endswith(":")]
        succ = []
        last = ins[-1][1] if ins else ""
        op = last.split()[0] if last else ""
        if op == "s_endpgm":
            pass
        elif op == "s_branch":
            succ = [block_of[label_at[last.split()[1]]]]
        else:
            if op.startswith("s_cbranch"):
                succ.append(block_of[label_at[last.split()[1]]])
            if en < len(lines):
                succ.append(block_of[en])
        blocks.append((ins, succ))
    return blocks


def check_kernel(lines):
    """violations [(index, instruction, load it collides with)] of one kernel: every path through the control flow
    graph is followed with its own in-order vmcnt queue (states are memoised per block)"""
    blocks = _blocks(lines)
    bad, seen, work, origin = {}, set(), [(0, ())], {}
    while work:
        b, queue = work.pop()
        if (b, queue) in seen:
            continue
        seen.add((b, queue))
        if len(seen) > 200000:
            raise RuntimeError("state explosion")
        q = list(queue)
        ins, succ = blocks[b]
        for i, line in ins:
            op = line.split()[0]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", line)
                if m and len(q) > int(m.group(1)):
                    q = q[len(q) - int(m.group(1)):]
                continue
            touched = _regs(line)
            is_vmem = re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", op) is not None
            for dst in q:
                text = origin.get(dst, "?")
                hit = dst & touched
                # a younger load into the same registers is fine (in-order return); its address operands are not
                if hit and is_vmem and not (dst & _regs(line.split(",", 1)[1] if "," in line else "")):
                    continue
                if hit:
                    bad.setdefault(i, (i, line, text))
                    break
            if is_vmem:
                dst = frozenset()
                if "_load_" in op and "_lds_" not in op and not op.endswith("_lds"):
                    dst = frozenset(_regs(line.split(",")[0]))
                elif "_atomic_" in op and (" glc" in line or " sc0" in line):
                    dst = frozenset(_regs(line.split(",")[0]))
                origin[dst] = line
                q.append(dst)
                q = q[-63:]
        for sb in succ:
            work.append((sb, tuple(q)))
    return [bad[k] for k in sorted(bad)]


def compile_to_asm(src):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src],
                       capture_output=True, text=True, check=True)
    return r.stdout


def check_file(src, only=None):
    res = {}
    for name, lines in kernels_of(compile_to_asm(src)):
        if only and only not in name:
            continue
        res[name] = check_kernel(lines)
    return res


if __name__ == "__main__":
    res = check_file(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    nbad = 0
    for name, bad in res.items():
        print(f"{name}: {len(bad)} violation(s)")
        for i, line, text in bad[:10]:
            print(f"    #{i}: {line}    <- in flight: {text}")
        nbad += len(bad)
    print(f"{len(res)} kernels, {nbad} violations")
    sys.exit(1 if nbad else 0)
