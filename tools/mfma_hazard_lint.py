#!/usr/bin/env python3
"""Wait-state lint for the matrix instructions that fa_fwd_rp16_kernel.hpp spells out as inline asm (Mx::mfma_v*, marked
"; fa_qk" in the listing).  The compiler's hazard recogniser does not look inside an asm statement, so nothing inserts the
wait states gfx950 requires between an XDL instruction that writes architectural registers and the first NON-matrix
instruction that reads or overwrites them (8-pass 16x16x32: 11; we ask for 12).  This walks the listing from every marked
instruction along fall-through and branch edges and fails if such an instruction is reachable in fewer wait states.
The other direction is checked inside the basic block (and across the loop back-edge of a self-looping block): a vector
instruction (v_accvgpr_write/read copies the register allocator puts in, conversions, moves) that writes a source
register of a marked instruction fewer than NEED_IN wait states in front of it, or a matrix instruction that writes one
(other than the marked instruction's own accumulator chain) fewer than NEED wait states in front -- this is how a
128-row-wave d = 64 variant failed (the allocator re-copied Q fragments into accumulator registers right in front of
their use).
Counting is conservative: a matrix instruction = 4 (its minimum issue time), s_nop N = N + 1, anything else = 1.

    python tools/mfma_hazard_lint.py flashattention_kernel_project_amd/csrc/fa_fwd_rp16_d128w.hip [-DFLAG ...]
"""
import re
import subprocess
import sys
import tempfile

NEED = 12
NEED_IN = 4
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-fvisibility=hidden", "-S", "--cuda-device-only"]


def regs(tok, cls="v"):
    """registers of class cls named by an operand token: v7 -> {7}, v[4:7] -> {4..7}"""
    m = re.fullmatch(cls + r"(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(cls + r"\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def both(tok):
    return {("v", r) for r in regs(tok, "v")} | {("a", r) for r in regs(tok, "a")}


def inbound(lines, i, labels):
    """hazards in FRONT of the marked instruction lines[i]: (position, wait states, text) of too-close producers"""
    _, ops = operands(lines[i])
    dst = both(ops[0])
    srcs = set()
    for t in ops[1:4]:
        srcs |= both(t)
    chain = dst & both(ops[3]) if len(ops) > 3 else set()   # the accumulator chain (srcC == vDst) is interlocked
    out, ws, pos, wrapped = [], 0, i - 1, False
    while ws < NEED:
        if pos < 0:
            break
        l = lines[pos]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:   # top of the block: follow the back-edge of a self-looping block once, else stop (other predecessors are
            if wrapped:   # reached through at least a branch and the block's own instructions)
                break
            tail = [k for k in range(i, len(lines)) if re.match(r"\s+s_cbranch\S+\s+" + re.escape(m.group(1)) + r"\s*$", lines[k].split(";")[0])]
            if not tail:
                break
            pos, wrapped = tail[0] - 1, True
            continue
        op, ops2 = operands(l)
        if not op or op.endswith(":") or op.startswith(".") or op.startswith(";"):
            pos -= 1
            continue
        writes = both(ops2[0]) if ops2 and (op.startswith("v_") or op.startswith("ds_read") or op.startswith("buffer_load") or op.startswith("scratch_load")) else set()
        if op.startswith("v_mfma"):
            if (writes & srcs) - chain:
                out.append((pos, ws, l.strip()))
            ws += 4
        elif op.startswith("v_") and writes & srcs:
            if ws < NEED_IN:
                out.append((pos, ws, l.strip()))
            ws += 1
        elif op == "s_nop":
            ws += int(ops2[0]) + 1
        else:
            ws += 1
        pos -= 1
    return out


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", []
    return parts[0], [t.strip() for t in parts[1].split(",")]


def lint_function(name, lines):
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    worst, bad = 10 ** 9, []
    marked = [i for i, l in enumerate(lines) if "fa_qk" in l and "v_mfma" in l]
    for i in marked:
        _, ops = operands(lines[i])
        dst = regs(ops[0])
        # (position, wait states so far); stop a path at NEED
        stack, seen = [(i + 1, 0)], set()
        while stack:
            pos, ws = stack.pop()
            while pos < len(lines) and ws < NEED:
                if (pos, ws) in seen:
                    break
                seen.add((pos, ws))
                l = lines[pos]
                op, ops2 = operands(l)
                if not op or op.endswith(":") or op.startswith(".") or op.startswith(";"):
                    pos += 1
                    continue
                used = set()
                for t in ops2:
                    for w in t.split():
                        used |= regs(w)
                if op.startswith("v_mfma"):
                    # same-destination accumulation (srcC == vDst, same opcode) needs no wait; A/B reads of the scores do not occur
                    if used & dst and regs(ops2[0]) != dst:
                        bad.append((i, pos, ws, l.strip()))
                    ws += 4
                elif used & dst:
                    worst = min(worst, ws)
                    bad.append((i, pos, ws, l.strip()))
                    break
                elif op == "s_nop":
                    ws += int(ops2[0]) + 1
                else:
                    ws += 1
                if op in ("s_branch",):
                    pos = labels.get(ops2[0], len(lines))
                    continue
                if op.startswith("s_cbranch"):
                    tgt = labels.get(ops2[0])
                    if tgt is not None:
                        stack.append((tgt, ws))
                if op in ("s_endpgm", "s_setpc_b64"):
                    break
                pos += 1
    for i in marked:
        for pos, ws, text in inbound(lines, i, labels):
            bad.append((i, pos, ws, "(in front) " + text))
    # third check: an LDS read into a source register of a marked instruction has been waited for (the compiler's wait-count
    # pass does see asm operands; this only confirms it on the listing).  Per basic block: LDS operations return in
    # order, s_waitcnt lgkmcnt(n) leaves the n youngest outstanding.
    pending = []
    for i, l in enumerate(lines):
        if re.match(r"^\.LBB\d+_\d+:", l):
            pending = []   # (inside basic blocks only: what is outstanding at a join depends on the path taken)
            continue
        op, ops = operands(l)
        if not op or op.endswith(":") or op.startswith("."):
            continue
        if op.startswith("ds_"):
            pending.append((i, regs(ops[0]) if (op.startswith("ds_read") or "permute" in op or "swizzle" in op) and ops else set()))
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            pending.append((i, set()))
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n > 0 else []
        elif "fa_qk" in l and op.startswith("v_mfma"):
            used = set()
            for t in ops[1:]:
                used |= regs(t)
            for pi, dst in pending:
                if dst & used:
                    bad.append((i, pi, 0, "(LDS read not waited for) " + lines[pi].strip()))
    return len(marked), bad


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        inc = ["-I" + src.rsplit("/", 1)[0]] if "/" in src else []
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + inc + extra + [src, "-o", f.name], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:])
            return 2
        txt = open(f.name).read().split("\n")
    fn, cur, total_marked, total_bad = None, [], 0, 0
    for l in txt:
        m = re.match(r"^(_Z\S+):", l)
        if m and fn is None:
            fn, cur = m.group(1), []
            continue
        if fn is not None:
            if l.startswith(".Lfunc_end"):
                n, bad = lint_function(fn, cur)
                total_marked += n
                total_bad += len(bad)
                if n:
                    print(f"{fn[:110]}: {n} spelled-out matrix instructions, {len(bad)} too close")
                for i, pos, ws, text in bad[:10]:
                    print(f"    {cur[i].strip()[:70]}  ->  after {ws} wait states: {text[:90]}")
                fn = None
            else:
                cur.append(l)
    print(f"total: {total_marked} spelled-out matrix instructions, {total_bad} hazards (need {NEED} wait states)")
    return 1 if total_bad or not total_marked else 0


if __name__ == "__main__":
    sys.exit(main())
