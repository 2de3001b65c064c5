#!/usr/bin/env python3
"""ISA lint of the hand-placed LDS fragment reads (`make -C megatts2_hierspeechpp_amd/csrc check-isa`).

The MFMA consumer loops of hsp_conv1d_mfma_kernel.h, hsp_cprod3.hip, hsp_dftseg.hip, hsp_bgemm.hip and hsp_mhaproj.hip fetch
their operands with inline-asm `ds_read_b32` and make them valid with an explicit `s_waitcnt lgkmcnt(N)` placed by hand.
The compiler does not know that the "=v" output of such a read is not valid yet: nothing in the language stops it from
COPYING the register (v_mov), SPILLING it (scratch_store / buffer_store) or feeding it to an instruction between the read
and the wait -- silently wrong audio (hsp_bgemm.hip:69-72 records the one time it happened).  The sources bind every
fragment register behind its wait (hsp_conv1d_mfma_kernel.h: wait_frags), which removes the known cause; this script turns
the remaining "what if a future compiler / edit does it anyway" into a build error.

For every kernel of every assembly file given (hipcc -S --cuda-device-only): walk the instructions in program order with
the queue of outstanding LGKM operations (LDS and scalar-memory instructions: `lgkmcnt` counts both).  A VGPR written by an
inline-asm ds_read (between ;;#ASMSTART / ;;#ASMEND) is PENDING until an `s_waitcnt` whose lgkmcnt leaves fewer operations
outstanding than were issued after it (LDS operations of one wave return in order; a scalar load in the queue makes the
lint conservative: only lgkmcnt(0) retires past it).  Any other instruction that reads or writes a pending register is an
error.  The walk follows the control-flow graph (both sides of every conditional branch, states memoised per basic block):
a read issued only on the path that stays in the loop is not held against the epilogue behind the loop's exit.

    python tools/check_isa.py build/isa/*.s          exit status 1 and a report on any finding"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
WAIT = re.compile(r"lgkmcnt\((\d+)\)")
LDS_OP = re.compile(r"^(ds_|s_load_|s_buffer_load_|s_store_|s_memtime|s_memrealtime|s_sendmsg)")
# dwords an LDS read returns (destination register count)
NREG = {"b32": 1, "b64": 2, "b96": 3, "b128": 4, "u8": 1, "i8": 1, "u16": 1, "i16": 1, "b64_tr_b16": 2}


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(path):
    """-> {kernel: [(line no, text, inside inline asm)]} (instructions and labels in program order)"""
    funcs, cur, in_asm = {}, None, False
    for ln, raw in enumerate(open(path), 1):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line or (line.startswith(".") and not line.endswith(":")):
            continue
        if line.endswith(":") and not line.startswith(".L"):
            cur = funcs.setdefault(line[:-1], [])
            continue
        if cur is not None:
            cur.append((ln, line, in_asm))
    return funcs


SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
# s_cmp_<rel>: the subset of {'<', '=', '>'} (a against b) under which SCC becomes 1
CMP = {"lt": "<", "le": "<=", "gt": ">", "ge": ">=", "eq": "=", "lg": "<>"}


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint_kernel(path, kernel, ins, findings):
    """Path-sensitive walk of the kernel's control-flow graph.  State = the queue of outstanding LGKM operations in issue
    order, each (VGPRs an inline-asm ds_read will write | empty, is scalar-memory op), plus what the path knows about scalar
    comparisons: `s_cmp_lt_i32 s8, s56` taken as true makes a later `s_cmp_eq_u32 s8, s56` false while neither register is
    rewritten -- the reads a consumer issues for the NEXT chunk sit behind the same test as the loop's back edge, and
    without that the exit path would be charged with them.  States are memoised per basic block, so loops terminate; every
    finding is reported once."""
    # basic blocks: split at labels and behind branches
    labels, starts = {}, {0}
    for i, (_, line, _) in enumerate(ins):
        if line.endswith(":"):
            labels[line[:-1]] = i
            starts.add(i)
        op = line.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            starts.add(i + 1)
    seen, work, reported, counted = set(), [(0, (), ())], set(), set()
    while work:
        i, queue, known = work.pop()
        if (i, queue, known) in seen or i >= len(ins):
            continue
        seen.add((i, queue, known))
        queue, known = list(queue), dict(known)      # known: (a, b) -> possible relations, a subset of "<=>"
        last_cmp = None                               # (a, b, relations under which SCC = 1) of the s_cmp that set SCC
        while i < len(ins):
            ln, line, in_asm = ins[i]
            if line.endswith(":"):
                i += 1
                if i in starts and (i, tuple(queue), tuple(known.items())) in seen:
                    break
                last_cmp = None
                continue
            op = line.split()[0]
            operands = line[len(op):]
            if op == "s_endpgm" or op == "s_setpc_b64":
                break
            if op == "s_branch":
                work.append((labels[operands.strip()], tuple(queue), tuple(known.items())))
                break
            if op.startswith("s_cbranch"):
                sides = {True: dict(known), False: dict(known)}           # branch taken / not taken
                if op in ("s_cbranch_scc0", "s_cbranch_scc1") and last_cmp is not None:
                    a, b, rel1 = last_cmp
                    cur = known.get((a, b), "<=>")
                    for scc in (1, 0):
                        poss = "".join(c for c in cur if (c in rel1) == bool(scc))
                        taken = (op == "s_cbranch_scc1") == bool(scc)
                        if poss:
                            kk = {k: v for k, v in known.items() if k != (a, b)}
                            kk[(a, b)] = poss
                            while len(kk) > 2:                       # (bounded state: the two latest comparisons)
                                kk.pop(next(iter(kk)))
                            sides[taken] = kk
                        else:
                            sides[taken] = None
                if sides[True] is not None:
                    work.append((labels[operands.strip()], tuple(queue), tuple(sides[True].items())))
                if sides[False] is not None:
                    work.append((i + 1, tuple(queue), tuple(sides[False].items())))
                break
            if op == "s_waitcnt":
                m = WAIT.search(line)
                imm = re.fullmatch(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)", line)
                keep = int(m.group(1)) if m else ((int(imm.group(1), 0) >> 8) & 0xF if imm else None)
                if keep == 0:
                    queue = []
                elif keep is not None and not any(sc for _, sc in queue):   # LDS operations return in order; a scalar load
                    queue = queue[len(queue) - keep:] if keep < len(queue) else queue   # does not: only lgkmcnt(0) retires past it
                i += 1
                continue
            m = re.fullmatch(r"s_cmp_(lt|le|gt|ge|eq|lg)_[iu]32", op)
            if m:
                ab = [t.strip() for t in operands.split(",")]
                last_cmp = None
                if len(ab) == 2:
                    rel = CMP[m.group(1)]
                    if ab[0] > ab[1]:                           # one key per operand pair: `s_cmp_eq s28, s1` meets `s_cmp_ge s1, s28`
                        ab, rel = [ab[1], ab[0]], rel.translate(str.maketrans("<>", "><"))
                    last_cmp = (ab[0], ab[1], rel)
                i += 1
                continue
            if op.startswith("s_") and not op.startswith(("s_nop", "s_barrier", "s_sleep", "s_setprio", "s_sethalt")):
                dst = sregs(operands.split(",")[0])                          # a scalar result: forget what was known about it
                if dst:
                    known = {k: v for k, v in known.items() if not (sregs(k[0]) | sregs(k[1])) & dst}
                    if last_cmp is not None and (sregs(last_cmp[0]) | sregs(last_cmp[1])) & dst:
                        last_cmp = None
                if not op.startswith(("s_mov", "s_load", "s_buffer_load", "s_cselect", "s_movk", "s_cmov")):
                    last_cmp = None if op.startswith(("s_add", "s_sub", "s_and", "s_or", "s_xor", "s_andn2", "s_orn2", "s_lshl",
                                                      "s_lshr", "s_ashr", "s_min", "s_max", "s_mul", "s_bfe", "s_not", "s_abs",
                                                      "s_cmpk", "s_bitcmp", "s_addc", "s_subb", "s_addk", "s_ff", "s_flbit",
                                                      "s_bcnt")) else last_cmp   # these rewrite SCC
            pending = set()
            for rs, _ in queue:
                pending |= rs
            if in_asm and op.startswith("ds_read"):
                d = regs(operands.split(",")[0])
                bad = (regs(",".join(operands.split(",")[1:])) | d) & pending
                if (bad or not d) and ln not in reported:
                    reported.add(ln)
                    findings.append((path, kernel, ln, f"`{line}` touches v{sorted(bad)} whose ds_read is still outstanding"
                                     if d else f"cannot parse the destination of `{line}`"))
                queue.append((frozenset(d), False))
                counted.add(ln)
            else:
                touched = regs(operands) & pending
                if touched and ln not in reported:
                    reported.add(ln)
                    findings.append((path, kernel, ln, f"`{line}` uses v{sorted(touched)} before the s_waitcnt that validates "
                                                       f"the inline-asm ds_read that writes it"))
                if LDS_OP.match(op):
                    queue.append((frozenset(), op.startswith("s_")))
            if len(queue) > 96:                      # (bounded state: drop the oldest operations that define nothing)
                queue = [q for q in queue[:-64] if q[0]] + queue[-64:]
            i += 1
            if i in starts and i < len(ins):
                work.append((i, tuple(queue), tuple(known.items())))
                break
    return len(counted)


def lint(path):
    findings, n = [], 0
    for kernel, ins in parse(path).items():
        n += lint_kernel(path, kernel, ins, findings)
    return findings, n


def main(paths):
    total, bad = 0, []
    for p in paths:
        f, n = lint(p)
        total += n
        bad += f
        print(f"{p}: {n} inline-asm ds_reads, {len(f)} finding(s)")
    for path, kernel, ln, msg in bad:
        print(f"{path}:{ln}: [{kernel}] {msg}")
    if not total:
        print("no inline-asm ds_read found in any file: the lint checked nothing")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
