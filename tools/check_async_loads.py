#!/usr/bin/env python3
"""Static check of the hand-waited column loads (gpfq_device.h: load16_async / wait_landed).

The register-resident kernels issue their column loads as inline asm and wait for them with explicit
`s_waitcnt vmcnt(N)` statements.  That is only sound if nothing touches a destination register of such a load
while the load may still be in flight -- in particular nothing the compiler adds on its own (a copy, a spill, a
temporary parked in the register).  This script replays the ISA of those kernels
(`make -C quantized_neural_nets_amd/csrc asm` writes it to csrc/build/) in program order:

  * a kernel whose asm statements name registers in the window notation (base+offset sums) is a WINDOW kernel: its
    window -- from the lowest such register up -- may be named by no compiler-generated instruction, and it may not
    spill; this covers the kernels whose column buffers are asm-loaded registers AND those that stage their columns
    through LDS (`global_load_lds`) and keep only the residual rows in the window.  For the others:
  * an inline-asm `global_load_dwordx4 vD, ...` puts the registers of vD in flight, in issue order;
  * an inline-asm `s_waitcnt vmcnt(N)` lands every load except the N youngest asm loads (the counter retires in
    order; loads and stores the compiler issues itself only make the hardware wait longer);
  * any other instruction that names a register in flight -- as a source or as a destination -- is an error, and
    so is any scratch access (a spill) in such a kernel.
  * in every checked kernel: an asm load whose scalar base was written by a VALU instruction (v_readfirstlane ...) fewer
    than 5 wait states earlier is an error -- the hardware hazard the compiler pads only for its own loads; so is a
    v_readfirstlane / v_readlane right behind the VALU instruction that wrote its source where the two straddle the
    boundary of an asm statement.

The unrolled loop body is replayed twice, the second time starting from the state at the bottom of the loop, so
that registers in flight across the back edge are covered.

    python tools/check_async_loads.py [path/to/gpfq_capi-hip-amdgcn-amd-amdhsa-gfx950.s]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "quantized_neural_nets_amd", "csrc", "build", "gpfq_capi-hip-amdgcn-amd-amdhsa-gfx950.s")


def _sum(expr):
    """'176+4+3' -> 183 (the column-window statements write register numbers as base+offset sums)"""
    return sum(int(x) for x in expr.split("+"))


def regs_of(tok):
    """v12 -> {12}; v[4:7] -> {4,5,6,7}; v[176+4:176+4+3] -> {180..183}; v[176+2] -> {178}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[([\d+]+):([\d+]+)\]", tok)
    if m:
        return set(range(_sum(m.group(1)), _sum(m.group(2)) + 1))
    m = re.fullmatch(r"v\[([\d+]+)\]", tok)
    if m:
        return {_sum(m.group(1))}
    return set()


def vregs(line):
    body = line.split(";")[0]
    out = set()
    for tok in re.findall(r"v\[[\d+]+(?::[\d+]+)?\]|\bv\d+\b", body):
        out |= regs_of(tok)
    return out


def window_start(lines):
    """First register of a kernel's reserved window: the lowest register any asm statement names in the window
    statements' own notation, base+offset sums (v[64+2:64+3]) -- column buffers the asm loads write, and residual rows
    the asm sweeps update in place (the only window registers of the kernels that stage their columns through LDS).
    None: the kernel has no window."""
    lo = None
    in_asm = False
    for ln in lines:
        st = ln.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
        elif st.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm:
            for tok in re.findall(r"v\[[\d+]+(?::[\d+]+)?\]", st.split(";")[0]):
                if "+" in tok:
                    r = regs_of(tok)
                    lo = min(r) if lo is None else min(lo, min(r))
    return lo


def window_violations(name, lines):
    """Kernels that keep column buffers and / or residual rows in a reserved register window (gpfq_device.h win_*): the
    window is everything from window_start() upwards, and NO instruction outside an asm statement may name a register
    in it -- the compiler must not know those registers exist."""
    lo = window_start(lines)
    if lo is None:
        return []
    out = []
    in_asm = False
    for no, ln in enumerate(lines):
        st = ln.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm or not st or st.startswith((";", ".")) or st.endswith(":"):
            continue
        hit = [r for r in vregs(st) if r >= lo]
        if hit:
            out.append("%s: line %d: compiler code names column-window register(s) %s (window starts at v%d): %s" % (
                name, no, sorted(hit), lo, st))
    return out


def replay(name, lines, inflight, problems, report):
    in_asm = False
    for no, ln in enumerate(lines):
        st = ln.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not st or st.startswith((";", ".")) or st.endswith(":"):
            continue
        if in_asm and st.startswith("global_load_dwordx4"):
            dst = regs_of(st.split()[1].rstrip(","))
            busy = set().union(*inflight) if inflight else set()
            if report and dst & busy:
                problems.append("%s: line %d loads into register(s) still in flight %s: %s" % (name, no, sorted(dst & busy), st))
            inflight.append(dst)
            continue
        if in_asm and st.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", st)
            if m:
                n = int(m.group(1))
                del inflight[:max(0, len(inflight) - n)]
            continue
        if st.startswith("scratch_"):
            if report:
                problems.append("%s: line %d spills in a kernel with hand-waited loads: %s" % (name, no, st))
            continue
        if not inflight:
            continue
        busy = set().union(*inflight)
        hit = vregs(st) & busy
        if hit and st.startswith("v_readfirstlane_b32"):
            # LLVM materialises an UNDEF scalar (e.g. the unused half of an SGPR pair feeding a v_pk_* with
            # op_sel_hi 0) as a readfirstlane of whatever VGPR.  It cannot be a real read: the compiler itself
            # regards the register as holding the column buffer, and the kernels never readfirstlane a buffer.
            continue
        if hit and report:
            problems.append("%s: line %d touches register(s) in flight %s: %s" % (name, no, sorted(hit), st))


def sregs(tok):
    """s12 -> {12}; s[46:47] -> {46, 47}; vcc -> {"vcc"}"""
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    return set()


def valu_sgpr_hazards(name, lines, need=5):
    """gfx940 family: an SGPR written by a VALU instruction (v_readfirstlane, v_readlane -- every reload of a spilt SGPR
    is one --, a compare or carry with a scalar destination) may not be read by a VMEM instruction within the next 5
    wait states.  The compiler pads that for the VMEM instructions it emits -- not for a load inside an asm statement,
    which it does not look into.  Every instruction is one wait state, `s_nop N` is N + 1; an asm VMEM instruction whose
    scalar operands include a register written by VALU fewer than `need` states ago ON ANY PATH is an error: the scan
    follows the control flow (the "young VALU-written SGPRs" at a label are the union over its fall-through and every
    branch to it, youngest age wins), iterated to a fixed point."""
    insts = []          # (line number, text, in_asm)
    in_asm = False
    for no, ln in enumerate(lines):
        raw = ln.strip()
        if raw.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.startswith(";;#ASMEND"):
            in_asm = False
            continue
        st = ln.split(";")[0].strip()
        if not st or st.startswith("."):
            if not re.match(r"\.LBB\d+_\d+:", st):
                continue
        insts.append((no, st, in_asm))
    label_at = {st[:-1]: i for i, (no, st, a) in enumerate(insts) if st.endswith(":")}
    entry = {}          # instruction index -> {sgpr: age} merged from branches
    problems = {}

    def merge(dst, src):
        changed = False
        for r, a in src.items():
            if r not in dst or a < dst[r]:
                dst[r] = a
                changed = True
        return changed

    for _ in range(12):
        changed = False
        age = {}
        for i, (no, st, asm) in enumerate(insts):
            if i in entry:
                merge(age, entry[i])
            if st.endswith(":"):
                continue
            op = st.split()[0]
            toks = re.findall(r"s\[\d+:\d+\]|\bs\d+\b|\bvcc(?:_lo|_hi)?\b", st)
            if asm and op.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "flat_load", "flat_store")):
                for tok in toks:
                    for r in sregs(tok):
                        if r in age and age[r] < need:
                            problems[(no, r)] = ("%s: line %d: asm VMEM instruction reads s%s %d wait state(s) after a VALU "
                                                 "instruction wrote it (needs %d): %s" % (name, no, r, age[r], need, st))
            states = int(st.split()[1]) + 1 if op == "s_nop" else 1
            for r in list(age):
                age[r] += states
                if age[r] >= need:
                    del age[r]
            if op.startswith("v_"):
                first = st[len(op):].split(",")[0].strip()
                for r in sregs(first):
                    age[r] = 0
                if op.startswith(("v_cmp", "v_cmpx")) and op.endswith("_e32"):
                    age["vcc"] = 0
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = st.split()[-1]
                if tgt in label_at:
                    changed |= merge(entry.setdefault(label_at[tgt], {}), age)
                if op == "s_branch":
                    age = {}
            elif op in ("s_endpgm", "s_setpc_b64"):
                age = {}
        if not changed:
            break
    return [problems[k] for k in sorted(problems)]


def readlane_hazards(name, lines, need=1):
    """gfx940 family: a VGPR written by a VALU instruction may not be read by v_readlane / v_readfirstlane in the next
    wait state.  The compiler pads that for its own instructions; checked here wherever the pair straddles the
    boundary of an asm statement (either instruction inside one), which the compiler does not look into.  Linear scan
    (a branch in between is itself a wait state)."""
    out = []
    prev = None         # (registers written by the previous VALU instruction, in_asm) if the previous instruction was VALU
    in_asm = False
    for no, ln in enumerate(lines):
        raw = ln.strip()
        if raw.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.startswith(";;#ASMEND"):
            in_asm = False
            continue
        st = ln.split(";")[0].strip()
        if not st or st.startswith(".") or st.endswith(":"):
            continue
        op = st.split()[0]
        if op in ("v_readfirstlane_b32", "v_readlane_b32") and prev is not None and (in_asm or prev[1]):
            src = st[len(op):].split(",")[1].strip()
            hit = regs_of(src) & prev[0]
            if hit:
                out.append("%s: line %d: %s reads v%s in the wait state after a VALU instruction wrote it, across an asm "
                           "boundary (needs %d): %s" % (name, no, op, sorted(hit), need, st))
        if op.startswith("v_") and not op.startswith(("v_readfirstlane", "v_readlane", "v_cmp")):
            first = st[len(op):].split(",")[0].strip()
            prev = (regs_of(first), in_asm)
        else:
            prev = None
    return out


def is_window_kernel(lines):
    return window_start(lines) is not None


def uses_lds_dma(lines):
    return any(l.strip().startswith("global_load_lds") for l in lines)


def spills(name, lines):
    return ["%s: line %d spills in a kernel with a register window / hand-waited loads: %s" % (name, no, ln.strip())
            for no, ln in enumerate(lines) if ln.strip().startswith("scratch_")]


def check_kernel(name, lines):
    """-> (kind, problems); kind: "window" (reserved register window: column buffers and / or residual rows),
    "replay" (hand-waited asm loads into compiler-allocated registers), "lds-dma" (global_load_lds without a window:
    only the spill check applies), or None (nothing hand-managed: not checked)."""
    asm_loads = any(l.strip().startswith("global_load_dwordx4") and lines[i - 1].strip().startswith(";;#ASMSTART")
                    for i, l in enumerate(lines) if i > 0)
    if is_window_kernel(lines):
        # Window kernels (gpfq_device.h win_*): the registers with loads in flight, and the residual rows updated in place,
        # are never compiler values, so what has to hold on the ISA is (1) no instruction outside an asm statement names a
        # window register and (2) nothing is spilt.  That every sweep is behind a wait covering its columns is a property
        # of the SOURCE: all window statements are `asm volatile`, which the compiler keeps in program order on every
        # path, and the vmcnt arithmetic is the parity tests' job (a linear replay of the text cannot follow these
        # kernels' spin loops).
        return "window", window_violations(name, lines) + spills(name, lines) + valu_sgpr_hazards(name, lines) + readlane_hazards(name, lines)
    if asm_loads:
        problems = []
        inflight = []
        replay(name, lines, inflight, problems, True)
        # second pass from the loop-bottom state: only the loop body matters, duplicates are dropped below
        seen = set(problems)
        again = []
        replay(name, lines, inflight, again, True)
        problems += [p for p in again if p not in seen and "loads into register(s) still in flight" not in p]
        return "replay", problems + valu_sgpr_hazards(name, lines) + readlane_hazards(name, lines)
    if uses_lds_dma(lines):
        return "lds-dma", spills(name, lines)
    return None, []


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    text = open(path).read().split("\n")
    kernels = {}
    cur = None
    for ln in text:
        m = re.match(r"^(_ZN4gpfq\w+|gpfq_\w+):", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            # a kernel ends at its .Lfunc_end label, NOT at the first s_endpgm: kernels whose waves play different roles
            # (the pipelined kernels' reducer waves, the resident kernels' prefetch agent) return from the middle of the text
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                kernels[cur].append(ln)
    bad = []
    kinds = {}
    skipped = []
    nspill = 0
    for name, lines in kernels.items():
        kind, problems = check_kernel(name, lines)
        nspill += sum(1 for ln in lines if ln.strip().startswith("scratch_"))
        if kind is None:
            skipped.append(name)
        else:
            kinds[kind] = kinds.get(kind, 0) + 1
            bad += problems
    for b in bad:
        print(b)
    checked = sum(kinds.values())
    lds_unwindowed = kinds.get("lds-dma", 0)
    print("%d kernels, %d checked (%s), %d not checked (no asm loads, no register window, no LDS DMA: %s), "
          "%d skipped window kernels, %d scratch instruction(s) in all kernels, %d problem(s)" % (
              len(kernels), checked, ", ".join("%d %s" % (v, k) for k, v in sorted(kinds.items())), len(skipped),
              ", ".join(sorted(set(re.sub(r"_ZN4gpfq\d+|ILi.*|EN?S_.*|E?v?P.*", "", n) for n in skipped))) or "-",
              lds_unwindowed, nspill, len(bad)))
    return 1 if bad or checked == 0 or lds_unwindowed else 0


if __name__ == "__main__":
    sys.exit(main())
