#!/usr/bin/env python3
"""Where do a kernel's spilt SGPRs hurt?  An SGPR spill is a v_writelane / v_readlane pair into a lane of a spare VGPR: free in a
prologue or in a rare block, a VALU instruction (and its hazards) on the critical path if it sits in the steady-state loop of a
wave that runs a dependent chain.  For every loop of a kernel that contains a barrier (the per-column / per-phase loops of the
register-resident kernels) this prints the spill instructions inside it, innermost first, from the ISA the Makefile keeps
(quantized_neural_nets_amd/csrc/build/*.s).

    python tools/spill_hotpath.py [kernel-name-substring ...]      (default: the kernels that carry the benchmark workloads)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.path.join(ROOT, "quantized_neural_nets_amd", "csrc", "build", "gpfq_capi-hip-amdgcn-amd-amdhsa-gfx950.s")
DEFAULT = ["gpfq_pipe_rg1_m0_w8E", "gpfq_pipe_rg2_m0_w8E", "gpfq_pipe_rg2_m0_w8sE", "gpfq_pipel_m0_w8E", "gpfq_coop_rt2_m0_w8E",
           "gpfq_coop_rt4_m0_w16lqE", "gpfq_resident_rt1_m0_w8E", "gpfq_resident_rt2_m0_w8E"]


def kernels(text):
    cur, out = None, {}
    for ln in text:
        m = re.match(r"^(_ZN4gpfq\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None:
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                out[cur].append(ln)
    return out


def main():
    want = sys.argv[1:] or DEFAULT
    ks = kernels(open(ISA).read().split("\n"))
    for name, L in ks.items():
        if not any(w in name for w in want):
            continue
        labels = {m.group(1): i for i, ln in enumerate(L) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
        loops = []
        for i, ln in enumerate(L):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        rows = []
        for a, b in sorted(set(loops), key=lambda ab: ab[1] - ab[0]):
            body = L[a:b + 1]
            nb = sum("s_barrier" in x for x in body)
            if nb == 0 or b - a < 100:
                continue
            rows.append((b - a, nb, sum("v_readlane_b32" in x for x in body), sum("v_writelane_b32" in x for x in body)))
        short = re.sub(r"^_ZN4gpfq\d+|ENS_10SlabParamsE$", "", name)
        print("%-28s whole kernel: %3d reloads (v_readlane) %3d spills (v_writelane)" % (
            short, sum("v_readlane_b32" in x for x in L), sum("v_writelane_b32" in x for x in L)))
        for ln_, nb, rl, wl in rows[:6]:
            print("    loop of %4d lines, %2d barrier(s): %3d reloads %3d spills" % (ln_, nb, rl, wl))


if __name__ == "__main__":
    main()
