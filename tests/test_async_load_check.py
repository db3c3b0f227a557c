"""tools/check_async_loads.py is the gate that lets the register-resident kernels wait for their column loads by hand
(DESIGN.md 4.1): the build fails if the ISA touches a register whose inline-asm load may still be in flight.  These
tests feed it small hand-written ISA fragments: what must pass, and every kind of violation it must catch."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_async_loads", os.path.join(ROOT, "tools", "check_async_loads.py"))
cal = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cal)


def asm_load(dst, off=0):
    return ["\t;;#ASMSTART", "\tglobal_load_dwordx4 %s, v99, s[8:9] offset:%d" % (dst, off), "\t;;#ASMEND"]


def asm_wait(n):
    return ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(%d)" % n, "\t;;#ASMEND"]


def run(lines):
    has, problems = cal.check_kernel("k", lines + ["\ts_endpgm"])
    assert has
    return problems


def test_reads_behind_the_wait_pass():
    body = asm_load("v[0:3]") + asm_load("v[4:7]") + ["\tv_add_f32_e32 v20, v21, v22"] + asm_wait(0) + \
        ["\tv_pk_mul_f32 v[30:31], v[0:1], v[4:5]"]
    assert run(body) == []


def test_wait_lands_all_but_the_youngest_n():
    # two batches of two; vmcnt(2) lands the first batch only
    body = asm_load("v[0:3]") + asm_load("v[4:7]") + asm_load("v[8:11]") + asm_load("v[12:15]") + asm_wait(2) + \
        ["\tv_fmac_f32_e32 v40, v0, v4"]
    assert run(body) == []
    bad = run(body + ["\tv_fmac_f32_e32 v40, v8, v4"])
    assert len(bad) == 1 and "[8]" in bad[0]


def test_copy_of_a_register_in_flight_is_caught():
    bad = run(asm_load("v[0:3]") + ["\tv_mov_b64_e32 v[50:51], v[2:3]"] + asm_wait(0))
    assert len(bad) == 1 and "touches register(s) in flight [2, 3]" in bad[0]


def test_write_into_a_register_in_flight_is_caught():
    bad = run(asm_load("v[0:3]") + ["\tv_mov_b32_e32 v1, v60"] + asm_wait(0))
    assert len(bad) == 1 and "[1]" in bad[0]


def test_compiler_waits_do_not_count():
    # an s_waitcnt the compiler inserted (outside asm markers) lands nothing for the check
    bad = run(asm_load("v[0:3]") + ["\ts_waitcnt vmcnt(0)", "\tv_add_f32_e32 v9, v0, v0"] + asm_wait(0))
    assert len(bad) == 1


def test_spill_is_caught():
    bad = run(asm_load("v[0:3]") + asm_wait(0) + ["\tscratch_store_dwordx2 off, v[20:21], off offset:8"])
    assert len(bad) == 1 and "spills" in bad[0]


def test_reload_into_a_buffer_still_in_flight_is_caught():
    bad = run(asm_load("v[0:3]") + asm_load("v[0:3]", 1024) + asm_wait(0))
    assert any("still in flight" in b for b in bad)


def test_loop_back_edge_is_replayed():
    # the load at the bottom of the loop is still in flight when the top of the loop is reached again
    body = ["\tv_add_f32_e32 v30, v0, v1"] + asm_wait(0) + asm_load("v[0:3]")
    bad = run(body)
    assert len(bad) == 1 and "touches register(s) in flight [0, 1]" in bad[0]


def test_readfirstlane_of_an_undef_is_tolerated():
    # LLVM materialises undef scalars as v_readfirstlane of an arbitrary VGPR, possibly one in flight
    body = asm_load("v[0:3]") + ["\tv_readfirstlane_b32 s31, v0", "\tv_pk_mul_f32 v[8:9], v[10:11], s[30:31] op_sel_hi:[1,0]"] + asm_wait(0)
    assert run(body) == []
    # any other scalar read of a register in flight is still an error
    assert len(run(asm_load("v[0:3]") + ["\tv_readlane_b32 s31, v0, 3"] + asm_wait(0))) == 1


def test_kernels_without_asm_loads_are_skipped():
    has, problems = cal.check_kernel("k", ["\tglobal_load_dwordx4 v[0:3], v9, s[0:1]", "\tv_mov_b32_e32 v5, v0", "\ts_endpgm"])
    assert not has and problems == []


def test_the_built_isa_passes():
    """The library in the tree was published by a build whose ISA passed (the Makefile runs the check before it copies
    the library); re-check the kept ISA so that a hand-built library cannot slip through."""
    import pytest
    isa = cal.DEFAULT
    if not os.path.exists(isa):
        pytest.skip("no ISA kept (library not built with make in this tree)")
    lib = os.path.join(ROOT, "quantized_neural_nets_amd", "libgpfq_hip.so")
    if os.path.exists(lib) and os.path.getmtime(lib) + 5 < os.path.getmtime(isa):
        pytest.skip("ISA newer than the library: a build is in progress or failed")
    text = open(isa).read().split("\n")
    kernels, cur = {}, None
    for ln in text:
        if ln.startswith("_ZN4gpfq") and ln.rstrip().endswith(":") or (ln.startswith("_ZN4gpfq") and ":" in ln.split()[0]):
            cur = ln.split(":")[0]
            kernels[cur] = []
        elif cur is not None:
            if ln.startswith(".Lfunc_end"):              # (the whole function: role-playing waves return from the middle of it)
                cur = None
            else:
                kernels[cur].append(ln)
    checked, bad, unguarded = 0, [], []
    for name, lines in kernels.items():
        kind, problems = cal.check_kernel(name, lines)
        checked += kind is not None
        bad += problems
        # every register-resident kernel (resident_* / coop_*, the LDS-staged four-row family included) keeps its
        # residual rows in a reserved window: each of them must be under the window guard, none may be skipped
        if ("gpfq_resident_" in name or "gpfq_coop_" in name or "gpfq_pipe_" in name) and kind != "window":
            unguarded.append(name)
    assert checked >= 100 and bad == [] and unguarded == []
    assert sum(1 for n in kernels if "w16l" in n) == 12         # the LDS-staged family (l, lq, lh x four quantizers) was checked
    assert sum(1 for n in kernels if "gpfq_pipe_" in n) == 20   # the pipelined family: (rg1, rg2) x (w8, w8s) + rg2 w8sq, x four quantizers
    # ... to their END: waves that play another role (the reducer waves of a pipelined kernel, the prefetch agent of a
    # resident one) may return from the middle of the function text, and a check that stopped at the first s_endpgm would
    # skip the sweep loop behind it: every such kernel's checked text holds its sweeps and its last barrier
    for name, lines in kernels.items():
        if "gpfq_pipe_" in name or ("gpfq_resident_" in name and "_w1" not in name):
            body = [ln.strip() for ln in lines]
            assert any(ln.startswith(("v_pk_fma_f32", "v_fmac_f32")) for ln in body) and any(ln.startswith("s_barrier") for ln in body), name
            assert body.index(next(ln for ln in body if ln.startswith(("v_pk_fma_f32", "v_fmac_f32")))) > 0
    assert not any(ln.strip().startswith("scratch_") for lines in kernels.values() for ln in lines)   # no kernel spills


# ---- column-window kernels (registers written as base+offset sums, reserved from the compiler) ----------------------
def win_load(base, q):
    return ["\t;;#ASMSTART", "\tglobal_load_dwordx4 v[%d+%d:%d+%d+3], v8, s[38:39] offset:%d" % (base, 4 * q, base, 4 * q, 1024 * q),
            "\t;;#ASMEND"]


def test_register_expressions_are_parsed():
    assert cal.regs_of("v[176+4:176+4+3]") == {180, 181, 182, 183}
    assert cal.regs_of("v[192+14]") == {206}
    assert cal.vregs("\tv_pk_mul_f32 v[20:21], v[18:19], v[192+0:192+1] op_sel_hi:[0,1]") == {18, 19, 20, 21, 192, 193}


def test_window_kernel_passes_when_the_compiler_stays_below_the_window():
    body = win_load(176, 0) + win_load(192, 0) + ["\tv_add_f32_e32 v20, v21, v22"] + asm_wait(0) + \
        ["\t;;#ASMSTART", "\tv_fmac_f32 v15, v[240+0], v[176+0]", "\t;;#ASMEND", "\tv_add_f32_dpp v15, v15, v15 quad_perm:[1,0,3,2]"]
    assert run(body) == []


def test_compiler_code_in_the_window_is_caught():
    bad = run(win_load(176, 0) + asm_wait(0) + ["\tv_mov_b32_e32 v177, v3"])
    assert len(bad) == 1 and "column-window" in bad[0] and "[177]" in bad[0]
    bad = run(win_load(176, 0) + asm_wait(0) + ["\tglobal_store_dwordx4 v[4:5], v[252:255], off"])
    assert len(bad) == 1 and "window starts at v176" in bad[0]


def test_spill_in_a_window_kernel_is_caught():
    bad = run(win_load(176, 0) + asm_wait(0) + ["\tscratch_store_dword off, v3, off offset:4"])
    assert len(bad) == 1 and "spills" in bad[0]


# ---- window kernels that stage their columns through LDS (no asm loads into registers: the window is the residual rows)
def lds_sweep(base):
    return ["\t;;#ASMSTART", "\tv_pk_mul_f32 v[10:11], v[20:21], v[30:31] op_sel:[0,0] op_sel_hi:[1,0]",
            "\tv_pk_add_f32 v[%d+0:%d+1], v[%d+0:%d+1], v[10:11] neg_lo:[0,1] neg_hi:[0,1]" % (base, base, base, base), "\t;;#ASMEND"]


LDS_DMA = ["\tglobal_load_lds_dwordx4 v5, s[10:11] offset:1024"]


def test_lds_staged_kernel_is_a_window_kernel():
    kind, problems = cal.check_kernel("k", LDS_DMA + lds_sweep(64) + ["\tv_add_f32_e32 v20, v21, v22", "\ts_endpgm"])
    assert kind == "window" and problems == []
    assert cal.window_start(lds_sweep(64)) == 64


def test_lds_staged_kernel_compiler_code_in_the_window_is_caught():
    kind, problems = cal.check_kernel("k", LDS_DMA + lds_sweep(64) + ["\tv_mov_b32_e32 v70, v3", "\ts_endpgm"])
    assert kind == "window" and len(problems) == 1 and "window starts at v64" in problems[0]


def test_lds_staged_kernel_spill_is_caught():
    kind, problems = cal.check_kernel("k", LDS_DMA + lds_sweep(64) + ["\tscratch_store_dwordx2 off, v[12:13], off offset:20", "\ts_endpgm"])
    assert kind == "window" and len(problems) == 1 and "spills" in problems[0]


def test_lds_dma_without_a_window_is_flagged_not_skipped():
    kind, problems = cal.check_kernel("k", LDS_DMA + ["\tscratch_load_dword v0, off, off", "\ts_endpgm"])
    assert kind == "lds-dma" and len(problems) == 1


# ---- VALU-written SGPR -> asm VMEM hazard (the compiler pads it only for its own loads) -----------------------------
def test_readfirstlane_right_before_an_asm_load_is_caught():
    """what one build scheduled: `v_readfirstlane_b32 s49, v5` directly in front of the asm load that reads s[48:49] --
    the load went out with a stale s49 and the kernel faulted"""
    body = ["\tv_readfirstlane_b32 s38, v4", "\ts_addc_u32 s47, s47, s3", "\tv_readfirstlane_b32 s39, v5"] + win_load(208, 0)
    bad = cal.valu_sgpr_hazards("k", body + ["\ts_endpgm"])
    assert len(bad) == 2 and "reads s38 2 wait state(s)" in bad[0] and "reads s39 0 wait state(s)" in bad[1]
    kind, problems = cal.check_kernel("k", body + ["\ts_endpgm"])
    assert kind == "window" and len(problems) == 2


def test_five_wait_states_clear_the_hazard():
    ok = ["\tv_readfirstlane_b32 s38, v4", "\tv_readfirstlane_b32 s39, v5", "\ts_nop 4"] + win_load(208, 0)
    assert cal.valu_sgpr_hazards("k", ok + ["\ts_endpgm"]) == []
    short = ["\tv_readfirstlane_b32 s38, v4", "\tv_readfirstlane_b32 s39, v5", "\ts_nop 2"] + win_load(208, 0)
    bad = cal.valu_sgpr_hazards("k", short + ["\ts_endpgm"])
    assert len(bad) == 2 and "s38 4 wait" in bad[0] and "s39 3 wait" in bad[1]
    # scalar-ALU writes have no such hazard
    assert cal.valu_sgpr_hazards("k", ["\ts_add_u32 s38, s38, s2", "\ts_addc_u32 s39, s39, s3"] + win_load(208, 0) + ["\ts_endpgm"]) == []


def test_hazard_is_followed_across_a_branch():
    """a reload of a spilt SGPR (v_readlane) at the end of one block, the asm load at the top of the block it jumps to"""
    body = ["\tv_readlane_b32 s38, v63, 3", "\tv_readlane_b32 s39, v63, 4", "\ts_branch .LBB0_5",
            ".LBB0_4:", "\ts_nop 7", ".LBB0_5:"] + win_load(176, 0) + ["\ts_endpgm"]
    bad = cal.valu_sgpr_hazards("k", body)
    assert len(bad) == 2 and "s38" in bad[0] and "s39" in bad[1]
    # the fall-through path alone (through the s_nop 7) is clean
    clean = [b for b in body if "s_branch" not in b]
    assert cal.valu_sgpr_hazards("k", clean) == []
