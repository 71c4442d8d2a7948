#!/usr/bin/env python3
"""Generates pytrimal_amd/csrc/msastat_lgloop.inc: the round loop of the similarity kernel (similarity_lg, msastat_simx.hip) as ONE
inline-asm statement with hand-allocated registers.

Why not C++ with small asm statements, as rounds 2-4 had it: the loop keeps its global loads in flight and counts VMCNT by hand, and
the compiler knows nothing of that -- it is free to copy a register a load is still in flight to (it did, as soon as the loop got
a second exit: stale W values), to put its own spill loads between the counted ones, and it cannot pair two steps' multiplies.
Inside one asm statement nothing moves.

The loop (per wave: one column, 64 rows j = lanes, the column's compacted list of valid partner rows k):
    blocks of 16 steps on TWO sets of 16 W registers: block b's W rows are requested while block b - 2 is consumed -- 32 loads in
    flight at every step (16 until late in round 5: a wave alone on its SIMD then ran at one step per 61 cycles, a sixteenth of the
    load latency, and a column of 1000 rows is a chain of 16 rounds x 500 such steps: at 1000 x 4000 -- and at 1000 x 300 -- the kernel
    took as long as ONE wave needs for the heaviest column, tools/cu_loads.py, profiles/r05_sim_by_columns.jsonl);
    block b + 1's 16 table rows (ds_read_addtid_b32, address = M0 + 4 lane) are requested at the top of block b, together with the
    list entries further down (scalar loads: table-row offsets of block b + 2, W-row offsets of block b + 3); the last two blocks
    request nothing (no load of the lists' padding).
    per PAIR of steps:  s_waitcnt vmcnt(30) . v_pk_mul_f32 (two W x D products in one instruction) . four v_pk_add_f32 ({even, odd}
    accumulators of the numerator and of the denominator, the reference's order: step 2i, then 2i + 1) . two v_add_u32 +
    global_load_dword (the reloads): 3.5 VALU per step where the compiler's loop had 4.

Registers (fixed, clobbered):  v[64:79] v[80:95] W rows of even / odd blocks . v[96:111] / v[48:63] their table rows (then the products,
    then the reloads' addresses) -- 64 of the 128 a wave has at four waves per SIMD
    s[36:51] s[60:75] W-row offsets (two sets: the block being requested, the next being loaded) . s[52:59] s[76:83] table-row offsets
    (u16) . s84 s85 temps . s[86:87] s[88:89] the list pointers
Operands: %[an] %[ad] accumulators (f2, in/out) . %[joff] lane offset (4 (j0 + lane)) . %[wuni] base of wlow (s64) . %[base] LDS address
    of the wave's table . %[offp] %[trowp] addresses of the first block's entries in the two lists (s64) . %[nblk] blocks (>= 1; in/out)
"""
import os

W, D = (64, 80), (96, 48)
O, C, T0, T1, OFFP, TROWP = (36, 60), (52, 76), 84, 85, 86, 88


def issue_rows(d, c):
    """16 table rows of a block: M0 = 16-bit entry + table base, ds_read_addtid_b32 (no address register, no VALU).  An instruction
    between the M0 write and the LDS instruction that reads it (the hazard needs one wait state)."""
    out = [f"s_bfe_u32 s{T0}, s{c}, 0x100000"]
    for k in range(16):
        cur = T0 if k % 2 == 0 else T1
        nxt = T1 if k % 2 == 0 else T0
        out.append(f"s_add_u32 m0, s{cur}, %[base]")
        if k < 15:
            reg, sh = c + (k + 1) // 2, "0x100010" if (k + 1) % 2 else "0x100000"
            out.append(f"s_bfe_u32 s{nxt}, s{reg}, {sh}")
        else:
            out.append("s_nop 0")
        out.append(f"ds_read_addtid_b32 v{d + k}")
    return out


def load(w, i, o, vo):
    return [f"v_add_u32 v{vo}, s{o + i}, %[joff]", f"global_load_dword v{w + i}, v{vo}, %[wuni]"]


def pair(i, w, d, reload_o, wait):
    """steps 2i, 2i + 1 of a block: the products land in the two table-row registers (dead behind the multiply), which then serve
    as the address registers of the two reloads (dead behind the numerator's adds): no register beside W, D and the accumulators"""
    ww, dd = f"v[{w + 2 * i}:{w + 2 * i + 1}]", f"v[{d + 2 * i}:{d + 2 * i + 1}]"
    out = [f"s_waitcnt vmcnt({wait})",
           f"v_pk_mul_f32 {dd}, {ww}, {dd}",
           f"v_pk_add_f32 %[ad], %[ad], {ww} op_sel_hi:[1,0]",
           f"v_pk_add_f32 %[an], %[an], {dd} op_sel_hi:[1,0]",
           f"v_pk_add_f32 %[ad], %[ad], {ww} op_sel:[0,1] op_sel_hi:[1,1]",
           f"v_pk_add_f32 %[an], %[an], {dd} op_sel:[0,1] op_sel_hi:[1,1]"]
    if reload_o is not None:  # (behind the last read of the two W registers and of the products)
        out += load(w, 2 * i, reload_o, d + 2 * i) + load(w, 2 * i + 1, reload_o, d + 2 * i + 1)
    return out


def consume_reload(x, o):
    """a block with the block after next requested into the registers it frees: 32 loads in flight throughout"""
    out = []
    for i in range(8):
        out += pair(i, W[x], D[x], o, 30)
    return out


def consume_plain(x, behind):
    """a block that requests nothing; `behind` loads (the next block's) stay in flight behind its own"""
    out = []
    for i in range(8):
        out += pair(i, W[x], D[x], None, behind + 14 - 2 * i)
    return out


def sload_o(x, off_bytes):
    return [f"s_load_dwordx16 s[{O[x]}:{O[x] + 15}], s[{OFFP}:{OFFP + 1}], {hex(off_bytes)}"]


def sload_c(x, off_bytes):
    return [f"s_load_dwordx8 s[{C[x]}:{C[x] + 7}], s[{TROWP}:{TROWP + 1}], {hex(off_bytes)}"]


lines = []
L = lambda name: f".Llg_{name}_%="
lines += [f"s_mov_b64 s[{OFFP}:{OFFP + 1}], %[offp]", f"s_mov_b64 s[{TROWP}:{TROWP + 1}], %[trowp]"]
lines += sload_o(0, 0) + sload_c(0, 0)
lines += ["s_cmp_lt_u32 %[nblk], 2", f"s_cbranch_scc1 {L('pro')}"]
lines += sload_o(1, 0x40) + sload_c(1, 0x20)
lines += [f"{L('pro')}:", "s_waitcnt lgkmcnt(0)"]
for i in range(16):
    lines += load(W[0], i, O[0], D[1] + i)  # (the odd blocks' table-row registers are free until the first block's top)
lines += ["s_cmp_lt_u32 %[nblk], 2", f"s_cbranch_scc1 {L('pro2')}"]
for i in range(16):
    lines += load(W[1], i, O[1], D[0] + i)  # (... and the even blocks' until the rows below are requested)
lines += ["s_cmp_lt_u32 %[nblk], 3", f"s_cbranch_scc1 {L('pro2')}"]
lines += sload_o(0, 0x80)  # block 2's W rows are requested while block 0 is consumed
lines += [f"{L('pro2')}:"]
lines += issue_rows(D[0], C[0])
lines += [f"{L('loop')}:"]
# ---- an even block (nblk = blocks left, this one included)
lines += ["s_waitcnt lgkmcnt(0)", "s_cmp_eq_u32 %[nblk], 1", f"s_cbranch_scc1 {L('lastA')}"]
lines += issue_rows(D[1], C[1])
lines += ["s_cmp_lt_u32 %[nblk], 3", f"s_cbranch_scc1 {L('tailA')}"]
lines += sload_c(0, 0x40) + sload_o(1, 0xC0)  # (W-row offsets three blocks ahead: the lists' padding behind a round's end)
lines += consume_reload(0, O[0])
# ---- an odd block (nblk >= 3)
lines += ["s_waitcnt lgkmcnt(0)"]
lines += issue_rows(D[0], C[0])
lines += [f"s_add_u32 s{OFFP}, s{OFFP}, 0x80", f"s_addc_u32 s{OFFP + 1}, s{OFFP + 1}, 0", f"s_add_u32 s{TROWP}, s{TROWP}, 0x40", f"s_addc_u32 s{TROWP + 1}, s{TROWP + 1}, 0"]
lines += ["s_cmp_lt_u32 %[nblk], 4", f"s_cbranch_scc1 {L('tailB')}"]
lines += sload_c(1, 0x20) + sload_o(0, 0x80)
lines += consume_reload(1, O[1])
lines += ["s_sub_u32 %[nblk], %[nblk], 2", f"s_branch {L('loop')}"]
# ---- the ends: the last block but one requests nothing, the last block has nothing behind it
lines += [f"{L('tailA')}:"] + consume_plain(0, 16) + ["s_waitcnt lgkmcnt(0)"] + consume_plain(1, 0) + [f"s_branch {L('done')}"]
lines += [f"{L('tailB')}:"] + consume_plain(1, 16) + ["s_waitcnt lgkmcnt(0)"]
lines += [f"{L('lastA')}:"] + consume_plain(0, 0)
lines += [f"{L('done')}:"]

clob = [f"v{r}" for r in range(48, 112)] + [f"s{r}" for r in range(36, 90)] + ["m0", "scc", "memory"]
out = ["// GENERATED by tools/gen_lg_loop.py -- do not edit; the design is described there.", "#define LG_LOOP_ASM \\"]
for ln in lines:
    out.append(f'    "{ln}\\n\\t" \\')
out.append('    ""')
out.append("#define LG_LOOP_CLOBBERS " + ", ".join(f'"{c}"' for c in clob))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytrimal_amd", "csrc", "msastat_lgloop.inc")
with open(path, "w") as f:
    f.write("\n".join(out) + "\n")
print(path, len(lines), "instructions")
