#!/usr/bin/env python3
"""Generates pytrimal_amd/csrc/msastat_lgloop.inc: the round loop of the similarity kernel (similarity_lg, msastat_simx.hip) as ONE
inline-asm statement with hand-allocated registers.

Why not C++ with small asm statements, as rounds 2-4 had it: the loop keeps 16 global loads in flight and counts VMCNT by hand, and
the compiler knows nothing of that -- it is free to copy a register a load is still in flight to (it did, as soon as the loop got
a second exit: stale W values), to put its own spill loads between the counted ones, and it cannot pair two steps' multiplies.
Inside one asm statement nothing moves.

The loop (per wave: one column, 64 rows j = lanes, the column's compacted list of valid partner rows k):
    blocks of 16 steps; block b's W rows are requested while block b - 1 is consumed (16 loads in flight at every step), its 16 table
    rows (ds_read_addtid_b32, address = M0 + 4 lane) and block b + 1's list entries (scalar loads) while it waits for them; the LAST
    block requests nothing (until round 5 every block requested the next one's rows: 16 - 32 loads of the lists' padding per
    round, and the loop ran in steps of 32).
    per PAIR of steps:  s_waitcnt vmcnt(14) . v_pk_mul_f32 (two W x D products in one instruction) . four v_pk_add_f32 ({even, odd}
    accumulators of the numerator and of the denominator, the reference's order: step 2i, then 2i + 1) . two v_add_u32 +
    global_load_dword (the reloads): 3.5 VALU per step where the compiler's loop had 4.

Registers (fixed, clobbered):  v[64:79] W rows . v[80:95] / v[48:63] table rows of blocks A / B (then the products, then the reloads' addresses)
    s[36:51] + s[52:59] entries of block A (16 W-row offsets, 16 table-row offsets as u16) . s[60:75] + s[76:83] block B . s84 s85 temps . s[86:87] s[88:89] the list pointers
Operands: %[an] %[ad] accumulators (f2, in/out) . %[joff] lane offset (4 (j0 + lane)) . %[wuni] base of wlow (s64) . %[base] LDS address
    of the wave's table . %[offp] %[trowp] addresses of the first block's entries in the two lists (s64) . %[nblk] blocks (>= 1; in/out)
"""
import os

W0, DA, DB = 64, 80, 48
EA_O, EA_C, EB_O, EB_C, T0, T1, OFFP, TROWP = 36, 52, 60, 76, 84, 85, 86, 88


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


BIG = False  # (set per generated variant below)


def load(i, o, vo):
    if BIG:
        # the list holds ROW INDICES (beyond 32768 rows the byte offsets of the W rows no longer fit 32 bits): the offset relative to the
        # segment's first row, multiplied out on the scalar unit -- %[wuni] is then the address of THAT row, and the caller vouches that
        # (last row - first row) x row bytes stays below 2^32 (round_loop_lds checks; else the C++ loop)
        return [f"s_sub_u32 s{T0}, s{o + i}, %[k0]", f"s_mul_i32 s{T0}, s{T0}, %[rowb]", f"v_add_u32 v{vo}, s{T0}, %[joff]",
                f"global_load_dword v{W0 + i}, v{vo}, %[wuni]"]
    return [f"v_add_u32 v{vo}, s{o + i}, %[joff]", f"global_load_dword v{W0 + i}, v{vo}, %[wuni]"]


def pair(i, d, reload_o, wait):
    """steps 2i, 2i + 1 of a block: the products land in the two table-row registers (dead behind the multiply), which then serve
    as the address registers of the two reloads (dead behind the numerator's adds): no register beside W, D and the accumulators"""
    w, dd = f"v[{W0 + 2 * i}:{W0 + 2 * i + 1}]", f"v[{d + 2 * i}:{d + 2 * i + 1}]"
    out = [f"s_waitcnt vmcnt({wait})",
           f"v_pk_mul_f32 {dd}, {w}, {dd}",
           f"v_pk_add_f32 %[ad], %[ad], {w} op_sel_hi:[1,0]",
           f"v_pk_add_f32 %[an], %[an], {dd} op_sel_hi:[1,0]",
           f"v_pk_add_f32 %[ad], %[ad], {w} op_sel:[0,1] op_sel_hi:[1,1]",
           f"v_pk_add_f32 %[an], %[an], {dd} op_sel:[0,1] op_sel_hi:[1,1]"]
    if reload_o is not None:  # (behind the last read of the two W registers and of the products)
        out += load(2 * i, reload_o, d + 2 * i) + load(2 * i + 1, reload_o, d + 2 * i + 1)
    return out


def consume_reload(d, o):
    out = []
    for i in range(8):
        out += pair(i, d, o, 14)
    return out


def consume_last(d):
    out = []
    for i in range(8):
        out += pair(i, d, None, 14 - 2 * i)
    return out


def sload(o, c, off_bytes):
    return [f"s_load_dwordx16 s[{o}:{o + 15}], s[{OFFP}:{OFFP + 1}], {hex(off_bytes)}", f"s_load_dwordx8 s[{c}:{c + 7}], s[{TROWP}:{TROWP + 1}], {hex(off_bytes // 2)}"]


def build(big):
    global BIG
    BIG = big
    tag = "big_" if big else ""
    lines = []
    L = lambda name: f".Llg_{tag}{name}_%="
    lines += [f"s_mov_b64 s[{OFFP}:{OFFP + 1}], %[offp]", f"s_mov_b64 s[{TROWP}:{TROWP + 1}], %[trowp]"]
    lines += sload(EA_O, EA_C, 0)
    lines += ["s_cmp_lt_u32 %[nblk], 2", f"s_cbranch_scc1 {L('pro')}"]
    lines += sload(EB_O, EB_C, 0x40)
    lines += [f"{L('pro')}:", "s_waitcnt lgkmcnt(0)"]
    for i in range(16):
        lines += load(i, EA_O, DB + i)  # (block B's table-row registers are free until block A's top)
    lines += issue_rows(DA, EA_C)
    lines += [f"{L('loop')}:"]
    # ---- block A
    lines += ["s_waitcnt lgkmcnt(0)", "s_cmp_eq_u32 %[nblk], 1", f"s_cbranch_scc1 {L('lastA')}"]
    lines += issue_rows(DB, EB_C)
    lines += ["s_cmp_lt_u32 %[nblk], 3", f"s_cbranch_scc1 {L('noA')}"]
    lines += sload(EA_O, EA_C, 0x80)
    lines += [f"{L('noA')}:"]
    lines += consume_reload(DA, EB_O)
    # ---- block B
    lines += ["s_waitcnt lgkmcnt(0)", "s_cmp_eq_u32 %[nblk], 2", f"s_cbranch_scc1 {L('lastB')}"]
    lines += issue_rows(DA, EA_C)
    lines += [f"s_add_u32 s{OFFP}, s{OFFP}, 0x80", f"s_addc_u32 s{OFFP + 1}, s{OFFP + 1}, 0", f"s_add_u32 s{TROWP}, s{TROWP}, 0x40", f"s_addc_u32 s{TROWP + 1}, s{TROWP + 1}, 0"]
    lines += ["s_cmp_lt_u32 %[nblk], 4", f"s_cbranch_scc1 {L('noB')}"]
    lines += sload(EB_O, EB_C, 0x40)
    lines += [f"{L('noB')}:"]
    lines += consume_reload(DB, EA_O)
    lines += ["s_sub_u32 %[nblk], %[nblk], 2", f"s_branch {L('loop')}"]
    lines += [f"{L('lastA')}:"] + consume_last(DA) + [f"s_branch {L('done')}"]
    lines += [f"{L('lastB')}:"] + consume_last(DB)
    lines += [f"{L('done')}:"]

    return lines


clob = [f"v{r}" for r in range(48, 96)] + [f"s{r}" for r in range(36, 90)] + ["m0", "scc", "memory"]
out = ["// GENERATED by tools/gen_lg_loop.py -- do not edit; the design is described there."]
total = 0
for big, name in ((False, "LG_LOOP_ASM"), (True, "LG_LOOP_BIG_ASM")):
    lines = build(big)
    total += len(lines)
    out.append(f"#define {name} \\")
    for ln in lines:
        out.append(f'    "{ln}\\n\\t" \\')
    out.append('    ""')
out.append("#define LG_LOOP_CLOBBERS " + ", ".join(f'"{c}"' for c in clob))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytrimal_amd", "csrc", "msastat_lgloop.inc")
with open(path, "w") as f:
    f.write("\n".join(out) + "\n")
print(path, total, "instructions")
