#!/usr/bin/env python3
"""Generates the hand-written partner loop of the column-tile similarity kernel (gfx950 assembly, as the body of
one inline-asm statement) -> pytrimal_amd/csrc/simtile_loop.inc.

One wave owns C columns and 64 consecutive rows j (lane = row).  It walks ALL rows k behind the round's first row
(dense: no per-column lists); W[k][j0 + lane] is loaded ONCE per k (strip layout: consecutive k are 256 bytes apart,
so 16 rows hang off one SGPR base through the instruction's immediate offset) and serves the C columns from a
register.  Per (k, column) slot the column's 16-bit entry says whether row k takes part and, if so, which of the
lane's table registers T_c[a_k] = D[a_k][a_j(lane)] multiplies W:

    s_bfe_u32 / s_lshr_b32 m0, entry      M0 = 0x2000 | a_k: VGPR-index mode for SRC1 + the index; SCC = (entry != 0)
    s_cbranch_scc0 skip                   row k holds no residue in this column: nothing to add
    v_mul_f32    x, w_k, v[T_c]           SRC1 relative: v[T_c + a_k]
    v_pk_fma_f32 {num_even, num_odd} += x * 1.0      (src1 is an SGPR pair: never indexed; fl(x*1 + s) = fl(s + x))
    v_pk_fma_f32 {den_even, den_odd} += w_k * 1.0

3 VALU + 1 SALU + 1 branch per valid slot, one vector load per k for all C columns, no LDS access.

Usage: gen_tile_loop.py C [K] > file.inc     (C columns per wave, K = rows per chunk: 16)
The text is a sequence of C string literals; operands are named below (see simtile kernel / tools/ubench_tile.hip).
"""
import sys


def gen(C, K=16, NT=21, skip="branch", table_base=64, wbase=48, xa=46, xb=47):
    """Registers (physical, clobbered): v[table_base : table_base + NT*C) tables, v[wbase : wbase+K) W ring,
    v46 / v47 products.  SGPRs: s[40:41] W base, s[42:43] {1.0, 1.0}, s44 entry offset, s45 chunk counter,
    entries A at s[48 : 48 + 8C), entries B behind it.  Column pointers come as operands %[e0].. (SGPR pairs)."""
    out = []
    emit = out.append
    nent = K // 2  # SGPRs of entries per column and chunk
    ea = 48
    eb = ea + nent * C
    assert eb + nent * C <= 102, "out of SGPRs"

    def sload(buf):
        for c in range(C):
            lo = buf + c * nent
            w = {4: "dwordx4", 8: "dwordx8"}[nent]
            emit(f"s_load_{w} s[{lo}:{lo + nent - 1}], %[e{c}], s44")

    def chunk(buf, other):
        emit("s_waitcnt lgkmcnt(0)")
        emit(f"s_add_u32 s44, s44, {2 * K}")
        sload(other)  # the chunk after this one
        emit("s_add_u32 s40, s40, %d" % (K * 256))
        emit("s_addc_u32 s41, s41, 0")
        n = 0
        for k in range(K):
            wk = wbase + k
            if skip not in ("noload", "purevalu", "purefma", "bfesgpr"):
                emit(f"s_waitcnt vmcnt({K - 1})")
            for c in range(C):
                se = buf + c * nent + k // 2
                x = xa if n % 2 == 0 else xb
                n += 1
                mdst = "s46" if skip == "bfesgpr" else "m0"
                if skip in ("purevalu", "purefma"):
                    pass
                elif k % 2 == 0:
                    emit(f"s_bfe_u32 {mdst}, s{se}, 0x100000")
                else:
                    emit(f"s_lshr_b32 {mdst}, s{se}, 16")
                if skip in ("branch", "plainfma", "noidx", "noload", "mfma", "mfma2", "mfmaadd"):
                    emit("s_cbranch_scc0 1f")
                elif skip in ("nobranch", "purevalu", "purefma", "bfesgpr"):
                    pass
                else:  # neutralised: the 1/0 flag multiplies W in the denominator (invalid: M0 = 0 reads T[0] = 0)
                    emit("s_cselect_b32 s46, 1.0, 0")
                xs = x & 1
                ws = wk & 1
                if skip in ("mfma", "mfma2", "mfmaadd"):
                    emit(f"v_mul_f32 v{x}, v{table_base + NT * c}, v{wk}")  # SRC0 relative
                    emit("s_movk_i32 m0, 0x1000")                           # index 0 for everything behind
                    if skip == "mfma":
                        emit(f"v_pk_fma_f32 %[d{c}], v[{wk & ~1}:{(wk & ~1) + 1}], s[42:43], %[d{c}] op_sel:[{ws},0,0] op_sel_hi:[{ws},1,1]")
                    elif skip == "mfmaadd":
                        emit(f"v_add_f32 %[d{c}l], v{wk}, %[d{c}l]")
                        emit(f"v_add_f32 %[d{c}h], v{wk}, %[d{c}h]")
                    else:
                        emit(f"v_mfma_f32_4x4x1_16b_f32 %[q{c}], %[one], v{wk}, %[q{c}]")
                    emit(f"v_mfma_f32_4x4x1_16b_f32 %[m{c}], %[one], v{x}, %[m{c}]")
                    emit("1:")
                    continue
                emit(f"v_mul_f32 v{x}, v{wk}, v{table_base + NT * c}")
                if skip in ("plainfma", "purefma"):
                    emit(f"v_fma_f32 %[n{c}l], v{x}, 1.0, %[n{c}l]")
                    emit(f"v_fma_f32 %[n{c}h], v{x}, 1.0, %[n{c}h]")
                    emit(f"v_fma_f32 %[d{c}l], v{wk}, 1.0, %[d{c}l]")
                    emit(f"v_fma_f32 %[d{c}h], v{wk}, 1.0, %[d{c}h]")
                else:
                    emit(f"v_pk_fma_f32 %[n{c}], v[{x & ~1}:{(x & ~1) + 1}], s[42:43], %[n{c}] op_sel:[{xs},0,0] op_sel_hi:[{xs},1,1]")
                    one = "s[46:47]" if skip == "neutral" else "s[42:43]"
                    oh = 0 if skip == "neutral" else 1
                    emit(f"v_pk_fma_f32 %[d{c}], v[{wk & ~1}:{(wk & ~1) + 1}], {one}, %[d{c}] op_sel:[{ws},0,0] op_sel_hi:[{ws},{oh},1]")
                if skip in ("branch", "plainfma", "noidx", "noload"):
                    emit("1:")
            if skip not in ("noload", "purevalu", "purefma", "bfesgpr"):
                emit(f"global_load_dword v{wk}, %[joff], s[40:41] offset:{k * 256}")

    # prologue: constants, first entries, the first K rows of W
    emit("s_mov_b64 s[40:41], %[wrow]")
    emit("s_mov_b32 s42, 1.0")
    emit("s_mov_b32 s43, 1.0")
    emit("s_mov_b32 s44, 0")
    emit("s_mov_b32 s45, %[nch2]")
    emit("s_mov_b32 s47, 0")
    sload(ea)
    for k in range(K):
        emit(f"global_load_dword v{wbase + k}, %[joff], s[40:41] offset:{k * 256}")
    if skip not in ("noidx", "purevalu", "purefma", "bfesgpr"):
        emit("s_set_gpr_idx_on s47, gpr_idx(%s)" % ("SRC0" if skip.startswith("mfma") else "SRC1"))
    emit("2:")
    chunk(ea, eb)
    chunk(eb, ea)
    emit("s_sub_u32 s45, s45, 1")
    emit("s_cmp_lg_u32 s45, 0")
    emit("s_cbranch_scc1 2b")
    emit("s_set_gpr_idx_off")
    if skip.startswith("mfma"):
        emit("s_nop 7")
        emit("s_nop 7")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return out


if __name__ == "__main__":
    C = int(sys.argv[1])
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    skip = sys.argv[3] if len(sys.argv) > 3 else "branch"
    name = sys.argv[4] if len(sys.argv) > 4 else None
    lines = gen(C, K, skip=skip)
    if name:  # as one macro
        print("#define %s \\" % name)
        print(" \\\n".join('    "%s\\n\\t"' % line for line in lines))
    else:
        for line in lines:
            print('"%s\\n\\t"' % line)
