#!/usr/bin/env python3
"""Generates the round loop of the strip-sharing similarity kernel (gfx950 assembly, the body of ONE inline-asm
statement): the waves of a workgroup (one column each, lane = row j of the round) share every 64 x K block of W
through LDS instead of each streaming it through the vector-memory pipeline.

Per strip of K partner rows k (W in strip layout: [j block][k][64 lanes], consecutive k 256 bytes apart):
    every wave copies K/G rows of the NEXT strip straight into LDS (global_load_lds_dword, M0 = destination),
    one s_barrier, then walks ITS column's valid rows inside the strip.  Per step (one valid partner row):
        s_mov_b32 m0, lds offset of row k        (list entry; a multiple of 256, so the VGPR index it implies is 0)
        ds_read_addtid_b32 w                     W[k][j0 + lane], 8 steps ahead
        s_mov_b32 m0, 0x2000 | a_k               VGPR-index mode SRC1 + the table row
        v_mul_f32 x, w, v[T + a_k]               the lane's table column lives in registers
        2 x v_pk_fma_f32 (or 4 x v_fma_f32)      {even, odd} accumulators: acc = x * 1.0 + acc
    -- no vector-memory instruction and no LDS table read in the step: 2 SALU + 1 LDS + 3 (5) VALU.
The list is not padded: a strip's entries are walked in batches of 8 and a tail of 1..8 steps that leaves through
compare-and-branch after every step.

Usage: gen_lstrip_loop.py G K variant NAME   (variant: pk | fma)
"""
import sys

RING_A, RING_B = 32, 40
XA, XB = 48, 49
LA, LB, IA, IB = 48, 56, 64, 72  # SGPR sets (8 each)
TABLE = 64


def gen(G, K, variant="pk", fill=True, barrier=True, grouped=False):
    out = []
    emit = out.append
    rows = K // G
    assert rows <= 16

    def fills():
        if not fill:
            return
        emit("s_mov_b32 m0, s38")
        emit("s_nop 0")
        for r in range(rows):
            emit(f"global_load_lds_dword %[joff], s[36:37] offset:{r * 256}")
        emit(f"s_add_u32 s36, s36, {K * 256}")
        emit("s_addc_u32 s37, s37, 0")
        emit(f"s_xor_b32 s38, s38, {K * 256}")

    def num_add(x):
        xs = x & 1
        if variant == "pk":
            emit(f"v_pk_fma_f32 %[n], v[{x & ~1}:{(x & ~1) + 1}], s[42:43], %[n] op_sel:[{xs},0,0] op_sel_hi:[{xs},1,1]")
        else:
            emit(f"v_fma_f32 %[nl], v{x}, 1.0, %[nl]")
            emit(f"v_fma_f32 %[nh], v{x}, 1.0, %[nh]")

    def den_add(w):
        ws = w & 1
        if variant == "pk":
            emit(f"v_pk_fma_f32 %[d], v[{w & ~1}:{(w & ~1) + 1}], s[42:43], %[d] op_sel:[{ws},0,0] op_sel_hi:[{ws},1,1]")
        else:
            emit(f"v_fma_f32 %[dl], v{w}, 1.0, %[dl]")
            emit(f"v_fma_f32 %[dh], v{w}, 1.0, %[dh]")

    def half(ring_use, idx_use, ring_next, lds_next, idx_load, lds_load, tail_label):
        emit("s_cmp_lt_u32 s41, 9")
        emit(f"s_cbranch_scc1 {tail_label}f")
        emit("s_waitcnt lgkmcnt(0)")
        emit(f"s_load_dwordx8 s[{idx_load}:{idx_load + 7}], %[ix], s40")
        emit("s_add_u32 s40, s40, 32")
        emit(f"s_load_dwordx8 s[{lds_load}:{lds_load + 7}], %[lo], s40")
        if grouped:
            for i in range(8):
                emit(f"s_mov_b32 m0, s{lds_next + i}")
                den_add(ring_use + i)
                emit(f"ds_read_addtid_b32 v{ring_next + i}")
            for i in range(8):
                emit(f"s_mov_b32 m0, s{idx_use + i}")
                emit(f"v_mul_f32 v{50 + i}, v{ring_use + i}, v{TABLE}")
            emit("s_mov_b32 m0, 0")
            for i in range(8):
                num_add(50 + i)
        else:
            for i in range(8):
                x = XA if i % 2 == 0 else XB
                emit(f"s_mov_b32 m0, s{lds_next + i}")
                if i == 0:
                    emit("s_nop 0")
                else:
                    den_add(ring_use + i - 1)
                emit(f"ds_read_addtid_b32 v{ring_next + i}")
                emit(f"s_mov_b32 m0, s{idx_use + i}")
                emit(f"v_mul_f32 v{x}, v{ring_use + i}, v{TABLE}")
                num_add(x)
            den_add(ring_use + 7)
        emit("s_sub_u32 s41, s41, 8")

    def tail(ring_use, idx_use):
        emit("s_waitcnt lgkmcnt(0)")
        for i in range(8):
            x = XA if i % 2 == 0 else XB
            emit(f"s_mov_b32 m0, s{idx_use + i}")
            emit(f"v_mul_f32 v{x}, v{ring_use + i}, v{TABLE}")
            num_add(x)
            den_add(ring_use + i)
            if i < 7:
                emit(f"s_cmp_eq_u32 s41, {i + 1}")
                emit("s_cbranch_scc1 9f")

    emit("s_mov_b64 s[36:37], %[wsrc]")
    emit("s_mov_b32 s38, %[ldsw]")
    emit("s_mov_b32 s39, %[nstr]")
    emit("s_mov_b32 s42, 1.0")
    emit("s_mov_b32 s43, 1.0")
    emit("s_mov_b32 s46, 0")
    emit("s_mov_b32 s47, 0")
    fills()
    emit("s_set_gpr_idx_on s47, gpr_idx(SRC1)")
    emit("1:")
    emit("s_load_dwordx2 s[44:45], %[cum], s46")
    emit("s_add_u32 s46, s46, 4")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_sub_u32 s41, s45, s44")
    emit("s_lshl_b32 s40, s44, 2")
    emit(f"s_load_dwordx8 s[{LA}:{LA + 7}], %[lo], s40")
    emit(f"s_load_dwordx8 s[{IA}:{IA + 7}], %[ix], s40")
    emit("s_add_u32 s40, s40, 32")
    emit(f"s_load_dwordx8 s[{LB}:{LB + 7}], %[lo], s40")
    emit("s_waitcnt vmcnt(0)")
    if barrier:
        emit("s_barrier")
    fills()
    emit("s_cmp_eq_u32 s41, 0")
    emit("s_cbranch_scc1 9f")
    emit("s_waitcnt lgkmcnt(0)")
    for i in range(8):
        emit(f"s_mov_b32 m0, s{LA + i}")
        emit("s_nop 0")
        emit(f"ds_read_addtid_b32 v{RING_A + i}")
    emit("2:")
    half(RING_A, IA, RING_B, LB, IB, LA, 5)
    half(RING_B, IB, RING_A, LA, IA, LB, 6)
    emit("s_branch 2b")
    emit("5:")
    tail(RING_A, IA)
    emit("s_branch 9f")
    emit("6:")
    tail(RING_B, IB)
    emit("9:")
    emit("s_sub_u32 s39, s39, 1")
    emit("s_cmp_lg_u32 s39, 0")
    emit("s_cbranch_scc1 1b")
    emit("s_set_gpr_idx_off")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if barrier:
        emit("s_barrier")
    return out


if __name__ == "__main__":
    G = int(sys.argv[1])
    K = int(sys.argv[2])
    variant = sys.argv[3]
    name = sys.argv[4]
    opts = sys.argv[5:]
    lines = gen(G, K, variant, fill="nofill" not in opts, barrier="nobarrier" not in opts, grouped="grouped" in opts)
    print("#define %s \\" % name)
    print(" \\\n".join('    "%s\\n\\t"' % line for line in lines))
    print()
