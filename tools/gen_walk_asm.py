#!/usr/bin/env python3
"""Writes linearham_amd/csrc/lh_prune_walk_asm.inc: the gfx950 assembly of K1's schedule walk (two sites per lane, no
N tips), used by prune_wave_ct through one inline-asm statement.  The text is generated because the 4x4 mat-vec and
the element-wise products are the same eight-row pattern over four register blocks; everything else is written out
below once.  Run from the repo root after editing:  python tools/gen_walk_asm.py

Why assembly: the walk is bound by instruction ISSUE (r03 PMC: every C++ version of the loop kept a wave's
instructions "active" for ~12.4k quad-cycles, whatever the memory latencies were set to), and the compiler's loop
carried ~145 instructions per op (copies at joins, flag juggling, spills) where ~85 are needed.

Register map inside the statement (the statement clobbers v5-v79 and s36-s99):
  v[8:23]  a     running CLV, site 0 in v[8:15], site 1 in v[16:23] (four doubles each)
  v[24:39] st0   pending sibling of stack slot 0 (already multiplied by its branch matrix)
  v[40:55] x     P a, or a tip column
  v[56:71] u     tip column / table entry / deep-slot sibling
  v72..v77 tip states sa0 sa1 sb0 sb1 sc0 sc1      v78 v79 temporaries
  v5 v6    clamped site numbers of the lane's two sites      v7 scaler counts (site 0 low half, site 1 high half)
  s[36:67] P (row-major)      s[68:69] this op's descriptor      s[70:71] the next op's      s[72:73] the one after (in flight)
  s[74:75] scratch region (P-matrices at 0, tables at ctoff)   s76 next table offset   s77 next matrix offset
  s[78:79] msa - L   s80 L   s81 LDS address of the tip table   s82 op counter   s83 op count   s[84:85] descriptors
  s86 offset of the descriptor to prefetch   s98 offset of the last descriptor   s87..s93, s99 temporaries
  s[94:95] exec save   s96 0x2ff00000 (high word of 2^-256)   s97 256
"""
import os

OPTS = set(os.environ.get("LH_ASM_OPTS", "").split(","))   # timing experiments (variants are built into lib_exp/, never the product)
A, ST0, X, U = 8, 24, 40, 56
P = 36


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


def spair(r):
    return "s[%d:%d]" % (r, r + 1)


def matvec(dst):
    """dst <- P a for both sites, rows interleaved (eight independent chains), in the order of lh::matvec:
    fma(p3, a3, fma(p2, a2, fma(p1, a1, p0 * a0)))."""
    out = []
    for j in range(4):
        for s in range(2):
            for i in range(4):
                d = pair(dst + 8 * s + 2 * i)
                a = pair(A + 8 * s + 2 * j)
                p = spair(P + 2 * (4 * i + j))
                out.append("v_mul_f64 %s, %s, %s" % (d, a, p) if j == 0 else "v_fmac_f64_e32 %s, %s, %s" % (d, p, a))
    return out


def product(f1, f2):
    return ["v_mul_f64 %s, %s, %s" % (pair(A + 2 * i), pair(f1 + 2 * i), pair(f2 + 2 * i)) for i in range(8)]


def p_load():
    if "phit" in OPTS:   # every matrix from one of two hot lines (results wrong)
        return ["s_and_b32 s92, s77, 0x80", "s_load_dwordx16 s[36:51], s[74:75], s92 offset:0x0",
                "s_load_dwordx16 s[52:67], s[74:75], s92 offset:0x40"]
    return ["s_load_dwordx16 s[36:51], s[74:75], s77 offset:0x0", "s_load_dwordx16 s[52:67], s[74:75], s77 offset:0x40"]


P_ADV = ["s_add_i32 s77, s77, 0x80"]               # the op's matrix is in registers: s77 names the next one


def tip_column(dst, s0, s1, tip_expr):
    """dst <- tip-table columns of the tip whose number `tip_expr` leaves in s87, for states v<s0>, v<s1>."""
    return tip_expr + [
        "s_lshl_b32 s87, s87, 7", "s_add_i32 s87, s87, s81",
        "v_lshl_add_u32 v78, v%d, 5, s87" % s0, "v_lshl_add_u32 v79, v%d, 5, s87" % s1,
        "ds_read_b128 v[%d:%d], v78" % (dst, dst + 3), "ds_read_b128 v[%d:%d], v78 offset:16" % (dst + 4, dst + 7),
        "ds_read_b128 v[%d:%d], v79" % (dst + 8, dst + 11), "ds_read_b128 v[%d:%d], v79 offset:16" % (dst + 12, dst + 15)]


TIP_A = ["s_lshr_b32 s87, s68, 16"]
TIP_B = ["s_and_b32 s87, s69, 0xffff"]
TIP_C = ["s_lshr_b32 s87, s69, 16"]


def table_entry(dst):
    """dst <- entry (sa, sb) of the next cherry table (global memory, written by this workgroup's prologue)."""
    toff = "%[ctoff]" if "tabhit" in OPTS else "s76"   # tabhit: every look-up in table 0 (results wrong)
    return ["v_lshl_add_u32 v78, v72, 2, v74", "v_lshl_add_u32 v79, v73, 2, v75",
            "v_lshl_add_u32 v78, v78, 5, %s" % toff, "v_lshl_add_u32 v79, v79, 5, %s" % toff,
            "global_load_dwordx4 v[%d:%d], v78, s[74:75]" % (dst, dst + 3),
            "global_load_dwordx4 v[%d:%d], v78, s[74:75] offset:16" % (dst + 4, dst + 7),
            "global_load_dwordx4 v[%d:%d], v79, s[74:75]" % (dst + 8, dst + 11),
            "global_load_dwordx4 v[%d:%d], v79, s[74:75] offset:16" % (dst + 12, dst + 15),
            "s_add_i32 s76, s76, 0x200"]


ROTATE = ["s_mov_b64 s[70:71], s[72:73]"]          # the next op's descriptor has arrived (behind an lgkmcnt(0))
PREFETCH = [                                       # the descriptor two ops ahead; the next P-matrix into the scalar cache
    "s_load_dwordx2 s[72:73], s[84:85], s86", "s_add_i32 s86, s86, 8", "s_min_u32 s86, s86, s98"] + (
    [] if "notouch" in OPTS else ["s_load_dword s99, s[74:75], s77 offset:0x0", "s_load_dword s93, s[74:75], s77 offset:0x40"])


def states(dx, dy):
    """Tip states of the op whose descriptor is s<dx>, s<dy>: tip A always, B and C on their flags."""
    if "nostate" in OPTS:   # states from arithmetic, no memory (results wrong)
        return ["s_lshr_b32 s87, s%d, 16" % dx, "v_add_u32_e32 v72, s87, v5", "v_and_b32_e32 v72, 3, v72", "v_add_u32_e32 v73, s87, v6",
                "v_and_b32_e32 v73, 3, v73", "v_mov_b32_e32 v74, v73", "v_mov_b32_e32 v75, v72", "v_mov_b32_e32 v76, v72",
                "v_mov_b32_e32 v77, v73"]
    return ["s_lshr_b32 s87, s%d, 16" % dx, "s_mul_i32 s87, s87, s80", "v_add_u32_e32 v78, s87, v5", "v_add_u32_e32 v79, s87, v6",
            "global_load_ubyte v72, v78, s[78:79]", "global_load_ubyte v73, v79, s[78:79]",
            "s_bitcmp1_b32 s%d, 13" % dx, "s_cbranch_scc0 1f",
            "s_and_b32 s87, s%d, 0xffff" % dy, "s_mul_i32 s87, s87, s80", "v_add_u32_e32 v78, s87, v5", "v_add_u32_e32 v79, s87, v6",
            "global_load_ubyte v74, v78, s[78:79]", "global_load_ubyte v75, v79, s[78:79]",
            "1:", "s_bitcmp1_b32 s%d, 14" % dx, "s_cbranch_scc0 2f",
            "s_lshr_b32 s87, s%d, 16" % dy, "s_mul_i32 s87, s87, s80", "v_add_u32_e32 v78, s87, v5", "v_add_u32_e32 v79, s87, v6",
            "global_load_ubyte v76, v78, s[78:79]", "global_load_ubyte v77, v79, s[78:79]", "2:"]


def deep_addr(slot_reg):
    return ["s_lshl_b32 s87, %s, 6" % slot_reg, "v_add_u32_e32 v78, s87, %[deep]"]


def block_io(op, base):
    return ["%s v78, v[%d:%d], off offset:%d" % (op, base + 4 * i, base + 4 * i + 3, 16 * i) if op.startswith("scratch_store")
            else "%s v[%d:%d], v78, off offset:%d" % (op, base + 4 * i, base + 4 * i + 3, 16 * i) for i in range(4)]


def push_block(label):
    """The op sets the accumulator aside first (descriptor bits 8:4 = slot + 1, in s88): slot <- P a."""
    return p_load() + ["s_waitcnt lgkmcnt(0)"] + ROTATE + P_ADV + [
        "s_cmp_eq_u32 s88, 1", "s_cbranch_scc0 %s_deep" % label] + matvec(ST0) + ["s_branch %s_pushed" % label,
        "%s_deep:" % label] + matvec(X) + ["s_add_i32 s89, s88, -2"] + deep_addr("s89") + block_io("scratch_store_dwordx4", X) + [
        "%s_pushed:" % label]


def generate():
    L = []
    add = L.extend
    add(["; ---- set-up -------------------------------------------------------------------------------------------",
         "s_mov_b32 s83, %[nw]", "s_mov_b64 s[84:85], %[wops]", "s_mov_b64 s[74:75], %[pm]", "s_mov_b32 s76, %[ctoff]",
         "s_mov_b32 s77, 0", "s_mov_b64 s[78:79], %[msa]", "s_mov_b32 s80, %[L]", "s_mov_b32 s81, %[tip]",
         "s_mov_b32 s96, 0x2ff00000", "s_movk_i32 s97, 0x100",
         "v_mbcnt_lo_u32_b32 v5, -1, 0", "v_mbcnt_hi_u32_b32 v5, -1, v5", "v_add_u32_e32 v5, %[site0], v5",
         "v_add_u32_e32 v6, 64, v5", "v_min_u32_e32 v5, %[last], v5", "v_min_u32_e32 v6, %[last], v6", "v_mov_b32_e32 v7, 0"])
    for i in range(8):
        add(["v_mov_b32_e32 v%d, 0" % (A + 2 * i), "v_mov_b32_e32 v%d, 0x3ff00000" % (A + 2 * i + 1)])
    add(["s_cmp_lt_i32 s83, 1", "s_cbranch_scc1 lh_walk_end",
         "s_lshl_b32 s98, s83, 3", "s_add_i32 s98, s98, -8",
         "s_load_dwordx2 s[68:69], s[84:85], 0x0", "s_min_u32 s87, s98, 8", "s_load_dwordx2 s[72:73], s[84:85], s87",
         "s_min_u32 s86, s98, 16", "s_waitcnt lgkmcnt(0)"])
    add(states(68, 69))
    add(["s_mov_b32 s82, 0",
         "; ---- one op per iteration ---------------------------------------------------------------------------",
         "lh_walk_top:", "s_waitcnt vmcnt(0)", "s_and_b32 s87, s68, 7",
         "s_cmp_eq_u32 s87, 1", "s_cbranch_scc1 lh_walk_tip", "s_cmp_eq_u32 s87, 4", "s_cbranch_scc1 lh_walk_ctab",
         "s_cmp_eq_u32 s87, 2", "s_cbranch_scc1 lh_walk_pop", "s_cmp_eq_u32 s87, 3", "s_cbranch_scc1 lh_walk_ctip"])
    # cherry: a = tipcol_A * tipcol_B (the accumulator pushed first if the op says so)
    add(["; cherry", "s_bfe_u32 s88, s68, 0x50004", "s_cmp_eq_u32 s88, 0", "s_cbranch_scc1 lh_walk_cherry_np"])
    add(push_block("lh_walk_cherry"))
    add(tip_column(U, 72, 73, TIP_A) + tip_column(X, 74, 75, TIP_B) + ["s_waitcnt lgkmcnt(0)"] + PREFETCH + product(U, X))
    add(states(70, 71) + ["s_branch lh_walk_tail"])
    add(["lh_walk_cherry_np:"] + tip_column(U, 72, 73, TIP_A) + tip_column(X, 74, 75, TIP_B) + ["s_waitcnt lgkmcnt(0)"] + ROTATE +
        PREFETCH + product(U, X) + states(70, 71) + ["s_branch lh_walk_tail"])
    # cherry table x tip column
    add(["; cherry table x tip column", "lh_walk_ctip:", "s_bfe_u32 s88, s68, 0x50004", "s_cmp_eq_u32 s88, 0",
         "s_cbranch_scc1 lh_walk_ctip_np"])
    add(push_block("lh_walk_ctip"))
    add(table_entry(U) + tip_column(X, 76, 77, TIP_C) + ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + PREFETCH + product(U, X))
    add(states(70, 71) + ["s_branch lh_walk_tail"])
    add(["lh_walk_ctip_np:"] + table_entry(U) + tip_column(X, 76, 77, TIP_C) + ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + ROTATE + PREFETCH +
        product(U, X) + states(70, 71) + ["s_branch lh_walk_tail"])
    # tip into accumulator: a = tipcol_A * (P a)
    add(["; tip into accumulator", "lh_walk_tip:"] + p_load() + tip_column(U, 72, 73, TIP_A) + ["s_waitcnt lgkmcnt(0)"] + ROTATE + P_ADV +
        PREFETCH + states(70, 71) + matvec(X) + product(U, X) + ["s_branch lh_walk_tail"])
    # cherry table into accumulator: a = table * (P a)
    add(["; cherry table into accumulator", "lh_walk_ctab:"] + p_load() + table_entry(U) + ["s_waitcnt lgkmcnt(0)"] + ROTATE + P_ADV + PREFETCH +
        matvec(X) + ["s_waitcnt vmcnt(0)"] + product(U, X) + states(70, 71) + ["s_branch lh_walk_tail"])
    # pop: a = pending sibling * (P a)
    add(["; pop", "lh_walk_pop:"] + p_load() + ["s_bfe_u32 s88, s68, 0x40009", "s_cmp_eq_u32 s88, 0", "s_cbranch_scc1 lh_walk_pop0",
                                                "s_add_i32 s89, s88, -1"] + deep_addr("s89") + block_io("scratch_load_dwordx4", U) +
        ["s_waitcnt lgkmcnt(0)"] + ROTATE + P_ADV + PREFETCH + matvec(X) + ["s_waitcnt vmcnt(0)"] + product(U, X) + states(70, 71) +
        ["s_branch lh_walk_tail",
         "lh_walk_pop0:", "s_waitcnt lgkmcnt(0)"] + ROTATE + P_ADV + PREFETCH + states(70, 71) + matvec(X) + product(ST0, X))
    # tail: 2^256 rescaling test on the high words (libpll's per-site scalers), next op
    hi = lambda s: [A + 8 * s + 2 * i + 1 for i in range(4)]
    add(["; ---- rescaling test, next op ------------------------------------------------------------------------",
         "lh_walk_tail:"])
    for s, t in ((0, 78), (1, 79)):
        h = hi(s)
        add(["v_max_u32_e32 v%d, v%d, v%d" % (t, h[0], h[1]), "v_max3_u32 v%d, v%d, v%d, v%d" % (t, h[2], h[3], t)])
    add(["v_min_u32_e32 v56, v78, v79", "v_cmp_gt_u32_e32 vcc, s96, v56", "s_cbranch_vccnz lh_walk_rescale",
         "lh_walk_back:", "s_mov_b64 s[68:69], s[70:71]", "s_add_i32 s82, s82, 1", "s_cmp_lt_i32 s82, s83",
         "s_cbranch_scc1 lh_walk_top", "s_branch lh_walk_end",
         "lh_walk_rescale:"])
    for s, t, inc in ((0, 78, "1"), (1, 79, "0x10000")):
        add(["v_cmp_gt_u32_e32 vcc, s96, v%d" % t, "s_nop 1", "s_and_saveexec_b64 s[94:95], vcc"])
        add(["v_ldexp_f64 %s, %s, s97" % (pair(A + 8 * s + 2 * i), pair(A + 8 * s + 2 * i)) for i in range(4)])
        add(["v_add_u32_e32 v7, %s, v7" % inc, "s_mov_b64 exec, s[94:95]"])
    add(["s_branch lh_walk_back",
         "; ---- results to the private array ------------------------------------------------------------------",
         "lh_walk_end:", "s_waitcnt vmcnt(0) lgkmcnt(0)"])
    add(["scratch_store_dwordx4 %%[out], v[%d:%d], off offset:%d" % (A + 4 * i, A + 4 * i + 3, 16 * i) for i in range(4)])
    add(["scratch_store_dword %[out], v7, off offset:64", "s_waitcnt vmcnt(0)"])
    return L


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = generate()
    out = os.environ.get("LH_ASM_OUT") or os.path.join(root, "linearham_amd", "csrc", "lh_prune_walk_asm.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_walk_asm.py (register map and rationale there) -- do not edit by hand.\n")
        f.write("// gfx950 assembly of K1's schedule walk, two sites per lane, alignments without N: the body of one asm statement.\n")
        import re
        for ln in lines:
            ln = ln.replace("%%", "%")
            ln = re.sub(r"(lh_walk_\w+)", r"\1_%=", ln)   # one copy of the labels per instantiation of the statement
            f.write('"%s\\n"\n' % ln)
    n_v = sum(1 for l in lines if l.startswith("v_"))
    print("wrote %s: %d lines (%d vector instructions in the text)" % (out, len(lines), n_v))


if __name__ == "__main__":
    main()
