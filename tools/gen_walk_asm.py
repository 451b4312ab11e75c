#!/usr/bin/env python3
"""Writes linearham_amd/csrc/lh_prune_walk_asm_s2.inc, _s2n.inc, _s1.inc, _s1n.inc (+ the clobber lists _clobbers_s2.inc,
_clobbers_s1.inc): the gfx950 assembly of K1's schedule walk for two sites per lane and for one (the wave that carries a
remainder of up to 64 patterns), for alignments without N and (n) for alignments that mix N with bases, each used by
prune_wave_asm through one inline-asm statement.  (Round 3 also generated a four-sites-per-lane walk, one with the tip
columns gathered from the scratch region and one that walks a schedule segment per statement; all three were measured
slower and left the product with round 4 -- git 0b72f3b has them.)  tests/test_structured_cpu.py checks that this script
reproduces the committed files byte for byte.  The text is generated because the 4x4 mat-vec and the element-wise products
are the same row pattern over four register blocks and S sites; everything else is written out below once.  Run from the
repo root after editing:    python tools/gen_walk_asm.py

Why assembly: the compiler's loop carried ~145 instructions per op (copies at joins, flag juggling, spills) where
~85 are needed, and with four sites per lane it spilled 50 registers; here every register is placed by hand.

Register map inside the statement, S sites per lane (clobbers v5 .. v<last>, s8-s31 and s36-s99):
  scal   1 reg (v5)            scaler counts, two sites per register (16 bits each)
  c128 c256 (v6 v7)            the constants 128, 256 (table look-up of alignments without N); N-aware walk: v6 = LDS address of
                               the four ones an N tip reads
  a      8S regs (v8 ..)       running CLV: site s in 8 consecutive registers (four doubles)
  st0    8S                    pending sibling of stack slot 0 (already multiplied by its branch matrix)
  st1    8S                    ... of stack slot 1 (round 4; deeper slots live in scratch memory)
  u      8S                    tip column / table entry / deep-slot sibling, both sites (their loads are in flight together)
  x      8                     P a of ONE site: the sites' mat-vecs run one after the other, each followed by its product; between
                               ops the block is free and serves as the temporaries of the address arithmetic and of the
                               rescaling test (a memory instruction has read its address registers when it issues)
  s[8:15] s[16:23] s[24:31]    tip states of the op's tips A, B, C as BIT PLANES: per tip the wave's 128 sites take 32 bytes of
                               the family's 2-bit MSA (lh_family_create: planes[tip][block of 128 sites][site set s][bit]),
                               one s_load_dwordx8 per tip and op; lane l's state of site set s is bit l of the two 64-bit
                               masks s[.. + 4 s : .. + 4 s + 1] (bit 0) and s[.. + 4 s + 2 : .. + 4 s + 3] (bit 1)
  s[36:67] P (row-major)      s[68:69] this op's descriptor      s[70:71] the next op's      s[72:73] the one after (in flight)
  s[74:75] scratch region (P-matrices at 0, tables at ctoff)   s76 next table offset   s77 next matrix offset
  s[78:79] state planes of the wave's block, tip 0   s80 bytes per tip   s81 LDS address of the tip table   s82 op counter   s83 op count   s[84:85] descriptors
  s86 offset of the descriptor to prefetch   s98 offset of the last descriptor   s87..s93, s99 temporaries
  s[94:95] exec save   s96 0x2ff00000 (high word of 2^-256)   s97 256
S = 2: scal v5, c128 v6, c256 v7, a v[8:23], st0 v[24:39], st1 v[40:55], u v[56:71], x v[72:79] (80 registers: six waves per
SIMD).  S = 1: a v[8:15], st0 v[16:23], st1 v[24:31], u v[32:39], x v[40:47].
Two stack slots in registers (round 4): configs[2]-like trees push 8.2 times per walk at level 0, 4.5 times at level 1 and 0.3
times deeper; with level 1 in scratch memory its 4 KB per wave and push went out to HBM and came back (tables and matrices
stream through L2 in between): 9 % of K1's time and a quarter of its HBM-side traffic (profiles/r04_k1_programme.txt).  The
sixteen registers come from computing one site's mat-vec at a time (x: 8 instead of 16) and from the temporaries moving into x.
Round 4: the tip states come through the SCALAR memory path.  Round 3's walk fetched them with one global_load_ubyte per
site and tip child; with them replaced by arithmetic the walk ran 12.5 % faster although it issued 9 % more vector
instructions (profiles/r03_k1_state_loads.txt): the byte loads' issue through the CU's one vector-memory address unit,
not their latency, was the cost.  An alignment without N needs 2 bits per state, the wave's sites are 128 consecutive
patterns, and the tip is wave-uniform: 32 bytes per tip and op, fetched into SGPRs, from which a lane takes its bits with
v_cndmask (the mask operand IS the 64-bit plane).  The only vector-memory instructions left in the walk are the cherry-table
gathers and the rare deep stack slot.
"""
import os
import re

OPTS = set(os.environ.get("LH_ASM_OPTS", "").split(","))   # timing experiments (variants are built into lib_exp/, never the product)
P = 36
SA, SB, SC = 8, 16, 24       # first SGPR of the state planes of the op's tips A, B, C (eight registers each; s32 is the stack pointer)
# N-aware walk (alignments that mix N with bases): a third plane per site set says "this lane's state is N" (the two state
# bits are 0 there), 12 registers per tip: A in s[8:19], B in s[20:31]; tip C of a table-x-tip op is fetched into the
# P-matrix registers s[36:47], which such an op no longer needs once a push (if any) is done: the load is in flight during
# the table look-up's address arithmetic.  An N tip reads the four ones behind the tip table(s) instead of a
# column (lh_prune.hip tip_column); a cherry table has 5 x 5 entries.
NSA, NSB = 8, 20


class Regs:
    def __init__(self, S):
        self.S = S
        self.scal = 5
        self.c128 = 6
        self.usite = 6                        # (N-aware walk: LDS address of the ones; that walk has no use for c128 / c256)
        self.c256 = 7
        self.A = 8
        self.ST0 = self.A + 8 * S
        self.ST1 = self.ST0 + 8 * S
        self.U = self.ST1 + 8 * S
        self.X = self.U + 8 * S
        # temporaries live in the x block while it holds no mat-vec result
        self.tmp = self.X
        self.t2 = self.X + 2
        self.t3 = self.X + 4
        self.last = self.X + 7


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


def weave(valu, other, ratio=2):
    """The two independent instruction lists as one stream, `ratio` vector instructions per other one: a wave whose
    scalar / memory instructions sit between its vector ones keeps feeding the vector unit while they issue."""
    if "noweave" in OPTS:
        return other + valu
    out, i, j = [], 0, 0
    while i < len(valu) or j < len(other):
        out += valu[i:i + ratio]
        i += ratio
        if j < len(other):
            out.append(other[j])
            j += 1
    return out


def spair(r):
    return "s[%d:%d]" % (r, r + 1)


P_ADV = ["s_add_i32 s77, s77, 0x80"]               # the op's matrix is in registers: s77 names the next one
ROTATE = ["s_mov_b64 s[70:71], s[72:73]"]          # the next op's descriptor has arrived (behind an lgkmcnt(0))
PREFETCH = [                                       # the descriptor two ops ahead; the next P-matrix into the scalar cache
    "s_load_dwordx2 s[72:73], s[84:85], s86", "s_add_i32 s86, s86, 8", "s_min_u32 s86, s86, s98"] + (
    [] if "notouch" in OPTS else ["s_load_dword s99, s[74:75], s77 offset:0x0", "s_load_dword s93, s[74:75], s77 offset:0x40"])


class Gen:
    def __init__(self, S=2, n_aware=False):
        self.r = Regs(S)
        self.S = S
        self.n_aware = n_aware
        self.sa, self.sb = (NSA, NSB) if n_aware else (SA, SB)
        self.sc = P if n_aware else SC

    def matvec(self, dst):
        """dst <- P a for all sites, rows interleaved (4 S independent chains), in the order of lh::matvec:
        fma(p3, a3, fma(p2, a2, fma(p1, a1, p0 * a0)))."""
        out = []
        for j in range(4):
            for s in range(self.S):
                for i in range(4):
                    d = pair(dst + 8 * s + 2 * i)
                    a = pair(self.r.A + 8 * s + 2 * j)
                    p = spair(P + 2 * (4 * i + j))
                    out.append("v_mul_f64 %s, %s, %s" % (d, a, p) if j == 0 else "v_fmac_f64_e32 %s, %s, %s" % (d, p, a))
        return out

    def product(self, f1, f2):
        return ["v_mul_f64 %s, %s, %s" % (pair(self.r.A + 2 * i), pair(f1 + 2 * i), pair(f2 + 2 * i)) for i in range(4 * self.S)]

    def mv_prod(self, f, wait=None):
        """a <- f o (P a), one site after the other: x <- P a_s (four independent chains, the order of lh::matvec), then
        a_s <- f_s o x.  wait: s_waitcnt placed before the product of each site (f's loads: site 0's were issued first)."""
        X, A = self.r.X, self.r.A
        out = []
        for s in range(self.S):
            for j in range(4):
                for i in range(4):
                    d, a, p = pair(X + 2 * i), pair(A + 8 * s + 2 * j), spair(P + 2 * (4 * i + j))
                    out.append("v_mul_f64 %s, %s, %s" % (d, a, p) if j == 0 else "v_fmac_f64_e32 %s, %s, %s" % (d, p, a))
            if wait:
                out.append(wait[s])
            out += ["v_mul_f64 %s, %s, %s" % (pair(A + 8 * s + 2 * i), pair(f + 8 * s + 2 * i), pair(X + 2 * i)) for i in range(4)]
        return out

    def p_load(self):
        if "phit" in OPTS:   # every matrix from one of two hot lines (results wrong)
            return ["s_and_b32 s92, s77, 0x80", "s_load_dwordx16 s[36:51], s[74:75], s92 offset:0x0",
                    "s_load_dwordx16 s[52:67], s[74:75], s92 offset:0x40"]
        return ["s_load_dwordx16 s[36:51], s[74:75], s77 offset:0x0", "s_load_dwordx16 s[52:67], s[74:75], s77 offset:0x40"]

    def plane(self, st, s, bit):
        """The 64-bit mask of bit `bit` (2: the N flag of the N-aware walk) of site set s in the state planes that start at
        SGPR `st`."""
        per_set = 6 if self.n_aware else 4
        return "s[%d:%d]" % (st + per_set * s + 2 * bit, st + per_set * s + 2 * bit + 1)

    def tip_column(self, dst, st, tip_expr):
        """dst <- tip-table columns of the tip whose number `tip_expr` leaves in s87, for the states in the planes s<st>..:
        LDS address = table + 128 tip + 32 state."""
        r = self.r
        out = tip_expr + ["s_lshl_b32 s87, s87, 7", "s_add_i32 s87, s87, s81"]
        for s in range(self.S):
            out += ["v_cndmask_b32_e64 v%d, 0, 32, %s" % (r.tmp + s, self.plane(st, s, 0)),
                    "v_cndmask_b32_e64 v%d, 0, 64, %s" % (r.t2 + s, self.plane(st, s, 1)),
                    "v_add3_u32 v%d, v%d, v%d, s87" % (r.tmp + s, r.tmp + s, r.t2 + s)]
            if self.n_aware:   # state N: the vector of ones (its LDS address in v<usite>)
                out += ["v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (r.tmp + s, r.tmp + s, r.usite, self.plane(st, s, 2))]
        for s in range(self.S):   # s81 = LDS address of the tip table
            out += ["ds_read_b128 v[%d:%d], v%d" % (dst + 8 * s, dst + 8 * s + 3, self.r.tmp + s),
                    "ds_read_b128 v[%d:%d], v%d offset:16" % (dst + 8 * s + 4, dst + 8 * s + 7, self.r.tmp + s)]
        return out

    def table_entry(self, dst):
        """dst <- entry (state A, state B) of the next cherry table (global memory, written by this workgroup's prologue)."""
        toff = "%[ctoff]" if "tabhit" in OPTS else "s76"   # tabhit: every look-up in table 0 (results wrong)
        r = self.r
        out = []
        pol = " nt" if "tabnt" in OPTS else " sc1" if "tabsc1" in OPTS else " sc0 sc1" if "tabsc" in OPTS else ""   # cache-policy experiments
        for s in range(self.S):
            if self.n_aware:   # 5 x 5 entries: byte offset = table + 160 state A + 32 state B, N = 4 (s92 holds 160)
                out += ["v_cndmask_b32_e64 v%d, 0, 32, %s" % (r.tmp + s, self.plane(self.sb, s, 0)),
                        "v_cndmask_b32_e64 v%d, 0, 64, %s" % (r.t2 + s, self.plane(self.sb, s, 1)),
                        "v_add3_u32 v%d, v%d, v%d, %s" % (r.tmp + s, r.tmp + s, r.t2 + s, toff),
                        "v_cndmask_b32_e64 v%d, 0, 1, %s" % (r.t2 + s, self.plane(self.sb, s, 2)),
                        "v_lshl_add_u32 v%d, v%d, 7, v%d" % (r.tmp + s, r.t2 + s, r.tmp + s),
                        "v_cndmask_b32_e64 v%d, 0, 1, %s" % (r.t2 + s, self.plane(self.sa, s, 0)),
                        "v_cndmask_b32_e64 v%d, 0, 2, %s" % (r.t3 + s, self.plane(self.sa, s, 1)),
                        "v_or_b32_e32 v%d, v%d, v%d" % (r.t2 + s, r.t2 + s, r.t3 + s),
                        "v_cndmask_b32_e64 v%d, v%d, 4, %s" % (r.t2 + s, r.t2 + s, self.plane(self.sa, s, 2)),
                        "v_mad_u32_u24 v%d, v%d, s92, v%d" % (r.tmp + s, r.t2 + s, r.tmp + s)]
            else:              # byte offset = table + 128 state A + 32 state B
                out += ["v_cndmask_b32_e64 v%d, 0, 32, %s" % (r.tmp + s, self.plane(self.sb, s, 0)),
                        "v_cndmask_b32_e64 v%d, 0, 64, %s" % (r.t2 + s, self.plane(self.sb, s, 1)),
                        "v_add3_u32 v%d, v%d, v%d, %s" % (r.tmp + s, r.tmp + s, r.t2 + s, toff),
                        "v_cndmask_b32_e64 v%d, 0, v%d, %s" % (r.t2 + s, r.c128, self.plane(self.sa, s, 0)),
                        "v_cndmask_b32_e64 v%d, 0, v%d, %s" % (r.t3 + s, r.c256, self.plane(self.sa, s, 1)),
                        "v_add3_u32 v%d, v%d, v%d, v%d" % (r.tmp + s, r.tmp + s, r.t2 + s, r.t3 + s)]
        for s in range(self.S):
            out += ["global_load_dwordx4 v[%d:%d], v%d, s[74:75]%s" % (dst + 8 * s, dst + 8 * s + 3, r.tmp + s, pol),
                    "global_load_dwordx4 v[%d:%d], v%d, s[74:75] offset:16%s" % (dst + 8 * s + 4, dst + 8 * s + 7, r.tmp + s, pol)]
        return out + ["s_add_i32 s76, s76, %s" % ("0x320" if self.n_aware else "0x200")]

    def plane_load(self, dst):
        """The planes of the tip whose number is in s87 into s<dst>.. (s80: bytes per tip)."""
        o = ["s_mul_i32 s87, s87, s80", "s_load_dwordx8 s[%d:%d], s[78:79], s87" % (dst, dst + 7)]
        if self.n_aware:
            o += ["s_load_dwordx4 s[%d:%d], s[78:79], s87 offset:0x20" % (dst + 8, dst + 11)]
        return o

    def states(self, dx, dy, split=False):
        """Tip states of the op whose descriptor is s<dx>, s<dy>, as bit planes into s[8:31]: tip A always, B and C on
        their flags (the N-aware walk fetches C later: table_x_tip).  Scalar loads: they are complete behind the
        s_waitcnt lgkmcnt(0) at the top of the next op."""
        if "nostate" in OPTS:   # no state traffic at all (the planes keep what they hold; results wrong)
            return ([], []) if split else []
        part_a = ["s_lshr_b32 s87, s%d, 16" % dx] + self.plane_load(self.sa)
        part_bc = ["s_bitcmp1_b32 s%d, 13" % dx, "s_cbranch_scc0 1f", "s_and_b32 s87, s%d, 0xffff" % dy] + self.plane_load(self.sb) + ["1:"]
        if not self.n_aware:
            part_bc += ["s_bitcmp1_b32 s%d, 14" % dx, "s_cbranch_scc0 2f", "s_lshr_b32 s87, s%d, 16" % dy] + self.plane_load(self.sc) + ["2:"]
        if split:
            return part_a, part_bc
        return part_a + part_bc

    def deep_addr(self, slot_reg):
        sh = {1: 5, 2: 6}[self.S]                         # a slot is 32 S bytes per lane
        return ["s_lshl_b32 s87, %s, %d" % (slot_reg, sh), "v_add_u32_e32 v%d, s87, %%[deep]" % self.r.tmp]

    def block_io(self, op, base):
        if "nodeep" in OPTS:   # no traffic for stack slots beyond the first (results wrong): what those slots cost
            return []
        t = self.r.tmp
        return ["%s v%d, v[%d:%d], off offset:%d" % (op, t, base + 4 * i, base + 4 * i + 3, 16 * i) if op.startswith("scratch_store")
                else "%s v[%d:%d], v%d, off offset:%d" % (op, base + 4 * i, base + 4 * i + 3, t, 16 * i) for i in range(2 * self.S)]

    def push_block(self, label):
        """The op sets the accumulator aside first (descriptor bits 8:4 = level + 1, in s88): slot <- P a -- levels 0 and 1 are
        register blocks, deeper ones scratch memory (through the u block, free at this point of the op)."""
        r = self.r
        return self.p_load() + ["s_waitcnt lgkmcnt(0)"] + ROTATE + P_ADV + [
            "s_cmp_eq_u32 s88, 1", "s_cbranch_scc0 %s_lvl1" % label] + self.matvec(r.ST0) + ["s_branch %s_pushed" % label,
            "%s_lvl1:" % label, "s_cmp_eq_u32 s88, 2", "s_cbranch_scc0 %s_deep" % label] + self.matvec(r.ST1) + ["s_branch %s_pushed" % label,
            "%s_deep:" % label] + self.matvec(r.U) + ["s_add_i32 s89, s88, -3"] + self.deep_addr("s89") + \
            self.block_io("scratch_store_dwordx4", r.U) + ["s_waitcnt vmcnt(0)", "%s_pushed:" % label]

    def generate(self):
        r, S = self.r, self.S
        assert S in (1, 2)
        A, ST0, ST1, X, U = r.A, r.ST0, r.ST1, r.X, r.U
        tip_a, tip_b, tip_c = ["s_lshr_b32 s87, s68, 16"], ["s_and_b32 s87, s69, 0xffff"], ["s_lshr_b32 s87, s69, 16"]
        L = []
        add = L.extend
        add(["; ---- set-up -------------------------------------------------------------------------------------------",
             "s_mov_b32 s83, %[nw]", "s_mov_b64 s[84:85], %[wops]", "s_mov_b64 s[74:75], %[pm]", "s_mov_b32 s76, %[ctoff]",
             "s_mov_b32 s77, 0", "s_mov_b64 s[78:79], %[planes]", "s_mov_b32 s80, %[pstride]", "s_mov_b32 s81, %[tip]",
             "s_mov_b32 s96, 0x2ff00000", "s_movk_i32 s97, 0x100"])
        add(["v_mov_b32_e32 v%d, 0" % r.scal])
        if self.n_aware:
            add(["v_mov_b32_e32 v%d, %%[ones]" % r.usite, "s_movk_i32 s92, 0xa0"])
        else:
            add(["v_mov_b32_e32 v%d, 0x80" % r.c128, "v_mov_b32_e32 v%d, 0x100" % r.c256])
        for i in range(4 * S):
            add(["v_mov_b32_e32 v%d, 0" % (A + 2 * i), "v_mov_b32_e32 v%d, 0x3ff00000" % (A + 2 * i + 1)])
        add(["s_cmp_lt_i32 s83, 1", "s_cbranch_scc1 lh_walk_end",
             "s_lshl_b32 s98, s83, 3", "s_add_i32 s98, s98, -8",
             "s_load_dwordx2 s[68:69], s[84:85], 0x0", "s_min_u32 s87, s98, 8", "s_load_dwordx2 s[72:73], s[84:85], s87",
             "s_min_u32 s86, s98, 16", "s_waitcnt lgkmcnt(0)"])
        add(self.states(68, 69))
        add(["s_mov_b32 s82, 0",
             "; ---- one op per iteration ---------------------------------------------------------------------------",
             "lh_walk_top:", "s_waitcnt lgkmcnt(0)", "s_and_b32 s87, s68, 7",
             "s_cmp_eq_u32 s87, 1", "s_cbranch_scc1 lh_walk_tip", "s_cmp_eq_u32 s87, 4", "s_cbranch_scc1 lh_walk_ctab",
             "s_cmp_eq_u32 s87, 2", "s_cbranch_scc1 lh_walk_pop", "s_cmp_eq_u32 s87, 3", "s_cbranch_scc1 lh_walk_ctip"])
        # cherry: a = tipcol_A * tipcol_B (the accumulator pushed first if the op says so); the second column lands in a itself
        cherry = self.tip_column(U, self.sa, tip_a) + self.tip_column(A, self.sb, tip_b)
        add(["; cherry", "s_bfe_u32 s88, s68, 0x50004", "s_cmp_eq_u32 s88, 0", "s_cbranch_scc1 lh_walk_cherry_np"])
        add(self.push_block("lh_walk_cherry"))
        tipwait = "s_waitcnt lgkmcnt(0)"
        add(cherry + [tipwait] + PREFETCH + self.product(U, A) + self.states(70, 71) + ["s_branch lh_walk_tail"])
        add(["lh_walk_cherry_np:"] + cherry + [tipwait] + ROTATE + PREFETCH + self.product(U, A) + self.states(70, 71) +
            ["s_branch lh_walk_tail"])
        # cherry table x tip column (the column lands in a: the old accumulator has been pushed or never was)
        # (N-aware: tip C's planes into the P-matrix registers, free here; requested before the look-up, awaited behind it)
        c_load = (["s_lshr_b32 s87, s69, 16"] + self.plane_load(self.sc)) if self.n_aware else []
        c_wait = ["s_waitcnt lgkmcnt(0)"] if self.n_aware else []
        ctip = c_load + self.table_entry(U) + c_wait + self.tip_column(A, self.sc, tip_c)
        add(["; cherry table x tip column", "lh_walk_ctip:", "s_bfe_u32 s88, s68, 0x50004", "s_cmp_eq_u32 s88, 0",
             "s_cbranch_scc1 lh_walk_ctip_np"])
        add(self.push_block("lh_walk_ctip"))
        add(ctip + ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + PREFETCH + self.product(U, A) + self.states(70, 71) + ["s_branch lh_walk_tail"])
        add(["lh_walk_ctip_np:"] + ctip + ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + ROTATE + PREFETCH + self.product(U, A) + self.states(70, 71) +
            ["s_branch lh_walk_tail"])
        # tip into accumulator: a = tipcol_A * (P a)
        st_a, st_bc = self.states(70, 71, split=True)
        half = 20 if S == 2 else 8
        mvp = self.mv_prod(U)
        add(["; tip into accumulator", "lh_walk_tip:"] + self.p_load() + self.tip_column(U, self.sa, tip_a) + ["s_waitcnt lgkmcnt(0)"] +
            ROTATE + weave(mvp[:half], P_ADV + PREFETCH + st_a) + st_bc + mvp[half:] + ["s_branch lh_walk_tail"])
        # cherry table into accumulator: a = table * (P a); a site's entry is awaited in front of its product (site 0's two
        # gathers were issued first: vmcnt counts the loads still out)
        waits = ["s_waitcnt vmcnt(%d)" % (2 * (S - 1 - s)) for s in range(S)]
        mvw = self.mv_prod(U, waits)
        add(["; cherry table into accumulator", "lh_walk_ctab:"] + self.p_load() + self.table_entry(U) + ["s_waitcnt lgkmcnt(0)"] + ROTATE +
            weave(mvw, P_ADV + PREFETCH, 4) + self.states(70, 71) + ["s_branch lh_walk_tail"])
        # pop: a = pending sibling * (P a); level 0 / 1 from their register blocks, deeper ones from scratch memory through u
        mv0, mv1 = self.mv_prod(ST0), self.mv_prod(ST1)
        add(["; pop", "lh_walk_pop:"] + self.p_load() + ["s_bfe_u32 s88, s68, 0x40009", "s_cmp_eq_u32 s88, 0",
                                                        "s_cbranch_scc1 lh_walk_pop0", "s_cmp_eq_u32 s88, 1", "s_cbranch_scc1 lh_walk_pop1",
                                                        "s_add_i32 s89, s88, -2"] +
            self.deep_addr("s89") + self.block_io("scratch_load_dwordx4", U) + ["s_waitcnt lgkmcnt(0)"] + ROTATE + P_ADV + PREFETCH +
            self.mv_prod(U, waits) + self.states(70, 71) + ["s_branch lh_walk_tail"])
        add(["lh_walk_pop1:", "s_waitcnt lgkmcnt(0)"] + ROTATE + weave(mv1[:half], P_ADV + PREFETCH + st_a) + st_bc + mv1[half:] +
            ["s_branch lh_walk_tail"])
        add(["lh_walk_pop0:", "s_waitcnt lgkmcnt(0)"] + ROTATE + weave(mv0[:half], P_ADV + PREFETCH + st_a) + st_bc + mv0[half:])
        # tail: 2^256 rescaling test on the high words (libpll's per-site scalers), next op.
        # The test runs after every FOURTH op and after the last one, and rescales until the largest entry is back above
        # 2^-256: multiplying by 2^256 is exact, an op lowers the largest entry by 2^-73 at worst (two tip columns across
        # 1e-6 branches at the slowest rate), so four ops stay 400 binades clear of the subnormals and the final (value,
        # count) pair is the one the test after every op leaves -- libpll's per-node scaling reaches the same pair
        # (round 4: six of ~65 vector instructions per op were this test).
        add(["; ---- rescaling test, next op ------------------------------------------------------------------------",
             "lh_walk_tail:", "s_add_i32 s87, s82, 1", "s_cmp_ge_i32 s87, s83", "s_cbranch_scc1 lh_walk_test",
             "s_and_b32 s87, s82, 3", "s_cmp_eq_u32 s87, 3", "s_cbranch_scc0 lh_walk_back",
             "lh_walk_test:"])
        for s in range(S):
            h = [A + 8 * s + 2 * i + 1 for i in range(4)]
            add(["v_max_u32_e32 v%d, v%d, v%d" % (r.tmp + s, h[0], h[1]),
                 "v_max3_u32 v%d, v%d, v%d, v%d" % (r.tmp + s, h[2], h[3], r.tmp + s)])
        add(["v_min_u32_e32 v%d, v%d, v%d" % (U, r.tmp, r.tmp + 1) if S > 1 else "v_mov_b32_e32 v%d, v%d" % (U, r.tmp)])   # the u block is free here
        add(["v_cmp_gt_u32_e32 vcc, s96, v%d" % U, "s_cbranch_vccnz lh_walk_rescale",
             "lh_walk_back:", "s_mov_b64 s[68:69], s[70:71]", "s_add_i32 s82, s82, 1", "s_cmp_lt_i32 s82, s83",
             "s_cbranch_scc1 lh_walk_top", "s_branch lh_walk_end",
             "lh_walk_rescale:"])
        for s in range(S):
            # (an all-zero CLV -- impossible data -- is rescaled once and left: its high words stay 0, the loop would not end)
            add(["v_cmp_gt_u32_e32 vcc, s96, v%d" % (r.tmp + s), "s_nop 1", "s_and_saveexec_b64 s[94:95], vcc"])
            add(["v_ldexp_f64 %s, %s, s97" % (pair(A + 8 * s + 2 * i), pair(A + 8 * s + 2 * i)) for i in range(4)])
            add(["v_add_u32_e32 v%d, %s, v%d" % (r.scal + s // 2, "0x10000" if s & 1 else "1", r.scal + s // 2), "s_mov_b64 exec, s[94:95]"])
        # again, unless nothing but zeros is left below the threshold (max high word 0)
        for s in range(S):
            h = [A + 8 * s + 2 * i + 1 for i in range(4)]
            add(["v_max_u32_e32 v%d, v%d, v%d" % (r.tmp + s, h[0], h[1]),
                 "v_max3_u32 v%d, v%d, v%d, v%d" % (r.tmp + s, h[2], h[3], r.tmp + s)])
            add(["v_cmp_eq_u32_e32 vcc, 0, v%d" % (r.tmp + s), "v_cndmask_b32_e64 v%d, v%d, -1, vcc" % (r.tmp + s, r.tmp + s)])
        add(["v_min_u32_e32 v%d, v%d, v%d" % (U, r.tmp, r.tmp + 1) if S > 1 else "v_mov_b32_e32 v%d, v%d" % (U, r.tmp),
             "v_cmp_gt_u32_e32 vcc, s96, v%d" % U, "s_cbranch_vccnz lh_walk_rescale"])
        add(["s_branch lh_walk_back",
             "; ---- results to the private array: a (32 S bytes), then the packed scaler counts ------------------",
             "lh_walk_end:", "s_waitcnt vmcnt(0) lgkmcnt(0)"])
        add(["scratch_store_dwordx4 %%[out], v[%d:%d], off offset:%d" % (A + 4 * i, A + 4 * i + 3, 16 * i) for i in range(2 * S)])
        add(["scratch_store_dword %%[out], v%d, off offset:%d" % (r.scal, 32 * S)])
        add(["s_waitcnt vmcnt(0)"])
        return L


def render(S=2, n_aware=False):
    """(text of lh_prune_walk_asm_s<S>[n].inc, text of lh_prune_walk_clobbers_s<S>.inc, summary line)"""
    g = Gen(S, n_aware)
    lines = g.generate()
    body = ["// GENERATED by tools/gen_walk_asm.py (register map and rationale there) -- do not edit by hand.\n",
            "// gfx950 assembly of K1's schedule walk, %d sites per lane, alignments %s: the body of one asm statement.\n" % (
                S, "that mix N with bases" if n_aware else "without N"),
            "// Vector registers v5 .. v%d (lh_prune_walk_clobbers_s%d.inc lists them for the statement).\n" % (g.r.last, S)]
    for ln in lines:
        ln = ln.replace("%%", "%")
        ln = re.sub(r"(lh_walk_\w+)", r"\1_%=", ln)   # one copy of the labels per instantiation of the statement
        body.append('"%s\\n"\n' % ln)
    clob = ["// GENERATED by tools/gen_walk_asm.py: registers the %d-site walk statement clobbers.\n" % S]
    regs = ['"v%d"' % i for i in range(5, g.r.last + 1)] + ['"s%d"' % i for i in list(range(SA, SC + 8)) + list(range(36, 100))]
    for i in range(0, len(regs), 16):
        clob.append(", ".join(regs[i:i + 16]) + (",\n" if i + 16 < len(regs) else "\n"))
    n_v = sum(1 for l in lines if l.startswith("v_"))
    return "".join(body), "".join(clob), "%d lines (%d vector instructions in the text), v5..v%d" % (len(lines), n_v, g.r.last)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outdir = os.environ.get("LH_ASM_OUT") or os.path.join(root, "linearham_amd", "csrc")
    for S in (2, 1):   # (one site per lane: the wave that carries a remainder of up to 64 patterns)
        for n_aware in (False, True):
            body, clob, summary = render(S, n_aware)
            out = os.path.join(outdir, "lh_prune_walk_asm_s%d%s.inc" % (S, "n" if n_aware else ""))
            with open(out, "w") as f:
                f.write(body)
            with open(os.path.join(outdir, "lh_prune_walk_clobbers_s%d.inc" % S), "w") as f:   # (the same registers in both)
                f.write(clob)
            print("wrote %s: %s" % (out, summary))


if __name__ == "__main__":
    main()
