#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 pair-count loops of K1 (k1_loop_gen.inc).

    python3 gen_k1_loop.py > k1_loop_gen.inc        (the Makefile does this; the .inc is committed as well)

One loop = one inline-asm statement that counts, for ONE wave and one range of 32-sample blocks,
    n(i, j) = #{s : pos_j(s) < edge_i(s)}     for the 32 gene rows i of a tile and the wave's 64 * RJ genes j
bit-sliced as in kernels.hip (count_pass): per row and block NB v_bitop3_b32 per chain (borrow of edge_i - pos_j,
plane by plane) and one v_bcnt_u32_b32.  What the hand schedule does that the compiler's did not (ISA of round 2):
  * no wait states between the plane steps (the compiler padded every asm statement with s_nop 0);
  * the tile operand (edge planes of the 32 rows, wave-uniform) is read from LDS one row AHEAD, quad by quad as the
    registers of the running row are consumed, with counted s_waitcnt lgkmcnt(N) -- no LDS latency in the row loop;
  * the tile operand reaches LDS by LDS-DMA (global_load_lds_dwordx4) one block ahead, into a 2-slot ring that belongs
    to the wave alone: no barrier anywhere, every wave is an independent work item;
  * the lane operand (pos planes of the wave's genes, 60-64 VGPRs) of the NEXT block is requested during the LAST row of
    the running block, each register quad as soon as its last reader has issued;
  * counted s_waitcnt vmcnt(N) throughout (the generator keeps the queues and asserts the loop invariant).

Register file of a loop (RJ = 4 genes per lane, one band edge per pass): acc 64 (two 16-bit counts per register: rows 2h,
2h+1), pos planes 64, row operand 16 + 4 (the quad that holds plane 0 is double-buffered), chains 4, addresses 2: v8..v161,
three waves per SIMD.  Tie-rich data runs the loop twice (hi planes, then lo planes: kernels.hip, k1w_pairs).
The library's loops: NB = 12, 15, 16 on the 16-plane layout -- P uint4 [nblk][4][Gp] (plane quad q of gene g in block b
at (b*4+q)*Gp + g), AL / AH uint4 [nblk][Gp][4], plane k in word (k + 15) % 16 -- and NB = 17, 18 on the big layout
(more than 65 535 genes: P [nblk][5][Gp], AL / AH [nblk][Gp][8], plane k in word k; 180 registers, two waves per SIMD).
`ties=True` (two edges and RJ = 2 inside one loop: round 2's tiling) and `ri=16` (four waves per SIMD) are kept for the
probe (tools/k1w_probe.hip); both measured slower and neither is generated for the library.
"""
import sys

RI = 32


def a_word(k):
    return (k + 15) & 15


class Loop:
    def __init__(self, nb, ties, name, lshl_add=True, opts=(), ri=RI, std=False):
        # opts: timing experiments only (tools/k1w_probe.hip) -- never set for the library's loops
        self.nb, self.ties, self.name, self.opts = nb, ties, name, set(opts)
        # big: more than 16 planes (more than 65 535 genes).  Five pos quads per gene and block, P [nblk][5][Gp]; edge rows
        # of 8 uint4 (20 words used, plane k in word k: the generator places the registers, no skew needed), A [nblk][Gp][8].
        self.ri = ri                       # gene rows per item: 32; 16 with std = the half-height items of a launch's last round
                                           # (same registers as the 32-row form, 32 counts); 16 without = the four-waves-per-SIMD
                                           # experiment of tools/k1w_probe.hip (128 registers)
        self.big = nb > 16
        assert not (self.big and ties)
        self.rj = 2 if ties else 4
        self.lines = []
        self.lq = []   # outstanding LDS reads, oldest first (tags)
        self.vq = []   # outstanding vector-memory operations, oldest first (tags)
        self.lshl_add = lshl_add
        self.label_n = 0
        self.in_loop = False
        # ---- VGPRs owned by the statement (physical) ----
        self.ACC = 8                       # v[8:71]: four tuples of 16, pinned outputs
        self.P = 72                        # pos planes: gene r, plane k at P + PSTR r + k   (bank = k % 4)
        nedge = 2 if ties else 1
        self.PSTR = 20 if self.big else 16
        self.LQ = 5 if self.big else 4     # pos quads per block in the layout
        self.ROWB = 128 if self.big else 64  # bytes of one edge row in memory and in LDS
        self.NDMA = self.ri * self.ROWB // 1024   # 1 KiB pieces per block and edge
        if self.big:
            top = 72 + self.PSTR * self.rj
            self.L = top
            self.VLDS = top + 4
            self.VA2 = top + 5             # (piece p > 0 of a DMA'd block: lane * 16 + 1024 p at VA2 + p - 1 -- see VAP)
            self.A = [top + 6]             # 20 words, plane k in word k
            self.vtop = top + 6 + 20 + 2   # + two more piece offsets
            self.VAP = [None, self.VA2, self.vtop - 2, self.vtop - 1]
            assert self.vtop <= 256 - 8, self.vtop
            for a in self.A:   # pos plane k in bank k % 4, edge plane k in bank (A + k) % 4
                assert a % 2 == 0 and a % 4 != 0
        elif self.ri == 16 and not std:   # 128 registers: 32 counts, the two addresses in the unused 16th word of the pos planes (NB <= 15)
            assert nb <= 15 and not ties
            self.P = 8 + 32
            self.A = [self.P + 64]
            self.L = self.A[0] + 20
            self.VLDS = self.P + 15
            self.VA2 = self.P + 31
            self.vtop = self.L + 4
            self.VAP = [None]
            assert self.vtop <= 128
        else:
            self.A = [72 + 16 * self.rj + 20 * e for e in range(nedge)]   # row operand words 0..15, +16..19 second copy of quad 3
            top = 72 + 16 * self.rj + 20 * nedge
            self.L = top                       # 4 chain registers
            self.VLDS = top + 4                # LDS byte address of the slot being read
            self.VA2 = top + 5                 # lane * 16 + 1024 (second half of a DMA'd block)
            self.vtop = top + 6
            self.VAP = [None, self.VA2]
            assert self.vtop <= 168 - 4, self.vtop
            # bank rule (measured, tools/microbench_bank.hip): a v_bitop3_b32 whose three sources sit in ONE bank issues
            # at half rate.  pos plane k is in bank k % 4, edge plane k in bank (A + k + 3) % 4: never the same.
            for a in self.A:
                assert a % 2 == 0 and (a + 3) % 4 != 0
        # ---- SGPRs owned by the statement ----
        s = 36
        self.SB = [s + 2 * q for q in range(self.LQ)]; s += 2 * self.LQ   # pos quad bases of the next block to load
        self.SA = [s, s + 2][:nedge]; s += 4                 # edge block to DMA next (lo, hi)
        self.S_M0 = s; self.S_REM = s + 1; self.S_LEFT = s + 2; self.S_SLOT = s + 3; self.S_T = s + 4; self.S_T2 = s + 5
        self.S_PS4 = s + 6; s += 7
        self.stop = s
        # planes -> row operand quads
        self.pq = (nb + 3) // 4            # pos quads per gene
        self.aq = sorted({self.word(k) // 4 for k in range(nb)})
        self.first_use = {q: min(k for k in range(nb) if self.word(k) // 4 == q) for q in self.aq}
        self.last_use = {q: max(k for k in range(nb) if self.word(k) // 4 == q) for q in self.aq}
        self.dbl = None if self.big else 3   # the quad that holds plane 0 AND the last planes: second copy
        self.slot_bytes = self.ri * self.ROWB * nedge

    def word(self, k):
        return k if self.big else a_word(k)

    # ---------------------------------------------------------------- emit helpers
    def e(self, s):
        self.lines.append(s)

    def lds_read(self, tag, dst, off):
        if "nolds" not in self.opts or not self.in_loop:   # (the probe keeps the prologue's reads: same data in the registers)
            self.e(f"ds_read_b128 v[{dst}:{dst + 3}], v{self.VLDS} offset:{off}")
        self.lq.append(tag)

    def lds_need(self, tags):
        """wait until every read in `tags` has returned (reads return in order)"""
        idx = max((self.lq.index(t) for t in tags if t in self.lq), default=-1)
        if idx < 0:
            return
        n = len(self.lq) - 1 - idx
        assert n <= 15
        if not ({"nolds", "nowait_lds"} & self.opts) or not self.in_loop:
            self.e(f"s_waitcnt lgkmcnt({n})")
        self.lq = self.lq[idx + 1:]

    def vm_need(self, tags):
        idx = max((self.vq.index(t) for t in tags if t in self.vq), default=-1)
        if idx < 0:
            return
        n = len(self.vq) - 1 - idx
        assert n <= 63
        if "nowait_vm" not in self.opts or all(t[0] == "dma" for t in tags):
            self.e(f"s_waitcnt vmcnt({n})")
        self.vq = self.vq[idx + 1:]

    def areg(self, e, k, row):
        """register of plane k of edge e in row `row`"""
        w = self.word(k)
        if w // 4 == self.dbl and (row & 1):
            return self.A[e] + 16 + (w & 3)
        return self.A[e] + w

    def aquad_reg(self, e, q, row):
        return self.A[e] + 16 if (q == self.dbl and (row & 1)) else self.A[e] + 4 * q

    def read_row_quad(self, row, q):
        """request quad q of the row operand(s) of row `row` (row 32 = row 0 of the next block, other slot)"""
        for e in range(len(self.A)):
            self.lds_read(("a", row & 1 if q == self.dbl else 0, e, q), self.aquad_reg(e, q, row),
                          self.ri * self.ROWB * e + (row % self.ri) * self.ROWB + q * 16)

    def need_row_quad(self, row, q):
        self.lds_need([("a", row & 1 if q == self.dbl else 0, e, q) for e in range(len(self.A))])

    def need_row_quads(self, row, qs):
        self.lds_need([("a", row & 1 if q == self.dbl else 0, e, q) for q in qs if q in self.aq for e in range(len(self.A))])

    def dma_block(self):
        """LDS-DMA of the next edge block(s) into the slot s_slot, then advance (clamped at the last block)"""
        for e in range(len(self.A)):
            sa = self.SA[e]
            for piece in range(self.NDMA):   # 1 KiB each: lane * 16 + 1024 piece, to the same offset of the slot
                off = self.ri * self.ROWB * e + 1024 * piece
                if off == 0:
                    self.e(f"s_mov_b32 m0, s{self.S_SLOT}")
                else:
                    self.e(f"s_add_u32 m0, s{self.S_SLOT}, {off}")
                self.e("s_nop 0")
                self.e(f"global_load_lds_dwordx4 {'%[aoff]' if piece == 0 else 'v' + str(self.VAP[piece])}, s[{sa}:{sa + 1}]")
                self.vq.append(("dma", e, piece))
        # advance the source by one block unless it is the last one; toggle the slot
        self.e(f"s_cmp_gt_u32 s{self.S_LEFT}, 1")
        self.e(f"s_cselect_b32 s{self.S_T}, %[astride], 0")
        self.e(f"s_cselect_b32 s{self.S_T2}, 1, 0")
        for e in range(len(self.A)):
            sa = self.SA[e]
            self.e(f"s_add_u32 s{sa}, s{sa}, s{self.S_T}")
            self.e(f"s_addc_u32 s{sa + 1}, s{sa + 1}, 0")
        self.e(f"s_sub_u32 s{self.S_LEFT}, s{self.S_LEFT}, s{self.S_T2}")
        self.e(f"s_xor_b32 s{self.S_SLOT}, s{self.S_SLOT}, {self.slot_bytes}")

    def load_pos_quad(self, q, only=None):
        n = min(4, self.nb - 4 * q)
        op = {4: "global_load_dwordx4", 3: "global_load_dwordx3", 2: "global_load_dwordx2", 1: "global_load_dword"}[n]
        for r in range(self.rj):
            if only is not None and r != only:
                continue
            d = self.P + self.PSTR * r + 4 * q
            dst = f"v[{d}:{d + n - 1}]" if n > 1 else f"v{d}"
            if "noreload" in self.opts and self.in_loop:
                self.vq.append(("p", q, r))
                continue
            self.e(f"{op} {dst}, %[poff], s[{self.SB[q]}:{self.SB[q] + 1}] offset:{1024 * r}")
            self.vq.append(("p", q, r))

    def advance_pos(self):
        for q in range(self.pq):
            self.e(f"s_add_u32 s{self.SB[q]}, s{self.SB[q]}, s{self.S_PS4}")
            self.e(f"s_addc_u32 s{self.SB[q] + 1}, s{self.SB[q] + 1}, 0")

    def chains(self, k, row):
        """one plane step of the four chains"""
        if self.ties:   # chains: (gene 0, lo) (gene 1, lo) (gene 0, hi) (gene 1, hi)
            ch = [(0, 0), (1, 0), (0, 1), (1, 1)]
        else:
            ch = [(r, 0) for r in range(4)]
        for c, (r, e) in enumerate(ch):
            p = self.P + self.PSTR * r + k
            a = self.areg(e, k, row)
            l = self.L + c
            if k == 0:
                self.e(f"v_bitop3_b32 v{l}, v{p}, v{a}, v{p} bitop3:0x0c")
            else:
                self.e(f"v_bitop3_b32 v{l}, v{p}, v{a}, v{l} bitop3:0x8e")

    def popcounts(self, row):
        h = row >> 1
        if "nopop" in self.opts:
            return
        accs = [self.ACC + (self.ri // 2) * c + h for c in range(4)]   # tie-free: gene c; ties: gt[0], gt[1], ge[0], ge[1]
        if row & 1:
            for c in range(4):
                self.e(f"v_bcnt_u32_b32 v{self.L + c}, v{self.L + c}, 0")
            for c in range(4):
                if "lshl" not in self.opts:   # high half += count, low half kept (SDWA: one full-rate op)
                    self.e(f"v_add_u32_sdwa v{accs[c]}, v{self.L + c}, v{accs[c]} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_1")
                else:
                    self.e(f"v_lshl_add_u32 v{accs[c]}, v{self.L + c}, 16, v{accs[c]}")
        else:
            for c in range(4):
                self.e(f"v_bcnt_u32_b32 v{accs[c]}, v{self.L + c}, v{accs[c]}")

    def skip_if_last(self):
        self.label_n += 1
        lab = f".Lk1skip_{self.name}_{self.label_n}_%="
        self.e(f"s_cmp_eq_u32 s{self.S_REM}, 1")
        self.e(f"s_cbranch_scc1 {lab}")
        return lab

    # ---------------------------------------------------------------- the loop
    def row(self, i):
        nb = self.nb
        last = i == self.ri - 1
        nxt = i + 1
        if last:
            # the next row is row 0 of the next block: its operand must have landed in the other slot
            self.vm_need([t for t in self.vq if t[0] == "dma"])
            self.e(f"v_xor_b32 v{self.VLDS}, {self.slot_bytes}, v{self.VLDS}")
        for k in range(nb):
            q = self.word(k) // 4
            if "wait4" in self.opts or self.big:
                if k == self.first_use[q] or (q == 3 and k == 0):
                    self.need_row_quad(i, q)
            elif k == 0:        # two waits per row (an s_waitcnt costs the wave an issue slot even when it has nothing to
                self.need_row_quads(i, [3, 0])   # wait for): quads 3 and 0 at the row start, the others before plane 5
            elif k == 5:
                self.need_row_quads(i, [q for q in self.aq if q not in (3, 0)])
            if i == 0 and k % 4 == 0:
                self.vm_need([("p", k // 4, r) for r in range(self.rj)])
            self.chains(k, i)
            # requests for the next row, as the registers of this row become free
            if self.dbl is not None and k == 0 and self.dbl in self.aq:
                self.read_row_quad(nxt, self.dbl)      # (second copy of that quad: free since row i - 1)
            if q != self.dbl and k == self.last_use[q]:
                self.read_row_quad(nxt, q)
            if last and "spread" in self.opts:
                # (probe) one reload per plane group instead of four in a row: gene r's quad q after plane 4 q + 3 + r
                todo = [(q2, r) for q2 in range(self.pq) for r in range(self.rj)
                        if min(4 * q2 + 3 + r, nb - 1) == k and (4 * q2 + 3 <= k or k == nb - 1)]
                if k == nb - 1:
                    todo = [(q2, r) for q2 in range(self.pq) for r in range(self.rj) if 4 * q2 + 3 + r >= nb - 1]
                if todo:
                    lab = self.skip_if_last()
                    for q2, r in todo:
                        self.load_pos_quad(q2, only=r)
                    self.e(lab + ":")
            elif last and (k % 4 == 3 or k == nb - 1):
                lab = self.skip_if_last()
                self.load_pos_quad(k // 4)
                self.e(lab + ":")
        self.popcounts(i)

    def generate(self):
        nedge = len(self.A)
        e = self.e
        e(f"s_mov_b32 s{self.S_M0}, m0")
        for q in range(self.pq):
            if q == 0:
                e(f"s_mov_b64 s[{self.SB[0]}:{self.SB[0] + 1}], %[pbase]")
            else:
                e(f"s_add_u32 s{self.SB[q]}, s{self.SB[q - 1]}, %[pstride]")
                e(f"s_addc_u32 s{self.SB[q] + 1}, s{self.SB[q - 1] + 1}, 0")
        if self.LQ == 4:
            e(f"s_lshl_b32 s{self.S_PS4}, %[pstride], 2")
        else:
            e(f"s_mul_i32 s{self.S_PS4}, %[pstride], {self.LQ}")
        e(f"s_mov_b64 s[{self.SA[0]}:{self.SA[0] + 1}], %[albase]")
        if nedge > 1:
            e(f"s_mov_b64 s[{self.SA[1]}:{self.SA[1] + 1}], %[ahbase]")
        e(f"s_mov_b32 s{self.S_REM}, %[nblk]")
        e(f"s_mov_b32 s{self.S_LEFT}, %[nblk]")
        e(f"s_mov_b32 s{self.S_SLOT}, %[ldsbase]")
        e(f"v_mov_b32 v{self.VLDS}, %[ldsbase]")
        for piece in range(1, self.NDMA):
            e(f"v_add_u32 v{self.VAP[piece]}, {1024 * piece}, %[aoff]")
        for r in range(2 * self.ri):
            e(f"v_mov_b32 v{self.ACC + r}, 0")
        self.dma_block()
        for q in range(self.pq):
            self.load_pos_quad(q)
        self.advance_pos()
        self.dma_block()
        # first row operand: block 0 has to be in LDS
        self.vm_need([t for t in self.vq if t[0] == "dma"][:self.NDMA * nedge])
        for q in ([self.dbl] if self.dbl in self.aq else []) + [q for q in self.aq if q != self.dbl]:
            self.read_row_quad(0, q)
        top_l, top_v = list(self.lq), list(self.vq)
        self.in_loop = True
        e(f".Lk1loop_{self.name}_%=:")
        for i in range(self.ri):
            self.row(i)
        # end of the block: the slot just read is free -- DMA of the block after the next one; pos bases move on
        lab = self.skip_if_last()
        self.dma_block()
        self.advance_pos()
        e(lab + ":")
        # loop invariant of the queues (a last block issues nothing and leaves the loop)
        assert self.lq == top_l, (self.lq, top_l)
        assert self.vq == top_v, (self.vq, top_v)
        e(f"s_sub_u32 s{self.S_REM}, s{self.S_REM}, 1")
        e(f"s_cmp_lg_u32 s{self.S_REM}, 0")
        e(f"s_cbranch_scc1 .Lk1loop_{self.name}_%=")
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e(f"s_mov_b32 m0, s{self.S_M0}")

    def cxx(self):
        nedge = len(self.A)
        out = []
        out.append(f"// {self.name}: NB = {self.nb}, {'with ties (RJ = 2, edges lo and hi)' if self.ties else ('more than 65 535 genes: five pos quads, 128-byte edge rows' if self.big else 'one edge per pass (RJ = 4)')};"
                   f" {len(self.lines)} instructions, VGPRs v8..v{self.vtop - 1}")
        ty = "u32x16" if self.ri == 32 else "u32x8"
        out.append(f"__device__ __forceinline__ void {self.name}({ty} &acc0, {ty} &acc1, {ty} &acc2, {ty} &acc3,")
        out.append("    const void *pbase, uint32_t pstride, const void *albase, const void *ahbase, uint32_t astride, uint32_t nblk,")
        out.append("    uint32_t poff, uint32_t aoff, uint32_t ldsbase)")
        out.append("{")
        out.append("    asm volatile(")
        for l in self.lines:
            out.append(f'        "{l}\\n\\t"')
        w = self.ri // 2
        out.append("        : " + ", ".join(f'"=&{{v[{self.ACC + w * c}:{self.ACC + w * c + w - 1}]}}"(acc{c})' for c in range(4)))
        ins = '[pbase] "s"(pbase), [pstride] "s"(pstride), [albase] "s"(albase), '
        if nedge > 1:
            ins += '[ahbase] "s"(ahbase), '
        ins += '[astride] "s"(astride), [nblk] "s"(nblk), [poff] "v"(poff), [aoff] "v"(aoff), [ldsbase] "s"(ldsbase)'
        out.append("        : " + ins)
        clob = [f'"v{r}"' for r in range(self.P, self.vtop)] + [f'"s{r}"' for r in range(36, self.stop)] + ['"vcc"', '"scc"', '"memory"']
        out.append("        : " + ", ".join(clob) + ");")
        out.append("}")
        return "\n".join(out)


VARIANTS = [(12, False), (15, False), (16, False), (17, False), (18, False)]   # (tie-rich data: two passes of the same loop, edges hi then lo)


def main():
    print("// GENERATED by gen_k1_loop.py -- do not edit; see that file for the schedule.")
    print("typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));")
    print("typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));")
    for nb, ties in VARIANTS:
        for ri in (32, 16):   # 16: half-height items (the last, partly filled round of a launch), name suffix _h
            lp = Loop(nb, ties, f"k1_loop_nb{nb}_{'ties' if ties else 'free'}{'' if ri == 32 else '_h'}", ri=ri, std=True)
            lp.generate()
            print()
            print(lp.cxx())


if __name__ == "__main__":
    main()
