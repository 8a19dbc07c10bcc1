#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of the large-N attention kernel (edtr_amd/csrc/attention.hip,
flash_attn64_v3_kernel): writes edtr_amd/csrc/attn_v3_loop.inc, a C macro whose body is ONE inline-asm string.

Why generated asm: at head width 64 the softmax needs more vector-ISSUE cycles than the two matrix products need matrix-pipe
cycles, so what decides the kernel's speed is which vector instruction sits in which MFMA gap — hipcc's scheduler clusters the
MFMAs, moves accumulators between the VGPR and AGPR files around every branch and packs f32 adds (measured on the C++ form of
the same algorithm, v2).  The stream below owns its registers (clobber list) and places every instruction.

Structure per wave (4 waves = 256 queries of one (image, head); a wave = two 32-query blocks A and B; one wave per SIMD):

  tile t:  s_waitcnt vmcnt(0); s_barrier; LDS-DMA of tile t+2 (4 pieces per wave)
           phase 1:  softmax(A, t)   beside   S_B(t) = K(t) Q_B   and   O_B += V(t-1) P_B(t-1)      (fragments already in registers)
           phase 2:  softmax(B, t)   beside   S_A(t+1) = K(t+1) Q_A   and   O_A += V(t) P_A(t)      (16 ds_read_b128 of K(t+1), V(t))
  every MFMA slot carries the softmax of two scores: v_exp x2, v_add x2 (row-sum), v_cvt_pk x1 — the additions and the pack of
  a pair run one slot behind its exponentials.  Scores arrive pre-scaled and with the running maximum already subtracted (the
  MFMA chain starts from the -max vector), the row maximum is only revisited when a half-row tile sum exceeds 2^14 (slow path).

Registers (v = arch VGPR, a = accumulator file):
  a[0:31] O_A  a[32:63] O_B   a[96:127] K fragments  a[128:159] V^T fragments  a[160:175] Q_A  a[176:191] Q_B
  v[96:127] S_A  v[128:159] S_B  v[160:175] -max_A  v[176:191] -max_B  v[192:207] P_A  v[208:223] P_B
  v[224:227] K fragment LDS addresses  v[228:231] V fragment LDS addresses  v[232:235] DMA lane offsets
  v[236:243] exp temporaries  v[244:247] tile row-sum accumulators  v[248] l_A  v[249] l_B  v[250:255] scratch
  s[84] tile counter  s[85]/s[86] K / V^T DMA scalar offsets  s[87] saved M0  s[88:91] scratch  s[92:93] slow-path return
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "edtr_amd", "csrc", "attn_v3_loop.inc")

OA, OB = 0, 32
KF, VF = 96, 128
QA, QB = 160, 176
SA, SB = 96, 128
NMA, NMB = 160, 176
PA, PB = 192, 208
KAD, VAD, DOFF = 224, 228, 232
T = 236
ACC = {"A": (244, 245), "B": (246, 247)}
LRUN = {"A": 248, "B": 249}
TMP = 250
S_T, S_SOFFK, S_SOFFV, S_M0, S_X0, S_X1, S_X2, S_X3 = 84, 85, 86, 87, 88, 89, 90, 91
STAGE_BYTES, TILE_BYTES = 16384, 8192
SUM_LIMIT = "0x46800000"        # 16384.0f


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=1):
    return f"a{base}" if n == 1 else f"a[{base}:{base + n - 1}]"


class Gen:
    def __init__(self, ahead=3, dma_spread=False, stamps=False, drop=""):
        self.lines = []
        self.label_id = 0
        self.ahead, self.dma_spread, self.stamps = ahead, dma_spread, stamps
        self.drop = set(drop.split(",")) if drop else set()      # timing experiments only (results become garbage)

    def stamp(self, acc):
        """Diagnostic build only: add the cycles since the previous stamp to SGPR accumulator `acc` (s78 sync, s79 DMA issue,
        s80 phase 1, s81 phase 2).  s_memtime returns through the scalar cache: the lgkmcnt(0) it needs also drains LDS reads."""
        if not self.stamps:
            return
        self.e("s_memtime s[74:75]")
        self.e("s_waitcnt lgkmcnt(0)")
        self.e("s_sub_u32 s77, s74, s76")
        self.e(f"s_add_u32 s{acc}, s{acc}, s77")
        self.e("s_mov_b32 s76, s74")

    def e(self, s):
        op = s.split()[0]
        if ("exp" in self.drop and op == "v_exp_f32") or ("add" in self.drop and op == "v_add_f32") or \
                ("cvt" in self.drop and op == "{CVT}") or ("dma" in self.drop and op.startswith("buffer_load")):
            return
        if "qkacc" in self.drop and op == "{MFMA}" and s.split()[1].startswith("v["):
            # score MFMAs write (and chain from) the accumulator file instead of arch VGPRs: a[64:95]
            parts = s.split(", ")
            d = parts[0].split()[1]
            base = int(d[2:d.index(":")])
            acc = f"a[{64 + (base - 96) % 32}:{64 + (base - 96) % 32 + 15}]"
            s = f"{{MFMA}} {acc}, {parts[1]}, {parts[2]}, {acc}"
        self.lines.append(s)

    def label(self, name):
        self.lines.append(name + ":")

    def new_label(self, stem):
        self.label_id += 1
        return f".Lattn3_{stem}_{self.label_id}_%="

    # ---- building blocks -------------------------------------------------------------------------------------------
    def dma_piece(self, which, lds_off):
        """One LDS-DMA instruction (8 tile rows = 1 KiB): which = 0/1 K pieces, 2/3 V^T pieces."""
        srd = "%[srdk]" if which < 2 else "%[srdv]"
        soff = f"s{S_SOFFK}" if which < 2 else f"s{S_SOFFV}"
        self.e(f"s_add_u32 s{S_X1}, s{S_X0}, {lds_off}")
        self.e(f"s_mov_b32 m0, s{S_X1}")
        self.e("s_nop 0")
        self.e(f"buffer_load_dwordx4 {vr(DOFF + which)}, {srd}, {soff} offen lds")

    def dma_begin(self, stage):
        """s[S_X0] = LDS byte address of this wave's slice of `stage`'s K tile."""
        self.e(f"s_add_u32 s{S_X0}, %[ldsb], {stage * STAGE_BYTES}")

    def dma_advance(self):
        self.e(f"s_add_u32 s{S_SOFFK}, s{S_SOFFK}, %[stepk]")
        self.e(f"s_add_u32 s{S_SOFFV}, s{S_SOFFV}, 128")

    DMA_LDS_OFF = [0, 1024, TILE_BYTES, TILE_BYTES + 1024]

    def kread(self, stage, kb, ks):
        self.e(f"ds_read_b128 {ar(KF + (kb * 4 + ks) * 4, 4)}, {vr(KAD + ks)} offset:{stage * STAGE_BYTES + kb * 4096}")

    def vread(self, stage, db, kb, st):
        self.e(f"ds_read_b128 {ar(VF + ((db * 2 + kb) * 2 + st) * 4, 4)}, {vr(VAD + kb * 2 + st)} "
               f"offset:{stage * STAGE_BYTES + TILE_BYTES + db * 4096}")

    def mfma_qk(self, blk, kb, ks):
        s, q, nm = (SA, QA, NMA) if blk == "A" else (SB, QB, NMB)
        d = vr(s + kb * 16, 16)
        c = vr(nm, 16) if ks == 0 else d
        self.e(f"{{MFMA}} {d}, {ar(KF + (kb * 4 + ks) * 4, 4)}, {ar(q + ks * 4, 4)}, {c}")

    def mfma_pv(self, blk, db, kb, st):
        o, p = (OA, PA) if blk == "A" else (OB, PB)
        d = ar(o + db * 16, 16)
        self.e(f"{{MFMA}} {d}, {ar(VF + ((db * 2 + kb) * 2 + st) * 4, 4)}, {vr(p + (kb * 2 + st) * 4, 4)}, {d}")

    # softmax of score pair i (i = 0..15; scores 2i, 2i+1 of the block's 32): exponentials in slot i, sums + pack in slot i+1
    def sm_exp(self, blk, i):
        s = SA if blk == "A" else SB
        t0 = T + (i & 1) * 2
        if i == 0:       # the first pair initialises the two row-sum accumulators
            a0, a1 = ACC[blk]
            self.e(f"v_exp_f32 {vr(a0)}, {vr(s)}")
            self.e(f"v_exp_f32 {vr(a1)}, {vr(s + 1)}")
        else:
            self.e(f"v_exp_f32 {vr(t0)}, {vr(s + 2 * i)}")
            self.e(f"v_exp_f32 {vr(t0 + 1)}, {vr(s + 2 * i + 1)}")

    def sm_fin(self, blk, i):
        p = PA if blk == "A" else PB
        a0, a1 = ACC[blk]
        t0 = T + (i & 1) * 2
        # pair i covers registers 2i, 2i+1 of score tile kb = i >> 3: packed word (kb*2 + st)*4 + w, r = 2i & 15, st = r >> 3, w = (r & 7) >> 1
        kb, r = i >> 3, (2 * i) & 15
        word = p + (kb * 2 + (r >> 3)) * 4 + ((r & 7) >> 1)
        if i == 0:
            self.e(f"{{CVT}} {vr(word)}, {vr(a0)}, {vr(a1)}")
        else:
            self.e(f"v_add_f32 {vr(a0)}, {vr(a0)}, {vr(t0)}")
            self.e(f"v_add_f32 {vr(a1)}, {vr(a1)}, {vr(t0 + 1)}")
            self.e(f"{{CVT}} {vr(word)}, {vr(t0)}, {vr(t0 + 1)}")

    def rowmax(self, s_base, dst):
        """dst = max over the block's 32 scores and over the lane pair (lane, lane^32)."""
        self.e(f"v_max3_f32 {vr(dst)}, {vr(s_base)}, {vr(s_base + 1)}, {vr(s_base + 2)}")
        for r in range(3, 31, 2):
            self.e(f"v_max3_f32 {vr(dst)}, {vr(dst)}, {vr(s_base + r)}, {vr(s_base + r + 1)}")
        self.e(f"v_max_f32 {vr(dst)}, {vr(dst)}, {vr(s_base + 31)}")
        self.e(f"v_mov_b32 {vr(dst + 1)}, {vr(dst)}")
        self.e("s_nop 1")
        self.e(f"v_permlane32_swap_b32 {vr(dst + 1)}, {vr(dst)}")
        self.e("s_nop 1")
        self.e(f"v_max_f32 {vr(dst)}, {vr(dst)}, {vr(dst + 1)}")

    def shift_scores(self, blk, m):
        s, nm = (SA, NMA) if blk == "A" else (SB, NMB)
        for r in range(32):
            self.e(f"v_sub_f32 {vr(s + r)}, {vr(s + r)}, {vr(m)}")
        for r in range(16):
            self.e(f"v_sub_f32 {vr(nm + r)}, {vr(nm + r)}, {vr(m)}")

    def remax_first(self, blk):
        s = SA if blk == "A" else SB
        self.rowmax(s, TMP)
        self.shift_scores(blk, TMP)

    def slow_path(self, blk):
        """Out of line: raise the block's maximum by the tile's row maximum (never lowered), rescale O and l, shift the scores,
        re-form the tile's probabilities and row sum; return through s[92:93]."""
        s, o, p = (SA, OA, PA) if blk == "A" else (SB, OB, PB)
        a0, a1 = ACC[blk]
        self.rowmax(s, TMP)
        self.e(f"v_max_f32 {vr(TMP)}, 0, {vr(TMP)}")
        self.e(f"v_sub_f32 {vr(TMP + 2)}, 0, {vr(TMP)}")
        self.e(f"v_exp_f32 {vr(TMP + 2)}, {vr(TMP + 2)}")          # alpha = 2^-mt
        self.shift_scores(blk, TMP)
        self.e("s_nop 0")
        self.e(f"v_mul_f32 {vr(LRUN[blk])}, {vr(LRUN[blk])}, {vr(TMP + 2)}")
        for r in range(32):
            self.e(f"v_accvgpr_read_b32 {vr(TMP + 3)}, {ar(o + r)}")
            self.e("s_nop 0")
            self.e(f"v_mul_f32 {vr(TMP + 3)}, {vr(TMP + 3)}, {vr(TMP + 2)}")
            self.e("s_nop 0")
            self.e(f"v_accvgpr_write_b32 {ar(o + r)}, {vr(TMP + 3)}")
        for i in range(16):
            self.sm_exp(blk, i)
            self.e("s_nop 0")
            self.sm_fin(blk, i)
        self.e(f"v_add_f32 {vr(a0)}, {vr(a0)}, {vr(a1)}")
        self.e("s_nop 7")
        self.e("s_nop 7")
        self.e("s_setpc_b64 s[92:93]")

    def check(self, blk, slow_label):
        """Row-sum check after a phase: any half-row tile sum above 2^14 takes the slow path, then l += tile sum."""
        a0, a1 = ACC[blk]
        ret = self.new_label("ret")
        self.e(f"v_add_f32 {vr(a0)}, {vr(a0)}, {vr(a1)}")
        self.e(f"v_mov_b32 {vr(TMP + 4)}, {SUM_LIMIT}")
        self.e(f"v_cmp_lt_f32 vcc, {vr(TMP + 4)}, {vr(a0)}")
        self.e(f"s_cbranch_vccz {ret}")
        self.e("s_getpc_b64 s[92:93]")
        self.e("s_add_u32 s92, s92, 12")         # return to the instruction after the s_branch: getpc yields the address of this s_add; s_add + s_addc + s_branch = 12 bytes
        self.e("s_addc_u32 s93, s93, 0")
        self.e(f"s_branch {slow_label}")
        self.label(ret)
        self.e(f"v_add_f32 {vr(LRUN[blk])}, {vr(LRUN[blk])}, {vr(a0)}")

    # ---- one tile ---------------------------------------------------------------------------------------------------
    def tile(self, stage, slowA, slowB):
        nxt, cur = (stage + 1) % 4, stage
        self.e("s_waitcnt vmcnt(0)")
        self.e("s_barrier")
        self.stamp(78)
        # tile t+2 goes into the stage tile t-2 left; s[S_X3] = 1 while such a tile exists
        self.e(f"s_add_u32 s{S_X2}, s{S_T}, 2")
        self.e(f"s_cmp_lt_u32 s{S_X2}, %[nt]")
        self.e(f"s_cselect_b32 s{S_X3}, 1, 0")
        self.dma_begin((stage + 2) % 4)

        def dma(w, last):
            skip = self.new_label("nodma")
            self.e(f"s_cmp_eq_u32 s{S_X3}, 1")
            self.e(f"s_cbranch_scc0 {skip}")
            self.dma_piece(w, self.DMA_LDS_OFF[w])
            if last:
                self.dma_advance()
            self.label(skip)
        if not self.dma_spread:
            for w in range(4):
                dma(w, w == 3)
        self.stamp(79)
        if stage == 0:
            nf = self.new_label("nofirstA")
            self.e(f"s_cmp_eq_u32 s{S_T}, 0")
            self.e(f"s_cbranch_scc0 {nf}")
            self.e("s_nop 15")
            self.e("s_nop 15")
            self.remax_first("A")
            self.label(nf)
        # ---- phase 1: softmax(A) beside S_B(t) and O_B += V(t-1) P_B(t-1); fragments are in registers
        mf = [("qk", "B", kb, ks) for kb in range(2) for ks in range(4)] + \
             [("pv", "B", db, kb, st) for db in range(2) for kb in range(2) for st in range(2)]
        for i, m in enumerate(mf):
            if m[0] == "qk":
                self.mfma_qk(*m[1:])
            else:
                self.mfma_pv(*m[1:])
            self.sm_exp("A", i)
            if i > 0:
                self.sm_fin("A", i - 1)
            if self.dma_spread and i in (1, 5, 9, 13):
                dma((i - 1) // 4, i == 13)
        self.sm_fin("A", 15)
        self.check("A", slowA)
        self.stamp(80)
        if stage == 0:
            nf = self.new_label("nofirstB")
            self.e(f"s_cmp_eq_u32 s{S_T}, 0")
            self.e(f"s_cbranch_scc0 {nf}")
            self.e("s_nop 15")
            self.remax_first("B")
            self.label(nf)
        # ---- phase 2: softmax(B) beside S_A(t+1) (K(t+1) fragments) and O_A += V(t) P_A(t) (V(t) fragments), read AHEAD slots ahead
        reads = [("k", nxt, kb, ks) for kb in range(2) for ks in range(4)] + \
                [("v", cur, db, kb, st) for db in range(2) for kb in range(2) for st in range(2)]
        mf = [("qk", "A", kb, ks) for kb in range(2) for ks in range(4)] + \
             [("pv", "A", db, kb, st) for db in range(2) for kb in range(2) for st in range(2)]
        AHEAD = self.ahead

        def issue(j):
            r = reads[j]
            if r[0] == "k":
                self.kread(*r[1:])
            else:
                self.vread(*r[1:])
        for j in range(min(AHEAD, 16)):
            issue(j)
        for i, m in enumerate(mf):
            if i + AHEAD < 16:
                issue(i + AHEAD)
            self.sm_exp("B", i)
            if i > 0:
                self.sm_fin("B", i - 1)
            self.e(f"s_waitcnt lgkmcnt({min(AHEAD, 15 - i)})")
            if m[0] == "qk":
                self.mfma_qk(*m[1:])
            else:
                self.mfma_pv(*m[1:])
        self.sm_fin("B", 15)
        self.check("B", slowB)
        self.stamp(81)
        self.e(f"s_add_u32 s{S_T}, s{S_T}, 1")

    # ---- whole stream ------------------------------------------------------------------------------------------------
    def build(self):
        e = self.e
        slowA, slowB, done = ".Lattn3_slowA_%=", ".Lattn3_slowB_%=", ".Lattn3_done_%="
        e(f"s_mov_b32 s{S_M0}, m0")
        # inputs -> owned registers
        for i in range(4):
            e(f"v_mov_b32 {vr(KAD + i)}, %[ka{i}]")
            e(f"v_mov_b32 {vr(VAD + i)}, %[va{i}]")
            e(f"v_mov_b32 {vr(DOFF + i)}, %[do{i}]")
        for f in range(4):
            for w in range(4):
                e(f"v_accvgpr_write_b32 {ar(QA + f * 4 + w)}, %[qa{f}{w}]")
                e(f"v_accvgpr_write_b32 {ar(QB + f * 4 + w)}, %[qb{f}{w}]")
        for r in range(64):
            e(f"v_accvgpr_write_b32 {ar(OA + r)}, 0")
        for r in range(32):
            e(f"v_accvgpr_write_b32 {ar(VF + r)}, 0")
        for r in range(16):
            e(f"v_mov_b32 {vr(PB + r)}, 0")
            e(f"v_mov_b32 {vr(NMA + r)}, 0")
            e(f"v_mov_b32 {vr(NMB + r)}, 0")
        e(f"v_mov_b32 {vr(LRUN['A'])}, 0")
        e(f"v_mov_b32 {vr(LRUN['B'])}, 0")
        e(f"s_mov_b32 s{S_T}, 0")
        e(f"s_mov_b32 s{S_SOFFK}, 0")
        e(f"s_mov_b32 s{S_SOFFV}, 0")
        e("s_nop 4")
        # tiles 0 and 1 in flight
        for stg in (0, 1):
            self.dma_begin(stg)
            for w in range(4):
                self.dma_piece(w, self.DMA_LDS_OFF[w])
            self.dma_advance()
        e("s_waitcnt vmcnt(4)")
        e("s_barrier")
        for kb in range(2):
            for ks in range(4):
                self.kread(0, kb, ks)
        e("s_waitcnt lgkmcnt(0)")
        for kb in range(2):
            for ks in range(4):
                self.mfma_qk("A", kb, ks)
        if self.stamps:
            for r in (78, 79, 80, 81):
                e(f"s_mov_b32 s{r}, 0")
            e("s_memtime s[74:75]")
            e("s_waitcnt lgkmcnt(0)")
            e("s_mov_b32 s76, s74")
        loop = ".Lattn3_loop_%="
        self.label(loop)
        for stage in range(4):
            self.tile(stage, slowA, slowB)
        e(f"s_cmp_lt_u32 s{S_T}, %[nt]")
        e(f"s_cbranch_scc1 {loop}")
        # O_B += V(nt-1) P_B(nt-1)
        for db in range(2):
            for kb in range(2):
                for st in range(2):
                    self.mfma_pv("B", db, kb, st)
        e(f"s_branch {done}")
        self.label(slowA)
        self.slow_path("A")
        self.label(slowB)
        self.slow_path("B")
        self.label(done)
        e("s_nop 15")
        e("s_nop 15")
        for r in range(16):
            e(f"v_accvgpr_read_b32 %[oa0{r:02d}], {ar(OA + r)}")
            e(f"v_accvgpr_read_b32 %[oa1{r:02d}], {ar(OA + 16 + r)}")
            e(f"v_accvgpr_read_b32 %[ob0{r:02d}], {ar(OB + r)}")
            e(f"v_accvgpr_read_b32 %[ob1{r:02d}], {ar(OB + 16 + r)}")
        e(f"v_mov_b32 %[la], {vr(LRUN['A'])}")
        e(f"v_mov_b32 %[lb], {vr(LRUN['B'])}")
        if self.stamps:
            for i, r in enumerate((78, 79, 80, 81)):
                e(f"s_mov_b32 %[st{i}], s{r}")
        e(f"s_mov_b32 m0, s{S_M0}")
        e("s_nop 1")


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--ahead", type=int, default=int(os.environ.get("ATTN3_AHEAD", "3")))
    ap.add_argument("--dma-spread", type=int, default=int(os.environ.get("ATTN3_DMA_SPREAD", "0")))
    ap.add_argument("--stamps", type=int, default=0)
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--drop", default="", help="timing experiments: comma list of exp,add,cvt,dma,qkacc (results become garbage)")
    args = ap.parse_args()
    g = Gen(ahead=args.ahead, dma_spread=bool(args.dma_spread), stamps=bool(args.stamps), drop=args.drop)
    g.build()
    out = ["// GENERATED by tools/gen_attn_v2.py — do not edit.  One inline-asm string: the main loop of flash_attn64_v3_kernel.",
           "// MFMA / CVT are string literals naming the dtype's instructions (v_mfma_f32_32x32x16_{bf16,f16}, v_cvt_pk_{bf16,f16}_f32).",
           "#define EDTR_ATTN_V3_ASM(MFMA, CVT) \\"]
    for ln in g.lines:
        parts = ln.replace("{MFMA}", '" MFMA "').replace("{CVT}", '" CVT "')
        out.append(f'    "{parts}\\n\\t" \\')
    out.append('    ""')
    clob = [f'"v{i}"' for i in range(96, 256)] + [f'"a{i}"' for i in range(0, 256)] + \
           [f'"s{i}"' for i in range(74 if args.stamps else 84, 94)] + ['"vcc"', '"scc"', '"memory"']
    out.append("#define EDTR_ATTN_V3_CLOBBERS " + ", ".join(clob))
    # operand lists over fixed C++ names: float oa0[16], oa1[16], ob0[16], ob1[16], la, lb; int kad[4], vad[4]; uint32_t doff[4];
    # U4 qA[4], qB[4]; u32x4 srd_k, srd_v; uint32_t ldsb, stepk; int nt
    outs = []
    for r in range(16):
        for nm in ("oa0", "oa1", "ob0", "ob1"):
            outs.append(f'[{nm}{r:02d}] "=v"({nm}[{r}])')
    outs += ['[la] "=v"(la)', '[lb] "=v"(lb)']
    if args.stamps:
        outs += [f'[st{i}] "=s"(stamps[{i}])' for i in range(4)]
    out.append("#define EDTR_ATTN_V3_OUTS " + ", ".join(outs))
    ins = []
    for i in range(4):
        ins += [f'[ka{i}] "v"(kad[{i}])', f'[va{i}] "v"(vad[{i}])', f'[do{i}] "v"(doff[{i}])']
    for f in range(4):
        for w, c in enumerate("xyzw"):
            ins += [f'[qa{f}{w}] "v"(qA[{f}].{c})', f'[qb{f}{w}] "v"(qB[{f}].{c})']
    ins += ['[srdk] "s"(srd_k)', '[srdv] "s"(srd_v)', '[ldsb] "s"(ldsb)', '[stepk] "s"(stepk)', '[nt] "s"(nt)']
    out.append("#define EDTR_ATTN_V3_INS " + ", ".join(ins))
    with open(args.out, "w") as f:
        f.write("\n".join(out) + "\n")
    print(f"{args.out}: {len(g.lines)} asm lines (ahead {args.ahead}, dma_spread {args.dma_spread}, stamps {args.stamps})")


if __name__ == "__main__":
    main()
