#!/usr/bin/env python3
"""Generates sparsearray_amd/csrc/pbc_dma_asm.inc: the hand-scheduled panel loop of
crossprod_pbc_dma_kernel (kernels_mult_pbc.hip) as one inline-asm string.

Why generated: the loop is a 3-stage software pipeline over three rotating SGPR
blocks and YSETS sets of y registers (3 * YSETS phases), with a resume stub per
phase for panel boundaries and an out-of-line DMA issue routine; the variants
differ only in register numbers.

Record stream ("format 1", built by pbc_pass_kernel<.., 1>): batches of 8 records,
96 bytes each = 8 x u32 meta followed by 8 x f64 value.
    meta = (8 * row_in_panel) << 16 | flag << 15 | 2 * column_in_group
The low byte is consumed as is by s_set_gpr_idx_* (they read bits 7:0), the high
half as is by an SDWA add (src0_sel:WORD_1): no scalar work goes into decoding.
flag (record 0 of a batch only) = "last batch of its tile"; every tile has at least
one batch, so the loop needs no tile table.

Pipeline, per phase (one batch of 8 records each):
    L(k+2)  s_load_dwordx8 + s_load_dwordx16        -> block X2
    D(k+1)  8 LDS addresses from block X1 (computed in the low half of the
            destination register pair), 8 ds_read_b64 -> y set SD
    F(k)    acc[c_j] += a_j * y_j in VGPR-index mode, y set SF
    one s_waitcnt lgkmcnt(0), flag test, DMA-stagger countdown
YSETS = 2: D runs ahead of F into the other y set, so the LDS latency hides under
the 16 instructions of F.  YSETS = 1: the read that refills y_j is issued right
behind the FMA that consumed it (16 VGPRs fewer, LDS latency of the last read
exposed at the wait).
DEFER (two y sets): a phase looks at its batch's flags BEFORE the batch; the last batch of a tile keeps its
FMAs for the far side of the barrier, where they run under the latency of the next tile's first LDS reads --
the one stretch in which all 16 wavefronts wait for the LDS at once.  Same sums in the same order (identical
result checksum); 1.735 -> 1.719 ms at config 2a, A/B on one box (tools/debug/r2_rot.sh).
Measured (tools/debug/exp_pbc.sh): what a phase costs is the round trips behind its
single wait, not its instructions -- twice the work per phase ran 1.5 % slower --
hence 8 records per phase (the SGPR file allows no more: 3 x 24).  The order inside a phase -- the 8 address
adds, the 8 LDS reads, then the 8 FMAs -- beats dealing the reads between the FMAs, one by one or in pairs
(1.720 / 1.733 / 1.728 ms at config 2a, same box).  Letting the wavefronts whose lanes lie past the workgroup's
share of the panel skip the finiteness check costs more in its branch than the check (1.715 -> 1.721 ms).

DMA issue: all 4 pieces of a wavefront at once, after the batch the layout flags with bit 14
(batch w & 3 of the tile, or its last one), so that the phases pay one flag test, not two.
Alternatives measured at config 2a: every piece on its own between phases 2.85 ms,
in pairs 2.50 ms, staggered batches of four 2.14 ms (a lone piece stalls its
wavefront ~260 cycles, four back to back ~130 each); all 64 at the barrier 2.32 ms.

Register map (fixed physical registers; the kernel pins C++ vectors to them):
  SGPR  s[28:51] s[52:75] s[76:99]   record blocks A, B, C (+0..7 meta, +8..23 values)
        s[8:9] record base           s10 stream byte offset of the current trip
        s11 next panel index         s12 end panel
        s13 LDS byte offset of the buffer holding the current panel (0 / BUF)
        s14 LDS byte address of this wavefront's first DMA piece in buffer 1
        s15 w >> 2 (PBC_SKEW experiment)   s16, s17 partial last panel (index, byte shift)  s19 finite-check iterations - 1
        s18 sticky "non-finite seen"
        s[20:27] DMA source bases of the wavefront's 4 dense columns (start of the row split)
        vcc_lo scratch, vcc_hi return selector of the issue routine, m0 scratch
  VGPR  v0 lane*ROW (lane base, buffer 0)   v1 lane*16 (DMA lane offset)
        v2 record-touch lane offset (+ look-ahead distance)
        v3 finite-check address (buffer 0)  v4 lane base of the current buffer
        v5 touch destination (never read)   v6 scratch address
        v7 constants by lane: [0] finite-check iterations, [1] index of the partial
           last panel (or ~0), [2] byte shift of that panel's window
        v[8:9] check value  v10 class mask
        v11 DMA offset: lane*16 + bytes from the split's first panel to the one to stage (>= 1024, so
            that the backward shift of a partial last panel, < 1024, keeps it non-negative: the
            global saddr form takes the VGPR offset as unsigned)
        v[12:27] y set a, v[28:43] y set b (YSETS = 2)  /  v[12:19] addresses,
        v[20:35] y (YSETS = 1)
        v[ACC:...] partial sums (register-indexed: v[ACC + 2*column]), ACC = 44 / 36
"""
import os

ROW = 1032            # bytes per dense column in an LDS buffer: (128 + 1) * 8
BUF = 64 * ROW        # bytes per buffer
CHK = 8 * ROW         # 1024 threads = 8 dense columns per finite-check step
BATCH = 96            # bytes per batch of 8 records

EXP = os.environ.get("PBC_EXP", "")      # timing experiments only (results are wrong)
YSETS = int(os.environ.get("PBC_YSETS", "2"))   # 2: LDS reads run a phase ahead of the FMAs (1.7 % faster than 1 once the scalar work was trimmed)
NPH = 3 * YSETS
TRIP = NPH * BATCH
BLK = {"A": 28, "B": 52, "C": 76}
if YSETS == 2:
    YSET = [12, 28]
    ACC = 44
else:
    ADDR = 12
    YSET = [20]
    ACC = 36
DEFER = YSETS == 2 and not EXP and os.environ.get("PBC_DEFER", "1") == "1"
SKEW = int(os.environ.get("PBC_SKEW", "0"))     # experiment: post-barrier skew of the wavefronts that share a SIMD
PROF = False
out = []


def e(s=""):
    out.append(s)


def stamp(bucket):
    """Tuning build only: add the cycles since the previous stamp to counter
    v[116 + bucket] (wave-uniform values kept in VGPRs; v126 = previous time)."""
    if not PROF:
        return
    e("s_memtime s[100:101]")
    e("s_waitcnt lgkmcnt(0)")
    e("v_sub_u32 v127, s100, v126")
    e(f"v_add_u32 v{116 + bucket}, v127, v{116 + bucket}")
    e("v_mov_b32 v126, s100")


def dispatch(reg, labels, direction):
    """Branch to labels[reg] (compare chain; the last label is the fall-through)."""
    n = len(labels)
    for i, lab in enumerate(labels[:-1]):
        e(f"s_cmp_eq_u32 {reg}, {i}")
        e(f"s_cbranch_scc1 {lab}{direction}")
    e(f"s_branch {labels[n - 1]}{direction}")


def load(blk, off):
    r = BLK[blk]
    o = f" offset:{off}" if off else ""
    if "nosmem" in EXP:                            # (experiment: fixed 7 batches per tile, no record loads)
        return
    e(f"s_load_dwordx8 s[{r}:{r + 7}], s[8:9], s10{o}")
    e(f"s_load_dwordx16 s[{r + 8}:{r + 23}], s[8:9], s10 offset:{off + 32}")


X0 = ["A", "B", "C"]      # block whose FMAs run in phase i (i % 3)
X1 = ["B", "C", "A"]      # block whose LDS reads are issued
X2 = ["C", "A", "B"]      # block being loaded


def sd(i):                # y set written by the LDS reads of phase i
    return YSET[(i + 1) % YSETS]


def sf(i):                # y set consumed by the FMAs of phase i
    return YSET[i % YSETS]


def sdwa_add(dst, src_s):
    e(f"v_add_u32_sdwa v{dst}, s{src_s}, v4 dst_sel:DWORD dst_unused:UNUSED_PAD "
      f"src0_sel:WORD_1 src1_sel:DWORD")


def d8(i, part="both"):
    """LDS reads of block X1(i) into y set sd(i) (two-set form and resume code);
    part = "addr": the address adds only, "read": the reads only."""
    r = BLK[X1[i % 3]]
    y = sd(i)
    if "nolds" in EXP:
        return
    for j in range(8):
        if part != "read":
            sdwa_add(y + 2 * j if YSETS == 2 else ADDR + j, r + j)
    for j in range(8):
        if part != "addr":
            a = y + 2 * j if YSETS == 2 else ADDR + j
            e(f"ds_read_b64 v[{y + 2 * j}:{y + 2 * j + 1}], v{a}")


def f8(i, interleave):
    r = BLK[X0[i % 3]]
    y = sf(i)
    for j in range(8):
        if "nofma" not in EXP:
            if j == 0:
                e(f"s_set_gpr_idx_on s{r}, gpr_idx(SRC2,DST)")
            else:
                e(f"s_set_gpr_idx_idx s{r + j}")
            e(f"v_fma_f64 v[{ACC}:{ACC + 1}], s[{r + 8 + 2 * j}:{r + 9 + 2 * j}], "
              f"v[{y + 2 * j}:{y + 2 * j + 1}], v[{ACC}:{ACC + 1}]")
        if interleave and "nolds" not in EXP:   # single y set: refill y_j for the next batch right away
            e(f"ds_read_b64 v[{y + 2 * j}:{y + 2 * j + 1}], v{ADDR + j}")
    if "nofma" not in EXP:
        e("s_set_gpr_idx_off")


def gen(prof):
    global PROF, out
    PROF = prof
    out = []
    if prof:
        e("s_memtime s[100:101]")
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v126, s100")
    ph = [f"{20 + i}" for i in range(NPH)]          # phase labels
    # ---------------------------------------------------------------- setup
    e("v_mov_b32 v10, 0x207")                      # class mask: sNaN | qNaN | -Inf | +Inf
    if "nosmem" in EXP:                            # constant, valid records (row j, column 5j, value 1.0)
        for r in BLK.values():
            for j in range(8):
                e(f"s_mov_b32 s{r + j}, {((8 * j) << 16) | (2 * 5 * j)}")
                e(f"s_mov_b32 s{r + 8 + 2 * j}, 0")
                e(f"s_mov_b32 s{r + 9 + 2 * j}, 0x3ff00000")
    load("A", 0)
    load("B", BATCH)
    e(f"s_add_u32 s10, s10, {2 * BATCH}")
    e("s_waitcnt lgkmcnt(0)")                      # (the boundary code itself never has a load in flight)
    e(f"s_branch {(50 if DEFER else 30) + NPH - 1}f")   # first panel: enter through the last phase's boundary

    # ---- panel boundary after a tile that ended in phase i (one copy per phase: no
    # ---- dispatch chains; the scalar unit is the busiest pipe of this kernel)
    def boundary(i, entry=False):
        # DEFER (two y sets): the FMAs of the tile's last batch run behind the barrier, under the latency
        # of the first LDS reads of the next tile -- the one stretch in which every wavefront of the
        # workgroup waits for the LDS at once (the phase that found the end-of-tile flag skipped them).
        # `entry`: the copy the first panel enters through (nothing to multiply yet).
        defer = DEFER and not entry
        e(f"{(50 if entry else 30) + i}:")
        if "nosmem" in EXP:
            e("s_mov_b32 s99, 7")
        stamp(7)                                   # phases
        e("s_cmp_ge_u32 s11, s12")
        e(f"s_cbranch_scc1 {(40 + i) if defer else 90}f")
        # bookkeeping and the LDS addresses of the next tile's first batch come before the
        # barrier: wavefronts that arrive early do them while they would wait anyway
        e(f"s_xor_b32 s13, s13, {BUF}")
        e("v_add_u32 v4, s13, v0")
        e("s_cmp_lg_u32 s11, s16")
        e("s_cbranch_scc1 16f")
        e("v_add_u32 v4, s17, v4")                 # partial last panel: its window was moved back to end
        e("16:")                                   # at the last row; the rows sit further in
        e("s_add_u32 s11, s11, 1")
        # finite check of this workgroup's share of the panel: the first read rides on
        # the LDS wait of the reads below, the rest (few column blocks only) loop at 12
        e("v_add_u32 v6, s13, v3")
        stamp(4)                                   # boundary bookkeeping
        d8(i, "addr")
        # own pieces of this panel (issued inside the previous tile; the younger touch may fly)
        if "nodmawait" not in EXP:                 # (experiment: pretend the pieces always landed in time)
            e(f"s_waitcnt vmcnt({0 if 'notouch' in EXP else 1})")
        stamp(1)                                   # own DMA pieces
        if "nobarrier" not in EXP:
            e("s_barrier")                         # everybody's pieces; everybody done with the previous panel
        if SKEW:
            # Round-3 experiment (PBC_SKEW=n): the barrier releases the four wavefronts of a SIMD in the same
            # state, and they stay aligned for the tile -- all reading LDS, then all multiplying (the
            # 'work' leg takes 1.40 ms behind a barrier against 0.95 ms free-running).
            # Wavefront w waits (w >> 2) * n * 64 cycles behind the barrier (s15 = w >> 2).
            e("s_mov_b32 vcc_lo, s15")
            e("17:")
            e("s_cmp_eq_u32 vcc_lo, 0")
            e("s_cbranch_scc1 18f")
            e(f"s_sleep {SKEW}")
            e("s_sub_u32 vcc_lo, vcc_lo, 1")
            e("s_branch 17b")
            e("18:")
        stamp(2)                                   # barrier
        d8(i, "read")                              # LDS reads of the batch the next phase multiplies
        if defer:
            f8(i, False)
        e("s_mov_b32 m0, s19")                     # (after the FMAs: the register-index mode writes m0)
        stamp(5)
        if "nofinite" in EXP:
            e("s_waitcnt lgkmcnt(0)")
        else:
            e("12:")
            e("ds_read_b64 v[8:9], v6")
            e(f"v_add_u32 v6, {CHK}, v6")
            e("s_waitcnt lgkmcnt(0)")
            e("v_cmp_class_f64 vcc, v[8:9], v10")
            if "nocheck" not in EXP:
                e("s_or_b32 s18, s18, vcc_lo")
                e("s_or_b32 s18, s18, vcc_hi")
            e("s_sub_u32 m0, m0, 1")               # m0 = iterations - 1: loop until it borrows
            e("s_cbranch_scc0 12b")
        stamp(6)                                   # resume reads + finiteness prescan
        e(f"s_branch {ph[(i + 1) % NPH]}b")

    # ---- DMA pieces of panel s11 (the one after the current) + the record touch
    def issue():
        stamp(7)
        e("s_cmp_ge_u32 s11, s12")
        e("s_cbranch_scc1 61f")
        e("s_cmp_lg_u32 s11, s16")
        e("s_cbranch_scc1 15f")
        e("v_subrev_u32 v11, s17, v11")            # partial last panel: its window ends at the last row
        e("15:")
        e("s_sub_u32 m0, s14, s13")                # first piece, other buffer
        for q in range(4):
            if q:
                e(f"s_add_u32 m0, m0, {ROW}")
            e("s_nop 0")
            if "nodma" not in EXP:
                e(f"global_load_lds_dwordx4 v11, s[{20 + 2 * q}:{21 + 2 * q}]")
        e("s_nop 1")
        e("v_add_u32 v11, 1024, v11")              # next panel (the bases stay; 32-bit offsets per row split)
        e("v_add_u32 v6, s10, v2")                 # records ~2 panels ahead towards L2 (all lanes:
        if "notouch" not in EXP:                   # those past the tile repeat its last line)
            e("global_load_dword v5, v6, s[8:9]")
        e("61:")
        stamp(3)                                   # DMA + touch issue

    # ---------------------------------------------------------------- the phases
    for i in range(NPH):
        if i == 0 and "align" in EXP:
            e(".p2align 6")                        # (experiment: loop head on a 64-byte boundary)
        e(f"{ph[i]}:")
        load(X2[i % 3], BATCH * i)
        if i == NPH - 1 and DEFER:
            e(f"s_add_u32 s10, s10, {TRIP}")
        if DEFER:
            # bit 15 of the batch's first meta word: last batch of the tile; bit 14: issue the
            # LDS-DMA of the next panel after this batch.  One test on the fast path, ahead of the
            # batch: the end of a tile keeps its FMAs for the far side of the barrier.
            e(f"s_and_b32 vcc_lo, s{BLK[X0[i % 3]]}, 0xc000")
            e(f"s_cbranch_scc1 {70 + i}f")
        if YSETS == 2:
            d8(i)
            f8(i, False)
        else:
            if "nolds" not in EXP:
                for j in range(8):
                    sdwa_add(ADDR + j, BLK[X1[i % 3]] + j)
            f8(i, True)
        e("s_waitcnt lgkmcnt(0)")
        if DEFER:
            continue
        if i == NPH - 1:
            e(f"s_add_u32 s10, s10, {TRIP}")
        if "nosmem" in EXP:                        # (experiment: 7 batches per tile, issue after the third)
            e("s_sub_u32 s99, s99, 1")
            e("s_cmp_eq_u32 s99, 4")
            e(f"s_cbranch_scc1 {70 + i}f")
            e("s_cmp_eq_u32 s99, 0")
            e(f"s_cbranch_scc1 {30 + i}f")
        else:
            # bit 15 of the batch's first meta word: last batch of the tile; bit 14: issue the
            # LDS-DMA of the next panel after this batch.  One test on the fast path.
            e(f"s_and_b32 vcc_lo, s{BLK[X0[i % 3]]}, 0xc000")
            e(f"s_cbranch_scc1 {70 + i}f")
    e(f"s_branch {ph[0]}b")
    for i in range(NPH):                           # slow paths: a flagged batch
        r0 = BLK[X0[i % 3]]
        e(f"{70 + i}:")
        if DEFER:
            e(f"s_bitcmp1_b32 s{r0}, 15")
            e(f"s_cbranch_scc1 {80 + i}f")
            d8(i)                                  # the issue flag alone: the batch as in the phase, then the DMA
            f8(i, False)
            e("s_waitcnt lgkmcnt(0)")
            issue()
            e(f"s_branch {ph[(i + 1) % NPH]}b")
            e(f"{80 + i}:")                        # end of the tile (the FMAs wait for the boundary)
            e(f"s_bitcmp1_b32 s{r0}, 14")
            e(f"s_cbranch_scc0 {30 + i}f")
            issue()
            e(f"s_branch {30 + i}f")
            continue
        if "nosmem" not in EXP:
            e(f"s_bitcmp1_b32 s{r0}, 14")
            e(f"s_cbranch_scc0 {30 + i}f")         # only the end-of-tile flag
        issue()
        if "nosmem" not in EXP:
            e(f"s_bitcmp1_b32 s{r0}, 15")
            e(f"s_cbranch_scc1 {30 + i}f")
        e(f"s_branch {ph[(i + 1) % NPH]}b")
    for i in range(NPH):
        boundary(i)
    if DEFER:
        boundary(NPH - 1, entry=True)
        for i in range(NPH):                       # no panel left: the last tile's last batch, then out
            e(f"{40 + i}:")
            f8(i, False)
            e("s_branch 90f")
    e("90:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return out


if EXP and not os.environ.get("PBC_ASM_OUT"):
    raise SystemExit("PBC_EXP builds compute wrong results: write them to PBC_ASM_OUT (a tuning build's include), "
                     "never over the product's pbc_dma_asm.inc")
dst = os.environ.get("PBC_ASM_OUT") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sparsearray_amd", "csrc", "pbc_dma_asm.inc")
with open(dst, "w") as f:
    f.write("// Generated by tools/gen_pbc_asm.py -- do not edit; see that file for the register map.\n")
    f.write(f"#define PBC_DMA_ROW {ROW}\n#define PBC_DMA_BUF {BUF}\n#define PBC_DMA_BATCH_BYTES {BATCH}\n")
    f.write(f"#define PBC_DMA_YSETS {YSETS}\n")
    for name, prof in (("PBC_DMA_ASM_TEXT", False), ("PBC_DMA_ASM_TEXT_PROF", True)):
        lines = gen(prof)
        if prof:
            f.write("#ifdef SVT_TUNING\n")
        f.write(f"#define {name} \\\n")
        for ln in lines:
            sep = "\\n" if ln.endswith(":") else "\\n\\t"
            f.write(f'\t"{ln}{sep}" \\\n')
        f.write('\t""\n')
        if prof:
            f.write("#endif\n")
        print("wrote", name, len(lines), "lines")
