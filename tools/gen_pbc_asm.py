#!/usr/bin/env python3
"""Generates sparsearray_amd/csrc/pbc_dma_asm.inc: the hand-scheduled main loop of
crossprod_pbc_dma_kernel (kernels_mult_pbc.hip) as one inline-asm string.

Why generated: the loop is a 3-stage software pipeline over three rotating SGPR
blocks and two LDS-value sets (6 phases), with a resume stub per phase for panel
boundaries; writing the 6 + 6 + 6 variants by hand invites slips.

Register map (fixed physical registers; the kernel pins C++ vectors to them):
  SGPR  s[36:51] s[52:67] s[68:83]   record blocks A, B, C (4 records x 4 dwords:
                                     +0 LDS byte offset of the row, +1 2*column,
                                     +2..3 value)
        s[84:85] rec base            s[86:87] -> next tile_ptr entry to load
        s88 stream byte offset of the current 6-phase trip
        s89 panel index p            s90 end panel
        s91..s94 tile_ptr[p], [p+1], [p+2], [p+3] (low dwords)
        s95 LDS byte offset of the buffer holding panel p (0 / BUF)
        s96 LDS byte address of this wavefront's first DMA piece in buffer 1
        s97 finite-check iterations  s98 index of the partial last panel (or ~0)
        s99 byte shift of that panel's window (it is moved back to end at the last row)
        s[20:27] DMA source bases of the 4 pieces of the NEXT panel
        s28 phases a wavefront lets pass before it issues its DMA pieces (stagger)
        s29 lane mask of the record touch   s30 countdown of s28 (negative: issued)
        s31 sticky "non-finite seen"        s[32:33] dense-operand touch base (panel p+3)
        s[100:101] time stamps (tuning build only)
        s34 batches left - 1                s35 phase to resume at
  VGPR  v0 lane*ROW (lane base, buffer 0)   v1 lane*16 (DMA lane offset)
        v2 record-touch lane offset         v3 dense-touch lane offset
        v4 finite-check address (buffer 0)  v5 lane base of the current buffer
        v6 touch destination (never read)   v7 running check address
        v8..v11 LDS addresses               v[12:13] check value   v14 class mask
        v15 constants by lane: [0] lane mask of the dense touch, [1] last panel that advances its base
        v[16:23] value set a                v[24:31] value set b
        v[32:...] partial sums (register-indexed: v[32 + 2*column])
"""
import os

ROW = 1032            # bytes per dense column in an LDS buffer: (128 + 1) * 8
BUF = 64 * ROW        # bytes per buffer
CHK = 8 * ROW         # 1024 threads = 8 dense columns per finite-check step

BLK = {"A": 36, "B": 52, "C": 68}
YSET = {"a": [16, 18, 20, 22], "b": [24, 26, 28, 30]}

out = []


def e(s=""):
    out.append(s)


def load(blk, off):
    r = BLK[blk]
    o = f" offset:{off}" if off else ""
    e(f"s_load_dwordx16 s[{r}:{r + 15}], s[84:85], s88{o}")


def d4(blk, ys):
    r = BLK[blk]
    for q in range(4):
        e(f"v_add_u32 v{8 + q}, s{r + 4 * q}, v5")
    for q in range(4):
        y = YSET[ys][q]
        if EXP != "nolds":
            e(f"ds_read_b64 v[{y}:{y + 1}], v{8 + q}")


EXP = os.environ.get("PBC_EXP", "")      # timing experiments only (results are wrong)


def f4(blk, ys):
    r = BLK[blk]
    if EXP == "nofma":
        return
    for q in range(4):
        y = YSET[ys][q]
        if EXP == "noidx":
            e(f"v_fma_f64 v[{32 + 2 * q}:{33 + 2 * q}], s[{r + 4 * q + 2}:{r + 4 * q + 3}], v[{y}:{y + 1}], v[{32 + 2 * q}:{33 + 2 * q}]")
            continue
        if q == 0:
            e(f"s_set_gpr_idx_on s{r + 1}, gpr_idx(SRC2,DST)")
        elif EXP != "idx1":
            e(f"s_set_gpr_idx_idx s{r + 4 * q + 1}")
        e(f"v_fma_f64 v[32:33], s[{r + 4 * q + 2}:{r + 4 * q + 3}], v[{y}:{y + 1}], v[32:33]")
    if EXP != "noidx":
        e("s_set_gpr_idx_off")


X0 = ["A", "B", "C"]      # block whose FMAs run in phase i
X1 = ["B", "C", "A"]      # block whose LDS reads are issued
X2 = ["C", "A", "B"]      # block being loaded


def sd(i):                # value set written by the LDS reads of phase i
    return "b" if i % 2 == 0 else "a"


def sf(i):                # value set consumed by the FMAs of phase i
    return "a" if i % 2 == 0 else "b"



PROF = False


def stamp(bucket):
    """Tuning build only: add the cycles since the previous stamp to counter
    v[96 + bucket] (wave-uniform values kept in VGPRs; v112 = previous time)."""
    if not PROF:
        return
    e("s_memtime s[100:101]")
    e("s_waitcnt lgkmcnt(0)")
    e("v_sub_u32 v113, s100, v112")
    e(f"v_add_u32 v{96 + bucket}, v113, v{96 + bucket}")
    e("v_mov_b32 v112, s100")


def gen(prof):
    global PROF, out
    PROF = prof
    out = []
    if prof:
        e("s_memtime s[100:101]")
        e("s_waitcnt lgkmcnt(0)")
        e("v_mov_b32 v112, s100")
    # ---------------------------------------------------------------- setup
    e("v_mov_b32 v14, 0x207")                      # class mask: sNaN | qNaN | -Inf | +Inf
    load("A", 0)
    load("B", 64)
    e("s_add_u32 s88, s88, 128")
    e("s_mov_b32 s35, 5")
    # ---------------------------------------------------------------- panel boundary
    e("10:")
    stamp(7)                                       # phases
    e("s_waitcnt lgkmcnt(0)")
    stamp(0)                                       # tail wait
    e("s_mov_b32 s91, s92")
    e("s_mov_b32 s92, s93")
    e("s_mov_b32 s93, s94")
    e("s_cmp_ge_u32 s89, s90")
    e("s_cbranch_scc1 90f")
    # pieces of panel p not issued yet (tile shorter than the stagger)? do it now
    e("s_cmp_lt_i32 s30, 0")
    e("s_cbranch_scc1 17f")
    e("s_mov_b32 vcc_hi, 6")
    e("s_branch 60f")
    e("17:")
    NT = 2 if ("ytouch" in EXP and "noytouch" not in EXP) else 1
    e(f"s_waitcnt vmcnt({NT})")                    # own pieces of panel p (the younger touches may fly)
    stamp(1)                                       # own DMA pieces
    e("s_barrier")                                 # everybody's pieces; everybody done with p-1
    stamp(2)                                       # barrier
    e(f"s_xor_b32 s95, s95, {BUF}")
    e("v_add_u32 v5, s95, v0")
    e("s_cmp_lg_u32 s89, s98")
    e("s_cbranch_scc1 16f")
    e("v_add_u32 v5, s99, v5")                     # partial last panel: rows sit s99 bytes further in
    e("16:")
    e("s_mov_b32 s30, s28")                        # arm the staggered issue of panel p+1
    e("s_load_dword s94, s[86:87], 0x0")
    e("s_add_u32 s86, s86, 8")
    e("s_addc_u32 s87, s87, 0")
    # finite check of this workgroup's share of panel p
    e("s_cmp_eq_u32 s97, 0")
    if "nocheck" in EXP:
        e("s_branch 13f")
    e("s_cbranch_scc1 13f")
    e("v_add_u32 v7, s95, v4")
    e("s_mov_b32 s34, s97")
    e("12:")
    e("ds_read_b64 v[12:13], v7")
    e(f"v_add_u32 v7, {CHK}, v7")
    e("s_waitcnt lgkmcnt(0)")
    e("v_cmp_class_f64 vcc, v[12:13], v14")
    e("s_or_b32 s31, s31, vcc_lo")
    e("s_or_b32 s31, s31, vcc_hi")
    e("s_sub_u32 s34, s34, 1")
    e("s_cmp_lg_u32 s34, 0")
    e("s_cbranch_scc1 12b")
    e("13:")
    stamp(4)                                       # finiteness prescan
    e("s_sub_u32 s34, s92, s91")                   # records of tile p
    e("s_lshr_b32 s34, s34, 2")
    e("s_add_u32 s89, s89, 1")
    e("s_sub_u32 s34, s34, 1")                     # batches - 1; borrow: empty tile
    e("s_cbranch_scc1 10b")
    e("s_cmp_lt_u32 s35, 3")
    e("s_cbranch_scc1 50f")
    e("s_cmp_eq_u32 s35, 3")
    e("s_cbranch_scc1 43f")
    e("s_cmp_eq_u32 s35, 4")
    e("s_cbranch_scc1 44f")
    e("s_branch 45f")
    e("50:")
    e("s_cmp_eq_u32 s35, 0")
    e("s_cbranch_scc1 40f")
    e("s_cmp_eq_u32 s35, 1")
    e("s_cbranch_scc1 41f")
    e("s_branch 42f")
    # ---------------------------------------------------------------- resume stubs
    for i in range(6):
        e(f"{40 + i}:")
        stamp(5)                                   # tile bookkeeping + dispatch
        d4(X1[i % 3], sd(i))
        e("s_waitcnt lgkmcnt(0)")
        stamp(6)                                   # resume stub
        e(f"s_branch {20 + (i + 1) % 6}f")
    # ---------------------------------------------------------------- the 6 phases
    for i in range(6):
        e(f"{20 + i}:")
        if "empty" in EXP:
            if i == 5:
                e("s_add_u32 s88, s88, 384")
            e("s_sub_u32 s34, s34, 1")
            e(f"s_cbranch_scc1 {30 + i}f")
            continue
        if EXP not in ("noload", "noloadwait"):
            load(X2[i % 3], 64 * i)
        if i == 0 and EXP != "nokpf":
            # Scalar-cache prefetch, batched: the 6 lines of the NEXT trip miss
            # together under this phase's wait; the record loads of the next
            # trip then hit the scalar cache instead of paying an L2 round trip
            # in every phase.
            for j in range(6):
                e(f"s_load_dword vcc_lo, s[84:85], s88 offset:{384 + 64 * j}")
        d4(X1[i % 3], sd(i))
        f4(X0[i % 3], sf(i))
        if EXP in ("nowait", "noloadwait"):
            out.append("PLACEHOLDER_NOWAIT")
        if EXP.startswith("nop"):
            for _ in range(int(EXP[3:])):
                e("s_nop 0")
        if EXP.startswith("valu"):
            for j in range(int(EXP[4:])):
                e(f"v_add_u32 v{12 + (j & 1)}, 1, v{12 + (j & 1)}")
        if EXP.startswith("salu"):
            for j in range(int(EXP[4:])):
                e("s_add_u32 vcc_lo, vcc_lo, 1")
        if EXP.startswith("fma"):
            for j in range(int(EXP[3:])):
                e(f"v_fma_f64 v[{12}:{13}], v[16:17], v[16:17], v[12:13]")
        if EXP == "double":                        # same SMEM traffic, twice the record work
            d4(X1[i % 3], sd(i))
            f4(X0[i % 3], sf(i))
        if out[-1] == "PLACEHOLDER_NOWAIT":
            out.pop()
        else:
            e("s_waitcnt lgkmcnt(0)")
        if i == 5:
            e("s_add_u32 s88, s88, 384")
        e("s_sub_u32 s34, s34, 1")
        e(f"s_cbranch_scc1 {30 + i}f")
        e("s_sub_u32 s30, s30, 1")                 # stagger expired: issue the next panel's pieces
        e(f"s_cbranch_scc1 {70 + i}f")
    e("s_branch 20b")
    for i in range(6):
        e(f"{30 + i}:")
        e(f"s_mov_b32 s35, {i}")
        e("s_branch 10b")
    for i in range(6):
        e(f"{70 + i}:")
        e(f"s_mov_b32 vcc_hi, {i}")
        e("s_branch 60f")
    # ---- issue the DMA pieces of panel s89 (the one after the current) + the
    # ---- record touch; returns to phase vcc_hi+1 (0..5) or to the boundary (6)
    e("60:")
    stamp(7)
    e("s_mov_b32 s30, -1")
    e("s_cmp_ge_u32 s89, s90")
    e("s_cbranch_scc1 61f")
    e("s_cmp_lg_u32 s89, s98")
    e("s_cbranch_scc1 15f")
    for q in range(4):                             # partial last panel: its window ends at the
        e(f"s_sub_u32 s{20 + 2 * q}, s{20 + 2 * q}, s99")   # last row (rows nrow-128 .. nrow-1)
        e(f"s_subb_u32 s{21 + 2 * q}, s{21 + 2 * q}, 0")
    e("15:")
    e("s_sub_u32 m0, s96, s95")                    # first piece, other buffer
    for q in range(4):
        if q:
            e(f"s_add_u32 m0, m0, {ROW}")
        e("s_nop 0")
        e(f"global_load_lds_dwordx4 v1, s[{20 + 2 * q}:{21 + 2 * q}]")
    for q in range(4):
        e(f"s_add_u32 s{20 + 2 * q}, s{20 + 2 * q}, 1024")
        e(f"s_addc_u32 s{21 + 2 * q}, s{21 + 2 * q}, 0")
    e("v_lshl_add_u32 v7, s93, 4, v2")             # records two panels ahead towards L2
    e("s_mov_b32 exec_lo, s29")
    e("s_mov_b32 exec_hi, 0")
    e("global_load_dword v6, v7, s[84:85]")
    if "ytouch" in EXP and "noytouch" not in EXP:   # measured: no gain (4.17 vs 4.08 ms), off
        # this wavefront's lines of the workgroup's share of the Y panel 3 ahead
        e("v_readlane_b32 exec_lo, v15, 0")
        e("s_nop 3")
        e("global_load_dword v6, v3, s[32:33]")
        e("v_readlane_b32 vcc_lo, v15, 1")         # last panel index that advances the base
        e("s_cmp_lt_u32 s89, vcc_lo")
        e("s_cselect_b32 vcc_lo, 1024, 0")
        e("s_add_u32 s32, s32, vcc_lo")
        e("s_addc_u32 s33, s33, 0")
    e("s_mov_b64 exec, -1")
    e("61:")
    stamp(3)                                       # DMA + touch issue
    e("s_cmp_lt_u32 vcc_hi, 3")
    e("s_cbranch_scc1 62f")
    e("s_cmp_eq_u32 vcc_hi, 3")
    e("s_cbranch_scc1 24b")
    e("s_cmp_eq_u32 vcc_hi, 4")
    e("s_cbranch_scc1 25b")
    e("s_cmp_eq_u32 vcc_hi, 5")
    e("s_cbranch_scc1 20b")
    e("s_branch 17b")
    e("62:")
    e("s_cmp_eq_u32 vcc_hi, 0")
    e("s_cbranch_scc1 21b")
    e("s_cmp_eq_u32 vcc_hi, 1")
    e("s_cbranch_scc1 22b")
    e("s_branch 23b")
    e("90:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")

    return out


dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "sparsearray_amd", "csrc", "pbc_dma_asm.inc")
with open(dst, "w") as f:
    f.write("// Generated by tools/gen_pbc_asm.py -- do not edit; see that file for the register map.\n")
    f.write(f"#define PBC_DMA_ROW {ROW}\n#define PBC_DMA_BUF {BUF}\n")
    for name, prof in (("PBC_DMA_ASM_TEXT", False), ("PBC_DMA_ASM_TEXT_PROF", True)):
        lines = gen(prof)
        f.write(f"#define {name} \\\n")
        for ln in lines:
            sep = "\\n" if ln.endswith(":") else "\\n\\t"
            f.write(f'\t"{ln}{sep}" \\\n')
        f.write('\t""\n')
        print("wrote", name, len(lines), "lines")
