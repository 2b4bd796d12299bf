#!/usr/bin/env python3
"""Generates sparsearray_amd/csrc/pbgx_asm.inc: the pass loop of crossprod_pbc_gatherx_kernel
(kernels_mult_pbc.hip) -- the gather product with two dense columns per lane, run as ONE software pipeline
over a wavefront's whole record stream of a pass (all the tiles of its column group inside its XCD's row
range), with the wavefronts of an XCD kept within a few row panels of each other so that the rows of Yt
they gather are hits in that XCD's L2.

Records: format-0 batches of 4 x {u32 8 * row_in_panel, u32 2 * column | tile-start flag << 15, f64 value}
in three rotating 16-SGPR blocks, pipeline load(k+2) / D(k+1) / F(k) as in tools/gen_pbg2_asm.py.  The first
batch of every tile has bit 15 of its first column word set (every tile has at least one batch); when that
batch reaches the D stage the boundary code runs -- nothing in it waits for the loads in flight:
    pb (s[86:87], base of the current panel of Yt)  += panel stride
    publish: progress[own entry] = ++step             (one lane, plain store)
    L2 touch of the wavefront's own record stream 2 KiB ahead of its load cursor (20 lanes, one dword per
    128-byte line: more than the longest tile it will meet before the next boundary repeats the touch) --
    records are read once, by scalar loads issued two batches ahead: without the touch every other one
    waits for HBM (4.50 -> 4.31 ms at one rank's share of BASELINE config 4)
    check the snapshot of all progress entries of the XCD taken at the PREVIOUS boundary (it has landed: at
    least one `s_waitcnt vmcnt(4)` lies in between): somebody started and more than `dsync` tiles behind ->
    spin (fresh snapshots, bounded: after %[spin] polls the wavefront stops synchronising for good)
    issue the next snapshot (global_load_dwordx4 sc1 into v[56:59]: 256 entries per XCD, 4 per lane)
The protocol is a pacing hint only: results never depend on it.

Register map: accumulators v[64:143] (lo) and v[144:223] (hi), y sets v[224:239] / v[240:255], snapshot
v[56:59]; s[36:83] record blocks, s85 M0 save, s[86:87] pb, s[88:89] touch pointer, s90 / s94 / s95 scratch.

Measured at one rank's share of BASELINE config 4 (round 4, tools/debug/config4_pacing.py): an L2 touch of the
panel ahead by the wavefronts themselves (one dword per 128-byte line, spread over the XCD's wavefronts) bought
nothing (4.85 ms with it, 4.72 without: the load sits in the wavefront's in-order vmcnt queue and its HBM miss
stalls the pipeline once per tile) -- the first gather of a line pulls it into L2 for the other 255 wavefronts.
"""
import os

BLK = {"A": 36, "B": 52, "C": 68}
RECTOUCH = int(os.environ.get("PBGX_RECTOUCH", "2048"))   # 0: no L2 touch of the record stream
FAR = 8                                                    # tiles behind beyond which a wavefront is not waited for
YSET = {"a": 224, "b": 240}
out = []
nlab = [0]


def e(s):
    out.append(s)


def load(blk):
    r = BLK[blk]
    e(f"s_load_dwordx16 s[{r}:{r + 15}], %[base], %[lo]")


def snapshot_min():
    # v56..v59 = entries; 0 = not started -> wraps to the maximum and is ignored; so is a wavefront more than FAR
    # tiles behind (s90 = the smallest entry - 1 that still counts): it gathers from rows that left this XCD's L2
    # long ago, nothing is gained by waiting for it -- workgroups that START late (CUs held by another kernel at
    # launch, e.g. a collective's) pace themselves among each other and do not hold the others back
    for v in (56, 57, 58, 59):
        e(f"v_add_u32 v{v}, -1, v{v}")
        e(f"v_cmp_gt_u32 vcc, s90, v{v}")
        e(f"v_cndmask_b32_e64 v{v}, v{v}, -1, vcc")
    e("v_min3_u32 %[vt], v56, v57, v58")
    e("v_min_u32 %[vt], %[vt], v59")
    e("v_cmp_gt_u32 vcc, s94, %[vt]")


def boundary(blk):
    """Runs when batch `blk` (about to enter the D stage) starts a tile."""
    r = BLK[blk]
    n = nlab[0]
    nlab[0] += 1
    e(f"s_bitcmp1_b32 s{r + 1}, 15")
    e(f"s_cbranch_scc0 2{n}f")
    e("s_add_u32 s86, s86, %[pst]")
    e("s_addc_u32 s87, s87, 0")
    e("s_add_u32 %[step], %[step], 1")
    e("v_mov_b32 %[vt], %[step]")
    e("s_mov_b64 exec, 1")
    e("global_store_dword %[vz], %[vt], %[pgm]")
    if RECTOUCH:
        # L2 touch of this wavefront's record stream RECTOUCH bytes ahead: 20 lines (one dword each)
        e("s_add_u32 s88, %[rbl], %[lo]")
        e("s_addc_u32 s89, %[rbh], 0")
        e(f"s_add_u32 s88, s88, {RECTOUCH}")
        e("s_addc_u32 s89, s89, 0")
        e("s_mov_b64 exec, 0xfffff")
        e("global_load_dword %[vd], %[l128], s[88:89]")
    e("s_mov_b64 exec, -1")
    # s94 = max(step - dsync, 1) - 1: the smallest (entry - 1) a started wavefront may show
    e("s_sub_u32 s94, %[step], %[dsync]")
    e("s_cselect_b32 s94, 0, s94")
    e("s_max_u32 s94, s94, 1")
    e("s_sub_u32 s94, s94, 1")
    e(f"s_sub_u32 s90, %[step], {FAR}")
    e("s_cselect_b32 s90, 0, s90")
    e("s_max_u32 s90, s90, 1")
    e("s_sub_u32 s90, s90, 1")
    snapshot_min()
    e(f"s_cbranch_vccz 5{n}f")
    e("s_mov_b32 s95, %[spin]")
    e(f"4{n}:")
    e("s_sleep 4")
    e("global_load_dwordx4 v[56:59], %[l16], %[pga] sc1")
    e("s_waitcnt vmcnt(0)")
    snapshot_min()
    e(f"s_cbranch_vccz 5{n}f")
    e("s_sub_u32 s95, s95, 1")
    e("s_cmp_lg_u32 s95, 0")
    e(f"s_cbranch_scc1 4{n}b")
    e("s_mov_b32 %[dsync], 0x7fffffff")
    e(f"5{n}:")
    e("global_load_dwordx4 v[56:59], %[l16], %[pga] sc1")
    e(f"2{n}:")


def d4(blk, ys):
    r, y = BLK[blk], YSET[ys]
    boundary(blk)
    for j in range(4):
        e(f"v_mad_u32_u24 %[t{j}], s{r + 4 * j}, %[kp], %[l16]")
    for j in range(4):
        e(f"global_load_dwordx4 v[{y + 4 * j}:{y + 4 * j + 3}], %[t{j}], s[86:87]")


def f4(blk, ys):
    r, y = BLK[blk], YSET[ys]
    for j in range(4):
        if j == 0:
            e(f"s_set_gpr_idx_on s{r + 1}, gpr_idx(SRC2,DST)")
        else:
            e(f"s_set_gpr_idx_idx s{r + 4 * j + 1}")
        v = f"s[{r + 4 * j + 2}:{r + 4 * j + 3}]"
        e(f"v_fma_f64 v[64:65], {v}, v[{y + 4 * j}:{y + 4 * j + 1}], v[64:65]")
        e(f"v_fma_f64 v[144:145], {v}, v[{y + 4 * j + 2}:{y + 4 * j + 3}], v[144:145]")
    e("s_set_gpr_idx_off")


def phase(lb, db, dys, fb, fys):
    e("s_add_u32 %[lo], %[lo], 64")
    load(lb)
    d4(db, dys)
    e("s_waitcnt vmcnt(4)")          # everything but the 4 loads just issued has landed
    f4(fb, fys)
    e("s_waitcnt lgkmcnt(0)")
    e("s_sub_u32 %[nb], %[nb], 1")
    e("s_cmp_eq_u32 %[nb], 0")
    e("s_cbranch_scc1 9f")


e("s_mov_b32 s85, m0")
e("s_mov_b32 s86, %[pbl]")
e("s_mov_b32 s87, %[pbh]")
e("s_cmp_eq_u32 %[nb], 0")
e("s_cbranch_scc1 9f")
load("A")
e("s_waitcnt lgkmcnt(0)")
e("s_add_u32 %[lo], %[lo], 64")
load("B")
d4("A", "a")
e("s_waitcnt lgkmcnt(0)")
e("1:")
phase("C", "B", "b", "A", "a")
phase("A", "C", "a", "B", "b")
phase("B", "A", "b", "C", "a")
phase("C", "B", "a", "A", "b")
phase("A", "C", "b", "B", "a")
phase("B", "A", "a", "C", "b")
e("s_branch 1b")
e("9:")
e("s_waitcnt vmcnt(0)")
e("s_mov_b32 m0, s85")

dst = os.environ.get("PBGX_ASM_OUT") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sparsearray_amd", "csrc", "pbgx_asm.inc")
with open(dst, "w") as f:
    f.write("// Generated by tools/gen_pbgx_asm.py -- do not edit; see that file for the register map.\n")
    f.write("#define PBGX_PASS_TXT \\\n")
    for ln in out:
        sep = "\\n" if ln.endswith(":") else "\\n\\t"
        f.write(f'\t"{ln}{sep}" \\\n')
    f.write('\t""\n')
print("wrote", dst, len(out), "lines")
