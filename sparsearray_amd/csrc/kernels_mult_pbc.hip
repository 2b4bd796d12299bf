// crossprod(A, Y) fast path: panel-blocked layout + LDS row panels.
//
// Why: the v1 gather kernel (kernels_mult.hip) moves ~37x the algorithmic
// bytes because every nonzero drags its own run of the dense operand from
// HBM (profiles/r01_v1_*).  Here the contracted dimension (rows) is cut into
// panels of R rows whose slice of Y sits in LDS, each wavefront keeps the
// partial sums of its own CBW columns in registers across all panels, and
// every nonzero costs one conflict-free 512-byte LDS read (lane = dense
// column) plus one FP64 FMA per lane.
//
// Device layout "PBC" (panel-blocked columns), built once per sparse operand
// from the CSC layout (svt_dev_pbc_build), the device analogue of the
// reference's per-call "preprocessing" of leaves (src/SparseMatrix_mult.c:
// 632-724):
//   group      = CBW consecutive columns = the partial sums one wavefront keeps
//                in registers (CBW/16 register-indexable vectors, contiguous)
//   tile(g, p) = the nonzeros of group g in row panel p, in CSC order
//                (column-major, rows ascending), stored contiguously and
//                zero-padded to a multiple of PBC_BATCH records of 16 bytes:
//                { u32 8*(row - p*R)   byte offset of the row in an LDS column,
//                  u32 2*(local column) VGPR index of the partial sum,
//                  f64 value }
//                (pre-scaled so that the product loop spends no scalar ALU
//                work on decoding: the scalar unit is shared by the 4 SIMDs of
//                a CU and is the first thing that saturates)
//   tile order = (group, panel): everything one wavefront reads is one
//                sequential stream
//   tile_ptr[g*npanels + p] = first record of tile (g, p)
// A workgroup = WPB wavefronts; grid = (row split, 64-wide tile of dense
// columns, column block).  Row splits give >= 256
// workgroups; their partial results are summed in a fixed order by
// pbc_reduce_kernel (deterministic, no atomics).
//
// Special values: the kernel is the reference's *finite* path
// (_dotprod_doubleSV_finite_doubles, src/SparseVec_dotprod.c:28-43).  While
// staging Y into LDS the column-block-0 workgroups evaluate the prescan
// predicate of src/SparseMatrix_mult.c:23-28 for free; if any dense entry is
// NaN/Inf/NA the general kernels of kernels_mult.hip (which implement the
// slow-path semantics) are run afterwards for the whole product -- they are
// always enqueued and exit at once when the flag is clear, so the call stays
// asynchronous.  A leaf holding an R NA gives NA_real_ (flag computed at
// build time, applied by the reduce kernel).
//
// Roofline: HBM for the algorithmic bytes (12 B/nz + Y + out); the practical
// bound is LDS bandwidth: 8 B of LDS read per (nonzero, dense column) pair.
#include "svt_common.h"

#include <string.h>

#include "svt_scan.h"

#include <vector>

typedef double d16 __attribute__((ext_vector_type(16)));
typedef double d8 __attribute__((ext_vector_type(8)));

struct svt_dev_pbc {
	int64_t nrow, ncol, nnz, nrec;
	int CBW, WPB, logR;
	int fmt;               // record format: 0 = 16-byte records, batches of 4 (register-staged
	                       // kernel); 1 = 12-byte records, batches of 8 (LDS-DMA kernel)
	int gather;            // format 0 with tall panels, read by crossprod_pbc_gather_kernel (very sparse operands)
	int64_t ngroups, nblocks, npanels;
	uint4 *rec;            // [nrec] 16-byte records
	int64_t *tile_ptr;     // [ngroups*npanels + 1]
	int *col_has_na;       // [ncol]
	int64_t max_leaf_nnz;  // the longest leaf (a dense column with more non-finite entries than that makes every cell NaN / NA)
	// streams that have read the layout: svt_dev_pbc_release() records an event on each and frees behind
	// those events instead of synchronising the device (a handle is used from one host thread)
	mutable hipStream_t use_s[4];
	mutable int nuse;
	mutable bool use_overflow;  // more than 4 streams: release falls back to hipDeviceSynchronize()
	int device;
};

#define PCH 256            // panels per build chunk
// Tuning builds only (make TUNING=1 -> -DSVT_TUNING; build() does not produce one): knobs of
// tools/tune_pbc.py.  The product library has constants here and exports none of the setters.
// After which batch of its tile wavefront w of a workgroup issues its LDS-DMA pieces of the next panel
// (bit 14 of that batch's first meta word; nb = batches of the tile, the caller clamps).  Pattern 7 --
// wavefronts 4g..4g+3 after batch g -- is the product's; 12 and 15 time the same, all at once (0) is 3 % slower.
__device__ inline int64_t pbc_stagger_batch(int stag_mode, int w, int64_t nb)
{
	const int g = w >> 2;
	switch (stag_mode) {
	case 0: return 0;
	case 1: return w & 1;
	case 2: return w & 3;
	case 3: return g & 1;
	case 4: return w & 7;
	case 5: return (w * nb) >> 4;
	case 6: return (w & 3) * 2;
	case 7: return g;
	case 8: return g * 2;
	case 9: return w >> 1;
	case 10: return g + 1;
	case 11: return w >> 3;
	case 12: return g < 2 ? g : 2;
	case 13: return g % 3;
	case 14: return g * 3 / 4;
	case 15: return g == 0 ? 0 : g == 3 ? 2 : 1;
	case 16: return g < 1 ? g : 1;
	case 17: return 1 + (w >> 3);
	default: return g;
	}
}

#ifdef SVT_TUNING
static int g_pbc_debug = 0;
static int g_pbc_nsplit = 0;
static int g_pbc_stagger = 7;
static int g_pbc_ahead10 = 20;
#else
static constexpr int g_pbc_debug = 0;
static constexpr int g_pbc_nsplit = 0;
static constexpr int g_pbc_stagger = 7;   // DMA issue: wavefronts 4g..4g+3 after batch g of their tile (patterns 12 and 15 time the same in the bench, tools/debug/r2_rot.sh)
static constexpr int g_pbc_ahead10 = 20;  // record touch: look-ahead in tenths of a tile
#endif

// ---------------------------------------------------------------------------
// layout build
// ---------------------------------------------------------------------------
__device__ inline int64_t lower_bound_row(const int32_t *__restrict__ row, int64_t lo,
					  int64_t hi, int64_t key)
{
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if ((int64_t) row[mid] < key) lo = mid + 1; else hi = mid;
	}
	return lo;
}

#define PBC_BATCH 4        // records per scalar-load batch; tiles are padded to it
#define PBC_AHEAD 2        // panels of look-ahead of the record prefetch into L2
#define PBC_SLACK 320       // zeroed records past the end: look-ahead loads and L2 touches land here
#define PBC_TP_PAD 8        // tile_ptr entries past the end (= nrec): the kernels read up to [p + 3]

// One wavefront per (group of CBW columns, chunk of PCH panels).  MODE 0: count
// the records of each tile (rounded up to PBC_BATCH).  MODE 1: write records to
// their final position and zero-fill the padding.
//
// FMT 1 (see tools/gen_pbc_asm.py): batches of 8 records, 96 bytes = 8 x u32 meta
// (8*row << 16 | last-batch-of-tile flag << 15 | 2*column) then 8 x f64 value; every
// tile has at least one batch (an all-zero one if it holds no nonzero).
__device__ inline void pbc1_store(uint4 *rec, int64_t ridx, uint32_t meta, double x)
{
	char *base = (char *) rec + (ridx >> 3) * 96;
	((uint32_t *) base)[ridx & 7] = meta;
	((double *) (base + 32))[ridx & 7] = x;
}

template <int MODE, int FMT>
__global__ void __launch_bounds__(64)
pbc_pass_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
		const double *__restrict__ val, int64_t ncol, int CBW, int logR,
		int64_t npanels, int64_t *__restrict__ counts_or_ptr,
		uint4 *__restrict__ rec, int *__restrict__ col_has_na, int stag_mode, int tile_flags)
{
	constexpr int BATCH = FMT == 1 ? 8 : PBC_BATCH;
	__shared__ int64_t fill[PCH];
	const int lane = threadIdx.x;
	const int64_t wv = blockIdx.x;
	const int64_t p0 = (int64_t) blockIdx.y * PCH;
	const int64_t p1 = p0 + PCH < npanels ? p0 + PCH : npanels;
#define TILE_OF(P) (wv * npanels + (P))
	for (int i = lane; i < PCH; i += 64)
		fill[i] = (MODE == 1 && p0 + i < npanels) ? counts_or_ptr[TILE_OF(p0 + i)] : 0;
	__syncthreads();
	const int64_t c0 = wv * CBW;
	const int64_t c1 = c0 + CBW < ncol ? c0 + CBW : ncol;
	for (int64_t c = c0; c < c1; c++) {
		const int64_t beg = col_ptr[c], end = col_ptr[c + 1];
		const int64_t lo = lower_bound_row(row_idx, beg, end, p0 << logR);
		const int64_t hi = lower_bound_row(row_idx, lo, end, p1 << logR);
		int saw_na = 0;
		for (int64_t k0 = lo; k0 < hi; k0 += 64) {
			const int64_t k = k0 + lane;
			const bool active = k < hi;
			const int32_t r = active ? row_idx[k] : 0;
			const int p = active ? (int) (((int64_t) r >> logR) - p0) : -1;
			const int prev_p = __shfl_up(p, 1, 64);
			const int next_p = __shfl_down(p, 1, 64);
			const bool is_first = active && (lane == 0 || p != prev_p);
			const bool is_last = active && (lane == 63 || p != next_p);
			const unsigned long long firsts = __ballot(is_first);
			const unsigned long long below = firsts & ((lane == 63) ? ~0ULL : ((2ULL << lane) - 1));
			const int seg_lane = 63 - __clzll(below | 1ULL);
			const int rank = lane - seg_lane;
			int64_t pos = 0;
			if (active) pos = fill[p] + rank;
			__syncthreads();
			if (MODE == 1 && active) {
				const double x = val[k];
				const uint32_t ro = (uint32_t) (r - (int32_t) ((p + p0) << logR)) * 8u;
				if (FMT == 1) {
					pbc1_store(rec, pos, (ro << 16) | (uint32_t) (c - c0) * 2u, x);
				} else {
					uint4 t;
					t.x = ro;
					t.y = (uint32_t) (c - c0) * 2u;
					t.z = (uint32_t) __double2loint(x);
					t.w = (uint32_t) __double2hiint(x);
					rec[pos] = t;
				}
				if (svt_is_na(x)) saw_na = 1;
			}
			if (is_last) fill[p] += rank + 1;
			__syncthreads();
		}
		if (MODE == 1 && __any(saw_na) && lane == 0)
			col_has_na[c] = 1;
	}
	for (int i = lane; i < PCH; i += 64) {
		if (p0 + i >= npanels) continue;
		if (MODE == 0) {
			int64_t n = (fill[i] + BATCH - 1) / BATCH * BATCH;
			if ((FMT == 1 || tile_flags) && n == 0) n = BATCH;  // no tile without a batch
			counts_or_ptr[TILE_OF(p0 + i)] = n;
		} else {
			// fill[i] = one past the last real record; pad up to the next tile
			const int64_t stop = counts_or_ptr[TILE_OF(p0 + i) + 1];
			for (int64_t q = fill[i]; q < stop; q++) {
				if (FMT == 1) pbc1_store(rec, q, 0u, 0.0);
				else rec[q] = make_uint4(0, 0, 0, 0);
			}
			// gather layouts: bit 15 of the first column word marks the start of a tile (the kernel that runs a
			// whole pass as one pipeline changes panels there; the register index is the word's low byte)
			if (FMT == 0 && tile_flags)
				atomicOr(&((unsigned int *) rec)[counts_or_ptr[TILE_OF(p0 + i)] * 4 + 1], 0x8000u);
			if (FMT == 1) {                             // flag the tile's last batch ...
				atomicOr((unsigned int *) ((char *) rec + ((stop - 8) >> 3) * 96), 0x8000u);
				// ... and the batch after which the wavefront issues the LDS-DMA of the
				// next panel: staggered over the wavefronts of a workgroup, never past
				// the end of the tile
				const int64_t start = counts_or_ptr[TILE_OF(p0 + i)];
				const int w = (int) (wv % 16);
				const int64_t nb = (stop - start) >> 3;
				int64_t bi = pbc_stagger_batch(stag_mode, w, nb);
				if (bi > nb - 1) bi = nb - 1;
				atomicOr((unsigned int *) ((char *) rec + ((start >> 3) + bi) * 96), 0x4000u);
			}
		}
	}
#undef TILE_OF
}

// ---------------------------------------------------------------------------
// Scatter pass of the format-1 layout through LDS.  pbc_pass_kernel<1, 1> writes every record
// with two scattered stores (4 + 8 bytes into 96-byte batches of some other tile each time):
// 4.8 GB leave the chip for 1.3 GB of records at BASELINE config 2
// (profiles/r01_dma_summary.txt).  Here a workgroup owns (column group g, PBC_SUBP consecutive
// panels): their tiles are contiguous in the record array, so the workgroup assembles the byte
// image of that stretch in LDS -- padding zeros and batch flags included -- and writes it out with
// 16-byte stores, every byte once.
//   1. lanes 0 .. nc-1 find their column's nonzeros inside the row range (two binary searches)
//   2. all threads walk the flattened list of those nonzeros: count per (panel, column), and the
//      first position of each (panel, column) run (rows ascend inside a column)
//   3. one thread per panel turns the counts into offsets inside the tile, columns in order:
//      the tile keeps the CSC order of the old builder, whatever the thread timing
//   4. second walk: every record goes to its final slot of the image
//   5. batch flags, copy-out
// A stretch larger than the LDS image (a very dense column group) falls back to global stores.
// ---------------------------------------------------------------------------
#define PBC_IMG_BYTES 24576
#define PBC_SC_KEEP 4

__device__ inline void pbc1_put(char *img, int64_t ridx, uint32_t meta, double x)
{
	char *b = img + (ridx >> 3) * 96;
	((uint32_t *) b)[ridx & 7] = meta;
	((double *) (b + 32))[ridx & 7] = x;
}

// Count pass of the format-1 build as ONE stream over the offsets (round 3; pbc_pass_kernel<0, 1> walks every column
// once per chunk of 256 panels with two binary searches and two barriers per 64 offsets: 0.46 ms at BASELINE
// config 2 for 0.4 GB).  Workgroup = one column group (CBW columns, a wavefront per column at a time, four
// loads per lane in flight): the group's records per panel are counted in LDS (npanels counters), rounded up to
// whole batches (at least one) and written as the group's row of the tile table; the same pass writes the
// table of stretch bounds that pbc_bounds_kernel builds with a binary search per entry (0.15 ms): the offset at
// which a column enters each chunk of `subp` panels is where two neighbouring elements differ in row >> cshift.
#define PBC_COUNT_MAXPANELS 36000
__global__ void __launch_bounds__(1024)
pbc_count_stream_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx, int64_t ncol,
			int CBW, int logR, int64_t npanels, int64_t *__restrict__ counts,
			int32_t *__restrict__ bounds, int64_t nchunks, int cshift)
{
	extern __shared__ uint32_t pbc_hist[];          // [npanels]
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t g = blockIdx.x;
	for (int64_t p = threadIdx.x; p < npanels; p += 1024) pbc_hist[p] = 0;
	__syncthreads();
	const int64_t c0 = g * CBW, c1 = c0 + CBW < ncol ? c0 + CBW : ncol;
	for (int64_t c = c0 + w; c < c1; c += 16) {
		const int64_t beg = col_ptr[c], end = col_ptr[c + 1];
		int32_t *__restrict__ bc = bounds + c * (nchunks + 1);
		int64_t carry = -1;                             // chunk of the element before this trip
		for (int64_t k0 = beg; k0 < end; k0 += 256) {
			int32_t r4[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int64_t k = k0 + u * 64 + lane;
				r4[u] = k < end ? row_idx[k] : 0x7FFFFFFF;
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int64_t kb = k0 + u * 64, k = kb + lane;
				if (kb >= end) break;
				const bool active = k < end;
				if (active) atomicAdd(&pbc_hist[r4[u] >> logR], 1u);
				// chunks that start at this position: those after the previous element's, up to this one's
				// (the first lane past the end closes the column's remaining chunks)
				const int64_t q = active ? (int64_t) (r4[u] >> cshift) : nchunks;
				int64_t pq = __shfl_up(q, 1, 64);
				if (lane == 0) pq = carry;
				carry = __shfl(q, 63, 64);
				if (k <= end)
					for (int64_t i = pq + 1; i <= q; i++) bc[i] = (int32_t) (k - beg);
			}
		}
		// a column whose length is a multiple of 64 (or zero) has not closed its last chunks
		if (lane == 0 && ((end - beg) & 63) == 0) {
			const int64_t ql = end > beg ? (int64_t) (row_idx[end - 1] >> cshift) : -1;
			for (int64_t i = ql + 1; i <= nchunks; i++) bc[i] = (int32_t) (end - beg);
		}
	}
	__syncthreads();
	for (int64_t p = threadIdx.x; p < npanels; p += 1024) {
		int64_t n = ((int64_t) pbc_hist[p] + 7) / 8 * 8;
		if (n == 0) n = 8;                              // no tile without a batch
		counts[g * npanels + p] = n;
	}
}

// bounds[c * (nchunks + 1) + q] = number of nonzeros of column c above row q * subp * R: where the
// stretches of `subp` panels start inside every column.  One independent binary search per entry
// (millions in flight), so that the scatter workgroups below start from two table reads instead of
// a chain of 2 x 14 dependent loads each (38 -> ~8 us per workgroup).
__global__ void pbc_bounds_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
				  int64_t ncol, int64_t nchunks, int64_t rows_per_chunk,
				  int32_t *__restrict__ bounds)
{
	const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= ncol * (nchunks + 1)) return;
	const int64_t c = t / (nchunks + 1), q = t - c * (nchunks + 1);
	const int64_t beg = col_ptr[c], end = col_ptr[c + 1];
	bounds[t] = (int32_t) (lower_bound_row(row_idx, beg, end, q * rows_per_chunk) - beg);
}

// dynamic LDS: [image PBC_IMG_BYTES][cnt subp * nc ints][first subp * nc ints]
__global__ void __launch_bounds__(256)
pbc_scatter_lds_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
		       const double *__restrict__ val, int64_t ncol, int CBW, int logR, int64_t npanels,
		       const int64_t *__restrict__ tile_ptr, char *__restrict__ rec,
		       int *__restrict__ col_has_na, int stag_mode, int subp,
		       const int32_t *__restrict__ bounds, int64_t nchunks)
{
	extern __shared__ __attribute__((aligned(16))) char lds_raw[];
	char *img = lds_raw;
	int *s_cnt = (int *) (lds_raw + PBC_IMG_BYTES);         // counts, then offsets inside the image
	int *s_first = s_cnt + subp * CBW;                      // first flattened index of the (panel, column) run
	__shared__ int64_t s_lo[64];
	__shared__ int s_cum[65];
	__shared__ int s_tile0[257];                            // first record of each tile, relative to the stretch
	const int tid = threadIdx.x;
	const int64_t g = blockIdx.x;
	const int64_t ps = (int64_t) blockIdx.y * subp;
	const int np = (int) (ps + subp <= npanels ? subp : npanels - ps);
	const int64_t c0 = g * CBW;
	// (the last workgroup's trailing groups may lie past the last column: nc = 0, tiles of padding only)
	const int nc = (int) (c0 + CBW <= ncol ? CBW : (c0 < ncol ? ncol - c0 : 0));
	const int64_t base = tile_ptr[g * npanels + ps];
	const int64_t nr = tile_ptr[g * npanels + ps + np] - base;            // records, a multiple of 8
	const int64_t nbytes = (nr >> 3) * 96;
	const bool in_lds = nbytes <= PBC_IMG_BYTES;
	char *dst = in_lds ? img : rec + (base >> 3) * 96;
	// ---- 1
	if (tid < nc) {
		const int64_t c = c0 + tid;
		const int32_t *bc = bounds + c * (nchunks + 1) + blockIdx.y;
		s_lo[tid] = col_ptr[c] + bc[0];
		s_cum[tid + 1] = bc[1] - bc[0];
	}
	if (tid <= np) s_tile0[tid] = (int) (tile_ptr[g * npanels + ps + tid] - base);
	for (int i = tid; i < np * nc; i += 256) { s_cnt[i] = 0; s_first[i] = 0x7fffffff; }
	if (in_lds)
		for (int64_t i = tid; i < nbytes / 16; i += 256) ((uint4 *) img)[i] = make_uint4(0, 0, 0, 0);
	__syncthreads();
	if (tid == 0) {
		s_cum[0] = 0;
		for (int j = 0; j < nc; j++) s_cum[j + 1] += s_cum[j];
	}
	__syncthreads();
	const int total = s_cum[nc];
	// ---- 2  (a thread's first PBC_SC_KEEP elements stay in its registers for step 4: column, row, value --
	// the second walk then repeats neither the search nor the loads; ~820 elements per workgroup at config 2)
	int ka[PBC_SC_KEEP], kr[PBC_SC_KEEP];
	double kx[PBC_SC_KEEP];
#pragma unroll
	for (int it = 0; it < PBC_SC_KEEP; it++) {
		const int i = tid + it * 256;
		ka[it] = 0; kr[it] = 0; kx[it] = 0.0;
		if (i < total) {
			int a = 0, b = nc;                              // column j with cum[j] <= i < cum[j + 1]
			while (b - a > 1) { const int m = (a + b) >> 1; if (s_cum[m] <= i) a = m; else b = m; }
			const int64_t k = s_lo[a] + (i - s_cum[a]);
			ka[it] = a; kr[it] = row_idx[k]; kx[it] = val[k];
		}
	}
#pragma unroll
	for (int it = 0; it < PBC_SC_KEEP; it++) {
		const int i = tid + it * 256;
		if (i < total) {
			const int p = (int) (((int64_t) kr[it] >> logR) - ps);
			atomicAdd(&s_cnt[p * nc + ka[it]], 1);
			atomicMin(&s_first[p * nc + ka[it]], i);
		}
	}
	for (int i = tid + PBC_SC_KEEP * 256; i < total; i += 256) {
		int a = 0, b = nc;
		while (b - a > 1) { const int m = (a + b) >> 1; if (s_cum[m] <= i) a = m; else b = m; }
		const int r = row_idx[s_lo[a] + (i - s_cum[a])];
		const int p = (int) (((int64_t) r >> logR) - ps);
		atomicAdd(&s_cnt[p * nc + a], 1);
		atomicMin(&s_first[p * nc + a], i);
	}
	__syncthreads();
	// ---- 3
	if (tid < np) {
		int run = s_tile0[tid];
		for (int j = 0; j < nc; j++) { const int n = s_cnt[tid * nc + j]; s_cnt[tid * nc + j] = run; run += n; }
		if (!in_lds) {                                      // global fallback: the padding is written here
			for (int q = run; q < s_tile0[tid + 1]; q++) pbc1_put(dst, q, 0u, 0.0);
		}
	}
	__syncthreads();
	// ---- 4
#pragma unroll
	for (int it = 0; it < PBC_SC_KEEP; it++) {
		const int i = tid + it * 256;
		if (i < total) {
			const int a = ka[it], r = kr[it];
			const double x = kx[it];
			const int p = (int) (((int64_t) r >> logR) - ps);
			const int pos = s_cnt[p * nc + a] + (i - s_first[p * nc + a]);
			const uint32_t ro = (uint32_t) (r - (int32_t) ((p + ps) << logR)) * 8u;
			pbc1_put(dst, pos, (ro << 16) | (uint32_t) a * 2u, x);
			if (svt_is_na(x)) col_has_na[c0 + a] = 1;
		}
	}
	for (int i = tid + PBC_SC_KEEP * 256; i < total; i += 256) {
		int a = 0, b = nc;
		while (b - a > 1) { const int m = (a + b) >> 1; if (s_cum[m] <= i) a = m; else b = m; }
		const int64_t k = s_lo[a] + (i - s_cum[a]);
		const int r = row_idx[k];
		const double x = val[k];
		const int p = (int) (((int64_t) r >> logR) - ps);
		const int pos = s_cnt[p * nc + a] + (i - s_first[p * nc + a]);
		const uint32_t ro = (uint32_t) (r - (int32_t) ((p + ps) << logR)) * 8u;
		pbc1_put(dst, pos, (ro << 16) | (uint32_t) a * 2u, x);
		if (svt_is_na(x)) col_has_na[c0 + a] = 1;
	}
	__syncthreads();
	// ---- 5: flag the tile's last batch, and the batch after which the wavefront issues the
	// LDS-DMA of the next panel (staggered over the wavefronts of a workgroup, as pbc_pass_kernel)
	if (tid < np) {
		const int start = s_tile0[tid], stop = s_tile0[tid + 1];
		uint32_t *last = (uint32_t *) (dst + ((stop - 8) >> 3) * 96);
		*last |= 0x8000u;
		const int w = (int) (g % 16);
		const int nb = (stop - start) >> 3;
		int bi = (int) pbc_stagger_batch(stag_mode, w, nb);
		if (bi > nb - 1) bi = nb - 1;
		uint32_t *issue = (uint32_t *) (dst + ((start >> 3) + bi) * 96);
		*issue |= 0x4000u;
	}
	if (!in_lds)
		return;
	__syncthreads();
	uint4 *out = (uint4 *) (rec + (base >> 3) * 96);
	for (int64_t i = tid; i < nbytes / 16; i += 256) out[i] = ((const uint4 *) img)[i];
}

__global__ void pbc_max_leaf_kernel(const int64_t *__restrict__ col_ptr, int64_t ncol, unsigned long long *__restrict__ out)
{
	// (grid-stride: one atomic per wavefront of a SMALL grid -- 15625 atomics on one word cost 0.14 ms for the 1e6
	// leaves of t(A) at BASELINE config 2)
	unsigned long long n = 0ULL;
	for (int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; c < ncol; c += (int64_t) gridDim.x * blockDim.x) {
		const unsigned long long m = (unsigned long long) (col_ptr[c + 1] - col_ptr[c]);
		n = m > n ? m : n;
	}
	for (int off = 32; off > 0; off >>= 1) {
		const unsigned long long o = __shfl_xor(n, off, 64);
		n = o > n ? o : n;
	}
	if ((threadIdx.x & 63) == 0 && n > 0) atomicMax(out, n);
}

// flag = 1 if some column group's record stream does not fit a 32-bit byte cursor
__global__ void pbc_group_limit_kernel(const int64_t *__restrict__ tile_ptr, int64_t npanels,
				       int64_t ngroups, int *__restrict__ flag)
{
	const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= ngroups) return;
	const int64_t n = tile_ptr[(g + 1) * npanels] - tile_ptr[g * npanels];
	if ((n + PBC_SLACK) * 16 >= ((int64_t) 1 << 32)) *flag = 1;
}

// End of a build, on the device: the pad entries of the tile table and the zeroed slack behind the records
// (the look-ahead stages of the kernels read up to 3 batches past the end), and what the host wants to know
// -- meta[0] = records, meta[1] = longest leaf (pbc_max_leaf_kernel), meta[2] = "a column group too large"
// (pbc_group_limit_kernel) -- so that one read-back ends the build instead of five synchronous calls.
__global__ void pbc_build_finish_kernel(int64_t *__restrict__ tile_ptr, int64_t ntiles, char *__restrict__ rec,
					int rbytes, long long *__restrict__ meta)
{
	const int64_t nrec = tile_ptr[ntiles];
	if (threadIdx.x == 0) meta[0] = nrec;
	if (threadIdx.x < PBC_TP_PAD) tile_ptr[ntiles + 1 + threadIdx.x] = nrec;
	uint32_t *slack = (uint32_t *) (rec + (size_t) nrec * rbytes);          // (records are 12 or 16 bytes: 4-byte aligned)
	for (int x = threadIdx.x; x < PBC_SLACK * 4; x += blockDim.x) slack[x] = 0u;
}

// The layout's buffers come from a stream-ordered pool of the library's own, one per device, which keeps what
// is freed up to PBC_POOL_KEEP bytes: hipMalloc of the 1.4 GB record array of BASELINE config 2 maps pages
// for ~0.5 ms on every build, and a build used to spend as long in allocation and its five synchronous calls
// as in its kernels (2.24 ms for 1.1 ms of kernels).  Not the device's default pool and not "keep everything":
// memory the pool holds is invisible to torch's caching allocator, so what it may keep is bounded (a rebuild of
// a config-2 layout stays warm; a 6-12 GB layout of config 4 goes back to the driver when it is released) and
// svt_dev_pbc_trim() returns all of it.
#define PBC_POOL_KEEP ((uint64_t) 3 << 30)
#define PBC_MAX_DEV 16
static hipMemPool_t g_pbc_pool[PBC_MAX_DEV];
static bool g_pbc_pool_ok[PBC_MAX_DEV];

static hipMemPool_t pbc_pool(int dev)
{
	if (dev < 0 || dev >= PBC_MAX_DEV)
		return NULL;
	if (!g_pbc_pool_ok[dev]) {
		hipMemPoolProps props;
		memset(&props, 0, sizeof(props));
		props.allocType = hipMemAllocationTypePinned;
		props.handleTypes = hipMemHandleTypeNone;
		props.location.type = hipMemLocationTypeDevice;
		props.location.id = dev;
		hipMemPool_t pool = NULL;
		if (hipMemPoolCreate(&pool, &props) != hipSuccess) {
			(void) hipGetLastError();
			return NULL;
		}
		uint64_t keep = PBC_POOL_KEEP;
		(void) hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
		g_pbc_pool[dev] = pool;
		g_pbc_pool_ok[dev] = true;
	}
	return g_pbc_pool[dev];
}

static hipError_t pbc_alloc(void **p, size_t n)
{
	int dev = 0;
	(void) hipGetDevice(&dev);
	hipMemPool_t pool = pbc_pool(dev);
	if (pool == NULL)
		return hipMallocAsync(p, n ? n : 16, 0);
	return hipMallocFromPoolAsync(p, n ? n : 16, pool, 0);
}
static void pbc_free(void *p) { if (p) (void) hipFreeAsync(p, 0); }

// Returns what the layout pools hold but no layout uses to the driver (all devices this process built on).
extern "C" void svt_dev_pbc_trim(void)
{
	for (int d = 0; d < PBC_MAX_DEV; d++)
		if (g_pbc_pool_ok[d])
			(void) hipMemPoolTrimTo(g_pbc_pool[d], 0);
}

// A product notes its stream here (host side only: an event record per product put a barrier packet behind every
// phase -- two 5.6 us bubbles per step, 4 % of the step at an eighth of the rows of config 2a,
// profiles/r05_share_trace.txt); the events are recorded when the handle is released.
static void pbc_note_use(const svt_dev_pbc *P, hipStream_t s)
{
	for (int i = 0; i < P->nuse; i++)
		if (P->use_s[i] == s) return;
	if (P->nuse == 4) { P->use_overflow = true; return; }
	P->use_s[P->nuse++] = s;
}

extern "C" void svt_dev_pbc_release(svt_dev_pbc *h)
{
	if (h == NULL) return;
	// products on other streams may still be reading the layout: the frees (stream 0, stream-ordered) go
	// behind an event recorded NOW on every stream that has run a product with this handle (it covers all
	// the work that stream has been given so far) -- no device-wide synchronisation, which would also stall
	// a collective or a peer copy in flight when a plan is dropped.  A stream that no longer exists has
	// finished its work; the runtime refuses the record and the device is synchronised instead.
	bool sync = h->use_overflow;
	for (int i = 0; i < h->nuse && !sync; i++) {
		if (h->use_s[i] == (hipStream_t) 0)
			continue;                                // the frees are ordered on stream 0 themselves
		hipEvent_t ev;
		if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { sync = true; break; }
		if (hipEventRecord(ev, h->use_s[i]) != hipSuccess || hipStreamWaitEvent(0, ev, 0) != hipSuccess) {
			(void) hipGetLastError();
			sync = true;
		}
		(void) hipEventDestroy(ev);
	}
	if (sync)
		(void) hipDeviceSynchronize();
	pbc_free(h->rec);
	pbc_free(h->tile_ptr);
	pbc_free(h->col_has_na);
	free(h);
}

extern "C" size_t svt_dev_pbc_bytes(const svt_dev_pbc *h)
{
	if (h == NULL) return 0;
	return (size_t) h->nrec * (h->fmt == 1 ? 12 : 16) + PBC_SLACK * 16 +
	       (size_t) (h->ngroups * h->npanels + 1 + PBC_TP_PAD) * 8 + (size_t) h->ncol * 4;
}

// The layout svt_dev_pbc_build(A, 0, 0, 0) picks.  The LDS-DMA kernel stages whole 128-row panels of
// Y for every block of 640 columns; below ~12 nonzeros per (40-column group, panel) tile -- density
// < 0.25 % -- most staged rows are never used and the gather kernel (rows of Y straight from L2,
// panels of 1024 rows) moves less.
void pbc_auto_layout(int64_t nrow, int64_t ncol, int64_t nnz, int *CBW, int *WPB, int *logR)
{
	*CBW = 40;
	const double per_tile = (double) nnz * 40.0 * 128.0 /
				((double) (nrow > 0 ? nrow : 1) * (double) (ncol > 0 ? ncol : 1));
	if (per_tile < 12.0 && nrow >= 4096) {
		// the tallest panels that leave the XCD-paced gather kernel its 64 panels (fewer tile starts, less
		// padding of tiles to whole batches; config 4's rank share: 2048 rows 4.73 ms, 1024: 4.82, 512: 5.2)
		*WPB = 4;
		*logR = (nrow >> 11) >= 64 ? 11 : (nrow >> 10) >= 64 ? 10 : 9;
	} else { *WPB = 16; *logR = 7; }
}

// Not on the launch path: allocates, synchronises.
extern "C" svt_dev_pbc *svt_dev_pbc_build(const svt_dev_csc *A, int CBW, int WPB, int logR)
{
	if (A->Rtype != SVT_REALSXP) {
		svt_set_error("svt_dev_pbc_build: f64 operands only");
		return NULL;
	}
	if (CBW == 0 && WPB == 0 && logR == 0)
		pbc_auto_layout(A->nrow, A->ncol, A->nnz, &CBW, &WPB, &logR);
	if (CBW <= 0 || CBW > 64 || WPB <= 0 || WPB > 16 || logR < 4 || logR > 15) {
		svt_set_error("svt_dev_pbc_build: bad parameters");
		return NULL;
	}
	svt_dev_pbc *h = (svt_dev_pbc *) calloc(1, sizeof(*h));
	h->nrow = A->nrow; h->ncol = A->ncol; h->nnz = A->nnz;
	h->CBW = CBW; h->WPB = WPB; h->logR = logR;
	// the LDS-DMA kernel wants 16 wavefronts, 128-row panels, <= 40 columns each
	h->fmt = (WPB == 16 && logR == 7 && CBW <= 40 && A->nrow >= 256 && g_pbc_debug != 9) ? 1 : 0;
	h->gather = (h->fmt == 0 && WPB == 4 && logR >= 9) ? 1 : 0;
	const int64_t CB = (int64_t) CBW * WPB;
	h->nblocks = (A->ncol + CB - 1) / CB;
	h->ngroups = h->nblocks * WPB;                // one group per wavefront
	h->npanels = (A->nrow + (1LL << logR) - 1) >> logR;
	if (h->npanels < 1) h->npanels = 1;
	const int64_t ntiles = h->ngroups * h->npanels;
	if (ntiles + 1 + PBC_TP_PAD >= (int64_t) 2147483647) {       // the scan below counts in int
		svt_set_error("svt_dev_pbc_build: too many tiles (%lld) for the panel-blocked layout",
			      (long long) ntiles);
		free(h);
		return NULL;
	}
	void *tmp = NULL;
	int32_t *bounds = NULL;
	size_t tmp_bytes = 0;
	bool ok = pbc_alloc((void **) &h->tile_ptr, (size_t) (ntiles + 1 + PBC_TP_PAD) * 8) == hipSuccess &&
		  pbc_alloc((void **) &h->col_has_na, (size_t) (A->ncol > 0 ? A->ncol : 1) * 4) == hipSuccess;
	if (ok) ok = hipMemsetAsync(h->tile_ptr, 0, (size_t) (ntiles + 1 + PBC_TP_PAD) * 8, 0) == hipSuccess &&
		     hipMemsetAsync(h->col_has_na, 0, (size_t) (A->ncol > 0 ? A->ncol : 1) * 4, 0) == hipSuccess;
	// format 1: the scatter workgroups take `subp` panels at a time -- the fewest (>= 16) that give a column ~8
	// nonzeros per stretch on average (very sparse operands: fewer, larger stretches; the table of stretch bounds
	// then stays smaller than the operand)
	int subp = 16;
	while (subp < 128 && (double) A->nnz * subp < 8.0 * (double) A->ncol * (double) h->npanels)
		subp *= 2;
	const int64_t nchunks = (h->npanels + subp - 1) / subp;
	const int64_t nb_entries = A->ncol * (nchunks + 1);
	bool bounds_done = false;
	if (ok && A->ncol > 0 && A->nnz > 0) {
		const int BATCH = h->fmt == 1 ? 8 : PBC_BATCH;
		const size_t rbytes = h->fmt == 1 ? 12 : 16;
		// Every tile is padded to whole batches (format 1: at least one): at most BATCH - 1 (BATCH)
		// records of padding per tile.  Allocating for that bound spares the build a round trip to
		// the host between its count and scatter passes; the exact count is read back at the end.
		const int64_t nrec_max = A->nnz + ntiles * (int64_t) (h->fmt == 1 || h->gather ? BATCH : BATCH - 1);
		dim3 grid((unsigned) h->ngroups, (unsigned) ((h->npanels + PCH - 1) / PCH));
		if (h->fmt == 1 && h->npanels <= PBC_COUNT_MAXPANELS) {
			// one stream over the offsets: tile counts and the table of stretch bounds together
			ok = pbc_alloc((void **) &bounds, (size_t) nb_entries * 4) == hipSuccess;
			if (ok) {
				int sl = 0;
				while ((1 << sl) < subp) sl++;
				const size_t lds = (size_t) h->npanels * 4;
				(void) hipFuncSetAttribute((const void *) pbc_count_stream_kernel,
							   hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
				hipLaunchKernelGGL(pbc_count_stream_kernel, dim3((unsigned) h->ngroups), dim3(1024), lds, 0,
						   A->col_ptr, A->row_idx, A->ncol, CBW, logR, h->npanels, h->tile_ptr,
						   bounds, nchunks, logR + sl);
				bounds_done = true;
			}
		} else if (h->fmt == 1)
			hipLaunchKernelGGL((pbc_pass_kernel<0, 1>), grid, dim3(64), 0, 0, A->col_ptr, A->row_idx,
					   (const double *) A->val, A->ncol, CBW, logR, h->npanels,
					   h->tile_ptr, (uint4 *) NULL, h->col_has_na, g_pbc_stagger, 0);
		else
			hipLaunchKernelGGL((pbc_pass_kernel<0, 0>), grid, dim3(64), 0, 0, A->col_ptr, A->row_idx,
					   (const double *) A->val, A->ncol, CBW, logR, h->npanels,
					   h->tile_ptr, (uint4 *) NULL, h->col_has_na, g_pbc_stagger, h->gather);
		// exclusive scan in place over ntiles+1 entries (last entry = total)
		tmp_bytes = exclusive_scan_ws_bytes(ntiles + 1);
		ok = pbc_alloc(&tmp, tmp_bytes) == hipSuccess &&
		     launch_exclusive_scan_i64(h->tile_ptr, ntiles + 1, tmp, 0) == 0;
		if (ok) ok = pbc_alloc((void **) &h->rec, (size_t) nrec_max * rbytes + PBC_SLACK * 16) == hipSuccess;
		if (ok) {
			if (h->fmt == 1) {
				if (!bounds_done) ok = pbc_alloc((void **) &bounds, (size_t) nb_entries * 4) == hipSuccess;
				if (ok) {
					if (!bounds_done)
						hipLaunchKernelGGL(pbc_bounds_kernel, dim3((unsigned) ((nb_entries + 255) / 256)), dim3(256), 0, 0,
								   A->col_ptr, A->row_idx, A->ncol, nchunks, (int64_t) subp << logR, bounds);
					const size_t lds = PBC_IMG_BYTES + (size_t) 2 * subp * CBW * 4;
					(void) hipFuncSetAttribute((const void *) pbc_scatter_lds_kernel,
								   hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
					dim3 sgrid((unsigned) h->ngroups, (unsigned) nchunks);
					hipLaunchKernelGGL(pbc_scatter_lds_kernel, sgrid, dim3(256), lds, 0, A->col_ptr, A->row_idx,
							   (const double *) A->val, A->ncol, CBW, logR, h->npanels,
							   h->tile_ptr, (char *) h->rec, h->col_has_na, g_pbc_stagger, subp,
							   bounds, nchunks);
				}
			} else {
				hipLaunchKernelGGL((pbc_pass_kernel<1, 0>), grid, dim3(64), 0, 0, A->col_ptr,
						   A->row_idx, (const double *) A->val, A->ncol, CBW, logR,
						   h->npanels, h->tile_ptr, h->rec, h->col_has_na, g_pbc_stagger, h->gather);
			}
		}
		// One read-back for the whole build: records, longest leaf, "a column group too large for the
		// kernels' 32-bit byte cursor" (tmp, the scan's scratch, is free again: >= 256 bytes).
		long long hmeta[3] = {0, 0, 0};
		bool too_big = false;
		if (ok) {
			long long *meta = (long long *) tmp;
			ok = hipMemsetAsync(meta, 0, 24, 0) == hipSuccess;
			if (ok) {
				hipLaunchKernelGGL(pbc_max_leaf_kernel, dim3((unsigned) ((A->ncol + 255) / 256 < 512 ? (A->ncol + 255) / 256 : 512)), dim3(256), 0, 0,
						   A->col_ptr, A->ncol, (unsigned long long *) (meta + 1));
				hipLaunchKernelGGL(pbc_group_limit_kernel, dim3((unsigned) ((h->ngroups + 255) / 256)),
						   dim3(256), 0, 0, h->tile_ptr, h->npanels, h->ngroups, (int *) (meta + 2));
				hipLaunchKernelGGL(pbc_build_finish_kernel, dim3(1), dim3(256), 0, 0, h->tile_ptr, ntiles,
						   (char *) h->rec, (int) rbytes, meta);
				ok = hipMemcpy(hmeta, meta, 24, hipMemcpyDeviceToHost) == hipSuccess;      // (synchronises)
			}
			h->nrec = hmeta[0];
			h->max_leaf_nnz = hmeta[1];
			too_big = hmeta[2] != 0;
		}
		if (ok && too_big) {
			svt_set_error("svt_dev_pbc_build: a column group too large for 32-bit record offsets");
			pbc_free(tmp);
			pbc_free(bounds);
			svt_dev_pbc_release(h);
			return NULL;
		}
	}
	pbc_free(tmp);
	pbc_free(bounds);
	if (!ok) {
		svt_set_error("svt_dev_pbc_build failed: %s", hipGetErrorString(hipGetLastError()));
		svt_dev_pbc_release(h);
		return NULL;
	}
	return h;
}

// ---------------------------------------------------------------------------
// main kernel
// CUs the LDS-DMA product kernel leaves idle (0 = none).  Its workgroups fill a CU each (132 KB of LDS,
// 16 x 128 VGPRs), so a collective's kernels (RCCL) cannot start beside it; with n > 0 the row splits are
// chosen so that at most 256 - n workgroups are launched -- 8 n / 256 CUs of every XCD stay free -- and
// every split's workgroups go round the XCDs (include/svt_hip.h: svt_dev_pbc_set_spare_cus).
static int g_pbc_spare_cus = 0;
extern "C" void svt_dev_pbc_set_spare_cus(int n)
{
	g_pbc_spare_cus = n < 0 ? 0 : n > 192 ? 192 : (n + 7) / 8 * 8;
}
extern "C" int svt_dev_pbc_spare_cus(void) { return g_pbc_spare_cus; }
// ---------------------------------------------------------------------------
#ifdef SVT_TUNING
// 0 = normal; 2 = timing-only build without the record loop (results wrong by construction);
// 3 = normal results + per-section cycle counts of workgroup 0 (svt_dev_pbc_read_prof);
// 100 + n = force n row splits; 200 + m = DMA issue stagger mode; 300 + t = record-touch look-ahead
extern "C" void svt_dev_pbc_set_debug(int mode)
{
	if (mode >= 300) g_pbc_ahead10 = mode - 300;
	else if (mode >= 200) g_pbc_stagger = mode - 200;
	else if (mode >= 100) g_pbc_nsplit = mode - 100;
	else g_pbc_debug = mode;
}
#endif

struct PbcFlags {
	int *y_nonfinite;    // [1] any NaN/Inf/NA in the dense operand
};

// DBG == 3 (tuning only): cycles per section, per wavefront of workgroup (0,0,0):
// [w][0] fetch issue, [1] record loop, [2] barrier after the loop, [3] commit,
// [4] barrier after commit, [5] panels
// (kept in the flag block at the head of the workspace, its last 1024 bytes)
#define PBC_SUBFLAG0 16           // flags[16 .. 63]: "a non-finite entry in block (row split, dense tile) of Y", index modulo 48
#define PBC_NSUBFLAG 48
#define PBC_FLAG_BYTES 16384     // [0, 256) flags; [256, 8192) per-column counters of the dirty-column fix-up when they fit;
                                 // [8192, 16384) progress words of the XCD-paced gather kernel (8 XCDs x 256)
#define PBC_PROG_OFFSET 8192
#ifdef SVT_TUNING
extern "C" int svt_dev_pbc_read_prof(const void *ws, unsigned long long *out)
{
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(out, (const char *) ws + 7168, 16 * 8 * 8, hipMemcpyDeviceToHost));
	return 0;
}
#endif

// A batch of PBC_BATCH (= 4) records = 16 dwords, held in a block of 16 SGPRs
// that is pinned to fixed physical registers (s[32:47] / s[48:63]) so that the
// hand-written step below can name the fields directly: no decoding, no copies.
typedef uint32_t batch_t __attribute__((ext_vector_type(16)));

// Staging of one Y panel through registers: NPF 16-byte pieces per thread,
// fetched while the previous panel is being consumed, written to LDS between
// two barriers.  LDS layout ylds[k][r], row stride RS = R + 1 doubles (odd, so
// that the lane = k reads of the product loop are bank-conflict free).
// Piece e (0 .. 32R-1) of a panel: column-major Y -> rows 2*(e % (R/2)), +1 of
// dense column e / (R/2); row-contiguous Y (TRY) -> dense columns 2*(e % 32),
// +1 of row e / 32.  Thread (w, lane) owns pieces (w*NPF + q)*64 + lane.
// Panels that stick out of Y (last rows, K not a multiple of 64) or that are
// not 16-byte aligned take the element-wise path in commit().
template <int NPF, int WPB, int LOGR, bool TRY>
struct Stager {
	double2 pf[NPF];
	bool fast = false;

	__device__ static inline void piece(int e, int &kk, int &rr)
	{
		constexpr int R = 1 << LOGR;
		if (!TRY) { kk = e / (R / 2); rr = (e % (R / 2)) * 2; }
		else { kk = (e % 32) * 2; rr = e / 32; }
	}

	__device__ inline void fetch(const double *__restrict__ Y, int64_t ldY, int64_t nrow,
				     int K, int k0, int64_t p, int w, int lane)
	{
		constexpr int R = 1 << LOGR;
		const int64_t r0 = p << LOGR;
		fast = (r0 + R <= nrow) && (k0 + 64 <= K) && ((ldY & 1) == 0) &&
		       ((((uintptr_t) Y) & 15) == 0);          // wave-uniform
		if (!fast)
			return;
#pragma unroll
		for (int q = 0; q < NPF; q++) {
			int kk, rr;
			piece((w * NPF + q) * 64 + lane, kk, rr);
			const double *src = TRY ? Y + (k0 + kk) + (r0 + rr) * ldY
						: Y + (r0 + rr) + (int64_t) (k0 + kk) * ldY;
			pf[q] = *(const double2 *) src;
		}
	}

	__device__ inline void commit(double *__restrict__ ylds, const double *__restrict__ Y,
				      int64_t ldY, int64_t nrow, int K, int k0, int64_t p,
				      int w, int lane, int &bad)
	{
		constexpr int R = 1 << LOGR, RS = R + 1;
		if (fast) {
#pragma unroll
			for (int q = 0; q < NPF; q++) {
				int kk, rr;
				piece((w * NPF + q) * 64 + lane, kk, rr);
				if (!svt_is_finite(pf[q].x) || !svt_is_finite(pf[q].y)) bad = 1;
				ylds[kk * RS + rr] = pf[q].x;
				if (!TRY) ylds[kk * RS + rr + 1] = pf[q].y;
				else ylds[(kk + 1) * RS + rr] = pf[q].y;
			}
			return;
		}
		const int64_t r0 = p << LOGR;
		for (int idx = w * 64 + lane; idx < 64 * R; idx += WPB * 64) {
			const int kk = TRY ? (idx & 63) : (idx >> LOGR);
			const int rr = TRY ? (idx >> 6) : (idx & (R - 1));
			const int64_t r = r0 + rr;
			double y = 0.0;
			if (r < nrow && k0 + kk < K)
				y = TRY ? Y[(k0 + kk) + r * ldY] : Y[r + (int64_t) (k0 + kk) * ldY];
			if (!svt_is_finite(y)) bad = 1;
			ylds[kk * RS + rr] = y;
		}
	}
};

// The record loop of one tile (the records of one wavefront in one panel),
// written by hand: the scalar unit is shared by the 4 SIMDs of a CU, so every
// SALU instruction per record counts, and the compiler's lowering of a
// register-indexed accumulator costs 2 mode switches + 4 v_mov per record
// (tools/micro/idx_bench.hip).
//
// Software pipeline over batches of 4 records, 3 stages:
//   L(k+2)  one 64-byte scalar load                       -> SGPR block
//   D(k+1)  4 x (LDS address = lane base + row offset), 4 x ds_read_b64
//           (lane = dense column: bank-conflict free)      -> y set
//   F(k)    4 x acc[c_q] += a_q * y_q in VGPR-index mode: all partial sums of
//           the wavefront are pinned to v[64 ...]; each v_fma_f64 addresses
//           source-2 and destination relative to M0 = 2*c_q
//           (1 SALU + 1 VALU per record)
// and ONE `s_waitcnt lgkmcnt(0)` per step, after the FMAs: the scalar load and
// the LDS reads of the step complete under them.  Three SGPR blocks (s[36:51],
// s[52:67], s[68:83]) and two y sets rotate, hence 6 phases per loop trip.
// The stream of a wavefront is contiguous (and has 3 batches of slack at the
// end of the array), so the look-ahead stages never need to know where the
// tile ends; row offsets of look-ahead records are valid LDS addresses.
// Record = {s+0: LDS byte offset of the row, s+1: 2*column, s[+2:+3]: value}.
#define PBC_D4(B0, B1, B2, B3, YS)                                                    \
	"v_add_u32 %[t0], s" #B0 ", %[lb]\n\t"                                          \
	"v_add_u32 %[t1], s" #B1 ", %[lb]\n\t"                                          \
	"v_add_u32 %[t2], s" #B2 ", %[lb]\n\t"                                          \
	"v_add_u32 %[t3], s" #B3 ", %[lb]\n\t"                                          \
	"ds_read_b64 %[" #YS "0], %[t0]\n\t"                                           \
	"ds_read_b64 %[" #YS "1], %[t1]\n\t"                                           \
	"ds_read_b64 %[" #YS "2], %[t2]\n\t"                                           \
	"ds_read_b64 %[" #YS "3], %[t3]\n\t"
#define PBC_F4(I0, A0L, A0H, I1, A1L, A1H, I2, A2L, A2H, I3, A3L, A3H, YS)              \
	"s_set_gpr_idx_on s" #I0 ", gpr_idx(SRC2,DST)\n\t"                              \
	"v_fma_f64 v[64:65], s[" #A0L ":" #A0H "], %[" #YS "0], v[64:65]\n\t"            \
	"s_set_gpr_idx_idx s" #I1 "\n\t"                                                \
	"v_fma_f64 v[64:65], s[" #A1L ":" #A1H "], %[" #YS "1], v[64:65]\n\t"            \
	"s_set_gpr_idx_idx s" #I2 "\n\t"                                                \
	"v_fma_f64 v[64:65], s[" #A2L ":" #A2H "], %[" #YS "2], v[64:65]\n\t"            \
	"s_set_gpr_idx_idx s" #I3 "\n\t"                                                \
	"v_fma_f64 v[64:65], s[" #A3L ":" #A3H "], %[" #YS "3], v[64:65]\n\t"            \
	"s_set_gpr_idx_off\n\t"
// block A = s[36:51], B = s[52:67], C = s[68:83]  (s32-s35 are the stack/frame
// pointer registers of the calling convention and are left alone)
#define PBC_LOAD_A "s_load_dwordx16 s[36:51], %[base], %[lo]\n\t"
#define PBC_LOAD_B "s_load_dwordx16 s[52:67], %[base], %[lo]\n\t"
#define PBC_LOAD_C "s_load_dwordx16 s[68:83], %[base], %[lo]\n\t"
#define PBC_D_A(YS) PBC_D4(36, 40, 44, 48, YS)
#define PBC_D_B(YS) PBC_D4(52, 56, 60, 64, YS)
#define PBC_D_C(YS) PBC_D4(68, 72, 76, 80, YS)
#define PBC_F_A(YS) PBC_F4(37, 38, 39, 41, 42, 43, 45, 46, 47, 49, 50, 51, YS)
#define PBC_F_B(YS) PBC_F4(53, 54, 55, 57, 58, 59, 61, 62, 63, 65, 66, 67, YS)
#define PBC_F_C(YS) PBC_F4(69, 70, 71, 73, 74, 75, 77, 78, 79, 81, 82, 83, YS)
// one phase: load into LB, LDS reads for DB into set YD, FMAs of FB with set YF
#define PBC_PHASE(LOADTXT, DTXT, FTXT)                                                 \
	"s_add_u32 %[lo], %[lo], 64\n\t"                                               \
	LOADTXT DTXT FTXT                                                              \
	"s_waitcnt lgkmcnt(0)\n\t"                                                     \
	"s_sub_u32 %[nb], %[nb], 1\n\t"                                                \
	"s_cmp_eq_u32 %[nb], 0\n\t"                                                    \
	"s_cbranch_scc1 9f\n\t"
#define PBC_PANEL_TXT                                                                  \
	"s_mov_b32 s85, m0\n\t"             /* VGPR-index mode rewrites M0 */           \
	"s_cmp_eq_u32 %[nb], 0\n\t"                                                    \
	"s_cbranch_scc1 9f\n\t"                                                        \
	/* prologue: batch 0 -> A, then its LDS reads + batch 1 -> B */                \
	PBC_LOAD_A                                                                     \
	"s_waitcnt lgkmcnt(0)\n\t"                                                     \
	"s_add_u32 %[lo], %[lo], 64\n\t"                                               \
	PBC_LOAD_B PBC_D_A(ya)                                                         \
	"s_waitcnt lgkmcnt(0)\n"                                                       \
	"1:\n\t"                                                                       \
	PBC_PHASE(PBC_LOAD_C, PBC_D_B(yb), PBC_F_A(ya))                                \
	PBC_PHASE(PBC_LOAD_A, PBC_D_C(ya), PBC_F_B(yb))                                \
	PBC_PHASE(PBC_LOAD_B, PBC_D_A(yb), PBC_F_C(ya))                                \
	PBC_PHASE(PBC_LOAD_C, PBC_D_B(ya), PBC_F_A(yb))                                \
	PBC_PHASE(PBC_LOAD_A, PBC_D_C(yb), PBC_F_B(ya))                                \
	PBC_PHASE(PBC_LOAD_B, PBC_D_A(ya), PBC_F_C(yb))                                \
	"s_branch 1b\n"                                                                \
	"9:\n\t"                                                                       \
	"s_mov_b32 m0, s85\n\t"
#define PBC_PANEL_OPS                                                                  \
	[lo] "+s"(lo_), [nb] "+s"(nb_),                                                \
	[t0] "=&v"(t0_), [t1] "=&v"(t1_), [t2] "=&v"(t2_), [t3] "=&v"(t3_),              \
	[ya0] "=&v"(ya0_), [ya1] "=&v"(ya1_), [ya2] "=&v"(ya2_), [ya3] "=&v"(ya3_),      \
	[yb0] "=&v"(yb0_), [yb1] "=&v"(yb1_), [yb2] "=&v"(yb2_), [yb3] "=&v"(yb3_)
#define PBC_PANEL_CLOBBERS "scc", "memory", "s85", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84"

// DBG: 0 = product build; 1 = skip staging of Y (timing only); 2 = skip the
// record loop (timing only).  Selected with svt_dev_pbc_set_debug().
template <int NV, int WPB, int LOGR, bool TRY, int DBG>
__global__ void __launch_bounds__(WPB * 64)
crossprod_pbc_kernel(const uint4 *__restrict__ rec,
		     const int64_t *__restrict__ tile_ptr, int64_t npanels,
		     const double *__restrict__ Y, int64_t ldY, int64_t nrow, int K,
		     int64_t ncol, int64_t panels_per_split, double *__restrict__ part,
		     int64_t Kp, PbcFlags fl, int CBW)
{
	// The only LDS object of this kernel: its byte offset is 0, which
	// lds_read_batch() relies on.
	extern __shared__ double ylds[];            // [64][R + 1]
	constexpr int R = 1 << LOGR;
	constexpr int RS = R + 1;
	constexpr int NPF = R / (2 * WPB);          // 16-byte pieces per thread per panel
	static_assert(NPF >= 1 && NPF * 2 * WPB == R, "panel must split evenly over the workgroup");
	const int tid = threadIdx.x, lane = tid & 63;
	// the wavefront id is wave-uniform; tell the compiler so that everything
	// derived from it (tile bounds, records) lives in SGPRs / scalar loads
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int split = blockIdx.x, kt = blockIdx.y;
	const int64_t b = blockIdx.z;
	const int64_t wv = b * WPB + w;             // global wavefront-group index
	const int64_t pa = (int64_t) split * panels_per_split;
	const int64_t pb = pa + panels_per_split < npanels ? pa + panels_per_split : npanels;
	const int k0 = kt * 64;
	if (pa >= pb)
		return;                                 // whole workgroup: no barrier crossed yet

	d16 acc[NV];
#pragma unroll
	for (int i = 0; i < NV; i++) acc[i] = 0.0;
	int bad = 0;
	uint32_t touch = 0, tv = 0;
	const uint32_t lane_base = (uint32_t) lane * (RS * 8);
	Stager<NPF, WPB, LOGR, TRY> st;

	// prologue: panel pa straight into LDS
	if (DBG != 1) {
		st.fetch(Y, ldY, nrow, K, k0, pa, w, lane);
		st.commit(ylds, Y, ldY, nrow, K, k0, pa, w, lane, bad);
	}
	__syncthreads();
	const int64_t *__restrict__ tb = tile_ptr + (wv * npanels + pa);
	// byte offset of the wavefront's current batch, relative to its own first record
	// (32-bit: svt_dev_pbc_build refuses layouts in which one group's stream reaches 4 GiB)
	const uint4 *__restrict__ rec_w = rec + tb[0];
	uint32_t off = 0;

	unsigned long long pr[6] = {0, 0, 0, 0, 0, 0}, tq = 0;
#define PBC_PROF(i) if (DBG == 3) { const unsigned long long t_ = __builtin_readcyclecounter(); pr[i] += t_ - tq; tq = t_; }
	if (DBG == 3) tq = __builtin_readcyclecounter();
	for (int64_t p = pa; p < pb; p++, tb += 1) {
		const int64_t tbeg = tb[0], tend = tb[1];
		// next panel of Y starts its trip from L2/HBM now, lands in registers
		if (DBG != 1 && p + 1 < pb) st.fetch(Y, ldY, nrow, K, k0, p + 1, w, lane);
		// Pull this wavefront's records of panel p + PBC_AHEAD towards L2: one
		// dword per 128-byte line.  The loaded word is only consumed one panel
		// later, so nothing waits for it here.
		touch ^= tv;
		tv = 0;
		if (DBG != 2 && p + PBC_AHEAD < npanels) {
			const int64_t ta = tb[PBC_AHEAD], te = tb[PBC_AHEAD + 1];
			const uint32_t toff = lane * 128u;
			const uint32_t len = (uint32_t) (te - ta) * 16u;
			if (toff < len)
				tv = *(const uint32_t *) ((const char *) (rec + ta) + toff);
		}

		PBC_PROF(0)
		// ---- this wavefront's records of the panel --------------------------
		if (DBG == 2) {
			acc[0][0] += (double) tbeg;
		} else {
			uint32_t nb_ = (uint32_t) ((tend - tbeg) / PBC_BATCH);
			uint32_t lo_ = off;                 // byte offset of the stream cursor
			off += nb_ * (PBC_BATCH * 16u);
			uint32_t t0_, t1_, t2_, t3_;
			double ya0_, ya1_, ya2_, ya3_, yb0_, yb1_, yb2_, yb3_;
			if constexpr (NV == 1) {
				asm volatile(PBC_PANEL_TXT
					     : "+{v[64:95]}"(acc[0]), PBC_PANEL_OPS
					     : [base] "s"(rec_w), [lb] "v"(lane_base)
					     : PBC_PANEL_CLOBBERS);
			} else if constexpr (NV == 2) {
				asm volatile(PBC_PANEL_TXT
					     : "+{v[64:95]}"(acc[0]), "+{v[96:127]}"(acc[NV > 1 ? 1 : 0]),
					       PBC_PANEL_OPS
					     : [base] "s"(rec_w), [lb] "v"(lane_base)
					     : PBC_PANEL_CLOBBERS);
			} else if constexpr (NV == 3) {
				asm volatile(PBC_PANEL_TXT
					     : "+{v[64:95]}"(acc[0]), "+{v[96:127]}"(acc[NV > 1 ? 1 : 0]),
					       "+{v[128:159]}"(acc[NV > 2 ? 2 : 0]), PBC_PANEL_OPS
					     : [base] "s"(rec_w), [lb] "v"(lane_base)
					     : PBC_PANEL_CLOBBERS);
			} else {
				asm volatile(PBC_PANEL_TXT
					     : "+{v[64:95]}"(acc[0]), "+{v[96:127]}"(acc[NV > 1 ? 1 : 0]),
					       "+{v[128:159]}"(acc[NV > 2 ? 2 : 0]),
					       "+{v[160:191]}"(acc[NV > 3 ? 3 : 0]), PBC_PANEL_OPS
					     : [base] "s"(rec_w), [lb] "v"(lane_base)
					     : PBC_PANEL_CLOBBERS);
			}
		}
		PBC_PROF(1)
		if (p + 1 < pb) {
			__syncthreads();                    // panel p fully consumed
			PBC_PROF(2)
			if (DBG != 1) st.commit(ylds, Y, ldY, nrow, K, k0, p + 1, w, lane, bad);
			PBC_PROF(3)
			__syncthreads();
			PBC_PROF(4)
		}
		if (DBG == 3) pr[5]++;
	}
	if (DBG == 3 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0)
		for (int i = 0; i < 6; i++) ((unsigned long long *) (fl.y_nonfinite + 1792))[w * 8 + i] = pr[i];
	if (b == 0 && __any(bad) && lane == 0)
		*fl.y_nonfinite = 1;
	touch ^= tv;
	if (touch == 0x9E3779B9u && K < 0)      // never true: keeps the prefetch loads alive
		fl.y_nonfinite[1] = 1;
	// ---- partial results: part[(split*Kp + k) * ncol + c] -----------------
	const int64_t c0 = wv * CBW;
	double *__restrict__ dst = part + ((int64_t) split * Kp + k0 + lane) * ncol + c0;
	if (CBW == 16 * NV && c0 + 16 * NV <= ncol) {   // wave-uniform: whole slabs inside
#pragma unroll
		for (int ii = 0; ii < NV; ii++)
#pragma unroll
			for (int jj = 0; jj < 16; jj++)
				dst[ii * 16 + jj] = acc[ii][jj];
	} else {
#pragma unroll
		for (int ii = 0; ii < NV; ii++)
#pragma unroll
			for (int jj = 0; jj < 16; jj++)
				if (ii * 16 + jj < CBW && c0 + ii * 16 + jj < ncol) dst[ii * 16 + jj] = acc[ii][jj];
	}
}

// ---------------------------------------------------------------------------
// Gather kernel: very sparse operands (BASELINE config 4: 0.1 %, 5 nonzeros per (40-column group,
// 128-row panel) tile).  Staging whole panels of Y for every block of columns moves 64 KB into LDS
// per ~80 nonzeros there; here every nonzero fetches the 512 bytes it needs -- row r of the
// row-major, K-padded copy Yt that prep_dense_kernel makes (kernels_mult.hip), 64 dense columns,
// lane = dense column -- straight from L2 into registers.  No LDS, no barrier: a wavefront streams
// the records of its column group (format 0, panels of 2^logR >= 512 rows, padded to batches of 4)
// through the same 3-stage pipeline as crossprod_pbc_kernel, with the LDS reads replaced by
// global loads (two batches = 8 loads in flight per wavefront, counted by vmcnt, which the scalar
// record loads do not share).  All wavefronts of a row split walk the rows in the same order, so
// a row of Yt is fetched from HBM once per XCD and hit in L2 by the other column groups.
// Traffic: 512 B per (nonzero, 64 dense columns) L2 -> CU, i.e. nnz * K * 8 bytes: the floor of an
// out-stationary product when a column block holds less than one nonzero per row.
// ---------------------------------------------------------------------------
#define PBG_D4(B0, B1, B2, B3, YS)                                                    \
	"v_mad_u32_u24 %[t0], s" #B0 ", %[kp], %[l8]\n\t"                               \
	"v_mad_u32_u24 %[t1], s" #B1 ", %[kp], %[l8]\n\t"                               \
	"v_mad_u32_u24 %[t2], s" #B2 ", %[kp], %[l8]\n\t"                               \
	"v_mad_u32_u24 %[t3], s" #B3 ", %[kp], %[l8]\n\t"                               \
	"global_load_dwordx2 %[" #YS "0], %[t0], %[pb]\n\t"                             \
	"global_load_dwordx2 %[" #YS "1], %[t1], %[pb]\n\t"                             \
	"global_load_dwordx2 %[" #YS "2], %[t2], %[pb]\n\t"                             \
	"global_load_dwordx2 %[" #YS "3], %[t3], %[pb]\n\t"
// accumulators pinned to v[64:143]: 144 VGPRs, 12 wavefronts per CU.  (Pinned to v[40:119] the kernel
// runs 16 per CU and takes 60 ms instead of 53 at BASELINE config 4: more wavefronts spread over more
// rows of Yt than the L2 holds.)
#define PBG_F4(I0, A0L, A0H, I1, A1L, A1H, I2, A2L, A2H, I3, A3L, A3H, YS)              \
	"s_set_gpr_idx_on s" #I0 ", gpr_idx(SRC2,DST)\n\t"                              \
	"v_fma_f64 v[64:65], s[" #A0L ":" #A0H "], %[" #YS "0], v[64:65]\n\t"            \
	"s_set_gpr_idx_idx s" #I1 "\n\t"                                                \
	"v_fma_f64 v[64:65], s[" #A1L ":" #A1H "], %[" #YS "1], v[64:65]\n\t"            \
	"s_set_gpr_idx_idx s" #I2 "\n\t"                                                \
	"v_fma_f64 v[64:65], s[" #A2L ":" #A2H "], %[" #YS "2], v[64:65]\n\t"            \
	"s_set_gpr_idx_idx s" #I3 "\n\t"                                                \
	"v_fma_f64 v[64:65], s[" #A3L ":" #A3H "], %[" #YS "3], v[64:65]\n\t"            \
	"s_set_gpr_idx_off\n\t"
#define PBG_F_A(YS) PBG_F4(37, 38, 39, 41, 42, 43, 45, 46, 47, 49, 50, 51, YS)
#define PBG_F_B(YS) PBG_F4(53, 54, 55, 57, 58, 59, 61, 62, 63, 65, 66, 67, YS)
#define PBG_F_C(YS) PBG_F4(69, 70, 71, 73, 74, 75, 77, 78, 79, 81, 82, 83, YS)
#define PBG_D_A(YS) PBG_D4(36, 40, 44, 48, YS)
#define PBG_D_B(YS) PBG_D4(52, 56, 60, 64, YS)
#define PBG_D_C(YS) PBG_D4(68, 72, 76, 80, YS)
// one phase: record load into LB, Y loads for DB into set YD, FMAs of FB with set YF (whose loads
// are the 4 older of the 8 in flight)
#define PBG_PHASE(LOADTXT, DTXT, FTXT)                                                 \
	"s_add_u32 %[lo], %[lo], 64\n\t"                                               \
	LOADTXT DTXT                                                                   \
	"s_waitcnt vmcnt(4)\n\t"                                                       \
	FTXT                                                                           \
	"s_waitcnt lgkmcnt(0)\n\t"                                                     \
	"s_sub_u32 %[nb], %[nb], 1\n\t"                                                \
	"s_cmp_eq_u32 %[nb], 0\n\t"                                                    \
	"s_cbranch_scc1 9f\n\t"
#define PBG_PANEL_TXT                                                                  \
	"s_mov_b32 s85, m0\n\t"                                                        \
	"s_cmp_eq_u32 %[nb], 0\n\t"                                                    \
	"s_cbranch_scc1 9f\n\t"                                                        \
	PBC_LOAD_A                                                                     \
	"s_waitcnt lgkmcnt(0)\n\t"                                                     \
	"s_add_u32 %[lo], %[lo], 64\n\t"                                               \
	PBC_LOAD_B PBG_D_A(ya)                                                         \
	"s_waitcnt lgkmcnt(0)\n"                                                       \
	"1:\n\t"                                                                       \
	PBG_PHASE(PBC_LOAD_C, PBG_D_B(yb), PBG_F_A(ya))                                \
	PBG_PHASE(PBC_LOAD_A, PBG_D_C(ya), PBG_F_B(yb))                                \
	PBG_PHASE(PBC_LOAD_B, PBG_D_A(yb), PBG_F_C(ya))                                \
	PBG_PHASE(PBC_LOAD_C, PBG_D_B(ya), PBG_F_A(yb))                                \
	PBG_PHASE(PBC_LOAD_A, PBG_D_C(yb), PBG_F_B(ya))                                \
	PBG_PHASE(PBC_LOAD_B, PBG_D_A(ya), PBG_F_C(yb))                                \
	"s_branch 1b\n"                                                                \
	"9:\n\t"                                                                       \
	"s_waitcnt vmcnt(0)\n\t"                                                       \
	"s_mov_b32 m0, s85\n\t"
#define PBG_PANEL_OPS                                                                  \
	[lo] "+s"(lo_), [nb] "+s"(nb_),                                                \
	[t0] "=&v"(t0_), [t1] "=&v"(t1_), [t2] "=&v"(t2_), [t3] "=&v"(t3_),              \
	[ya0] "=&v"(ya0_), [ya1] "=&v"(ya1_), [ya2] "=&v"(ya2_), [ya3] "=&v"(ya3_),      \
	[yb0] "=&v"(yb0_), [yb1] "=&v"(yb1_), [yb2] "=&v"(yb2_), [yb3] "=&v"(yb3_)
#define PBG_CHUNK_PANELS 256     // panels per row split and launch (16: 88 ms, 64: 59, 256: 53, all: 78-87 at config 4)
#define PBG2_CHUNK_PANELS 160    // the same for the kernel with two dense columns per lane (96: 51.8 ms, 128: 49.7, 160: 49.2, 192: 51.1, 256: 55)
#define PBG_PANEL_IN [base] "s"(rec_w), [pb] "s"(pbase), [kp] "v"(kp_v), [l8] "v"(lane8)

template <int NV>
__global__ void __launch_bounds__(256)
crossprod_pbc_gather_kernel(const uint4 *__restrict__ rec, const int64_t *__restrict__ tile_ptr,
			    int64_t npanels, const double *__restrict__ Yt, int64_t Ktp, int K,
			    int64_t ncol, int64_t panels_per_split, double *__restrict__ part, int64_t Kp,
			    int CBW, int logR, int64_t p_first, int64_t p_last, int accumulate)
{
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int split = blockIdx.x, kt = blockIdx.y;
	const int64_t wv = (int64_t) blockIdx.z * 4 + w;          // column group of this wavefront
	// this launch: panels [p_first, p_last) of the operand, cut into gridDim.x row splits
	const int64_t pa = p_first + (int64_t) split * panels_per_split;
	const int64_t pb = pa + panels_per_split < p_last ? pa + panels_per_split : p_last;
	if (pa >= pb)
		return;
	const int k0 = kt * 64;
	// NV == 3 keeps 40 columns (16 + 16 + 8)
	d16 acc[NV > 2 ? 2 : NV];
	d8 acc8 = 0.0;
#pragma unroll
	for (int i = 0; i < (NV > 2 ? 2 : NV); i++) acc[i] = 0.0;
	const uint32_t lane8 = (uint32_t) lane * 8u;
	const uint32_t kp_v = (uint32_t) Ktp;                    // (row offset in bytes = 8 * row * Ktp)
	const int64_t *__restrict__ tb = tile_ptr + (wv * npanels + pa);
	const uint4 *__restrict__ rec_w = rec + tb[0];
	uint32_t off = 0;
	for (int64_t p = pa; p < pb; p++, tb += 1) {
		const int64_t tbeg = tb[0], tend = tb[1];
		uint32_t nb_ = (uint32_t) ((tend - tbeg) / PBC_BATCH);
		uint32_t lo_ = off;
		off += nb_ * (PBC_BATCH * 16u);
		// first row of the panel, this wavefront's 64 dense columns (wave-uniform: an SGPR pair)
		const double *pbase = Yt + (p << logR) * Ktp + k0;
		uint32_t t0_, t1_, t2_, t3_;
		double ya0_, ya1_, ya2_, ya3_, yb0_, yb1_, yb2_, yb3_;
		if constexpr (NV == 1) {
			asm volatile(PBG_PANEL_TXT
				     : "+{v[64:95]}"(acc[0]), PBG_PANEL_OPS
				     : PBG_PANEL_IN
				     : PBC_PANEL_CLOBBERS);
		} else if constexpr (NV == 2) {
			asm volatile(PBG_PANEL_TXT
				     : "+{v[64:95]}"(acc[0]), "+{v[96:127]}"(acc[NV > 1 ? 1 : 0]), PBG_PANEL_OPS
				     : PBG_PANEL_IN
				     : PBC_PANEL_CLOBBERS);
		} else {
			asm volatile(PBG_PANEL_TXT
				     : "+{v[64:95]}"(acc[0]), "+{v[96:127]}"(acc[NV > 1 ? 1 : 0]),
				       "+{v[128:143]}"(acc8), PBG_PANEL_OPS
				     : PBG_PANEL_IN
				     : PBC_PANEL_CLOBBERS);
		}
	}
	// ---- partial results: part[(split*Kp + k) * ncol + c] -----------------
	const int64_t c0 = wv * CBW;
	double *__restrict__ dst = part + ((int64_t) split * Kp + k0 + lane) * ncol + c0;
#pragma unroll
	for (int ii = 0; ii < (NV > 2 ? 2 : NV); ii++)
#pragma unroll
		for (int jj = 0; jj < 16; jj++)
			if (ii * 16 + jj < CBW && c0 + ii * 16 + jj < ncol)
				dst[ii * 16 + jj] = accumulate ? dst[ii * 16 + jj] + acc[ii][jj] : acc[ii][jj];
	if constexpr (NV > 2) {
#pragma unroll
		for (int jj = 0; jj < 8; jj++)
			if (32 + jj < CBW && c0 + 32 + jj < ncol)
				dst[32 + jj] = accumulate ? dst[32 + jj] + acc8[jj] : acc8[jj];
	}
}

// ---------------------------------------------------------------------------
// Gather kernel, two dense columns per lane (K a multiple of 128): a nonzero fetches the 1 KiB of its
// row of Yt with ONE 16-byte load per lane and feeds two FMAs -- half the load instructions, half
// the record loads and half the index switches of the kernel above per (nonzero, dense column), and
// 16-byte loads move more bytes per L2 request cycle than 8-byte ones.  A lane's two partial sums of
// column c sit in v[64 + 2c] and v[144 + 2c] (the same register index serves both FMAs); the y
// registers are pinned to v[224:255] (two sets of four 4-register tuples): 256 VGPRs, 8 wavefronts per
// CU, each with 8 KiB in flight.
// ---------------------------------------------------------------------------
#include "pbg2_asm.inc"       // PBG2_PANEL_TXT, generated by tools/gen_pbg2_asm.py
#define PBG2_PANEL_OPS                                                                 \
	[lo] "+s"(lo_), [nb] "+s"(nb_),                                                \
	[t0] "=&v"(t0_), [t1] "=&v"(t1_), [t2] "=&v"(t2_), [t3] "=&v"(t3_),              \
	"+{v[224:255]}"(ysets)
#define PBG2_PANEL_IN [base] "s"(rec_w), [pb] "s"(pbase), [kp] "v"(kp_v), [l16] "v"(lane16)
typedef uint32_t u32x32_t __attribute__((ext_vector_type(32)));

template <int NV>
__global__ void __launch_bounds__(256)
crossprod_pbc_gather2_kernel(const uint4 *__restrict__ rec, const int64_t *__restrict__ tile_ptr,
			     int64_t npanels, const double *__restrict__ Yt, int64_t Ktp, int K,
			     int64_t ncol, int64_t panels_per_split, double *__restrict__ part, int64_t Kp,
			     int CBW, int logR, int64_t p_first, int64_t p_last, int accumulate)
{
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int split = blockIdx.x, kt = blockIdx.y;
	const int64_t wv = (int64_t) blockIdx.z * 4 + w;          // column group of this wavefront
	const int64_t pa = p_first + (int64_t) split * panels_per_split;
	const int64_t pb = pa + panels_per_split < p_last ? pa + panels_per_split : p_last;
	if (pa >= pb)
		return;
	const int k0 = kt * 128;
	// lo: dense column k0 + 2 * lane, hi: k0 + 2 * lane + 1; NV == 3 keeps 40 columns (16 + 16 + 8)
	d16 acc[NV > 2 ? 2 : NV], acch[NV > 2 ? 2 : NV];
	d8 acc8 = 0.0, acch8 = 0.0;
#pragma unroll
	for (int i = 0; i < (NV > 2 ? 2 : NV); i++) { acc[i] = 0.0; acch[i] = 0.0; }
	u32x32_t ysets = 0;
	const uint32_t lane16 = (uint32_t) lane * 16u;
	const uint32_t kp_v = (uint32_t) Ktp;                    // (row offset in bytes = 8 * row * Ktp)
	const int64_t *__restrict__ tb = tile_ptr + (wv * npanels + pa);
	const uint4 *__restrict__ rec_w = rec + tb[0];
	uint32_t off = 0;
	for (int64_t p = pa; p < pb; p++, tb += 1) {
		const int64_t tbeg = tb[0], tend = tb[1];
		uint32_t nb_ = (uint32_t) ((tend - tbeg) / PBC_BATCH);
		uint32_t lo_ = off;
		off += nb_ * (PBC_BATCH * 16u);
		const double *pbase = Yt + (p << logR) * Ktp + k0;
		uint32_t t0_, t1_, t2_, t3_;
		if constexpr (NV == 1) {
			asm volatile(PBG2_PANEL_TXT
				     : "+{v[64:95]}"(acc[0]), "+{v[144:175]}"(acch[0]), PBG2_PANEL_OPS
				     : PBG2_PANEL_IN
				     : PBC_PANEL_CLOBBERS);
		} else if constexpr (NV == 2) {
			asm volatile(PBG2_PANEL_TXT
				     : "+{v[64:95]}"(acc[0]), "+{v[96:127]}"(acc[NV > 1 ? 1 : 0]),
				       "+{v[144:175]}"(acch[0]), "+{v[176:207]}"(acch[NV > 1 ? 1 : 0]), PBG2_PANEL_OPS
				     : PBG2_PANEL_IN
				     : PBC_PANEL_CLOBBERS);
		} else {
			asm volatile(PBG2_PANEL_TXT
				     : "+{v[64:95]}"(acc[0]), "+{v[96:127]}"(acc[NV > 1 ? 1 : 0]), "+{v[128:143]}"(acc8),
				       "+{v[144:175]}"(acch[0]), "+{v[176:207]}"(acch[NV > 1 ? 1 : 0]), "+{v[208:223]}"(acch8),
				       PBG2_PANEL_OPS
				     : PBG2_PANEL_IN
				     : PBC_PANEL_CLOBBERS);
		}
	}
	// ---- partial results: part[(split*Kp + k) * ncol + c] -----------------
	const int64_t c0 = wv * CBW;
	double *__restrict__ dlo = part + ((int64_t) split * Kp + k0 + 2 * lane) * ncol + c0;
	double *__restrict__ dhi = dlo + ncol;
#pragma unroll
	for (int ii = 0; ii < (NV > 2 ? 2 : NV); ii++)
#pragma unroll
		for (int jj = 0; jj < 16; jj++)
			if (ii * 16 + jj < CBW && c0 + ii * 16 + jj < ncol) {
				dlo[ii * 16 + jj] = accumulate ? dlo[ii * 16 + jj] + acc[ii][jj] : acc[ii][jj];
				dhi[ii * 16 + jj] = accumulate ? dhi[ii * 16 + jj] + acch[ii][jj] : acch[ii][jj];
			}
	if constexpr (NV > 2) {
#pragma unroll
		for (int jj = 0; jj < 8; jj++)
			if (32 + jj < CBW && c0 + 32 + jj < ncol) {
				dlo[32 + jj] = accumulate ? dlo[32 + jj] + acc8[jj] : acc8[jj];
				dhi[32 + jj] = accumulate ? dhi[32 + jj] + acch8[jj] : acch8[jj];
			}
	}
}

// ---------------------------------------------------------------------------
// Gather kernel, XCD-paced (round 4; K a multiple of 128, >= 8 panels per XCD).  The two kernels above leave
// 64 GB of gathered rows at BASELINE config 4's per-rank share to the Infinity Cache (10.5 TB/s): their
// wavefronts start a tile with an empty pipeline (two scalar round trips, then the first rows: ~1.5 us of
// a ~5 us tile) and drift apart by more rows of Yt than an XCD's 4 MiB L2 holds.  Here
//   * the grid is persistent -- 2 workgroups per CU, workgroup L on XCD L % 8 (the dispatcher deals
//     workgroups round-robin over the XCDs) -- and XCD x owns the rows of panels [x * npx, (x + 1) * npx):
//     one partial result per XCD, summed in fixed order by pbc_reduce_kernel;
//   * a workgroup takes the column blocks slot, slot + nslots, ... (a "pass" each) and a wavefront runs the
//     whole record stream of a pass -- npx tiles, contiguous in the layout -- as ONE pipeline
//     (pbgx_asm.inc, tools/gen_pbgx_asm.py): the tile-start flag of the layout switches the panel base;
//   * at every tile start a wavefront publishes its tile count and looks at a snapshot of the counts of
//     the XCD's other wavefronts: it does not run more than `dsync` tiles ahead of the slowest one that has
//     started.  All the XCD's wavefronts then gather from a window of (dsync + 1) panels that stays in its
//     L2 (the first gather of a line brings it in for the other 255).  A wavefront more than 8 tiles behind is
//     not waited for (workgroups that could only start late -- CUs held by a collective's kernels at launch --
//     pace themselves among each other).  The protocol only paces: a wavefront that waits in vain (a spin
//     budget) stops looking, and no result depends on it.
// ---------------------------------------------------------------------------
#include "pbgx_asm.inc"       // PBGX_PASS_TXT, generated by tools/gen_pbgx_asm.py
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define PBGX_CLOBBERS "scc", "vcc", "memory", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s85", "s86", "s87", "s88", "s89", "s90", "s94", "s95"
#define PBGX_OPS                                                                       \
	[lo] "+s"(lo_), [nb] "+s"(nb_), [step] "+s"(step), [dsync] "+s"(dsync),          \
	[t0] "=&v"(t0_), [t1] "=&v"(t1_), [t2] "=&v"(t2_), [t3] "=&v"(t3_),              \
	[vt] "=&v"(vt_), [vd] "=&v"(vd_), "+{v[56:59]}"(snap), "+{v[224:255]}"(ysets)
#define PBGX_IN [base] "s"(rec_w), [pbl] "s"(pbl), [pbh] "s"(pbh), [pst] "s"(pst), [spin] "s"(spin_u), \
	[pgm] "s"(pg_mine), [pga] "s"(pg_all), [rbl] "s"(rbl), [rbh] "s"(rbh),            \
	[kp] "v"(kp_v), [l16] "v"(lane16), [l128] "v"(lane128), [vz] "v"(vzero)
#define PBGX_PROG_ENTRIES 256     // progress words per XCD: one per wavefront (64 workgroup slots x 4)
#define PBGX_DONE 0x7fffffffu

template <int NV>
__global__ void __launch_bounds__(256)
crossprod_pbc_gatherx_kernel(const uint4 *__restrict__ rec, const int64_t *__restrict__ tile_ptr,
			     int64_t npanels, int64_t ngroups, const double *__restrict__ Yt, int64_t Ktp,
			     int64_t ncol, double *__restrict__ part, int64_t Kp, int CBW, int logR,
			     int64_t npx, int nkt, int64_t nunits, unsigned int *__restrict__ prog,
			     int dsync_in, int spin_in)
{
	const int tid = threadIdx.x, lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
	const int64_t pa = (int64_t) xcd * npx;
	const int64_t pb = pa + npx < npanels ? pa + npx : npanels;
	if (pa >= pb)
		return;
	unsigned int *pg_all = prog + xcd * PBGX_PROG_ENTRIES;
	unsigned int *pg_mine = pg_all + slot * 4 + w;
	uint32_t step = 0, dsync = (uint32_t) dsync_in;
	const uint32_t spin_u = (uint32_t) spin_in;
	const uint32_t pst = (uint32_t) (((int64_t) 8 << logR) * Ktp);          // bytes of Yt per panel
	const uint32_t lane16 = (uint32_t) lane * 16u, lane128 = (uint32_t) lane * 128u, vzero = 0u;
	const uint32_t kp_v = (uint32_t) Ktp;                    // (row offset in bytes = 8 * row * Ktp)
	u32x4 snap = 0;
	u32x32_t ysets = 0;
	// units (column block, pair of dense tiles) whose column group exists for this wavefront (the last block may be short)
	int ulim = (int) ((ngroups - w + 3) >> 2) * nkt;
	if (ulim > (int) nunits) ulim = (int) nunits;
	for (int u = slot; u < ulim; u += nslots) {
		// (the division runs on the vector unit: back to scalar registers for the asm operands)
		const int kt = __builtin_amdgcn_readfirstlane(u % nkt);
		const int64_t wv = (int64_t) __builtin_amdgcn_readfirstlane(u / nkt) * 4 + w;   // column group of this wavefront
		const int k0 = kt * 128;
		// lo: dense column k0 + 2 * lane, hi: k0 + 2 * lane + 1; NV == 3 keeps 40 columns (16 + 16 + 8)
		d16 acc[NV > 2 ? 2 : NV], acch[NV > 2 ? 2 : NV];
		d8 acc8 = 0.0, acch8 = 0.0;
#pragma unroll
		for (int i = 0; i < (NV > 2 ? 2 : NV); i++) { acc[i] = 0.0; acch[i] = 0.0; }
		const int64_t tbeg = tile_ptr[wv * npanels + pa], tend = tile_ptr[wv * npanels + pb];
		const uint4 *__restrict__ rec_w = rec + tbeg;
		uint32_t nb_ = (uint32_t) ((tend - tbeg) / PBC_BATCH), lo_ = 0;
		// the first tile's flag steps the base onto panel pa
		const uint64_t pbase = (uint64_t) (uintptr_t) Yt + (uint64_t) (((pa - 1) << logR) * Ktp + k0) * 8u;
		const uint32_t pbl = (uint32_t) pbase, pbh = (uint32_t) (pbase >> 32);
		const uint32_t rbl = (uint32_t) (uintptr_t) rec_w, rbh = (uint32_t) ((uint64_t) (uintptr_t) rec_w >> 32);
		uint32_t t0_, t1_, t2_, t3_, vt_, vd_;
		if constexpr (NV == 1) {
			asm volatile(PBGX_PASS_TXT
				     : "+{v[64:95]}"(acc[0]), "+{v[144:175]}"(acch[0]), PBGX_OPS
				     : PBGX_IN
				     : PBGX_CLOBBERS);
		} else if constexpr (NV == 2) {
			asm volatile(PBGX_PASS_TXT
				     : "+{v[64:95]}"(acc[0]), "+{v[96:127]}"(acc[NV > 1 ? 1 : 0]),
				       "+{v[144:175]}"(acch[0]), "+{v[176:207]}"(acch[NV > 1 ? 1 : 0]), PBGX_OPS
				     : PBGX_IN
				     : PBGX_CLOBBERS);
		} else {
			asm volatile(PBGX_PASS_TXT
				     : "+{v[64:95]}"(acc[0]), "+{v[96:127]}"(acc[NV > 1 ? 1 : 0]), "+{v[128:143]}"(acc8),
				       "+{v[144:175]}"(acch[0]), "+{v[176:207]}"(acch[NV > 1 ? 1 : 0]), "+{v[208:223]}"(acch8),
				       PBGX_OPS
				     : PBGX_IN
				     : PBGX_CLOBBERS);
		}
		// ---- partial results of this XCD: part[(xcd*Kp + k) * ncol + c] -------
		const int64_t c0 = wv * CBW;
		double *__restrict__ dlo = part + ((int64_t) xcd * Kp + k0 + 2 * lane) * ncol + c0;
		double *__restrict__ dhi = dlo + ncol;
#pragma unroll
		for (int ii = 0; ii < (NV > 2 ? 2 : NV); ii++)
#pragma unroll
			for (int jj = 0; jj < 16; jj++)
				if (ii * 16 + jj < CBW && c0 + ii * 16 + jj < ncol) {
					dlo[ii * 16 + jj] = acc[ii][jj];
					dhi[ii * 16 + jj] = acch[ii][jj];
				}
		if constexpr (NV > 2) {
#pragma unroll
			for (int jj = 0; jj < 8; jj++)
				if (32 + jj < CBW && c0 + 32 + jj < ncol) {
					dlo[32 + jj] = acc8[jj];
					dhi[32 + jj] = acch8[jj];
				}
		}
	}
	// out of the race: nobody waits for this wavefront any more
	if (lane == 0)
		__hip_atomic_store(pg_mine, PBGX_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// DMA kernel: the same product on the same layout, restructured around the three
// things the register-staged kernel above spends its time on (measured with
// s_memtime per section, tools/tune_pbc.py --prof: per 128-row panel ~3000
// cycles in the record loop and ~4000 in fetch issue, two barriers and the
// LDS write pass):
//   * the Y panel goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, one
//     1 KiB piece = 128 rows of one dense column), double-buffered: the pieces of
//     panel p+1 fly while panel p is consumed, ONE barrier per panel, no VGPRs
//     and no ds_write pass.  LDS image: [dense column][128 rows + 1 pad], i.e.
//     odd columns start 8 mod 16 bytes (tools/micro/stage_bench.hip checks that
//     the DMA accepts that) so the lane = dense-column reads stay conflict-free.
//   * an L2 touch (one dword per 128-byte line) of this wavefront's records two
//     panels ahead, issued AFTER the DMA pieces so that the in-order vmcnt wait
//     for the pieces does not wait for it: without it every scalar load of the
//     record stream pays an HBM miss (7.6 ms instead of 5.2 at BASELINE config 2).
//   * the DMA pieces of the next panel are issued from inside the record loop,
//     after the batch the layout flags for this wavefront (wavefronts 4g .. 4g+3
//     after batch g of their tile), instead of all 64 at the barrier: a piece
//     blocks its wavefront for 100+ cycles and a burst blocks them all.
//   * the whole panel loop is one asm statement (pbc_dma_asm.inc, generated by
//     tools/gen_pbc_asm.py): the scalar-load pipeline of the record stream runs
//     on across panel boundaries (only the LDS reads restart), loads use
//     immediate offsets.  The CU's one scalar unit serves 16 wavefronts and is the
//     busiest pipe of the kernel, so the loop is built to spend few scalar
//     instructions: one flag test per batch, the boundary code once per resume
//     phase (no dispatch chains), staging offsets in a VGPR, bookkeeping ahead of
//     the barrier (DESIGN.md section 4 has the measurements).
// The finiteness prescan of Y (src/SparseMatrix_mult.c:23-28) reads this
// workgroup's 1/nblocks share of every landed panel back from LDS.
// Grid: 1-D; when the number of row splits is a multiple of 8 the decode keeps
// all column blocks of one (row split, dense tile) on one XCD, next to each
// other in launch order, so that they pull the same Y panels through one L2 at
// about the same time.
// Preconditions (checked by the launcher, else the kernel above runs):
// column-major Y, WPB = 16, logR = 7, nrow >= 256.
// ---------------------------------------------------------------------------
#include "pbc_dma_asm.inc"
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
#define PBC_DMA_CLOBBERS "memory", "scc", "vcc", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

template <int NV, int PROF>      // PROF (tuning build): 1 = cycles per section of the panel loop (NV <= 2), 2 = prologue / loop / epilogue only
__global__ void __launch_bounds__(1024)
crossprod_pbc_dma_kernel(const uint4 *__restrict__ rec, const int64_t *__restrict__ tile_ptr,
			 int64_t npanels, const double *__restrict__ Y, int64_t ldY, int64_t nrow,
			 int K, int64_t ncol, int CBW, int nfull, int nblocks, int block0,
			 int64_t panels_per_split, double *__restrict__ part, int64_t Kp,
			 PbcFlags fl, int rt_lines, int rt_ahead, int stag_mode, int64_t part_ld, int64_t part_c0)
{
	extern __shared__ double ylds[];            // 2 buffers x [64][129]
	const int tid = threadIdx.x;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int kt = (int) (Kp / 64);
	const int L = blockIdx.x;
#ifdef SVT_TUNING
	unsigned long long t_entry = 0, t_loop = 0, t_epi = 0;
	if (PROF) t_entry = __builtin_readcyclecounter();
#endif
	// nblocks column blocks are launched, the first of them is block0 of the layout (a symmetric
	// product only needs the blocks from its dense chunk's first column on); bl counts from 0
	int bl, kh, split;
	// nfull > 0: row splits dealt to the XCDs whole, eight at a time.  nfull < 0: CUs are kept free, -nfull
	// workgroups per XCD (workgroup L runs on XCD L % 8): the (split, dense tile, column block) triples are
	// numbered so that an XCD holds consecutive ones -- a (split, dense tile) group, whose workgroups pull
	// the same panels of Y, sits on one XCD or straddles two (all round the XCDs: 2.93 instead of 2.1 ms
	// at config 2a with 32 CUs kept free); numbers past the last split find no panels and leave.
	if (nfull < 0) {
		const int V = (L & 7) * (-nfull) + (L >> 3), u = V / nblocks;
		bl = V % nblocks; kh = u % kt; split = u / kt;
	} else if (L < nfull * kt * nblocks) {
		const int xcd = L & 7, j = L >> 3, u = j / nblocks;
		bl = j % nblocks; kh = u % kt; split = (u / kt) * 8 + xcd;
	} else {                                        // the other splits: their workgroups go round the XCDs
		const int Lr = L - nfull * kt * nblocks, u = Lr / nblocks;
		bl = Lr % nblocks; kh = u % kt; split = nfull + u / kt;
	}
	const int b = bl + block0;
	const int64_t pa = (int64_t) split * panels_per_split;
	const int64_t pb = pa + panels_per_split < npanels ? pa + panels_per_split : npanels;
	if (pa >= pb)
		return;
	const int k0 = kh * 64;
	const int64_t wv = (int64_t) b * 16 + w;
	const int64_t *__restrict__ tb = tile_ptr + (wv * npanels + pa);

	// NV == 3 keeps 40 columns (16 + 16 + 8)
	d16 acc[NV > 2 ? 2 : NV];
	d8 acc8 = 0.0;
#pragma unroll
	for (int i = 0; i < (NV > 2 ? 2 : NV); i++) acc[i] = 0.0;
	const bool partial = (nrow & 127) != 0;

	// Which 4 of the panel's 64 pieces (dense columns) this wavefront stages: its own.  (Rotating the
	// order by the column block -- so that the 16 workgroups of an XCD that pull the same panel do not ask
	// the L2 for the same lines at the same moment -- makes the staging ALONE 24 % faster, 1.28 -> 0.97 ms,
	// tools/micro/ceiling_bench.hip stage2; the product kernel does not notice, 1.730-1.745 ms either
	// way in tools/debug/r2_rot.sh: its workgroups drift apart within a few panels and its panel time is
	// set by the record work and the barrier, not by the staging.)
	const int wd = w;
	// ---- wave-uniform state (SGPR vectors pinned to s[8:11], s[12:27]) ------------
	u32x4 PA;
	u32x16 PB;
	{
		// record base = this wavefront's first record; the 32-bit stream cursor is relative to it
		const uint64_t recp = (uint64_t) (uintptr_t) rec + (uint64_t) tb[0] * 12u;
		PA[0] = (uint32_t) recp; PA[1] = (uint32_t) (recp >> 32);
		PA[2] = 0;                                      // stream cursor (bytes)
		PA[3] = (uint32_t) pa;
		PB[0] = (uint32_t) pb;
		PB[1] = PBC_DMA_BUF;                            // toggles to 0 for the first panel
		PB[2] = (uint32_t) (wd * 4 * PBC_DMA_ROW + PBC_DMA_BUF);
		PB[3] = (uint32_t) (w >> 2);                    // (s15: the wavefront's rank on its SIMD; read by PBC_SKEW builds only)
		// partial last panel (staged as rows nrow-128 .. nrow-1): its index, the byte shift
		PB[4] = partial ? (uint32_t) (npanels - 1) : 0xFFFFFFFFu;
		PB[5] = partial ? (uint32_t) ((128 - (nrow & 127)) * 8) : 0u;
		PB[6] = 0;
		{   // finiteness prescan: iterations - 1 over this block's share of a panel (see V0[3])
			const int chunk = (8192 + nblocks - 1) / nblocks;
			PB[7] = (uint32_t) ((chunk + 1023) / 1024 - 1);
		}
#pragma unroll
		for (int q = 0; q < 4; q++) {
			int kk = k0 + wd * 4 + q;
			if (kk > K - 1) kk = K - 1;                 // tail of K: a valid column, never stored
			// (bases one panel back, the offset register starts one panel in: the shift of a
			// partial last panel must not take the unsigned 32-bit offset below zero)
			const uint64_t src = (uint64_t) (uintptr_t) (Y + (int64_t) kk * ldY + pa * 128);
			PB[8 + 2 * q] = (uint32_t) src; PB[9 + 2 * q] = (uint32_t) (src >> 32);
		}
	}
	// ---- per-lane constants (VGPR vector pinned to v[0:15]) -----------------------
	u32x16 V0 = 0, V1 = 0;
	u32x4 V2 = 0;
	{
		const int lane = tid & 63;
		V0[0] = (uint32_t) lane * PBC_DMA_ROW;
		V0[1] = (uint32_t) lane * 16u;
		// record touch: one 128-byte line per lane, the lanes past rt_lines repeat the last one
		V0[2] = (uint32_t) (lane < rt_lines ? lane : rt_lines - 1) * 128u + (uint32_t) rt_ahead;
		V0[11] = (uint32_t) lane * 16u + 1024u;     // DMA lane offset + bytes from the split's first panel
		// finiteness prescan: this block's share of the 8192 doubles of a panel
		const int chunk = (8192 + nblocks - 1) / nblocks;
		const int nit = (chunk + 1023) / 1024;
		int e0 = bl * chunk;
		if (e0 > 8192 - nit * 1024) e0 = 8192 - nit * 1024;
		const int e = e0 + tid;
		V0[3] = (uint32_t) ((e >> 7) * PBC_DMA_ROW + (e & 127) * 8);
		V0[7] = lane == 0 ? (uint32_t) nit :
			lane == 1 ? (partial ? (uint32_t) (npanels - 1) : 0xFFFFFFFFu) :
				    (partial ? (uint32_t) ((128 - (nrow & 127)) * 8) : 0u);
	}
	// ---- first panel: straight into buffer 0 ---------------------------------------
	{
		const int lane = tid & 63;
		const bool clampme = partial && pa == npanels - 1;
#pragma unroll
		for (int q = 0; q < 4; q++) {
			int kk = k0 + wd * 4 + q;
			if (kk > K - 1) kk = K - 1;
			// a partial last panel is staged as rows nrow-128 .. nrow-1
			const int64_t r0 = clampme ? nrow - 128 : pa * 128;
			const double *src = Y + (int64_t) kk * ldY + r0 + lane * 2;
			double *dst = ylds + (wd * 4 + q) * 129;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) src,
							 (__attribute__((address_space(3))) void *) dst, 16, 0, 0);
		}
		// cold start of the record stream: the in-loop touch runs two tiles ahead, so the
		// first tiles' records would come from HBM one scalar load at a time (it shows with
		// many short workgroups: A %*% Y runs 3126 of them at BASELINE config 2b).  One
		// 128-byte line per lane, the first 4 KB of this wavefront's stream.
		{
			// (all lanes, no branch: lanes 32.. repeat line 31; the 4 KB stay inside the
			// record array's slack even for the last wavefront)
			uint32_t sink;
			const char *ptr = (const char *) rec + tb[0] * 12 + (int64_t) (lane < 32 ? lane : 31) * 128;
			// (load and wait in one statement: the register must not be handed to anything
			// else while the load is in flight)
			asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)"
				     : "=v"(sink) : "v"(ptr) : "memory");
		}
	}
#ifdef SVT_TUNING
	if (PROF) t_loop = __builtin_readcyclecounter();
#endif
#if PBC_DMA_YSETS == 2
	u32x8 V3 = 0;
#define PBC_DMA_STATE "+{v[0:15]}"(V0), "+{v[16:31]}"(V1), "+{v[32:39]}"(V3), "+{v[40:43]}"(V2), "+{s[8:11]}"(PA), "+{s[12:27]}"(PB)
#define PBC_ACC0 "+{v[44:75]}"
#define PBC_ACC1 "+{v[76:107]}"
#define PBC_ACC2 "+{v[108:123]}"
#else
#define PBC_DMA_STATE "+{v[0:15]}"(V0), "+{v[16:31]}"(V1), "+{v[32:35]}"(V2), "+{s[8:11]}"(PA), "+{s[12:27]}"(PB)
#define PBC_ACC0 "+{v[36:67]}"
#define PBC_ACC1 "+{v[68:99]}"
#define PBC_ACC2 "+{v[100:115]}"
#endif
#ifdef SVT_TUNING
	if constexpr (PROF == 1) {
		// tuning build (NV <= 2): cycles per section in v[116:123], see gen_pbc_asm.py
		u32x16 PV = 0;
		asm volatile(PBC_DMA_ASM_TEXT_PROF
			     : PBC_ACC0(acc[0]), PBC_ACC1(acc[NV > 1 ? 1 : 0]), PBC_DMA_STATE,
			       "+{v[112:127]}"(PV)
			     : : PBC_DMA_CLOBBERS, "s100", "s101");
		if (blockIdx.x == 0 && (tid & 63) == 0) {
			unsigned long long *o = (unsigned long long *) (fl.y_nonfinite + 1792) + w * 8;
			// [-, dma wait, barrier, issue, prescan, dispatch, stub, phases]
			for (int i = 0; i < 8; i++) o[i] = PV[4 + i];
		}
	} else
#endif
	if constexpr (NV == 1) {
		asm volatile(PBC_DMA_ASM_TEXT
			     : PBC_ACC0(acc[0]), PBC_DMA_STATE
			     : : PBC_DMA_CLOBBERS);
	} else if constexpr (NV == 2) {
		asm volatile(PBC_DMA_ASM_TEXT
			     : PBC_ACC0(acc[0]), PBC_ACC1(acc[NV > 1 ? 1 : 0]), PBC_DMA_STATE
			     : : PBC_DMA_CLOBBERS);
	} else {
		asm volatile(PBC_DMA_ASM_TEXT
			     : PBC_ACC0(acc[0]), PBC_ACC1(acc[NV > 1 ? 1 : 0]),
			       PBC_ACC2(acc8), PBC_DMA_STATE
			     : : PBC_DMA_CLOBBERS);
	}
#undef PBC_ACC0
#undef PBC_ACC1
#undef PBC_ACC2
#undef PBC_DMA_STATE
	if (PB[6] != 0 && (tid & 63) == 0) {
		*fl.y_nonfinite = 1;
		// which (row split, dense tile) block of Y it was in: the fix-up scans only those
		fl.y_nonfinite[PBC_SUBFLAG0 + (split * kt + kh) % PBC_NSUBFLAG] = 1;
	}
#ifdef SVT_TUNING
	if (PROF) t_epi = __builtin_readcyclecounter();
#endif
	// ---- partial results: part[(split*Kp + k) * ncol + c] ------------------------
	// A lane holds one dense column k of its wavefront's CBW sparse columns: stored
	// straight from registers that is 64 eight-byte writes ncol*8 bytes apart per
	// instruction.  The Y buffers are free now: 16 dense columns at a time go through
	// LDS as [k][16*CBW (+1)] and leave as whole rows of the workgroup's 16*CBW
	// columns (matters when the result is large: x %*% y writes nrow x K).  (All 64 dense columns of four
	// wavefronts at a time -- every lane storing to LDS, a quarter of the LDS store instructions, runs of
	// 4*CBW columns to memory -- was measured too: A %*% Y 2.48-2.52 against 2.38-2.46 ms on the same box.)
	const int lane2 = (int) __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	// (row stride even: the rows leave as 16-byte stores; 16 lanes writing one column of 16 rows still
	// hit 16 different bank groups: 2 * LS mod 64 = 4 for CBW = 40)
	const int CWG = 16 * CBW, LS = CWG + 2;
	const int64_t cwg0 = (int64_t) b * CWG;
	const int nvalid = (int) (ncol - cwg0 < CWG ? ncol - cwg0 : CWG);     // columns of this block inside the matrix
	__syncthreads();
#pragma unroll 1
	for (int q = 0; q < 4; q++) {
		if ((lane2 >> 4) == q) {
			double *row = ylds + (lane2 & 15) * LS + w * CBW;
#pragma unroll
			for (int ii = 0; ii < (NV > 2 ? 2 : NV); ii++)
#pragma unroll
				for (int jj = 0; jj < 16; jj++)
					if (ii * 16 + jj < CBW) row[ii * 16 + jj] = acc[ii][jj];
			if constexpr (NV > 2) {
#pragma unroll
				for (int jj = 0; jj < 8; jj++)
					if (32 + jj < CBW) row[32 + jj] = acc8[jj];
			}
		}
		__syncthreads();
		{       // wavefront w stores row w of the 16: one contiguous run, 16 bytes per lane where the row
			// is aligned (the epilogue took 35k cycles per workgroup with 8-byte stores and an integer
			// division per element: a tenth of a 79-panel workgroup of A %*% Y)
			// (part_ld / part_c0: the partials of a launch that covers only the column blocks from part_c0 on are
			// kept in a buffer of part_ld columns per row -- the row-split last round of A %*% Y)
			double *__restrict__ dstrow = part + ((int64_t) split * Kp + k0 + q * 16 + w) * part_ld + (cwg0 - part_c0);
			const double *__restrict__ src = ylds + w * LS;
			if ((((uintptr_t) dstrow) & 15) == 0) {
				// (non-temporal stores here were measured in round 5: the kernel that sums the partials reads them
				// back slower, 16.6 -> 22.5 us at an eighth of the rows of config 2a, the product kernel gains nothing)
				for (int cc = lane2 * 2; cc < nvalid; cc += 128) {
					if (cc + 1 < nvalid) *(double2 *) (dstrow + cc) = *(const double2 *) (src + cc);
					else dstrow[cc] = src[cc];
				}
			} else {
				for (int cc = lane2; cc < nvalid; cc += 64) dstrow[cc] = src[cc];
			}
		}
		__syncthreads();
	}
#ifdef SVT_TUNING
	if (PROF && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1 || blockIdx.x == gridDim.x / 2)) {
		const unsigned long long t_end = __builtin_readcyclecounter();
		printf("wg %d: prologue %llu, panel loop %llu, epilogue %llu cycles\n", (int) blockIdx.x,
		       t_loop - t_entry, t_epi - t_loop, t_end - t_epi);
	}
#endif
}

// (pbc_reduce_kernel / pbc_nafix_kernel: below, behind step 1 of the dirty-column fix-up that they host)

// ---------------------------------------------------------------------------
// Dense operand given by rows (tr_y: element (r, k) at Y[k + r * ldY], the tcrossprod() /
// transpose.x orientation of src/SparseMatrix_mult.c:411-421, which copies one row at a time
// into a column buffer): one tiled pass turns it into the column-major panel source of the
// product kernel.  1 GB in, 1 GB out at BASELINE config 2.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pbc_transpose_dense_kernel(const double *__restrict__ Yin, int64_t ldY, int64_t nrow, int K,
			   double *__restrict__ Yc)
{
	__shared__ double tile[64][65];
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	const int k0 = blockIdx.y * 64;
	for (int64_t r0 = (int64_t) blockIdx.x * 64; r0 < nrow; r0 += (int64_t) gridDim.x * 64) {
		if (r0 != (int64_t) blockIdx.x * 64) __syncthreads();
		for (int rr = ty; rr < 64; rr += 4) {        // lanes along k: contiguous in Yin
			const int64_t r = r0 + rr;
			const int k = k0 + tx;
			tile[rr][tx] = (r < nrow && k < K) ? Yin[k + r * ldY] : 0.0;
		}
		__syncthreads();
		for (int kk = ty; kk < 64; kk += 4) {        // lanes along r: contiguous in Yc
			const int64_t r = r0 + tx;
			const int k = k0 + kk;
			if (r < nrow && k < K) Yc[r + (int64_t) k * nrow] = tile[tx][kk];
		}
	}
}

// ---------------------------------------------------------------------------
// Dense columns that hold NaN / Inf / NA.  The reference switches to
// _dotprod_doubleSV_doubles (src/SparseVec_dotprod.c:48-65) for exactly those columns
// (src/SparseMatrix_mult.c:193-207): it multiplies the implicit zeros too, so per (leaf c, dirty
// column k):
//     the column or the leaf holds an R NA                 -> NA_real_
//     some non-finite y sits on a row where c has no entry -> NaN   (0 * Inf, 0 * NaN)
//     every non-finite y sits on a nonzero of c            -> the IEEE sum over the nonzeros, in
//                                                             ascending offset order
// The product kernel raises one flag when any staged entry is not finite.  Then (and only then:
// every kernel below returns at once while the flag is clear) one pass over Y counts the
// non-finite entries per column and lists their (row, column) positions; a leaf "hits" an entry
// if the row is among its offsets (binary search, offsets ascend inside a leaf); the fix-up
// rewrites the cells of the dirty columns by the rule above.  The clean columns are never
// touched.  A column with more non-finite entries than the longest leaf has nonzeros is NaN / NA in every
// cell without looking at its entries (a column of NAs).  More than PBC_DIRTY_CAP listed entries, more than
// PBC_DIRTY_COLS listed columns, or a column between PBC_DIRTY_LIGHT and the longest leaf: the leaf's wavefront
// walks its nonzeros once per such column, counts the non-finite y it meets and sums on the way (round 5; rounds
// 2-4 redid the WHOLE product with the general kernels in those cases, behind two gate launches per product).
// ---------------------------------------------------------------------------
#define PBC_DIRTY_CAP 8192
#define PBC_DIRTY_COLS 256    // light dense columns that get a slot (hit counters of a wavefront in LDS)
#define PBC_DIRTY_LIGHT 256     // a dense column lists at most this many of its non-finite entries
#define PBC_DIRTY_FIRST 4       // rows of a column's first non-finite entries kept apart from the list: a leaf that has no
                                // nonzero on one of them has its cell decided (NaN) without a walk
struct DirtyWs {
	int *flags;          // the flag block: [0] product kernel saw a non-finite y, [2] run the general
	                     // kernels, [3] number of listed non-finite entries
	int *col_nf;         // [Kp] non-finite entries per dense column
	int *has_na;         // [Kp] the column holds an R NA
	uint2 *list;         // [PBC_DIRTY_CAP] (row, column)
	int *first;          // [Kp][PBC_DIRTY_FIRST] rows of the first non-finite entries noted in every column (whatever the list holds)
};

static size_t dirty_ws_bytes(int64_t ncol, int64_t Kp)
{
	(void) ncol;
	return (size_t) Kp * 8 + (size_t) PBC_DIRTY_CAP * 8 + (size_t) Kp * PBC_DIRTY_FIRST * 4 + 256;
}

static DirtyWs dirty_ws_of(void *flag_block, void *tail, int64_t ncol, int64_t Kp)
{
	DirtyWs d;
	(void) ncol;
	d.flags = (int *) flag_block;
	// the counters must stay below the progress words of the paced gather kernel ([PBC_PROG_OFFSET, PBC_FLAG_BYTES): that kernel
	// runs between the clear of phase 1 and the readers of phase 2) and below the tuning builds' cycle counters (the KB before them)
	const bool in_block = 256 + Kp * 8 <= PBC_PROG_OFFSET - 1024;
	d.col_nf = in_block ? (int *) ((char *) flag_block + 256) : (int *) tail;   // then phase 1 clears them with the flags
	d.has_na = d.col_nf + Kp;
	d.list = (uint2 *) (((uintptr_t) (in_block ? (char *) tail : (char *) (d.has_na + Kp)) + 15) & ~(uintptr_t) 15);
	d.first = (int *) (d.list + PBC_DIRTY_CAP);
	return d;
}

// Two steps, no grid-wide barrier and no assumption about which workgroups are resident together (round 2
// ran all steps in one launch separated by grid barriers: beside an RCCL kernel that holds CUs those can
// starve -- VERDICT round 2).  Both steps cost nothing while the product kernel's flag is clear:
//   step 1 (scan) is the tail of the launch that sums the partial results (pbc_reduce_kernel /
//          pbc_nafix_kernel run after every product anyway): with the flag set their workgroups go on to
//          count the non-finite entries per dense column and list their (row, column) positions;
//   step 2 (pbc_dirty_leaf_kernel, one launch, returns at once while the flag is clear): one wavefront
//          per leaf finds which listed entries sit on its nonzeros (binary searches, hit counts stay in
//          the wavefront), rewrites its cells of the dirty columns, and where every non-finite entry of a
//          column sits on a nonzero of the leaf sums that cell again itself.
// The stream order between the two launches is the only synchronisation.

// step 1: non-finite entries per dense column and their (row, column) list.  Called by every workgroup of
// the hosting launch (block `wg` of `nwg`, any block size); d.flags[0] is the product kernel's flag.
__device__ inline void pbc_dirty_scan(const double *__restrict__ Y, int64_t rs, int64_t cs, int64_t nrow, int K,
				      const DirtyWs &d, int nsplit, int64_t pps, int kt, int64_t wg, int64_t nwg)
{
	auto note = [&](const double y, const int64_t r, const int k) {
		if (svt_is_finite(y))
			return;
		const int seen = atomicAdd(d.col_nf + k, 1);
		if (svt_is_na(y)) d.has_na[k] = 1;
		if (seen < PBC_DIRTY_FIRST) d.first[k * PBC_DIRTY_FIRST + seen] = (int) r;
		if (seen >= PBC_DIRTY_LIGHT)
			return;                                 // a heavy column: decided without its entries, or by the general kernels
		const int at = atomicAdd(d.flags + 3, 1);
		if (at < PBC_DIRTY_CAP) d.list[at] = make_uint2((unsigned) r, (unsigned) k);
	};
	// blocks of Y: (row split s, dense tile kh) as the product kernel staged them (pps = panels of 128 rows
	// per split; 0 = no such record, one block covers everything).  Consecutive threads take consecutive
	// row pairs of 8 dense columns (16-byte loads where the columns are aligned for them).
	const int nblk = pps > 0 ? nsplit * kt : 1;
	const bool pairs = rs == 1 && (cs & 1) == 0 && (((uintptr_t) Y) & 15) == 0;
	for (int blk = 0; blk < nblk; blk++) {
		int64_t b_lo = 0, b_hi = nrow;
		int kb = 0, ke = K;
		if (pps > 0) {
			if (d.flags[PBC_SUBFLAG0 + blk % PBC_NSUBFLAG] == 0)
				continue;
			const int sp = blk / kt, kh = blk % kt;
			b_lo = (int64_t) sp * pps * 128;                 // (even: row pairs stay aligned)
			b_hi = b_lo + pps * 128 < nrow ? b_lo + pps * 128 : nrow;
			kb = kh * 64; ke = kb + 64 < K ? kb + 64 : K;
		}
		if (b_hi <= b_lo || ke <= kb)
			continue;
		// The group of 8 dense columns is the same for a whole workgroup (wave-uniform counter addresses:
		// the compiler turns the 64 adds of a wavefront into one -- a column of NaNs is 1e6 adds on one
		// counter otherwise, 13 ms instead of 0.3); the workgroups that share a group stride over the rows.
		const int64_t ncg = (ke - kb + 7) / 8;
		const bool dealt = nwg >= ncg;
		const int64_t nsub = dealt ? nwg / ncg : nwg, sub = dealt ? wg / ncg : wg;
		if (sub >= nsub)
			continue;
		for (int64_t cg = dealt ? wg % ncg : 0; cg < ncg; cg += dealt ? ncg : 1) {
			const int k0 = kb + 8 * (int) cg;
			if (pairs) {
				const int64_t nrp = (b_hi - b_lo + 1) / 2;
				for (int64_t rp = sub * blockDim.x + threadIdx.x; rp < nrp; rp += nsub * blockDim.x) {
					const int64_t r = b_lo + 2 * rp;
					const bool pair = r + 1 < b_hi;
					double2 y[8];
#pragma unroll
					for (int u = 0; u < 8; u++) {
						const double *src = Y + r + (int64_t) (k0 + u) * cs;
						if (k0 + u >= ke) y[u] = make_double2(0.0, 0.0);
						else if (pair) y[u] = *(const double2 *) src;
						else y[u] = make_double2(*src, 0.0);
					}
#pragma unroll
					for (int u = 0; u < 8; u++) {
						note(y[u].x, r, k0 + u);
						note(y[u].y, r + 1, k0 + u);
					}
				}
				continue;
			}
			for (int64_t r = b_lo + sub * blockDim.x + threadIdx.x; r < b_hi; r += nsub * blockDim.x) {
				double y[8];
#pragma unroll
				for (int u = 0; u < 8; u++)
					y[u] = k0 + u < ke ? Y[r * rs + (int64_t) (k0 + u) * cs] : 0.0;      // element (r, k) at Y[r * rs + k * cs]
#pragma unroll
				for (int u = 0; u < 8; u++)
					note(y[u], r, k0 + u);
			}
		}
	}
}

// what the hosting launches need for step 1
struct DirtyScanArgs {
	DirtyWs d;
	const double *Y;         // element (r, k) at Y[r * rs + k * cs]; NULL: no scan in this launch
	int64_t rs, cs, nrow;
	int K, nsplit, kt;
	int64_t pps;
};

// out[c, k] = sum over splits (fixed order) ; NA_real_ for leaves holding an NA.
// Tail (both kernels): step 1 of the dirty-column fix-up when the product kernel raised its flag.
__device__ inline void pbc_dirty_scan_tail(const DirtyScanArgs &ds, int64_t wg, int64_t nwg)
{
	if (ds.Y == NULL || ds.d.flags[0] == 0)                 // (the same answer in every workgroup)
		return;
	pbc_dirty_scan(ds.Y, ds.rs, ds.cs, ds.nrow, ds.K, ds.d, ds.nsplit, ds.pps, ds.kt, wg, nwg);
}

__global__ void pbc_reduce_kernel(const double *__restrict__ part, int nsplit, int64_t Kp,
				  int K, int64_t ncol, const int *__restrict__ col_has_na,
				  double *__restrict__ out, int64_t sc, int64_t sk, int64_t c_begin,
				  const DirtyScanArgs ds, int pairs)
{
	const int k = blockIdx.y;
	if (pairs) {
		// two leaves per thread, 16-byte loads and stores (ncol, c_begin even, unit stride of the result, aligned bases:
		// checked by the launcher); the same additions in the same order per cell
		const int64_t c = c_begin + ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * 2;
		if (c < ncol && k < K) {
			double2 s = make_double2(0.0, 0.0);
			for (int t = 0; t < nsplit; t++) {
				const double2 v = *(const double2 *) (part + ((int64_t) t * Kp + k) * ncol + c);
				s.x += v.x; s.y += v.y;
			}
			const int2 na = *(const int2 *) (col_has_na + c);
			if (na.x) s.x = svt_na_real();
			if (na.y) s.y = svt_na_real();
			*(double2 *) (out + c + (int64_t) k * sk) = s;
		}
	} else {
		const int64_t c = c_begin + (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
		if (c < ncol && k < K) {
			double s = 0.0;
			for (int t = 0; t < nsplit; t++)
				s += part[((int64_t) t * Kp + k) * ncol + c];
			if (col_has_na[c]) s = svt_na_real();
			out[c * sc + (int64_t) k * sk] = s;
		}
	}
	pbc_dirty_scan_tail(ds, (int64_t) blockIdx.y * gridDim.x + blockIdx.x, (int64_t) gridDim.x * gridDim.y);
}

// Single row split: the product kernel wrote `out` itself; only the leaves
// holding an R NA remain to be patched.
__global__ void pbc_nafix_kernel(const int *__restrict__ col_has_na, int K, int64_t ncol,
				 double *__restrict__ out, int64_t sc, int64_t sk, int64_t c_begin,
				 const DirtyScanArgs ds)
{
	const int64_t c = c_begin + (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (c < ncol && col_has_na[c])
		for (int k = 0; k < K; k++) out[c * sc + (int64_t) k * sk] = svt_na_real();
	pbc_dirty_scan_tail(ds, blockIdx.x, gridDim.x);
}

// Class of every dirty dense column k (nf = its non-finite entries), recomputed by every workgroup from the scan's
// counters (K <= a few hundred entries): slot_lds[k] =
//   s >= 0   light (nf <= PBC_DIRTY_LIGHT, all of its entries listed): hit counter s of the wavefront;
//   -1       saturated (nf > the longest leaf: no leaf can have a nonzero on every non-finite row, every cell is NaN or
//            NA -- a column of NAs, the common case, costs no more than a single Inf);
//   -2       anything else (between the list and the longest leaf; more light columns than slots; more entries than
//            the list holds): the leaf's wavefront walks its nonzeros and looks at y itself.
// Returns the number of slots in use.  (Rounds 2-4 sent the -2 cases through the general kernels for the WHOLE product,
// 35 ms at config 2a, behind two more gate launches per product; round 5: one walk of the leaf per such column.)
__device__ inline int pbc_dirty_classes(int K, const DirtyWs &d, int *slot_lds, int64_t max_leaf_nnz)
{
	__shared__ int s_nslots;
	if (threadIdx.x == 0) {
		int n = 0;
		const bool listed_all = d.flags[3] <= PBC_DIRTY_CAP;
		for (int k = 0; k < K; k++) {
			const int nf = d.col_nf[k];
			int cls = -1;
			if (nf > 0) {
				if (nf <= PBC_DIRTY_LIGHT && listed_all && n < PBC_DIRTY_COLS) cls = n++;
				else if ((int64_t) nf > max_leaf_nnz) cls = -1;
				else cls = -2;
			}
			slot_lds[k] = cls;
		}
		s_nslots = n;
	}
	__syncthreads();
	return s_nslots;
}

// step 2: one wavefront per leaf c.
//   hits[slot] = listed entries of dirty column `slot` that sit on a nonzero of c (binary search: offsets
//   ascend inside a leaf); then per dirty column k the rule of the header comment; a cell whose column has
//   every non-finite entry on a nonzero of c is summed again here, over the leaf's nonzeros (lane-strided
//   partial sums, then a fixed-order butterfly: NaN / Inf class as in the sequential sum, finite parts
//   within rounding) -- the product kernel's own value cannot be kept: the zero records that pad its tiles
//   multiply row 0 of their panel, and 0 * Inf is NaN.
#define PBC_DIRTY_WPB 4
__global__ void __launch_bounds__(PBC_DIRTY_WPB * 64)
pbc_dirty_leaf_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
		      const double *__restrict__ val, const int *__restrict__ col_has_na,
		      const double *__restrict__ Y, int64_t rs, int64_t cs, int64_t ncol, int K,
		      DirtyWs d, double *__restrict__ out, int64_t sc, int64_t sk, int64_t max_leaf_nnz,
		      int *__restrict__ gen_counters, int n_gen_counters)
{
	extern __shared__ int slot_lds[];                       // [K] + [PBC_DIRTY_WPB][PBC_DIRTY_COLS] + [PBC_DIRTY_WPB][K]
	if (d.flags[0] == 0)                                    // (the same answer in every workgroup)
		return;
	(void) gen_counters; (void) n_gen_counters;
	const int nslots = pbc_dirty_classes(K, d, slot_lds, max_leaf_nnz);      // (the same answer in every workgroup)
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	int *hits = slot_lds + K + w * PBC_DIRTY_COLS;
	int *miss = slot_lds + K + PBC_DIRTY_WPB * PBC_DIRTY_COLS + w * K;
	int n = d.flags[3];
	if (n > PBC_DIRTY_CAP) n = 0;                            // (the list is incomplete: no column has a slot, all are walked)
	for (int64_t c = (int64_t) blockIdx.x * PBC_DIRTY_WPB + w; c < ncol; c += (int64_t) gridDim.x * PBC_DIRTY_WPB) {
		for (int i = lane; i < nslots; i += 64) hits[i] = 0;
		for (int i = lane; i < K; i += 64) miss[i] = 0;
		__builtin_amdgcn_wave_barrier();
		const int64_t beg = col_ptr[c], end = col_ptr[c + 1];
		// Columns of class -2 (the leaf would be walked for each of them): one of the column's first non-finite entries
		// on a row where the leaf holds nothing decides the cell -- 0 * Inf, 0 * NaN -- without the walk.  (Round 5
		// walked always: 1.7 ms per such column at config 2a, 215 ms for 128 of them where the general kernels of
		// rounds 2-4 took 35 for everything; a leaf of a 1 % operand misses the first entry 99 times in 100.)
		for (int t = lane; t < K * PBC_DIRTY_FIRST; t += 64) {
			const int k = t / PBC_DIRTY_FIRST, j = t % PBC_DIRTY_FIRST;
			if (slot_lds[k] != -2 || j >= d.col_nf[k])
				continue;
			const uint32_t r = (uint32_t) d.first[t];
			int64_t lo = beg, hi = end;
			while (lo < hi) {
				const int64_t mid = (lo + hi) >> 1;
				if ((uint32_t) row_idx[mid] < r) lo = mid + 1; else hi = mid;
			}
			if (!(lo < end && (uint32_t) row_idx[lo] == r))
				miss[k] = 1;                        // (LDS; several lanes may write the same 1)
		}
		for (int e = lane; e < n; e += 64) {
			const uint2 rk = d.list[e];
			const int sl = slot_lds[rk.y];
			if (sl < 0)
				continue;                           // (the first entries of a saturated column)
			int64_t lo = beg, hi = end;
			while (lo < hi) {
				const int64_t mid = (lo + hi) >> 1;
				if ((uint32_t) row_idx[mid] < rk.x) lo = mid + 1; else hi = mid;
			}
			if (lo < end && (uint32_t) row_idx[lo] == rk.x)
				atomicAdd(hits + sl, 1);            // (LDS)
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const bool leaf_na = col_has_na[c] != 0;
		for (int k = 0; k < K; k++) {                       // (wave-uniform control flow throughout)
			const int nf = d.col_nf[k];
			if (nf == 0)
				continue;
			double *cell = out + c * sc + (int64_t) k * sk;
			const int cls = slot_lds[k];
			if (d.has_na[k] || leaf_na) {
				if (lane == 0) *cell = svt_na_real();
			} else if (cls == -1 || (cls >= 0 && hits[cls] < nf) || (cls == -2 && miss[k] != 0)) {
				if (lane == 0) *cell = *cell + NAN;
			} else {
				// every non-finite entry of the column sits on a nonzero of the leaf (cls >= 0), or that is still to be
				// found out (cls == -2): one walk over the leaf's nonzeros gives the count of non-finite y met and the sum
				const double *__restrict__ y = Y + (int64_t) k * cs;
				double acc = 0.0;
				int met = 0;
				for (int64_t i = beg + lane; i < end; i += 64) {
					const double yy = y[(int64_t) row_idx[i] * rs];
					met += svt_is_finite(yy) ? 0 : 1;
					acc += val[i] * yy;
				}
				for (int off = 32; off > 0; off >>= 1) {
					acc += __shfl_xor(acc, off, 64);
					met += __shfl_xor(met, off, 64);
				}
				// (a non-finite entry on a row where the leaf holds nothing: 0 * Inf and 0 * NaN are NaN in the
				// reference's walk over all rows, src/SparseVec_dotprod.c:48-65)
				if (lane == 0) *cell = (cls == -2 && met < nf) ? acc + NAN : acc;
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
// Row splits: enough workgroups to fill the chip, every split non-empty.
static bool pbc_dma_ok(const svt_dev_pbc *P, int tr_y)
{
	// the layout was built for it; the kernel stages a row split through 32-bit byte offsets
	// (8 bytes per row per dense column), and a split is never longer than the matrix
	return !tr_y && P->fmt == 1 && P->nrow < ((int64_t) 1 << 28);
}

// rows of the staged dense operand the gather kernel may touch: its look-ahead loads run on into the
// records of the next tile, whose row offsets are relative to a panel that may be the last, partial one
static int64_t pbc_padded_rows(const svt_dev_pbc *P)
{
	// (one panel more: the XCD-paced kernel's look-ahead batch past its last tile may carry a tile-start flag)
	return P->gather ? ((P->npanels + 1) << P->logR) : P->nrow;
}

// Pacing of crossprod_pbc_gatherx_kernel: a wavefront runs at most `dsync` tiles ahead of the slowest started
// wavefront of its XCD and gives up after `spin` polls.  dsync < 0: the kernels with one launch per row
// chunk (crossprod_pbc_gather2_kernel) run instead.  (One rank's share of BASELINE config 4, 2048-row
// panels: dsync 0: 5.69 ms, 1: 4.77, 2: 4.73, 3: 4.80, no pacing at all: 7.9; with the L2 touch of the
// record stream 1: 4.31, 2: 4.43; spin 16: 6.7 -- a wavefront that gives up never paces itself again.)
static int g_pbgx_dsync = 1, g_pbgx_spin = 256;
static int g_pbc_rounds = 1;        // 0: one launch whatever the number of column blocks; 1: one launch per round of workgroups, the
                                    // last (partly filled) round cut by rows; 2: per round, last round whole (svt_dev_pbc_set_round_launches)
extern "C" void svt_dev_pbc_set_round_launches(int on) { g_pbc_rounds = on; }
extern "C" void svt_dev_pbc_set_gather_pacing(int dsync, int spin)
{
	g_pbgx_dsync = dsync;
	g_pbgx_spin = spin < 1 ? 1 : spin;
}

static int pbc_cus(void)
{
	static int cus = 0;
	if (cus == 0) {
		int dev = 0, n = 0;
		if (hipGetDevice(&dev) == hipSuccess &&
		    hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8)
			cus = n;
		else
			cus = 256;
	}
	return cus;
}

// the XCD-paced kernel takes whole pairs of 64-wide dense tiles and wants a few panels per XCD
static bool pbgx_ok(const svt_dev_pbc *P, int K)
{
	const int64_t Kp = ((int64_t) K + 63) / 64 * 64;
	return P->gather && g_pbgx_dsync >= 0 && Kp % 128 == 0 && P->npanels >= 64 && P->WPB == 4 &&
	       ((int64_t) 8 << P->logR) * Kp < ((int64_t) 1 << 31);
}

// gather kernel: no staging to share, so splits only have to fill the chip (~8192 wavefronts)
static int pick_nsplit_gather(const svt_dev_pbc *P, int K, int64_t *pps_out, bool unpaced = false)
{
	if (!unpaced && pbgx_ok(P, K)) {                                 // one partial result per XCD with rows
		const int64_t npx = (P->npanels + 7) / 8;
		if (pps_out) *pps_out = npx;
		return (int) ((P->npanels + npx - 1) / npx);
	}
	const int64_t kt = ((int64_t) K + 63) / 64;
	const bool wide = kt % 2 == 0;                       // crossprod_pbc_gather2_kernel: one wavefront per 128 dense columns
	const int64_t waves = P->ngroups * (wide ? kt / 2 : kt);
	// wavefronts per launch: ~4096 for both kernels (the 128-column one keeps 8 per CU resident, 8 KiB in
	// flight each: two rounds; config 4: 4 row splits 49.2 ms, 2 splits 59, 7 and more 83)
	int64_t target = 4096;
#ifdef SVT_TUNING
	if (getenv("SVT_PBG_WAVES")) target = atoll(getenv("SVT_PBG_WAVES"));
#endif
	int64_t s = waves >= target ? 1 : (target + waves - 1) / waves;
	if (s > P->npanels) s = P->npanels;
	if (s < 1) s = 1;
	const int64_t pps = (P->npanels + s - 1) / s;
	s = (P->npanels + pps - 1) / pps;
	if (pps_out) *pps_out = pps;
	return (int) s;
}

// CUs are kept free (svt_dev_pbc_set_spare_cus) when one round of workgroups fits the rest
static bool pbc_sparing(const svt_dev_pbc *P, int K)
{
	const int64_t units = P->nblocks * (((int64_t) K + 63) / 64);
	return g_pbc_spare_cus > 0 && units <= 256 - g_pbc_spare_cus && P->npanels >= 8 * 16;
}

static int pick_nsplit(const svt_dev_pbc *P, int K, bool dma, int64_t *pps_out, bool may_spare = true)
{
	const int64_t kt = ((int64_t) K + 63) / 64;
	const int64_t units = P->nblocks * kt;
	int64_t s;
	if (dma && may_spare && pbc_sparing(P, K)) {
		s = (256 - g_pbc_spare_cus) / units;    // one round on the CUs that may be used
		if (P->npanels / s < 16) s = P->npanels / 16;
	} else if (units >= 512) {
		s = 1;                                  // enough column blocks: no row split, no partials
	} else if (dma && P->npanels >= 8 * 16) {
		// One workgroup per CU (LDS, VGPRs), 32 CUs per XCD, the column blocks of a (split, dense
		// tile) pair share an XCD.  Candidates: sx splits per XCD (8 sx in all), plus, where CUs are
		// left over in the last round, further splits whose workgroups are dealt over all XCDs (28
		// workgroups per split: 8 x 28 = 224 leave 32 CUs for a ninth split; its panels come through
		// eight L2s instead of one: 1.757 -> 1.647 ms at 1e6 x 8960, tools/debug/ninth_split.py).
		// Cost of a candidate: rounds x panels per workgroup x ~1.8 us, plus the partial results (one
		// ncol x Kp block per split written and read again at ~4 TB/s) -- without that term 64 splits
		// in 7 full rounds beat 8 splits in one round that leaves an eighth of the CUs idle (2.2 ms).
		const double t_panel = 1.8e-6, t_split = 2.0 * (double) P->ncol * (double) (kt * 64) * 8.0 / 4e12;
		double best_t = 1e30;
		s = 8;
		for (int64_t sx = 1; sx <= 16 && P->npanels / (8 * sx) >= 16; sx++) {
			const int64_t u = units * sx, rounds = (u + 31) / 32;
			int64_t extra = (rounds * 32 - u) * 8 / units;
			if (extra > 7) extra = 7;
			for (int64_t ex = 0; ex <= extra; ex += (extra > 0 ? extra : 1)) {
				const int64_t cand = 8 * sx + ex;
				if (P->npanels / cand < 16) continue;
				const double t = (double) rounds * (double) ((P->npanels + cand - 1) / cand) * t_panel +
					(double) cand * t_split;
				if (t < best_t - 1e-9) { best_t = t; s = cand; }
			}
		}
	} else {
		s = (512 + units - 1) / units;          // aim for >= 512 workgroups
		s = (s + 7) / 8 * 8;                    // whole XCD rounds
	}
	if (g_pbc_nsplit > 0 && units < 512) s = g_pbc_nsplit;   // tuning override
	if (s > P->npanels) s = P->npanels;
	if (s < 1) s = 1;
	const int64_t pps = (P->npanels + s - 1) / s;
	s = (P->npanels + pps - 1) / pps;              // drop empty splits
	if (pps_out) *pps_out = pps;
	return (int) s;
}

extern "C" size_t svt_dev_crossprod_pbc_ws_bytes(const svt_dev_pbc *P, int K)
{
	const int64_t Kp = ((int64_t) K + 63) / 64 * 64;
	int ns = pick_nsplit(P, K, false, NULL);
	{                                               // (the knob may change between the query and the launch)
		const int n0 = pick_nsplit(P, K, true, NULL, false);
		if (pbc_dma_ok(P, 0) && n0 > ns) ns = n0;
	}
	if (pbc_dma_ok(P, 0)) {
		const int nd = pick_nsplit(P, K, true, NULL);
		if (nd > ns) ns = nd;
	}
	if (P->gather) {
		const int ng = pick_nsplit_gather(P, K, NULL, true);
		if (ng > ns) ns = ng;
		if (ns < 8) ns = 8;                         // (the pacing knob may change between the query and the launch)
	}
	// [flags][partials][general-path workspace; a row-major Y is transposed into it first]
	// [dirty-column scratch]
	return PBC_FLAG_BYTES + (size_t) ns * Kp * (P->ncol > 0 ? P->ncol : 1) * 8 +
	       crossprod_ws_bytes(pbc_padded_rows(P), P->ncol, K) + dirty_ws_bytes(P->ncol, Kp);
}

template <int NV, int WPB, int LOGR>
static void launch_main(const svt_dev_pbc *P, const double *Y, int64_t ldY, int tr_y, int K,
			int nsplit, int64_t pps, double *part, int64_t Kp, PbcFlags fl,
			hipStream_t s)
{
	const int R = 1 << LOGR;
	const size_t lds = (size_t) 64 * (R + 1) * 8;
	dim3 grid((unsigned) nsplit, (unsigned) (Kp / 64), (unsigned) P->nblocks);
	void (*kern)(const uint4 *, const int64_t *, int64_t, const double *,
		     int64_t, int64_t, int, int64_t, int64_t, double *, int64_t, PbcFlags, int);
	if (tr_y)
		kern = crossprod_pbc_kernel<NV, WPB, LOGR, true, 0>;
	else
#ifdef SVT_TUNING
		kern = g_pbc_debug == 2 ? crossprod_pbc_kernel<NV, WPB, LOGR, false, 2> :
		       g_pbc_debug == 3 ? crossprod_pbc_kernel<NV, WPB, LOGR, false, 3> :
					  crossprod_pbc_kernel<NV, WPB, LOGR, false, 0>;
#else
		kern = crossprod_pbc_kernel<NV, WPB, LOGR, false, 0>;
#endif
	(void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	hipLaunchKernelGGL(kern, grid, dim3(WPB * 64), lds, s, P->rec, P->tile_ptr, P->npanels,
			   Y, ldY, P->nrow, K, P->ncol, pps, part, Kp, fl, P->CBW);
}

template <int NV>
static void launch_dma(const svt_dev_pbc *P, const double *Y, int64_t ldY, int K, int nsplit,
		       int64_t pps, double *part, int64_t Kp, PbcFlags fl, hipStream_t s, int block0, int nb_launch = 0,
		       int64_t part_ld = -1, int64_t part_c0 = 0)
{
	if (part_ld < 0) part_ld = P->ncol;
	const size_t lds = (size_t) 2 * PBC_DMA_BUF;
	const int kt = (int) (Kp / 64);
	// L2 touches of the record stream: ~1.5 tiles' worth of 128-byte lines
	const double tile_bytes = P->ngroups * P->npanels > 0 ?
		(double) P->nrec * 12.0 / (double) (P->ngroups * P->npanels) : 0.0;
	int rt_lines = (int) (tile_bytes * 1.5 / 128.0) + 2;
	if (rt_lines > 31) rt_lines = 31;
	// the touch starts ~2 tiles past the scalar-load cursor (the PBC_SLACK records
	// past the end of the stream absorb it at the end: 320 * 16 = 5120 bytes)
	int rt_ahead = ((int) (g_pbc_ahead10 * 0.1 * tile_bytes) + 127) / 128 * 128;
	if (rt_ahead + rt_lines * 128 > 4608) rt_ahead = 4608 - rt_lines * 128;
	if (rt_ahead < 0) rt_ahead = 0;
#ifdef SVT_TUNING
	auto kern = g_pbc_debug != 3 ? crossprod_pbc_dma_kernel<NV, 0> :
		    NV <= 2 ? crossprod_pbc_dma_kernel<(NV <= 2 ? NV : 2), 1> : crossprod_pbc_dma_kernel<NV, 2>;
#else
	auto kern = crossprod_pbc_dma_kernel<NV, 0>;
#endif
	(void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	int nb = (int) P->nblocks - block0;
	if (nb_launch > 0 && nb_launch < nb) nb = nb_launch;
	// with CUs kept free: 8 x per_xcd workgroups, packed XCD by XCD (see the kernel's decode)
	const bool sparing = pbc_sparing(P, K) && part_c0 == 0 && part_ld == P->ncol;
	const int per_xcd = (256 - g_pbc_spare_cus) / 8;
	const int nfull = sparing ? -per_xcd : nsplit & ~7;
	const int64_t nwg = sparing ? (int64_t) 8 * per_xcd : (int64_t) nsplit * kt * nb;
	hipLaunchKernelGGL(kern, dim3((unsigned) nwg), dim3(1024), lds, s,
			   P->rec, P->tile_ptr, P->npanels, Y, ldY, P->nrow, K, P->ncol, P->CBW, nfull,
			   nb, block0, pps, part, Kp, fl, rt_lines, rt_ahead, g_pbc_stagger, part_ld, part_c0);
}

// out[k][c0 + c] = sum over the row splits of tail[(t * Kp + k) * ld + c], in split order (the row-split last round of a
// product without row splits: launch code below); leaves holding an R NA are patched by pbc_nafix_kernel afterwards
__global__ void __launch_bounds__(256)
pbc_tail_reduce_kernel(const double *__restrict__ tail, int nsplit, int64_t Kp, int K, int64_t ld, int64_t ncols,
		       double *__restrict__ out, int64_t out_ld, int64_t c0)
{
	const int64_t c = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * 2;
	const int k = blockIdx.y;
	if (c >= ncols || k >= K)
		return;
	// (ld and c0 are multiples of 16 * CBW = even, rows of `out` start 16-byte aligned when out_ld is even)
	if (c + 1 < ncols && ((out_ld | c0) & 1) == 0 && (((uintptr_t) out) & 15) == 0 && (((uintptr_t) tail) & 15) == 0) {
		double2 a = make_double2(0.0, 0.0);
		for (int t = 0; t < nsplit; t++) {
			const double2 v = *(const double2 *) (tail + ((int64_t) t * Kp + k) * ld + c);
			a.x += v.x; a.y += v.y;
		}
		*(double2 *) (out + (int64_t) k * out_ld + c0 + c) = a;
		return;
	}
	for (int64_t cc = c; cc < c + 2 && cc < ncols; cc++) {
		double a = 0.0;
		for (int t = 0; t < nsplit; t++) a += tail[((int64_t) t * Kp + k) * ld + cc];
		out[(int64_t) k * out_ld + c0 + cc] = a;
	}
}

// three word ranges to zero (a NULL range is skipped)
__global__ void __launch_bounds__(256)
pbc_clear_kernel(uint32_t *__restrict__ a, int na, uint32_t *__restrict__ b, int nb, uint32_t *__restrict__ c, int nc)
{
	for (int i = threadIdx.x; i < na; i += blockDim.x) a[i] = 0u;
	if (b != NULL)
		for (int i = threadIdx.x; i < nb; i += blockDim.x) b[i] = 0u;
	if (c != NULL)
		for (int i = threadIdx.x; i < nc; i += blockDim.x) c[i] = 0u;
}

int launch_crossprod_general_if(const CrossprodArgs &a, const int *flag, bool counters_cleared, hipStream_t s);
int *crossprod_general_counters(void *ws, int64_t nrow, int K, int *n);
int launch_dense_prepare_flag(const CrossprodArgs &a, int *any, hipStream_t s);

// phase 1: the LDS-panel product kernel (partials into ws); phase 2: sum the
// partials into `out`, then the general kernels if Y was not finite.
// first_col > 0: only the leaves of the workgroup column block that holds first_col and of the
// later ones are computed (their cells of `out` written); the cells of earlier leaves are left
// alone.  Unary crossprod(x) needs only the leaves c >= k of dense column k (the reference's
// compute_sym_dotprods_*, src/SparseMatrix_mult.c:263-296, computes ncol^2 / 2 dot products).
static int pbc_phase_impl(const svt_dev_pbc *P, const svt_dev_csc *A,
			  const double *Y, int64_t ldY, int K, int tr_y,
			  double *out, int64_t out_stride_c,
			  int64_t out_stride_k, void *ws, size_t ws_bytes,
			  void *stream, int phase, int64_t first_col);
static int pbc_phase(const svt_dev_pbc *P, const svt_dev_csc *A,
		     const double *Y, int64_t ldY, int K, int tr_y,
		     double *out, int64_t out_stride_c,
		     int64_t out_stride_k, void *ws, size_t ws_bytes,
		     void *stream, int phase, int64_t first_col)
{
	const int rc = pbc_phase_impl(P, A, Y, ldY, K, tr_y, out, out_stride_c, out_stride_k, ws, ws_bytes, stream,
				      phase, first_col);
	pbc_note_use(P, (hipStream_t) stream);
	return rc;
}
static int pbc_phase_impl(const svt_dev_pbc *P, const svt_dev_csc *A,
			  const double *Y, int64_t ldY, int K, int tr_y,
			  double *out, int64_t out_stride_c,
			  int64_t out_stride_k, void *ws, size_t ws_bytes,
			  void *stream, int phase, int64_t first_col)
{
	hipStream_t s = (hipStream_t) stream;
	if (P->ncol <= 0 || K <= 0)
		return 0;
	if (ws_bytes < svt_dev_crossprod_pbc_ws_bytes(P, K))
		return svt_set_error("svt_dev_crossprod_pbc: workspace too small");
	const int64_t Kp = ((int64_t) K + 63) / 64 * 64;
	int64_t pps = 1;
	// a dense operand given by rows is transposed on the device first (phase 1) and the product
	// runs on the column-major copy: same kernel, same speed + one 2 x |Y| pass
	const bool dma = pbc_dma_ok(P, 0);
	const bool gath = P->gather != 0;
	const bool via_copy = tr_y && dma;
	int block0 = 0;
	if (dma && first_col > 0) {
		block0 = (int) (first_col / ((int64_t) 16 * P->CBW));
		if (block0 > P->nblocks - 1) block0 = (int) P->nblocks - 1;
	}
	const int64_t c_begin = (int64_t) block0 * 16 * P->CBW;
	const int nsplit = gath ? pick_nsplit_gather(P, K, &pps) : pick_nsplit(P, K, dma, &pps);
	PbcFlags fl;
	fl.y_nonfinite = (int *) ws;
	double *part = (double *) ((char *) ws + PBC_FLAG_BYTES);
	void *gen_ws = (char *) part + (size_t) nsplit * Kp * P->ncol * 8;
	const size_t gen_bytes = crossprod_ws_bytes(pbc_padded_rows(P), P->ncol, K);
	DirtyWs dw = dirty_ws_of(ws, (char *) gen_ws + gen_bytes, P->ncol, Kp);
	const double *Yc = via_copy ? (const double *) gen_ws : Y;     // what the product kernel stages
	const int64_t ldc = via_copy ? P->nrow : ldY;
	// one split, whole 64-wide dense tiles, result laid out like the partials:
	// the product kernel writes `out` directly
	const bool direct = nsplit == 1 && Kp == K && out_stride_c == 1 && out_stride_k == P->ncol;
	if (direct) part = out;
	if (phase == 1) {
		// flags, the per-column counters and (paced gather kernel) the progress words: ONE small kernel.  A
		// hipMemsetAsync() between two kernels costs a step 5 us of blit kernel plus a 5-6 us bubble in front of it
		// (profiles/r05_share_trace.txt: the runtime orders its blit kernels behind a barrier packet) -- 4 % of the
		// step at an eighth of the rows of config 2a
		{
			const bool paced = gath && P->rec != NULL && pbgx_ok(P, K);
			hipLaunchKernelGGL(pbc_clear_kernel, dim3(1), dim3(256), 0, s,
					   (uint32_t *) ws, 64, (uint32_t *) dw.col_nf, (int) (Kp * 2),
					   paced ? (uint32_t *) ((char *) ws + PBC_PROG_OFFSET) : (uint32_t *) NULL,
					   8 * PBGX_PROG_ENTRIES);
		}
		if (P->rec == NULL) {
			// no nonzero at all (no record stream was built): the general kernels
			// produce the zeros -- or the NaNs, if Y is not finite
			HIP_TRY(hipMemsetAsync(ws, 1, 12, s));        // flags [0] and [2]
			return 0;
		}
		const int nv = (P->CBW + 15) / 16;
		if (gath) {
			// row-major, K-padded copy of the dense operand (either orientation) + "not finite" flag,
			// then every nonzero gathers its row from it
			CrossprodArgs pa;
			memset(&pa, 0, sizeof(pa));
			pa.Rtype = SVT_REALSXP; pa.nrow = P->nrow; pa.ncol = P->ncol; pa.Y = Y; pa.ldY = ldY; pa.K = K;
			pa.tr_y = tr_y; pa.ws = gen_ws; pa.ws_bytes = gen_bytes;
			if (launch_dense_prepare_flag(pa, fl.y_nonfinite, s))
				return -1;
			if (pbgx_ok(P, K)) {
				// persistent grid, one row range per XCD, paced (see crossprod_pbc_gatherx_kernel)
				unsigned int *prog = (unsigned int *) ((char *) ws + PBC_PROG_OFFSET);     // (cleared with the flags)
				int nslots = (pbc_cus() - g_pbc_spare_cus) / 8 * 2;
				if (nslots > PBGX_PROG_ENTRIES / 4) nslots = PBGX_PROG_ENTRIES / 4;
				if (nslots < 2) nslots = 2;
				const int nkt = (int) (Kp / 128);
				const int64_t nunits = P->nblocks * nkt;
				auto kx = nv == 1 ? crossprod_pbc_gatherx_kernel<1> : nv == 2 ? crossprod_pbc_gatherx_kernel<2>
										   : crossprod_pbc_gatherx_kernel<3>;
				hipLaunchKernelGGL(kx, dim3((unsigned) (8 * nslots)), dim3(256), 0, s, P->rec, P->tile_ptr,
						   P->npanels, P->ngroups, (const double *) gen_ws, Kp, P->ncol, part, Kp, P->CBW,
						   P->logR, pps, nkt, nunits, prog, g_pbgx_dsync, g_pbgx_spin);
				HIP_TRY(hipGetLastError());
				return 0;
			}
			// whole pairs of 64-wide dense tiles: the kernel with two dense columns per lane
			const bool wide = Kp % 128 == 0;
			dim3 grid((unsigned) nsplit, (unsigned) (wide ? Kp / 128 : Kp / 64), (unsigned) P->nblocks);
			auto kern = wide ? (nv == 1 ? crossprod_pbc_gather2_kernel<1> : nv == 2 ? crossprod_pbc_gather2_kernel<2>
										     : crossprod_pbc_gather2_kernel<3>)
					 : (nv == 1 ? crossprod_pbc_gather_kernel<1> : nv == 2 ? crossprod_pbc_gather_kernel<2>
										     : crossprod_pbc_gather_kernel<3>);
			// Row chunks as consecutive launches (the partial sums go through memory in between):
			// inside one launch the wavefronts of all column groups walk the same rows at about the
			// same pace, so a row of Yt fetched for one group is still in L2 for the others; over a
			// whole operand they drift apart by more rows than the L2 holds and every record's 512
			// bytes come from the Infinity Cache or HBM (81.8 ms at BASELINE config 4, i.e. HBM speed).
			int64_t cpanels = wide ? PBG2_CHUNK_PANELS : PBG_CHUNK_PANELS;
#ifdef SVT_TUNING
			if (getenv("SVT_PBG_CHUNK")) cpanels = atoll(getenv("SVT_PBG_CHUNK"));
#endif
			if (P->logR < 10) cpanels <<= 10 - P->logR;          // (the chunk sizes were measured with 1024-row panels)
			const int64_t chunk = (int64_t) nsplit * cpanels;
			for (int64_t p0 = 0; p0 < P->npanels; p0 += chunk) {
				const int64_t p1 = p0 + chunk < P->npanels ? p0 + chunk : P->npanels;
				const int64_t cpps = (p1 - p0 + nsplit - 1) / nsplit;
				hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, P->rec, P->tile_ptr, P->npanels,
						   (const double *) gen_ws, Kp, K, P->ncol, cpps, part, Kp, P->CBW, P->logR,
						   p0, p1, p0 > 0 ? 1 : 0);
			}
			HIP_TRY(hipGetLastError());
			return 0;
		}
		if (dma) {
			if (via_copy && P->nrow > 0) {
				const int64_t ntile = (P->nrow + 63) / 64;
				dim3 tg((unsigned) (ntile < 8192 ? ntile : 8192), (unsigned) (Kp / 64));
				hipLaunchKernelGGL(pbc_transpose_dense_kernel, tg, dim3(256), 0, s, Y, ldY, P->nrow, K,
						   (double *) gen_ws);
			}
			// Many column blocks, no row split (A %*% Y on the layout of t(A): 1563 blocks x 2 dense tiles of
			// 79 panels each): one launch per round of workgroups (a workgroup per CU).  In ONE launch the
			// workgroups of later rounds start whenever a CU frees up, spread over all panel positions, and
			// the 5 MB dense tile they all stage no longer fits the XCD's 4 MiB L2 (4030 cycles per panel in
			// the first round, 4650 later); launch by launch every round starts aligned.
			const int kt_ = (int) (Kp / 64);
			int per_launch = 0;
			if (nsplit == 1 && !pbc_sparing(P, K) && g_pbc_rounds != 0 &&
			    (int64_t) (P->nblocks - block0) * kt_ >= 2 * pbc_cus())
				per_launch = pbc_cus() / kt_ > 0 ? pbc_cus() / kt_ : 1;
			// The last round: `rem` column blocks (54 of 256 CUs busy at config 2b) are cut by rows so that they fill
			// the chip -- ts row splits of a quarter of the panels each instead of one more full round; their partial
			// sums go to the (unused: the product writes `out` directly) partials area of the workspace and are summed in
			// split order behind them.  12.2 rounds then cost 12 + 1 / ts instead of 13.
			int tail_blocks = 0, ts = 1;
			int64_t tpps = pps;
			if (per_launch > 0 && direct && g_pbc_rounds == 1) {
				const int rem = (int) ((P->nblocks - block0) % per_launch);
				if (rem > 0 && rem * kt_ * 2 <= pbc_cus()) {
					ts = pbc_cus() / (rem * kt_);
					if (ts > 8) ts = 8;
					while (ts > 1 && P->npanels / ts < 8) ts--;
					tpps = (P->npanels + ts - 1) / ts;
					ts = (int) ((P->npanels + tpps - 1) / tpps);
					const int64_t tcols = (int64_t) rem * 16 * P->CBW;
					if (ts > 1 && (int64_t) ts * tcols <= P->ncol)      // (fits the partials area: Kp * ncol doubles)
						tail_blocks = rem;
				}
			}
			const int last_full = (int) P->nblocks - tail_blocks;
			for (int b0 = block0; b0 < last_full; b0 += per_launch > 0 ? per_launch : (int) P->nblocks) {
				const int nbl = per_launch > 0 ? (b0 + per_launch <= last_full ? per_launch : last_full - b0)
							       : tail_blocks > 0 ? last_full - b0 : 0;
				if (nv == 1) launch_dma<1>(P, Yc, ldc, K, nsplit, pps, part, Kp, fl, s, b0, nbl);
				else if (nv == 2) launch_dma<2>(P, Yc, ldc, K, nsplit, pps, part, Kp, fl, s, b0, nbl);
				else launch_dma<3>(P, Yc, ldc, K, nsplit, pps, part, Kp, fl, s, b0, nbl);
			}
			if (tail_blocks > 0) {
				double *tail = (double *) ((char *) ws + PBC_FLAG_BYTES);
				const int64_t c0 = (int64_t) last_full * 16 * P->CBW;
				const int64_t tld = (int64_t) tail_blocks * 16 * P->CBW;
				const int64_t tcols = P->ncol - c0;                  // (the last block may be short)
				if (nv == 1) launch_dma<1>(P, Yc, ldc, K, ts, tpps, tail, Kp, fl, s, last_full, tail_blocks, tld, c0);
				else if (nv == 2) launch_dma<2>(P, Yc, ldc, K, ts, tpps, tail, Kp, fl, s, last_full, tail_blocks, tld, c0);
				else launch_dma<3>(P, Yc, ldc, K, ts, tpps, tail, Kp, fl, s, last_full, tail_blocks, tld, c0);
				dim3 tg((unsigned) ((tcols + 511) / 512), (unsigned) K);
				hipLaunchKernelGGL(pbc_tail_reduce_kernel, tg, dim3(256), 0, s, tail, ts, Kp, K, tld, tcols,
						   out, P->ncol, c0);
			}
			HIP_TRY(hipGetLastError());
			return 0;
		}
		// format-1 layouts have no register-staged kernel (row-major Y): general path
		const int key = P->fmt == 1 ? -1 : nv * 10000 + P->WPB * 100 + P->logR;
#define PBC_CASE(NV, WPB, LOGR) \
		case (NV) * 10000 + (WPB) * 100 + (LOGR): \
			launch_main<NV, WPB, LOGR>(P, Y, ldY, tr_y, K, nsplit, pps, part, Kp, fl, s); break;
		switch (key) {
		PBC_CASE(1, 16, 7) PBC_CASE(2, 16, 7)
		PBC_CASE(1, 16, 8) PBC_CASE(2, 16, 8)
		PBC_CASE(2, 8, 7) PBC_CASE(3, 8, 7) PBC_CASE(4, 8, 7)
		PBC_CASE(2, 8, 6) PBC_CASE(4, 4, 5)
		default:
			// no panel kernel for this layout: raise the flags that send the whole
			// product through the general kernels in phase 2
			HIP_TRY(hipMemsetAsync(ws, 1, 12, s));
		}
#undef PBC_CASE
		HIP_TRY(hipGetLastError());
		return 0;
	}
	// Dense columns with NaN / Inf / NA (step 1 rides on the launch below, step 2 is one launch that
	// returns at once while the product kernel's flag is clear).  The register-staged kernels read the
	// dense operand as it was given: their dirty columns always take the general kernels.
	const bool fast = dma || gath;
	// where the fix-up reads the dense operand: element (r, k) at Yd[r * yrs + k * ycs]
	const double *Yd = gath ? (const double *) gen_ws : Yc;
	const int64_t yrs = gath ? Kp : 1, ycs = gath ? 1 : ldc;
	DirtyScanArgs ds;
	memset(&ds, 0, sizeof(ds));
	ds.d = dw;
	if (fast && P->rec != NULL) {
		ds.Y = Yd; ds.rs = yrs; ds.cs = ycs; ds.nrow = P->nrow; ds.K = K;
		ds.nsplit = nsplit; ds.kt = (int) (Kp / 64); ds.pps = dma ? pps : 0;
	}
	if (direct) {
		hipLaunchKernelGGL(pbc_nafix_kernel, dim3((unsigned) ((P->ncol - c_begin + 255) / 256)), dim3(256), 0, s,
				   P->col_has_na, K, P->ncol, out, out_stride_c, out_stride_k, c_begin, ds);
	} else {
		const int pairs = (P->ncol % 2 == 0 && c_begin % 2 == 0 && out_stride_c == 1 && out_stride_k % 2 == 0 &&
				   (((uintptr_t) part | (uintptr_t) out) & 15) == 0 && ((uintptr_t) P->col_has_na & 7) == 0) ? 1 : 0;
		const int64_t per_wg = pairs ? 512 : 256;
		dim3 rgrid((unsigned) ((P->ncol - c_begin + per_wg - 1) / per_wg), (unsigned) K);
		hipLaunchKernelGGL(pbc_reduce_kernel, rgrid, dim3(256), 0, s, part, nsplit, Kp, K, P->ncol,
				   P->col_has_na, out, out_stride_c, out_stride_k, c_begin, ds, pairs);
	}
	HIP_TRY(hipGetLastError());
	if (fast && P->rec != NULL) {
		int n_gen_counters = 0;
		int *gen_counters = NULL;
		// one wavefront per leaf, at most ~8 wavefronts per CU's worth of workgroups in flight at a time
		int64_t nwg = (P->ncol + PBC_DIRTY_WPB - 1) / PBC_DIRTY_WPB;
		if (nwg > 4096) nwg = 4096;
		hipLaunchKernelGGL(pbc_dirty_leaf_kernel, dim3((unsigned) nwg), dim3(PBC_DIRTY_WPB * 64),
				   (size_t) (K + PBC_DIRTY_WPB * PBC_DIRTY_COLS + PBC_DIRTY_WPB * K) * 4, s,
				   A->col_ptr, A->row_idx, (const double *) A->val, P->col_has_na, Yd, yrs, ycs,
				   P->ncol, K, dw, out, out_stride_c, out_stride_k, P->max_leaf_nnz,
				   gen_counters, n_gen_counters);
		HIP_TRY(hipGetLastError());
	}
	if (fast && P->rec != NULL)
		return 0;                               // (round 5: the leaf kernel above handles every class of dirty column itself)
	// General (slow-path) semantics for the layouts without the fix-up (register-staged kernels, no record stream).
	CrossprodArgs a;
	memset(&a, 0, sizeof(a));
	a.col_ptr = A->col_ptr; a.row_idx = A->row_idx; a.val = A->val; a.Rtype = SVT_REALSXP;
	a.nrow = A->nrow; a.ncol = A->ncol; a.Y = Y; a.ldY = ldY; a.K = K; a.tr_y = tr_y;
	a.out = out; a.out_stride_c = out_stride_c; a.out_stride_k = out_stride_k;
	a.ws = gen_ws; a.ws_bytes = gen_bytes;
	return launch_crossprod_general_if(a, fl.y_nonfinite, false, s);
}

extern "C" int svt_dev_crossprod_pbc_phase(const svt_dev_pbc *P, const svt_dev_csc *A,
					   const double *Y, int64_t ldY, int K, int tr_y,
					   double *out, int64_t out_stride_c,
					   int64_t out_stride_k, void *ws, size_t ws_bytes,
					   void *stream, int phase)
{
	return pbc_phase(P, A, Y, ldY, K, tr_y, out, out_stride_c, out_stride_k, ws, ws_bytes, stream, phase, 0);
}

extern "C" int svt_dev_crossprod_pbc(const svt_dev_pbc *P, const svt_dev_csc *A, const double *Y,
				     int64_t ldY, int K, int tr_y, double *out,
				     int64_t out_stride_c, int64_t out_stride_k, void *ws,
				     size_t ws_bytes, void *stream)
{
	return svt_dev_crossprod_pbc_from(P, A, Y, ldY, K, tr_y, out, out_stride_c, out_stride_k,
					  ws, ws_bytes, stream, 0);
}

extern "C" int svt_dev_crossprod_pbc_from(const svt_dev_pbc *P, const svt_dev_csc *A, const double *Y,
					  int64_t ldY, int K, int tr_y, double *out,
					  int64_t out_stride_c, int64_t out_stride_k, void *ws,
					  size_t ws_bytes, void *stream, int64_t first_col)
{
	if (pbc_phase(P, A, Y, ldY, K, tr_y, out, out_stride_c, out_stride_k, ws, ws_bytes, stream, 1, first_col))
		return -1;
	return pbc_phase(P, A, Y, ldY, K, tr_y, out, out_stride_c, out_stride_k, ws, ws_bytes, stream, 2, first_col);
}
