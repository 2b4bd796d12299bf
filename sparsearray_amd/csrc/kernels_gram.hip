// crossprod(X, Y) for two SPARSE operands without a dense operand: out[c, j] = sum over the rows r where BOTH
// X[r, c] and Y[r, j] are nonzero of X[r, c] * Y[r, j].
//
// The reference (C_crossprod2_SVT_SVT / C_crossprod1_SVT, src/SparseMatrix_mult.c:1037-1140) expands the leaves
// of one operand, one at a time, into a dense buffer and walks ALL leaves of the other operand over it
// (crossprod2_Lpp_* / crossprod2_Rpp_* :728-820, compute_sym_dotprods_* :827-873, the loops K11-K13 :263-296):
// ncol(x) * nnz(y) multiply-adds, nearly all of them with the buffer's zeros.  With finite operands the cell is
// the sum over the rows where both leaves hold a nonzero (the others add exact zeros), and that is all this
// kernel multiplies: at BASELINE config-2 scale (crossprod(A), A 1e6 x 1e4 @ 1 %) 5e9 products for the upper
// triangle where the dense-buffer route -- rounds 2-5: densify 128 columns, panel product, 79 times -- does 5e11.
//
// Form (column-wise Gustavson on t(X)): the workgroup that owns result column j keeps its cells out[., j] in LDS
// and, for every nonzero (r, v) of Y[:, j], adds v * X[r, .] -- row r of X, i.e. LEAF r of t(X), a contiguous run
// of (column, value) pairs -- into them (ds_add_f64; two rows can meet in a cell).  A group of G lanes takes one
// nonzero of Y at a time.  t(X) is the caller's (svt_dev_transpose; for x %*% y the operand itself).
//   * nx = ncol(X) <= 20400 (symmetric form: 16384): all cells of a result column fit one workgroup's LDS (two
//     workgroups per CU up to 10200 cells, one beyond; gram_one_block() has the measurements).
//     Symmetric case (Y is X): only the cells c <= j are formed -- row r's pairs come in ascending column order,
//     a lane stops at its first column > j, so the prefix needs no search -- and columns j and nx - 1 - j share a
//     workgroup: nx + 1 cells, the same work for every workgroup (which keeps the workgroups in flight walking
//     the same stretch of rows: the runs several of them name can then come from the memory-side cache).
//   * wider results: panels of 8192 cells; the part of leaf r inside a panel comes from the table of run bounds
//     that the row-panel kernels use (launch_rowpanel_table, kernels_rowstats.hip); symmetric: panels above the
//     diagonal cell are skipped, the diagonal panel is cut as above.
// The lower triangle of a symmetric result is the mirror image of the upper one (gram_mirror_kernel, 64 x 64
// tiles through LDS), as compute_sym_dotprods_* writes out[k] and out[k * ncol] from one dot product.
//
// Not a sum in the reference's order: the additions of one cell come in the order the lane groups get to them
// (last-bit differences between runs; integer operands are exact below 2^53).  A non-finite value or an NA
// ANYWHERE in either operand changes what the reference computes (its dirty-leaf loops multiply the implicit
// zeros too, src/SparseVec_dotprod.c:48-65; an NA_integer_ fills whole rows / columns, :684-724): *flag goes up
// and the caller takes the dense-buffer route.  Every value of Y is looked at by the product itself, the values
// of X by a scan in front of it (symmetric: X is Y).
#include "svt_common.h"

#define GRAM_NT 1024
#define GRAM_U 4

static int g_gram_one = 20400, g_gram_ps = 13;
#define GRAM_SYM_ONE_MAX 16384          // the symmetric form keeps one block up to here (see gram_one_block)

void gram_set_panel(int one_block_max, int log2_panel)
{
	g_gram_one = one_block_max < 0 ? 20400 : (one_block_max > 20400 ? 20400 : one_block_max);   // (> 10200: one workgroup per CU)
	g_gram_ps = log2_panel < 4 || log2_panel > 14 ? 13 : log2_panel;
}

template <typename T> __device__ inline bool gram_bad(T v);
template <> __device__ inline bool gram_bad<double>(double v) { return !(fabs(v) <= 1.7976931348623157e308); }
template <> __device__ inline bool gram_bad<int>(int v) { return v == NA_INT; }

template <typename T>
__global__ void __launch_bounds__(256)
gram_scan_values_kernel(const T *__restrict__ val, int64_t n, int *__restrict__ flag)
{
	bool bad = false;
	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
		bad |= gram_bad<T>(val[i]);
	if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) *flag = 1;
}

// MODE 0: one block of cells per result column; 1: the same, symmetric, columns j and nx - 1 - j per workgroup;
//      2: panels of cells + table of run bounds; 3: the same, symmetric (cells c <= j only)
template <typename TA, typename TB, int MODE>
__global__ void __launch_bounds__(GRAM_NT)
gram_kernel(GramArgs a, int G)
{
	extern __shared__ double acc[];
	__shared__ int stop;
	const int tid = threadIdx.x, NT = blockDim.x;
	constexpr bool CUT = MODE == 1 || MODE == 3;      // entries at or past the task's last cell end a lane's walk
	constexpr bool TABLE = MODE >= 2;
	// the workgroup's tasks: result column, cells [clo, chi), where they sit in LDS
	int ntask = 1;
	int64_t tk[2] = { 0, 0 };
	int clo[2] = { 0, 0 }, chi[2] = { 0, 0 }, off[2] = { 0, 0 };
	int64_t q = 0;
	if (MODE == 0) {
		tk[0] = blockIdx.x; chi[0] = (int) a.nx;
	} else if (MODE == 1) {
		tk[0] = blockIdx.x; chi[0] = (int) tk[0] + 1;
		tk[1] = a.nx - 1 - tk[0]; chi[1] = (int) tk[1] + 1; off[1] = chi[0];
		if (tk[1] != tk[0]) ntask = 2;
	} else {
		const int64_t L = blockIdx.x;
		q = L % a.npan; tk[0] = L / a.npan;
		const int64_t c0 = q << a.ps;
		int64_t c1 = c0 + ((int64_t) 1 << a.ps);
		if (c1 > a.nx) c1 = a.nx;
		if (MODE == 3 && c1 > tk[0] + 1) c1 = tk[0] + 1;
		if (c1 <= c0)
			return;                         // (a panel above the diagonal cell: the whole workgroup leaves)
		clo[0] = (int) c0; chi[0] = (int) c1;
	}
	const int ncell = off[ntask - 1] + chi[ntask - 1] - clo[ntask - 1];
	for (int x = tid; x < ncell; x += NT) acc[x] = 0.0;
	if (tid == 0)
		stop = *(volatile const int *) a.flag;  // (ONE read per workgroup: its wavefronts must agree)
	__syncthreads();
	if (stop != 0)
		return;
	const int grp = tid / G, sl = tid % G, ngrp = NT / G;
	const TA *__restrict__ av = (const TA *) a.a_val;
	const TB *__restrict__ bv = (const TB *) a.b_val;
	const int32_t *__restrict__ pt0 = TABLE ? a.pt + q * a.nrow : NULL;
	const int32_t *__restrict__ pt1 = TABLE ? pt0 + a.nrow : NULL;
	bool bad = false;
	for (int ti = 0; ti < ntask; ti++) {
		const int64_t bb = a.b_ptr[tk[ti]];
		const int64_t npairs = a.b_ptr[tk[ti] + 1] - bb;
		const int lo = clo[ti], hi = chi[ti];
		double *__restrict__ cell = acc + off[ti] - lo;
		// GRAM_U nonzeros of Y per group in flight: their bounds are fetched together, then their runs
		for (int64_t t0 = grp; t0 < npairs; t0 += (int64_t) GRAM_U * ngrp) {
			int64_t xb[GRAM_U], xe[GRAM_U];
			double bval[GRAM_U];
#pragma unroll
			for (int u = 0; u < GRAM_U; u++) {
				const int64_t t = t0 + (int64_t) u * ngrp;
				xb[u] = xe[u] = 0; bval[u] = 0.0;
				if (t < npairs) {
					const int64_t r = a.b_idx[bb + t];
					const TB b = bv[bb + t];
					const int64_t base = a.a_ptr[r];
					if (TABLE) { xb[u] = base + pt0[r] + sl; xe[u] = base + pt1[r]; }
					else { xb[u] = base + sl; xe[u] = a.a_ptr[r + 1]; }
					bval[u] = (double) b;
					bad |= gram_bad<TB>(b);
				}
			}
			bool more = true;
			while (more) {
				TA v[GRAM_U];
				int c[GRAM_U];
#pragma unroll
				for (int u = 0; u < GRAM_U; u++)
					if (xb[u] < xe[u]) { c[u] = a.a_idx[xb[u]]; v[u] = av[xb[u]]; }
				more = false;
#pragma unroll
				for (int u = 0; u < GRAM_U; u++)
					if (xb[u] < xe[u]) {
						if (!CUT || c[u] < hi) {
							bad |= gram_bad<TA>(v[u]);
							atomicAdd(&cell[c[u]], (double) v[u] * bval[u]);
							xb[u] += G;
							more |= xb[u] < xe[u];
						} else
							xe[u] = 0;      // (ascending columns: the lane's later entries are past the cut too)
					}
			}
		}
	}
	if (__ballot(bad) != 0 && (tid & 63) == 0) *a.flag = 1;
	__syncthreads();
	for (int ti = 0; ti < ntask; ti++) {
		double *__restrict__ dst = a.out + tk[ti] * a.ldo + clo[ti];
		const double *__restrict__ src = acc + off[ti];
		const int n = chi[ti] - clo[ti];
		for (int x = tid; x < n; x += NT) dst[x] = src[x];
	}
}

// The symmetric one-block form.  A lane group's walk of row r for result column j ends at the first column > j:
// walks for one result column vary from nothing to the whole row, and a wavefront is as slow as its longest walk
// (first version, one walk per slot: 24.3 ms at config-2 scale where the general form, which reads twice as much,
// took 22.1).  Here a slot's unit of work is the t-th nonzero of column j TOGETHER with the t-th nonzero of
// column n - 1 - j: a prefix of one row up to column j and a prefix of another up to column n - 1 - j, about one
// row's length in all whatever j is: 18.3 ms; on 12-byte records (one run of memory per walk instead of two) 14.6;
// with the loop written as it stands below 13.8.  Lane groups of 16 (32 lanes read further past the cuts: 15.5; 8
// lanes make twice the trips: 16.5); two consecutive entries per lane and trip 19.5 (twice the over-read).
// Records of t(X) for the symmetric form: (column, value) side by side, 12 bytes (8 for integer values) -- a
// prefix of a row is then ONE run of memory instead of two (columns, values), read by one load per lane
// (Measured at config-2 scale, symmetric form: t(X) as it is, two arrays, 16.3 ms; 12-byte records 14.6 incl. the 0.45 ms
// that makes them; 10-byte records with a 16-bit column -- the value at a 2-byte boundary, global_load_dwordx2 at
// offset 2 -- 15.4.)
template <typename T, int AOS> struct GramRec;
template <> struct __attribute__((packed, aligned(4))) GramRec<double, 1> { double v; int c; };   // (value first: it lands in an even register pair)
template <> struct __attribute__((packed, aligned(4))) GramRec<int, 1> { int v; int c; };

// a record as the plain words a load brings in
template <typename T> struct GramRaw;
template <> struct GramRaw<double> {
	struct type { int a, b, c; };           // value (low, high word), column
	__device__ static inline int col(const type &q) { return q.c; }
	__device__ static inline double val(const type &q) { return __hiloint2double(q.b, q.a); }
	// (all three words are wanted HERE: without this the compiler loads the column alone and fetches the value
	// behind the cut test, a second, dependent trip to memory)
	__device__ static inline void pin(type &q) { asm volatile("" : "+v"(q.a), "+v"(q.b), "+v"(q.c)); }
};
template <> struct GramRaw<int> {
	struct type { int a, b; };              // value, column
	__device__ static inline int col(const type &q) { return q.b; }
	__device__ static inline int val(const type &q) { return q.a; }
	__device__ static inline void pin(type &q) { asm volatile("" : "+v"(q.a), "+v"(q.b)); }
};

template <typename T, int AOS>
__global__ void __launch_bounds__(256)
gram_pack_kernel(const int32_t *__restrict__ idx, const T *__restrict__ val, int64_t n, GramRec<T, AOS> *__restrict__ rec)
{
	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
		GramRec<T, AOS> r;
		r.c = idx[i]; r.v = val[i];
		rec[i] = r;
	}
}

template <typename T, int SU, int G>
__global__ void __launch_bounds__(GRAM_NT)
gram_sym_kernel(GramArgs a)
{
	extern __shared__ double acc[];
	__shared__ int stop;
	const int tid = threadIdx.x, NT = blockDim.x;
	const int64_t k1 = blockIdx.x, k2 = a.nx - 1 - k1;
	const int hi1 = (int) k1 + 1, hi2 = (int) k2 + 1, off2 = hi1;
	const int ncell = k2 != k1 ? hi1 + hi2 : hi1;
	for (int x = tid; x < ncell; x += NT) acc[x] = 0.0;
	if (tid == 0)
		stop = *(volatile const int *) a.flag;
	__syncthreads();
	if (stop != 0)
		return;
	const int grp = tid / G, sl = tid % G, ngrp = NT / G;
	const typename GramRaw<T>::type *__restrict__ rec = (const typename GramRaw<T>::type *) a.a_val;
	const T *__restrict__ bv = (const T *) a.b_val;
	const int64_t bb1 = a.b_ptr[k1];
	const int np1 = (int) (a.b_ptr[k1 + 1] - bb1);
	const int64_t bb2 = a.b_ptr[k2];
	const int np2 = k2 != k1 ? (int) (a.b_ptr[k2 + 1] - bb2) : 0;
	const int nunits = np1 > np2 ? np1 : np2;
	// Every stored value of x is the `b` of exactly one walk of some workgroup (Y is X): looking at the b's looks at
	// all of them, the entries of the walked rows need no look of their own.
	bool bad = false;
	// positions inside t(X) in 32 bits (the launcher sends operands with 2^31 nonzeros or more to gram_kernel).
	// What bounds the loop below (config-2 scale, 13.5-14 ms): memory.  Its first form issued 39 vector + 25 scalar
	// instructions per LDS add (rocprofv3 SQ_INSTS_*: 4.5e9 / 2.9e9 / 1.15e8 per launch) and waited behind every load;
	// this one -- selects instead of branches, one exec-masked region, the loads of all slots in flight together --
	// issues half of that and runs in the same time, with one unit per lane group in flight or two (13.8 / 14.0 ms;
	// three 15.9, four 17.6: more walks in flight only spread the reads of a row further apart).  The walks move
	// ~60 GB of row prefixes in 600-byte runs plus what the last trip of a walk reads past its cut; L2 serves 46 %
	// of the requests (TCC_HIT / TCC_REQ), the rest is the rate at which HBM and the memory-side cache deliver
	// such runs.  Workgroups deliberately started at different places of their sweeps: 18.0 ms (the ~1000 columns in
	// flight name every row ~10 times; side by side those reads share the caches).
	for (int t0 = grp; t0 < nunits; t0 += SU * ngrp) {
		unsigned x[SU], xe[SU], x2[SU], xe2[SU];
		double b[SU], b2[SU];
		int hi[SU], off[SU];
#pragma unroll
		for (int u = 0; u < SU; u++) {
			const int t = t0 + u * ngrp;
			x[u] = xe[u] = x2[u] = xe2[u] = 0; b[u] = b2[u] = 0.0; hi[u] = hi1; off[u] = 0;
			if (t < np1) {
				const int64_t r = a.b_idx[bb1 + t];
				const T w = bv[bb1 + t];
				x[u] = (unsigned) a.a_ptr[r] + sl; xe[u] = (unsigned) a.a_ptr[r + 1];
				b[u] = (double) w; bad |= gram_bad<T>(w);
			}
			if (t < np2) {
				const int64_t r = a.b_idx[bb2 + t];
				const T w = bv[bb2 + t];
				x2[u] = (unsigned) a.a_ptr[r] + sl; xe2[u] = (unsigned) a.a_ptr[r + 1];
				b2[u] = (double) w; bad |= gram_bad<T>(w);
			}
		}
		bool more = true;
		while (more) {
			typename GramRaw<T>::type raw[SU];
			bool act[SU];
			// all SU loads are issued before any is waited for: no branch around a load (a lane whose walk is over
			// reads record 0 again, one address for the whole wavefront), the record taken apart only where it is used
#pragma unroll
			for (int u = 0; u < SU; u++) {
				const bool sw = x[u] >= xe[u];          // this walk is over: on to the unit's second one (or to nothing)
				x[u] = sw ? x2[u] : x[u]; xe[u] = sw ? xe2[u] : xe[u]; b[u] = sw ? b2[u] : b[u];
				hi[u] = sw ? hi2 : hi[u]; off[u] = sw ? off2 : off[u];
				xe2[u] = sw ? 0u : xe2[u];
				act[u] = x[u] < xe[u];
				raw[u] = rec[act[u] ? x[u] : 0u];
			}
			more = false;
#pragma unroll
			for (int u = 0; u < SU; u++) GramRaw<T>::pin(raw[u]);
#pragma unroll
			for (int u = 0; u < SU; u++) {
				const int c = GramRaw<T>::col(raw[u]);
				const double p = (double) GramRaw<T>::val(raw[u]) * b[u];
				const bool ok = act[u] && c < hi[u];    // (ascending columns: past the cut once, past it for good)
				if (ok) atomicAdd(&acc[off[u] + c], p);
				x[u] = ok ? x[u] + G : xe[u];
				more |= x[u] < xe[u] || x2[u] < xe2[u];
			}
		}
	}
	if (__ballot(bad) != 0 && (tid & 63) == 0) *a.flag = 1;
	__syncthreads();
	{
		double *__restrict__ dst = a.out + k1 * a.ldo;
		for (int x = tid; x < hi1; x += NT) dst[x] = acc[x];
	}
	if (k2 != k1) {
		double *__restrict__ dst = a.out + k2 * a.ldo;
		for (int x = tid; x < hi2; x += NT) dst[x] = acc[off2 + x];
	}
}

// The general one-block form on records: one result column per workgroup, whole rows of x (no cut), the loop of
// gram_sym_kernel.  (gram_kernel<.., 0> on the two arrays of t(X): 22.1 ms for the config-2 operand against itself.)
// The values of x are looked at by the scan in front of the launch, those of y here.
template <typename TA, typename TB, int SU, int G>
__global__ void __launch_bounds__(GRAM_NT)
gram_gen_kernel(GramArgs a)
{
	extern __shared__ double acc[];
	__shared__ int stop;
	const int tid = threadIdx.x, NT = blockDim.x;
	const int64_t k = blockIdx.x;
	const int ncell = (int) a.nx;
	for (int x = tid; x < ncell; x += NT) acc[x] = 0.0;
	if (tid == 0)
		stop = *(volatile const int *) a.flag;
	__syncthreads();
	if (stop != 0)
		return;
	const int grp = tid / G, sl = tid % G, ngrp = NT / G;
	const typename GramRaw<TA>::type *__restrict__ rec = (const typename GramRaw<TA>::type *) a.a_val;
	const TB *__restrict__ bv = (const TB *) a.b_val;
	const int64_t bb = a.b_ptr[k];
	const int np = (int) (a.b_ptr[k + 1] - bb);
	bool bad = false;
	for (int t0 = grp; t0 < np; t0 += SU * ngrp) {
		unsigned x[SU], xe[SU];
		double b[SU];
#pragma unroll
		for (int u = 0; u < SU; u++) {
			const int t = t0 + u * ngrp;
			x[u] = xe[u] = 0; b[u] = 0.0;
			if (t < np) {
				const int64_t r = a.b_idx[bb + t];
				const TB w = bv[bb + t];
				x[u] = (unsigned) a.a_ptr[r] + sl; xe[u] = (unsigned) a.a_ptr[r + 1];
				b[u] = (double) w; bad |= gram_bad<TB>(w);
			}
		}
		bool more = true;
		while (more) {
			typename GramRaw<TA>::type raw[SU];
			bool act[SU];
#pragma unroll
			for (int u = 0; u < SU; u++) {
				act[u] = x[u] < xe[u];
				raw[u] = rec[act[u] ? x[u] : 0u];
			}
			more = false;
#pragma unroll
			for (int u = 0; u < SU; u++) GramRaw<TA>::pin(raw[u]);
#pragma unroll
			for (int u = 0; u < SU; u++) {
				const double p = (double) GramRaw<TA>::val(raw[u]) * b[u];
				if (act[u]) atomicAdd(&acc[GramRaw<TA>::col(raw[u])], p);
				x[u] += G;
				more |= x[u] < xe[u];
			}
		}
	}
	if (__ballot(bad) != 0 && (tid & 63) == 0) *a.flag = 1;
	__syncthreads();
	double *__restrict__ dst = a.out + k * a.ldo;
	for (int x = tid; x < ncell; x += NT) dst[x] = acc[x];
}

// The panel form on records (results wider than one block): workgroup = (panel of 2^ps cells, result column), the
// part of row r inside the panel from the table of run bounds, CUT: the symmetric product's cells c <= column only
// (panels above the diagonal cell leave at once).  The loop of gram_gen_kernel.
template <typename TA, typename TB, bool CUT, int SU, int G>
__global__ void __launch_bounds__(GRAM_NT)
gram_pan_kernel(GramArgs a)
{
	extern __shared__ double acc[];
	__shared__ int stop;
	const int tid = threadIdx.x, NT = blockDim.x;
	const int64_t L = blockIdx.x;
	const int64_t q = L % a.npan, k = L / a.npan;
	const int64_t c0 = q << a.ps;
	int64_t c1 = c0 + ((int64_t) 1 << a.ps);
	if (c1 > a.nx) c1 = a.nx;
	if (CUT && c1 > k + 1) c1 = k + 1;
	if (c1 <= c0)
		return;                                         // (a panel above the diagonal cell: the whole workgroup leaves)
	const int ncell = (int) (c1 - c0), lo = (int) c0, hi = (int) c1;
	for (int x = tid; x < ncell; x += NT) acc[x] = 0.0;
	if (tid == 0)
		stop = *(volatile const int *) a.flag;
	__syncthreads();
	if (stop != 0)
		return;
	const int grp = tid / G, sl = tid % G, ngrp = NT / G;
	const typename GramRaw<TA>::type *__restrict__ rec = (const typename GramRaw<TA>::type *) a.a_val;
	const TB *__restrict__ bv = (const TB *) a.b_val;
	const int32_t *__restrict__ pt0 = a.pt + q * a.nrow, *__restrict__ pt1 = pt0 + a.nrow;
	const int64_t bb = a.b_ptr[k];
	const int np = (int) (a.b_ptr[k + 1] - bb);
	bool bad = false;
	for (int t0 = grp; t0 < np; t0 += SU * ngrp) {
		unsigned x[SU], xe[SU];
		double b[SU];
#pragma unroll
		for (int u = 0; u < SU; u++) {
			const int t = t0 + u * ngrp;
			x[u] = xe[u] = 0; b[u] = 0.0;
			if (t < np) {
				const int64_t r = a.b_idx[bb + t];
				const TB w = bv[bb + t];
				const unsigned base = (unsigned) a.a_ptr[r];
				x[u] = base + (unsigned) pt0[r] + sl; xe[u] = base + (unsigned) pt1[r];
				b[u] = (double) w; bad |= gram_bad<TB>(w);
			}
		}
		bool more = true;
		while (more) {
			typename GramRaw<TA>::type raw[SU];
			bool act[SU];
#pragma unroll
			for (int u = 0; u < SU; u++) {
				act[u] = x[u] < xe[u];
				raw[u] = rec[act[u] ? x[u] : 0u];
			}
			more = false;
#pragma unroll
			for (int u = 0; u < SU; u++) GramRaw<TA>::pin(raw[u]);
#pragma unroll
			for (int u = 0; u < SU; u++) {
				const int c = GramRaw<TA>::col(raw[u]);
				const double p = (double) GramRaw<TA>::val(raw[u]) * b[u];
				const bool ok = act[u] && (!CUT || c < hi);
				if (ok) atomicAdd(&acc[c - lo], p);
				x[u] = ok ? x[u] + G : xe[u];
				more |= x[u] < xe[u];
			}
		}
	}
	if (__ballot(bad) != 0 && (tid & 63) == 0) *a.flag = 1;
	__syncthreads();
	double *__restrict__ dst = a.out + k * a.ldo + c0;
	for (int x = tid; x < ncell; x += NT) dst[x] = acc[x];
}

// out[j, i] = out[i, j] for i < j (n x n, column-major, leading dimension ld): 64 x 64 tiles through LDS, both
// the reads and the writes run along columns
__global__ void __launch_bounds__(256)
gram_mirror_kernel(double *__restrict__ out, int64_t n, int64_t ld)
{
	__shared__ double tile[64][65];
	const int64_t bi = blockIdx.x, bj = blockIdx.y;
	if (bi > bj)
		return;
	const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
	const int64_t i = bi * 64 + tx;
	for (int ty = ty0; ty < 64; ty += 4) {
		const int64_t j = bj * 64 + ty;
		if (i < n && j < n) tile[ty][tx] = out[i + j * ld];
	}
	__syncthreads();
	const int64_t row = bj * 64 + tx;
	for (int ty = ty0; ty < 64; ty += 4) {
		const int64_t col = bi * 64 + ty;
		if (row < n && col < n && row > col) out[row + col * ld] = tile[tx][ty];
	}
}

int launch_gram_mirror(double *out, int64_t n, int64_t ld, hipStream_t s)
{
	if (n <= 1)
		return 0;
	const unsigned nt = (unsigned) ((n + 63) / 64);
	if (nt > 65535)
		return svt_set_unsupported("sparse crossprod: result too wide to mirror");
	hipLaunchKernelGGL(gram_mirror_kernel, dim3(nt, nt), dim3(256), 0, s, out, n, ld);
	HIP_TRY(hipGetLastError());
	return 0;
}

static int g_gram_aos = 1, g_gram_su = 1;

// does the symmetric one-block form (gram_sym_kernel) apply?  (32-bit positions inside t(X))
static bool gram_sym_one(int64_t nx, int64_t a_nnz)
{
	return nx <= g_gram_one && a_nnz < (int64_t) 2147483647 - 64;
}

// *out += sum over the leaves of t(X) of len * (len + 1) / 2: the pairs of nonzeros the symmetric form multiplies, exactly
// (the entry points' route choice estimates them as nnz^2 / (2 nrow), which rows of very unequal length exceed)
__global__ void __launch_bounds__(256)
gram_pairs_kernel(const int64_t *__restrict__ a_ptr, int64_t nrow, double *__restrict__ out)
{
	double t = 0.0;
	for (int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; r < nrow; r += (int64_t) gridDim.x * blockDim.x) {
		const double len = (double) (a_ptr[r + 1] - a_ptr[r]);
		t += 0.5 * len * (len + 1.0);
	}
	t = wave_sum(t);
	if ((threadIdx.x & 63) == 0 && t != 0.0) atomicAdd(out, t);
}

int launch_gram_pairs(const int64_t *a_ptr, int64_t nrow, double *out, hipStream_t s)
{
	HIP_TRY(hipMemsetAsync(out, 0, 8, s));
	if (nrow > 0) {
		int64_t nb = (nrow + 255) / 256;
		if (nb > 2048) nb = 2048;
		hipLaunchKernelGGL(gram_pairs_kernel, dim3((unsigned) nb), dim3(256), 0, s, a_ptr, nrow, out);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// [256 bytes: flag words][table of run bounds, wide results only | records of t(X), one-block forms]
// One block of cells per result column, or panels?  Measured (x 2e5 x 1.2e4 @ 0.5 % / 1e5 x 2e4 @ 1 % / 1e5 x 3e4 @ 0.5 %, ms):
//   symmetric: one block (one workgroup per CU past 10200 cells) 1.8 / 8.2 / -, panels of 8192 2.3 / 6.1 / 7.4, of 16384 - / - / 7.0
//   general:   one block 1.9 / 7.6 / -, panels of 8192 2.6 / 9.4 / 6.9, of 16384 - / - / 7.8
// (the symmetric form skips the panels above the diagonal cell, whose work a single block only cuts short).
static bool gram_one_block(int64_t nx, bool sym)
{
	const int64_t lim = sym && g_gram_one > GRAM_SYM_ONE_MAX ? GRAM_SYM_ONE_MAX : g_gram_one;
	return nx <= lim;
}

size_t gram_ws_bytes(int64_t nx, int64_t nrow, int64_t a_nnz)
{
	const bool small = a_nnz < (int64_t) 2147483647 - 64;
	const size_t recs = small ? ((size_t) (a_nnz > 0 ? a_nnz : 1) * 12 + 255) / 256 * 256 : 0;
	size_t need = 0;
	if (gram_one_block(nx, false) || gram_one_block(nx, true))              // (either form may be asked for)
		need = gram_sym_one(nx, a_nnz) ? recs : 0;
	if (!gram_one_block(nx, false) || !gram_one_block(nx, true)) {
		const int64_t npan = (nx + ((int64_t) 1 << g_gram_ps) - 1) >> g_gram_ps;
		const size_t pan = ((size_t) (nrow > 0 ? nrow : 1) * (size_t) (npan + 1) * 4 + 255) / 256 * 256 + recs;
		if (pan > need) need = pan;
	}
	return 256 + need;
}

int launch_gram(GramArgs a, int64_t a_nnz, int64_t b_nnz, void *ws, hipStream_t s)
{
	(void) b_nnz;
	if (a.nx <= 0 || a.ny <= 0)
		return 0;
	int *flag = (int *) ws;
	a.flag = flag;
	HIP_TRY(hipMemsetAsync(flag, 0, 8, s));
	if (!a.sym && a_nnz > 0) {                      // the values of X (symmetric: every value is some workgroup's Y value)
		int64_t nb = (a_nnz + 256 * 8 - 1) / (256 * 8);
		if (nb > 256 * 16) nb = 256 * 16;
		if (a.a_type == SVT_REALSXP)
			hipLaunchKernelGGL(gram_scan_values_kernel<double>, dim3((unsigned) nb), dim3(256), 0, s,
					   (const double *) a.a_val, a_nnz, flag);
		else
			hipLaunchKernelGGL(gram_scan_values_kernel<int>, dim3((unsigned) nb), dim3(256), 0, s,
					   (const int *) a.a_val, a_nnz, flag);
	}
	const bool one = gram_one_block(a.nx, a.sym != 0);
	int mode;
	size_t lds;
	int64_t nwg;
	double run = a.nrow > 0 ? (double) a_nnz / (double) a.nrow : 0.0;        // mean length of a lane group's walk
	if (one) {
		mode = a.sym ? 1 : 0;
		lds = (size_t) (a.nx + (a.sym ? 1 : 0)) * 8;
		nwg = a.sym ? (a.nx + 1) / 2 : a.ny;
		a.npan = 1; a.ps = 0; a.pt = NULL;
	} else {
		mode = a.sym ? 3 : 2;
		a.ps = g_gram_ps;
		a.npan = (a.nx + ((int64_t) 1 << a.ps) - 1) >> a.ps;
		lds = ((size_t) 1 << a.ps) * 8;
		nwg = a.npan * a.ny;
		a.pt = (const int32_t *) ((char *) ws + 256);
		launch_rowpanel_table(a.a_ptr, a.a_idx, a.nrow, a_nnz, a.npan, a.ps, (int32_t *) a.pt, s);
		run /= (double) a.npan;
	}
	if (a.sym) run *= 0.5;
	if (nwg >= (int64_t) 2147483647)
		return svt_set_unsupported("sparse crossprod: too many workgroups for one launch");
	int G = 64;
	while (G > 8 && run < 2.0 * G) G >>= 1;
#ifdef SVT_TUNING
	if (getenv("SVT_GRAM_G")) G = atoi(getenv("SVT_GRAM_G"));
#endif
	int nt = GRAM_NT;
#ifdef SVT_TUNING
	if (getenv("SVT_GRAM_NT")) nt = atoi(getenv("SVT_GRAM_NT"));
#endif
	const dim3 grid((unsigned) nwg);
#define GRAM_GO(TA, TB, M) do { \
		(void) hipFuncSetAttribute((const void *) gram_kernel<TA, TB, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
		hipLaunchKernelGGL((gram_kernel<TA, TB, M>), grid, dim3(nt), lds, s, a, G); } while (0)
#define GRAM_MODES(TA, TB) do { \
		if (mode == 0) GRAM_GO(TA, TB, 0); else if (mode == 1) GRAM_GO(TA, TB, 1); \
		else if (mode == 2) GRAM_GO(TA, TB, 2); else GRAM_GO(TA, TB, 3); } while (0)
	int symk = 1, aos = g_gram_aos, su = g_gram_su;  // (tuning build: SVT_GRAM_SYMK=0 = the first forms, on t(X)'s two arrays)
#ifdef SVT_TUNING
	if (getenv("SVT_GRAM_SYMK")) symk = atoi(getenv("SVT_GRAM_SYMK"));
	if (getenv("SVT_GRAM_SU")) su = atoi(getenv("SVT_GRAM_SU"));
#endif
	(void) su;
	const bool recs = aos && symk && gram_sym_one(a.nx, a_nnz) && ((mode == 1 && a.a_type == a.b_type) || mode == 0);
	if (recs) {
		void *rec = (char *) ws + 256;
		int64_t nb = (a_nnz + 255) / 256;
		if (nb > 256 * 32) nb = 256 * 32;
		if (nb < 1) nb = 1;
#define GRAM_PACK(T, W) hipLaunchKernelGGL((gram_pack_kernel<T, W>), dim3((unsigned) nb), dim3(256), 0, s, a.a_idx, (const T *) a.a_val, a_nnz, (GramRec<T, W> *) rec)
		if (a.a_type == SVT_REALSXP) GRAM_PACK(double, 1); else GRAM_PACK(int, 1);
#undef GRAM_PACK
		a.a_val = rec; a.a_idx = NULL;
	}
	if (!one && aos && symk && a_nnz < (int64_t) 2147483647 - 64 && (mode == 2 || a.a_type == a.b_type)) {
		void *rec = (char *) ws + 256 + ((size_t) (a.nrow > 0 ? a.nrow : 1) * (size_t) (a.npan + 1) * 4 + 255) / 256 * 256;
		int64_t nb = (a_nnz + 255) / 256;
		if (nb > 256 * 32) nb = 256 * 32;
		if (nb < 1) nb = 1;
		if (a.a_type == SVT_REALSXP)
			hipLaunchKernelGGL((gram_pack_kernel<double, 1>), dim3((unsigned) nb), dim3(256), 0, s, a.a_idx, (const double *) a.a_val, a_nnz, (GramRec<double, 1> *) rec);
		else
			hipLaunchKernelGGL((gram_pack_kernel<int, 1>), dim3((unsigned) nb), dim3(256), 0, s, a.a_idx, (const int *) a.a_val, a_nnz, (GramRec<int, 1> *) rec);
		a.a_val = rec; a.a_idx = NULL;
#define GRAM_PAN_GO(TA, TB, CUT, GG) do { \
		(void) hipFuncSetAttribute((const void *) gram_pan_kernel<TA, TB, CUT, 2, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
		hipLaunchKernelGGL((gram_pan_kernel<TA, TB, CUT, 2, GG>), grid, dim3(nt), lds, s, a); } while (0)
#define GRAM_PAN_G(TA, TB, CUT) do { if (G >= 32) GRAM_PAN_GO(TA, TB, CUT, 32); else if (G <= 8) GRAM_PAN_GO(TA, TB, CUT, 8); else GRAM_PAN_GO(TA, TB, CUT, 16); } while (0)
#define GRAM_PAN_T(TA, TB) do { if (mode == 3) GRAM_PAN_G(TA, TB, true); else GRAM_PAN_G(TA, TB, false); } while (0)
		if (a.a_type == SVT_REALSXP && a.b_type == SVT_REALSXP) GRAM_PAN_T(double, double);
		else if (a.a_type == SVT_INTSXP && a.b_type == SVT_INTSXP) GRAM_PAN_T(int, int);
		else if (a.a_type == SVT_REALSXP && a.b_type == SVT_INTSXP) GRAM_PAN_T(double, int);
		else GRAM_PAN_T(int, double);
#undef GRAM_PAN_T
#undef GRAM_PAN_G
#undef GRAM_PAN_GO
	} else
	if (recs && mode == 0) {
#define GRAM_GEN_GO(TA, TB, GG) do { \
		(void) hipFuncSetAttribute((const void *) gram_gen_kernel<TA, TB, 2, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
		hipLaunchKernelGGL((gram_gen_kernel<TA, TB, 2, GG>), grid, dim3(nt), lds, s, a); } while (0)
#define GRAM_GEN_G(TA, TB) do { if (G >= 32) GRAM_GEN_GO(TA, TB, 32); else if (G <= 8) GRAM_GEN_GO(TA, TB, 8); else GRAM_GEN_GO(TA, TB, 16); } while (0)
		if (a.a_type == SVT_REALSXP && a.b_type == SVT_REALSXP) GRAM_GEN_G(double, double);
		else if (a.a_type == SVT_INTSXP && a.b_type == SVT_INTSXP) GRAM_GEN_G(int, int);
		else if (a.a_type == SVT_REALSXP && a.b_type == SVT_INTSXP) GRAM_GEN_G(double, int);
		else GRAM_GEN_G(int, double);
#undef GRAM_GEN_G
#undef GRAM_GEN_GO
	} else
	if (recs && mode == 1) {
#define GRAM_SYM_GO(T, SU, GG) do { \
		(void) hipFuncSetAttribute((const void *) gram_sym_kernel<T, SU, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
		hipLaunchKernelGGL((gram_sym_kernel<T, SU, GG>), grid, dim3(nt), lds, s, a); } while (0)
#ifdef SVT_TUNING
#define GRAM_SYM_G(T, SU) do { if (G >= 32) GRAM_SYM_GO(T, SU, 32); else if (G <= 8) GRAM_SYM_GO(T, SU, 8); else GRAM_SYM_GO(T, SU, 16); } while (0)
#define GRAM_SYM_SU(T) do { if (su == 4) GRAM_SYM_G(T, 4); else if (su == 3) GRAM_SYM_G(T, 3); else if (su == 2) GRAM_SYM_G(T, 2); else GRAM_SYM_G(T, 1); } while (0)
#else
#define GRAM_SYM_SU(T) do { if (G >= 32) GRAM_SYM_GO(T, 1, 32); else if (G <= 8) GRAM_SYM_GO(T, 1, 8); else GRAM_SYM_GO(T, 1, 16); } while (0)
#endif
		if (a.a_type == SVT_REALSXP) GRAM_SYM_SU(double); else GRAM_SYM_SU(int);
#ifdef SVT_TUNING
#undef GRAM_SYM_G
#endif
#undef GRAM_SYM_SU
#undef GRAM_SYM_GO
	} else
	if (a.a_type == SVT_REALSXP && a.b_type == SVT_REALSXP) GRAM_MODES(double, double);
	else if (a.a_type == SVT_INTSXP && a.b_type == SVT_INTSXP) GRAM_MODES(int, int);
	else if (a.a_type == SVT_REALSXP && a.b_type == SVT_INTSXP) GRAM_MODES(double, int);
	else if (a.a_type == SVT_INTSXP && a.b_type == SVT_REALSXP) GRAM_MODES(int, double);
	else return svt_set_error("sparse crossprod: unsupported operand types");
#undef GRAM_MODES
#undef GRAM_GO
	HIP_TRY(hipGetLastError());
	if (a.sym && launch_gram_mirror(a.out, a.nx, a.ldo, s))
		return -1;
	return 0;
}
