// row* matrixStats, rowsum() and colsum() on the CSC device layout.
//
// Reference: C_rowStats_SVT (src/SparseArray_matrixStats.c:1121-1205) walks
// the tree serially and scatters every leaf into a dense `out`
// (update_out_for_rowSums :599-634 etc.); rowsum_SVT_double/int and
// colsum_SVT_double/int (src/rowsum_methods.c:86-125, 204-255) do the same
// into group-indexed outputs.  On the device each leaf is owned by one
// wavefront and the scatter is done with memory-side atomics (or LDS atomics
// when the per-column target fits in LDS), which is legal because every
// update rule of the reference reduces to an order-independent form:
//   sum-like ops : IEEE addition (NA/NaN propagate by themselves)
//   min / max    : "any NA wins, else any NaN wins, else the extremum", plus
//                  the implicit zero when a cell is covered fewer than
//                  nstrata times (:914-961)
// Roofline: HBM (12 B per nonzero + the dense output); the atomic rate of the
// memory side is the practical bound for scattered 8-byte adds.
#include "svt_common.h"

#define RF_NA   1
#define RF_NAN  2
#define RF_HAVE 4

// --------------------------------------------------------------------------
// row stats
// --------------------------------------------------------------------------
struct MinMaxScratch {
	unsigned long long *best;   // ordered-double or sign-extended int
	int *flags;
	unsigned int *cov;
};

__host__ __device__ inline MinMaxScratch split_scratch(void *p, int64_t n)
{
	MinMaxScratch s;
	s.best = (unsigned long long *) p;
	s.flags = (int *) (s.best + n);
	s.cov = (unsigned int *) (s.flags + n);
	return s;
}

size_t rowstats_scratch_bytes(int opcode, int out_Rtype, int64_t out_len)
{
	(void) out_Rtype;
	if (opcode != SVT_OP_MIN && opcode != SVT_OP_MAX)
		return 16;
	return (size_t) out_len * 16 + 16;
}

__global__ void rowstats_init_kernel(RowStatsArgs a)
{
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.out_len)
		return;
	switch (a.opcode) {
	case SVT_OP_ANYNA:
		((int *) a.out)[i] = 0;
		break;
	case SVT_OP_COUNTNAS: case SVT_OP_SUM:
		((double *) a.out)[i] = 0.0;
		break;
	case SVT_OP_CENTERED_X2_SUM: {   // :1044-1066
		const double c = a.center ? a.center[i] : 0.0;
		((double *) a.out)[i] = a.center ? c * c * (double) a.nstrata : 0.0;
		break;
	}
	default: {
		MinMaxScratch s = split_scratch(a.scratch, a.out_len);
		s.best[i] = a.opcode == SVT_OP_MIN ? ~0ULL : 0ULL;
		s.flags[i] = 0;
		s.cov[i] = 0;
	}
	}
}

template <typename T>
__global__ void __launch_bounds__(256)
rowstats_scatter_kernel(RowStatsArgs a)
{
	const int lane = threadIdx.x & 63;
	const int64_t j = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (j >= a.ncol)
		return;
	const T *__restrict__ val = (const T *) a.val;
	const int32_t *__restrict__ row = a.row_idx;
	const int64_t beg = a.col_ptr[j], end = a.col_ptr[j + 1];
	const int64_t base = (j % a.inner) * a.nrow;
	const bool is_dbl = sizeof(T) == 8;
	const bool narm = a.na_rm != 0;
	const double NAr = svt_na_real();
	MinMaxScratch ms = split_scratch(a.scratch, a.out_len);

	for (int64_t k = beg + lane; k < end; k += SVT_WAVE) {
		const T v = val[k];
		const int64_t i = base + row[k];
		const bool miss = is_dbl ? (v != v) : ((int) v == NA_INT);
		const bool isna = is_dbl ? svt_is_na((double) v) : miss;
		switch (a.opcode) {
		case SVT_OP_ANYNA:       // :498-514
			if (miss) ((int *) a.out)[i] = 1;
			break;
		case SVT_OP_COUNTNAS:    // :516-535
			if (miss) atomicAdd((double *) a.out + i, 1.0);
			break;
		case SVT_OP_SUM: {       // :599-634 with :412-433
			if (miss && narm) break;
			const double x = (!is_dbl && miss) ? NAr : (double) v;
			atomicAdd((double *) a.out + i, x);
			break;
		}
		case SVT_OP_CENTERED_X2_SUM: {   // :636-696
			const double c = a.center ? a.center[i] : 0.0;
			if (miss && narm) {
				atomicAdd((double *) a.out + i, -(c * c));
				break;
			}
			const double x = (!is_dbl && miss) ? NAr : (double) v;
			atomicAdd((double *) a.out + i, x * (x - 2 * c));
			break;
		}
		default: {               // min / max, :537-597
			atomicAdd(ms.cov + i, 1u);
			if (miss) {
				atomicOr(ms.flags + i, isna ? RF_NA : RF_NAN);
				break;
			}
			atomicOr(ms.flags + i, RF_HAVE);
			unsigned long long key = is_dbl ?
				f64_to_ordered((double) v) :
				(unsigned long long) ((long long) (int) v + 0x80000000LL);
			if (a.opcode == SVT_OP_MIN) atomicMin(ms.best + i, key);
			else atomicMax(ms.best + i, key);
		}
		}
	}
}

template <typename T>
__global__ void rowstats_minmax_finish_kernel(RowStatsArgs a)
{
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.out_len)
		return;
	MinMaxScratch ms = split_scratch(a.scratch, a.out_len);
	const bool is_dbl = sizeof(T) == 8;
	const bool is_min = a.opcode == SVT_OP_MIN;
	const bool narm = a.na_rm != 0;
	const int fl = ms.flags[i];
	const bool partial = (int64_t) ms.cov[i] < a.nstrata;   // implicit zeros
	bool have = (fl & RF_HAVE) != 0;
	if (is_dbl) {
		double m = have ? ordered_to_f64(ms.best[i]) : 0.0;
		double r;
		if (!narm && (fl & RF_NA)) r = svt_na_real();
		else if (!narm && (fl & RF_NAN)) r = NAN;
		else {
			if (partial) {
				m = have ? (is_min ? (0.0 < m ? 0.0 : m) : (0.0 > m ? 0.0 : m)) : 0.0;
				have = true;
			}
			r = have ? m : (is_min ? INFINITY : -INFINITY);   // :956-957
		}
		((double *) a.out)[i] = r;
	} else {
		int m = have ? (int) ((long long) ms.best[i] - 0x80000000LL) : 0;
		int r;
		if (!narm && (fl & RF_NA)) r = NA_INT;
		else {
			if (partial) {
				m = have ? (is_min ? (0 < m ? 0 : m) : (0 > m ? 0 : m)) : 0;
				have = true;
			}
			if (have) r = m;
			else { r = NA_INT; if (a.warn_flag) *a.warn_flag = 1; }   // :930-931
		}
		((int *) a.out)[i] = r;
	}
}

// --------------------------------------------------------------------------
// Row-panel variant: no memory-side atomics.
//
// The output (inner x nrow cells) is cut into panels of 2048 or 8192 rows; one
// workgroup owns one (output column i, panel q) pair, keeps its cells in LDS
// (ds_add_f64 / ds_min_u64 ...), walks the nstrata leaves j = i + s*inner that
// land on it and stores the finished cells once, coalesced.  The part of leaf j
// inside panel q is a contiguous run of its (sorted) offsets; the run bounds
// come from a table built by one binary search per (leaf, panel boundary)
// (rowpanel_table_kernel).  Traffic: A once (12 B/nz) + the table + out once.
// --------------------------------------------------------------------------
// Panel length: 2048 rows by default (16 bytes of LDS per row serve every operation); the
// sum-like operations on a zero-background operand with many rows take 8192-row panels -- four
// times longer leaf segments (81 instead of 20 nonzeros at BASELINE config 2: whole 128-byte
// lines instead of fragments of them) -- and, when that leaves fewer workgroups than the chip
// has slots, cut the strata into `gridDim.z` ranges whose partial cells meet in `out` through
// memory-side atomics (one coalesced add per cell and range).
#define ROWPANEL_MIN 2048
#define ROWPANEL_SHIFT 11
#define ROWPANEL_BIG_SHIFT 13
#define ROWPANEL_NT 1024

size_t rowstats_panel_ws_bytes(int64_t nrow, int64_t ncol)
{
	const int64_t npan = (nrow + ROWPANEL_MIN - 1) / ROWPANEL_MIN;   // the shortest panels
	return (size_t) (ncol > 0 ? ncol : 1) * (size_t) (npan + 1) * 4 + 64;
}

// pt[q*ncol + j] = number of offsets of leaf j that are < q << ps, q = 0..npan.
// One wavefront per leaf streams its offsets once; the element that is the
// first of its leaf at or past a panel boundary writes that boundary's entry.
__global__ void __launch_bounds__(256)
rowpanel_table_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
		      int64_t ncol, int64_t npan, int ps, int32_t *__restrict__ pt)
{
	// long leaves: the whole workgroup on one leaf; short ones: a wavefront each
	const bool wide = gridDim.x == (unsigned) ncol;
	const int64_t j = wide ? (int64_t) blockIdx.x : (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
	const int tid = wide ? threadIdx.x : (threadIdx.x & 63);
	const int nt = wide ? 256 : SVT_WAVE;
	if (j >= ncol)
		return;
	const int64_t beg = col_ptr[j], end = col_ptr[j + 1];
	for (int64_t k = beg + tid; k < end; k += nt) {
		const int64_t p = row_idx[k] >> ps;
		// previous offset: the lane below holds it, except at the start of a wavefront
		int prev = __shfl_up((int) p, 1, 64);
		if ((threadIdx.x & 63) == 0 || k == beg)
			prev = k == beg ? -1 : row_idx[k - 1] >> ps;
		for (int64_t q = (int64_t) prev + 1; q <= p; q++)
			pt[q * ncol + j] = (int32_t) (k - beg);
	}
	// boundaries past the last offset (all of them for an empty leaf)
	const int64_t pl = end > beg ? row_idx[end - 1] >> ps : -1;
	for (int64_t q = pl + 1 + tid; q <= npan; q += nt)
		pt[q * ncol + j] = (int32_t) (end - beg);
}

// Same table, 16 leaves per workgroup (a wavefront each): the entries are collected in LDS
// and every table row gets one 64-byte run instead of 16 scattered 4-byte stores (the
// scattered form spends 222 us of a 0.65 ms rowSums at BASELINE config 2 on write
// amplification).  LDS: (npan + 1) * 64 bytes.
#define PT_LEAVES 16
#define PT_U 8                  // loads of 64 offsets a wavefront keeps in flight (4: 157 us for the pass at config 3, 8: see DESIGN.md)
// SCAN (the sparse x sparse product, kernels_spmm.hip): the pass also looks at the VALUES of the leaves it walks
// -- all of them (skip == NULL) or those with skip[j] == 0 -- and raises *flag at a NaN / Inf / NA (doubles) or an
// NA_integer_ (ints): one stream over the operand instead of two.  SCAN: 0 none, 1 doubles, 2 ints.
// S wavefronts share a leaf (each a contiguous S-th of its offsets; a workgroup then holds 16 / S leaves): leaves
// are the unit the chip is filled with, and 1e4 long leaves on 8192 wavefront slots are two rounds of which the
// second is a quarter full (BASELINE config 2/3: 101 us; in quarters 5 rounds of a quarter the length).
template <int SCAN, int S>
__global__ void __launch_bounds__(PT_LEAVES * 64)
rowpanel_table16_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
			int64_t ncol, int64_t npan, int ps, int32_t *__restrict__ pt,
			const void *__restrict__ val, const uint8_t *__restrict__ skip, int *__restrict__ flag)
{
	extern __shared__ int32_t tab[];            // [npan + 1][L]
	constexpr int L = PT_LEAVES / S;
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int wl = w / S, sg = w % S;           // leaf of the workgroup, segment of the leaf
	const int64_t j0 = (int64_t) blockIdx.x * L, j = j0 + wl;
	// (with a map of the leaves the product will ask for -- svt %*% svt2, skip[j] != 0 -- the others need no run bounds:
	// their offsets are not read, only their values are looked at below: 0.11 GB less at BASELINE config 3)
	if (j < ncol && (skip == NULL || skip[j] != 0)) {
		const int64_t beg = col_ptr[j], end = col_ptr[j + 1];
		const int64_t sb = beg + (end - beg) * sg / S, se = beg + (end - beg) * (sg + 1) / S;
		int carry = sb > beg ? row_idx[sb - 1] >> ps : -1;     // panel of the element before this trip
		for (int64_t k0 = sb; k0 < se; k0 += PT_U * 64) {
			int32_t r[PT_U];
#pragma unroll
			for (int u = 0; u < PT_U; u++) {    // PT_U coalesced loads in flight
				const int64_t k = k0 + u * 64 + lane;
				r[u] = k < se ? row_idx[k] : 0x7FFFFFFF;
			}
#pragma unroll
			for (int u = 0; u < PT_U; u++) {
				const int64_t k = k0 + u * 64 + lane;
				const int p = r[u] >> ps;
				int prev = __shfl_up(p, 1, 64);
				if (lane == 0) prev = carry;
				carry = __shfl(p, 63, 64);
				if (k < se)
					for (int q = prev + 1; q <= p; q++)
						tab[(int64_t) q * L + wl] = (int32_t) (k - beg);
			}
		}
		if (sg == S - 1) {
			const int64_t pl = end > beg ? row_idx[end - 1] >> ps : -1;
			for (int64_t q = pl + 1 + lane; q <= npan; q += 64)
				tab[q * L + wl] = (int32_t) (end - beg);
		}
	}
	if (SCAN != 0) {
		// the values of the workgroup's leaves that are looked at here, shared out over ALL its wavefronts (a
		// wavefront that scanned its own leaf alone would hold the others back)
		bool bad = false;
		for (int l = 0; l < L && j0 + l < ncol; l++) {
			if (skip != NULL && skip[j0 + l] != 0)
				continue;
			const int64_t b = col_ptr[j0 + l], e = col_ptr[j0 + l + 1];
			for (int64_t k = b + threadIdx.x; k < e; k += 4 * PT_LEAVES * 64) {
				if (SCAN == 1) {
					double x[4];
#pragma unroll
					for (int u = 0; u < 4; u++)
						x[u] = k + u * PT_LEAVES * 64 < e ? ((const double *) val)[k + u * PT_LEAVES * 64] : 0.0;
#pragma unroll
					for (int u = 0; u < 4; u++) bad |= !(fabs(x[u]) <= 1.7976931348623157e308);
				} else {
#pragma unroll
					for (int u = 0; u < 4; u++)
						bad |= k + u * PT_LEAVES * 64 < e && ((const int *) val)[k + u * PT_LEAVES * 64] == NA_INT;
				}
			}
		}
		if (__ballot(bad) != 0 && lane == 0) *flag = 1;
	}
	__syncthreads();
	const int64_t n = (npan + 1) * L;
	for (int64_t t = threadIdx.x; t < n; t += PT_LEAVES * 64) {
		const int64_t q = t / L, l = t % L;
		if (j0 + l < ncol && (skip == NULL || skip[j0 + l] != 0))
			pt[q * ncol + j0 + l] = tab[t];
	}
}

// G = lanes that share one leaf segment (power of two <= 64): the host picks
// it from the mean segment length so that short segments still fill the wave.
template <typename T>
__global__ void __launch_bounds__(ROWPANEL_NT)
rowstats_panel_kernel(RowStatsArgs a, const int32_t *__restrict__ pt, int64_t npan, int G, int ps)
{
	extern __shared__ unsigned long long lds64[];   // 1 << ps cells
	const int64_t q = blockIdx.x, i = blockIdx.y;
	const int tid = threadIdx.x, NT = blockDim.x;
	const int prow = 1 << ps;
	const int64_t r0 = q * prow;
	const int np = (int) (a.nrow - r0 < prow ? a.nrow - r0 : prow);   // rows in this panel
	// strata range of this workgroup (gridDim.z > 1: sum-like operations only, `out` zeroed)
	const bool split = gridDim.z > 1;
	const int64_t s_chunk = (a.nstrata + gridDim.z - 1) / gridDim.z;
	const int64_t s_lo = (int64_t) blockIdx.z * s_chunk;
	const int64_t s_hi = s_lo + s_chunk < a.nstrata ? s_lo + s_chunk : a.nstrata;
	const int64_t cell0 = i * a.nrow + r0;
	const bool is_dbl = sizeof(T) == 8;
	const bool narm = a.na_rm != 0;
	const int oc = a.opcode;
	const bool is_minmax = oc == SVT_OP_MIN || oc == SVT_OP_MAX;
	double *accd = (double *) lds64;
	double *cen = accd + prow;                           // centered_X2_sum only
	int *flg = (int *) (lds64 + prow);                   // min/max only
	unsigned int *cov = (unsigned int *) (flg + prow);
	const T *__restrict__ val = (const T *) a.val;
	const int32_t *__restrict__ row = a.row_idx;
	const double NAr = svt_na_real();
	// NaArray (a.na_bg): cells covered fewer than nstrata times hold implicit NAs:
	//   countNAs / anyNA: count the stored non-NA values, result nstrata - count (!= 0)
	//   sum, na.rm=FALSE: NA_real_ wherever the coverage is short (:612-634)
	//   min / max: the NA background joins instead of the implicit zero (:914-961)
	const bool nabg = a.na_bg != 0;

	for (int r = tid; r < np; r += NT) {
		if (is_minmax) {
			lds64[r] = oc == SVT_OP_MIN ? ~0ULL : 0ULL;
			flg[r] = 0;
			cov[r] = 0;
		} else if (oc == SVT_OP_CENTERED_X2_SUM) {
			const double c = a.center ? a.center[cell0 + r] : 0.0;
			cen[r] = c;
			accd[r] = a.center && blockIdx.z == 0 ? c * c * (double) a.nstrata : 0.0;
		} else if (oc == SVT_OP_ANYNA && !nabg) {
			((int *) lds64)[r] = 0;
		} else {
			accd[r] = 0.0;
			if (nabg) cov[r] = 0;
		}
	}
	__syncthreads();
	const int sub = tid / G, sl = tid % G, nsub = NT / G;
	const int32_t *__restrict__ pt0 = pt + q * a.ncol, *__restrict__ pt1 = pt0 + a.ncol;
	// one nonzero into its cell
	auto apply = [&](const T v, const int r) {
		const bool miss = is_dbl ? (v != v) : ((int) v == NA_INT);
		switch (oc) {
		case SVT_OP_ANYNA:
			if (nabg) { if (!miss) atomicAdd(accd + r, 1.0); }
			else if (miss) ((int *) lds64)[r] = 1;
			break;
		case SVT_OP_COUNTNAS:
			if (nabg ? !miss : miss) atomicAdd(accd + r, 1.0);
			break;
		case SVT_OP_SUM:
			if (nabg && !narm) atomicAdd(cov + r, 1u);
			if (miss && narm) break;
			atomicAdd(accd + r, (!is_dbl && miss) ? NAr : (double) v);
			break;
		case SVT_OP_CENTERED_X2_SUM: {
			const double c = cen[r];
			if (miss && narm) { atomicAdd(accd + r, -(c * c)); break; }
			const double x = (!is_dbl && miss) ? NAr : (double) v;
			atomicAdd(accd + r, x * (x - 2 * c));
			break;
		}
		default: {
			atomicAdd(cov + r, 1u);
			if (miss) {
				const bool isna = is_dbl ? svt_is_na((double) v) : true;
				atomicOr(flg + r, isna ? RF_NA : RF_NAN);
				break;
			}
			atomicOr(flg + r, RF_HAVE);
			const unsigned long long key = is_dbl ?
				f64_to_ordered((double) v) :
				(unsigned long long) ((long long) (int) v + 0x80000000LL);
			if (oc == SVT_OP_MIN) atomicMin(lds64 + r, key);
			else atomicMax(lds64 + r, key);
		}
		}
	};
	// RS_U leaf segments per lane group at a time, their loads in flight together (one segment
	// after the other leaves a wavefront with a single load pair outstanding: 2.1 TB/s at
	// BASELINE config 2); the next round's bounds are fetched a round ahead.
	constexpr int RS_U = 4;
	int64_t nb[RS_U], ne[RS_U];
	auto bounds = [&](const int64_t s0) {
#pragma unroll
		for (int u = 0; u < RS_U; u++) {
			const int64_t s = s0 + (int64_t) u * nsub;
			nb[u] = ne[u] = 0;
			if (s < s_hi) {
				const int64_t j = i + s * a.inner;
				const int64_t base = a.col_ptr[j];
				nb[u] = base + pt0[j] + sl; ne[u] = base + pt1[j];
			}
		}
	};
	bounds(s_lo + sub);
	for (int64_t s = s_lo + sub; s < s_hi; s += (int64_t) nsub * RS_U) {
		int64_t kb[RS_U], ke[RS_U];
#pragma unroll
		for (int u = 0; u < RS_U; u++) { kb[u] = nb[u]; ke[u] = ne[u]; }
		if (s + (int64_t) nsub * RS_U < s_hi)
			bounds(s + (int64_t) nsub * RS_U);
		bool more = true;
		while (more) {
			T v[RS_U];
			int r[RS_U];
#pragma unroll
			for (int u = 0; u < RS_U; u++)
				if (kb[u] < ke[u]) { v[u] = val[kb[u]]; r[u] = (int) (row[kb[u]] - r0); }
			more = false;
#pragma unroll
			for (int u = 0; u < RS_U; u++)
				if (kb[u] < ke[u]) {
					apply(v[u], r[u]);
					kb[u] += G;
					more |= kb[u] < ke[u];
				}
		}
	}
	__syncthreads();
	if (split) {                                 // partial cells of this strata range
		for (int r = tid; r < np; r += NT) {
			if (oc == SVT_OP_ANYNA) {
				if (((int *) lds64)[r]) ((int *) a.out)[cell0 + r] = 1;
			} else {
				atomicAdd((double *) a.out + cell0 + r, accd[r]);
			}
		}
		return;
	}
	for (int r = tid; r < np; r += NT) {
		const int64_t cell = cell0 + r;
		if (!is_minmax) {
			if (nabg && (oc == SVT_OP_ANYNA || oc == SVT_OP_COUNTNAS)) {
				const double nas = (double) a.nstrata - accd[r];
				if (oc == SVT_OP_ANYNA) ((int *) a.out)[cell] = nas != 0.0;
				else ((double *) a.out)[cell] = nas;
			} else if (oc == SVT_OP_ANYNA) {
				((int *) a.out)[cell] = ((int *) lds64)[r];
			} else if (nabg && oc == SVT_OP_SUM && !narm && (int64_t) cov[r] < a.nstrata) {
				((double *) a.out)[cell] = NAr;
			} else {
				((double *) a.out)[cell] = accd[r];
			}
			continue;
		}
		// NA > NaN > extremum; the implicit zero joins when the cell was
		// covered fewer than nstrata times (:914-961)
		const bool is_min = oc == SVT_OP_MIN;
		const int fl = flg[r];
		const bool partial = (int64_t) cov[r] < a.nstrata;
		bool have = (fl & RF_HAVE) != 0;
		if (is_dbl) {
			double m = have ? ordered_to_f64(lds64[r]) : 0.0, res;
			if (!narm && ((fl & RF_NA) || (nabg && partial))) res = NAr;
			else if (!narm && (fl & RF_NAN)) res = NAN;
			else {
				if (partial && !nabg) {
					m = have ? (is_min ? (0.0 < m ? 0.0 : m) : (0.0 > m ? 0.0 : m)) : 0.0;
					have = true;
				}
				res = have ? m : (is_min ? INFINITY : -INFINITY);
			}
			((double *) a.out)[cell] = res;
		} else {
			int m = have ? (int) ((long long) lds64[r] - 0x80000000LL) : 0, res;
			if (!narm && ((fl & RF_NA) || (nabg && partial))) res = NA_INT;
			else {
				if (partial && !nabg) {
					m = have ? (is_min ? (0 < m ? 0 : m) : (0 > m ? 0 : m)) : 0;
					have = true;
				}
				if (have) res = m;
				else { res = NA_INT; if (a.warn_flag) *a.warn_flag = 1; }
			}
			((int *) a.out)[cell] = res;
		}
	}
}

// Whole-column variant: many output columns of few, short leaves each and a first dimension that fits LDS
// (rowSums(x, dims = 2) of BASELINE config 5: 2e4 output columns x 64 leaves of ~100 nonzeros, 2e4 rows).
// One workgroup per output column keeps ALL its cells in LDS (nrow * 8 bytes) and reads its leaves whole:
// no row panels, no table pass over the offsets (0.41 ms of the 2.0 ms there), no segment bounds.
// Sum-like operations on a zero-background operand.
template <typename T>
__global__ void __launch_bounds__(ROWPANEL_NT)
rowstats_whole_kernel(RowStatsArgs a, int G)
{
	extern __shared__ unsigned long long lds64[];   // nrow cells (+ nrow centers)
	const int64_t i = blockIdx.x;
	const int tid = threadIdx.x, NT = blockDim.x;
	const int np = (int) a.nrow;
	const bool is_dbl = sizeof(T) == 8;
	const bool narm = a.na_rm != 0;
	const int oc = a.opcode;
	double *accd = (double *) lds64;
	double *cen = accd + np;                             // centered_X2_sum only
	const T *__restrict__ val = (const T *) a.val;
	const int32_t *__restrict__ row = a.row_idx;
	const double NAr = svt_na_real();
	const int64_t cell0 = i * a.nrow;
	for (int r = tid; r < np; r += NT) {
		if (oc == SVT_OP_CENTERED_X2_SUM) {
			const double c = a.center ? a.center[cell0 + r] : 0.0;
			cen[r] = c;
			accd[r] = a.center ? c * c * (double) a.nstrata : 0.0;
		} else if (oc == SVT_OP_ANYNA) {
			((int *) lds64)[r] = 0;
		} else {
			accd[r] = 0.0;
		}
	}
	__syncthreads();
	auto apply = [&](const T v, const int r) {
		const bool miss = is_dbl ? (v != v) : ((int) v == NA_INT);
		switch (oc) {
		case SVT_OP_ANYNA:
			if (miss) ((int *) lds64)[r] = 1;
			break;
		case SVT_OP_COUNTNAS:
			if (miss) atomicAdd(accd + r, 1.0);
			break;
		case SVT_OP_SUM:
			if (miss && narm) break;
			atomicAdd(accd + r, (!is_dbl && miss) ? NAr : (double) v);
			break;
		default: {                                   // centered_X2_sum
			const double c = cen[r];
			if (miss && narm) { atomicAdd(accd + r, -(c * c)); break; }
			const double x = (!is_dbl && miss) ? NAr : (double) v;
			atomicAdd(accd + r, x * (x - 2 * c));
		}
		}
	};
	// four leaves per lane group at a time (see rowstats_panel_kernel)
	constexpr int RS_U = 4;
	const int sub = tid / G, sl = tid % G, nsub = NT / G;
	for (int64_t s0 = sub; s0 < a.nstrata; s0 += (int64_t) nsub * RS_U) {
		int64_t kb[RS_U], ke[RS_U];
#pragma unroll
		for (int u = 0; u < RS_U; u++) {
			const int64_t s = s0 + (int64_t) u * nsub;
			kb[u] = ke[u] = 0;
			if (s < a.nstrata) {
				const int64_t j = i + s * a.inner;
				kb[u] = a.col_ptr[j] + sl; ke[u] = a.col_ptr[j + 1];
			}
		}
		bool more = true;
		while (more) {
			T v[RS_U];
			int r[RS_U];
#pragma unroll
			for (int u = 0; u < RS_U; u++)
				if (kb[u] < ke[u]) { v[u] = val[kb[u]]; r[u] = (int) row[kb[u]]; }
			more = false;
#pragma unroll
			for (int u = 0; u < RS_U; u++)
				if (kb[u] < ke[u]) {
					apply(v[u], r[u]);
					kb[u] += G;
					more |= kb[u] < ke[u];
				}
		}
	}
	__syncthreads();
	for (int r = tid; r < np; r += NT) {
		if (oc == SVT_OP_ANYNA) ((int *) a.out)[cell0 + r] = ((int *) lds64)[r];
		else ((double *) a.out)[cell0 + r] = accd[r];
	}
}

// The whole-column form as a persistent, pipelined loop (sums and NA counts of doubles / ints; at most 64 leaves of at
// most ~128 nonzeros per output column -- rowSums(x, dims = 2) of BASELINE config 5): a workgroup owns a CU (160 KB of
// LDS = its column's 2e4 cells) and with one workgroup per column nothing of the next column is in flight while the
// current one leaves (zero the cells -> read 64 leaves -> write 160 KB, strictly in a row: 14.5 us per column, 1.13 ms).
// Here a workgroup walks columns i, i + grid, ...: the leaf bounds and the (value, row) pairs of the NEXT column are
// loaded into registers before the current column's cells are written out (and zeroed again in the same sweep).
// ATOMIC: an output column has more leaves than one unit takes (64): its leaves are cut into chunks of 64, a workgroup takes a
// contiguous range of (column, chunk) units, keeps adding into its LDS cells while the column stays the same and adds the cells
// to `out` (zeroed by the launcher) when it changes -- rowSums(x, dims = 1) of an N-d array: one column, 1.28e6 leaves at config 5.
template <typename T, bool ATOMIC>
__global__ void __launch_bounds__(ROWPANEL_NT)
rowstats_whole_pipe_kernel(RowStatsArgs a, int64_t nchunks)
{
	extern __shared__ unsigned long long lds64[];   // nrow cells
	constexpr int U = 4, TT = 2;                    // leaves per wavefront, trips of 64 lanes fetched ahead
	const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, w = tid >> 6, nw = NT >> 6;
	const int np = (int) a.nrow;
	const bool is_dbl = sizeof(T) == 8;
	const bool narm = a.na_rm != 0;
	const int oc = a.opcode;
	double *accd = (double *) lds64;
	const T *__restrict__ val = (const T *) a.val;
	const int32_t *__restrict__ row = a.row_idx;
	const double NAr = svt_na_real();
	auto apply = [&](const T v, const int r) {
		const bool miss = is_dbl ? (v != v) : ((int) v == NA_INT);
		if (oc == SVT_OP_COUNTNAS) {
			if (miss) atomicAdd(accd + r, 1.0);
		} else if (!(miss && narm)) {
			atomicAdd(accd + r, (!is_dbl && miss) ? NAr : (double) v);
		}
	};
	int64_t kb[U], ke[U];
	T v[U][TT];
	int r[U][TT];
	auto fetch = [&](const int64_t unit) {
		const int64_t i = unit / nchunks, c = unit % nchunks;
#pragma unroll
		for (int u = 0; u < U; u++) {
			const int64_t s = c * (int64_t) (U * nw) + w + (int64_t) u * nw;
			kb[u] = ke[u] = 0;
			if (s < a.nstrata) {
				const int64_t j = i + s * a.inner;
				kb[u] = a.col_ptr[j] + lane; ke[u] = a.col_ptr[j + 1];
			}
		}
#pragma unroll
		for (int u = 0; u < U; u++)
#pragma unroll
			for (int t = 0; t < TT; t++) {
				const int64_t k = kb[u] + 64 * t;
				if (k < ke[u]) { v[u][t] = val[k]; r[u][t] = (int) row[k]; }
			}
	};
	const int64_t total = a.inner * nchunks, per = (total + gridDim.x - 1) / gridDim.x;
	const int64_t u0 = (int64_t) blockIdx.x * per, u1 = u0 + per < total ? u0 + per : total;
	if (u0 < u1) fetch(u0);
	for (int x = tid; x < np; x += NT) accd[x] = 0.0;
	__syncthreads();
	for (int64_t unit = u0; unit < u1; unit++) {
#pragma unroll
		for (int u = 0; u < U; u++) {
#pragma unroll
			for (int t = 0; t < TT; t++)
				if (kb[u] + 64 * t < ke[u]) apply(v[u][t], r[u][t]);
			for (int64_t k = kb[u] + 64 * TT; k < ke[u]; k += 64) apply(val[k], (int) row[k]);   // (leaves past 128 nonzeros)
		}
		const int64_t i = unit / nchunks;
		const bool flush = unit + 1 == u1 || (unit + 1) / nchunks != i;
		if (flush) __syncthreads();
		if (unit + 1 < u1) fetch(unit + 1);                             // in flight while this column leaves
		if (flush) {
			const int64_t cell0 = i * a.nrow;
			for (int x = tid; x < np; x += NT) {
				const double c = accd[x];
				if (ATOMIC) { if (c != 0.0) atomicAdd((double *) a.out + cell0 + x, c); }
				else ((double *) a.out)[cell0 + x] = c;
				accd[x] = 0.0;
			}
			__syncthreads();
		}
	}
}

// The same table for SHORT leaves (mean < 256 offsets: the 1.28e6 leaves of ~100 offsets of BASELINE config 5): 16 lanes per
// leaf, 64 leaves per workgroup -- a wavefront per leaf leaves three quarters of its lanes idle and its table rows leave in
// 64-byte pieces (0.45 ms of a 1.12 ms rowSums(dims = 2) at config 5); here a row of the LDS image is 256 bytes.
#define PTS_LEAVES 64
__global__ void __launch_bounds__(1024)
rowpanel_table_short_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
			    int64_t ncol, int64_t npan, int ps, int32_t *__restrict__ pt)
{
	extern __shared__ int32_t tab[];            // [npan + 1][PTS_LEAVES]
	const int g = threadIdx.x >> 4, sl = threadIdx.x & 15;
	const int64_t j0 = (int64_t) blockIdx.x * PTS_LEAVES, j = j0 + g;
	if (j < ncol) {
		const int64_t beg = col_ptr[j], end = col_ptr[j + 1];
		int carry = -1;                         // panel of the element before this trip
		for (int64_t k0 = beg; k0 < end; k0 += 4 * 16) {
			int32_t r[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int64_t k = k0 + u * 16 + sl;
				r[u] = k < end ? row_idx[k] : 0x7FFFFFFF;
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int64_t k = k0 + u * 16 + sl;
				const int p = r[u] >> ps;
				int prev = __shfl_up(p, 1, 16);
				if (sl == 0) prev = carry;
				carry = __shfl(p, 15, 16);
				if (k < end)
					for (int q = prev + 1; q <= p; q++)
						tab[q * PTS_LEAVES + g] = (int32_t) (k - beg);
			}
		}
		const int64_t pl = end > beg ? row_idx[end - 1] >> ps : -1;
		for (int64_t q = pl + 1 + sl; q <= npan; q += 16)
			tab[q * PTS_LEAVES + g] = (int32_t) (end - beg);
	}
	__syncthreads();
	const int64_t n = (npan + 1) * PTS_LEAVES;
	for (int64_t t = threadIdx.x; t < n; t += 1024) {
		const int64_t q = t / PTS_LEAVES, l = t % PTS_LEAVES;
		if (j0 + l < ncol)
			pt[q * ncol + j0 + l] = tab[t];
	}
}

// wavefronts per leaf of rowpanel_table16_kernel: the split (1, 2 or 4) with the fewest leaf-times of rounds on 8192
// wavefront slots; short leaves are not split
static int rowpanel_split(int64_t ncol, int64_t nnz_hint)
{
	if (ncol <= 0 || nnz_hint / ncol < 2048)
		return 1;
	int best = 1;
	double best_t = 1e30;
	for (int sp = 1; sp <= 4; sp *= 2) {
		const int64_t nwg = (ncol * sp + PT_LEAVES - 1) / PT_LEAVES, rounds = (nwg + 511) / 512;
		const double t = (double) rounds / sp;
		if (t < best_t - 1e-9) { best_t = t; best = sp; }
	}
	return best;
}

template <int SCAN>
static void launch_table16(const int64_t *col_ptr, const int32_t *row_idx, int64_t ncol, int64_t nnz_hint, int64_t npan,
			   int ps, int32_t *pt, const void *val, const uint8_t *skip, int *flag, hipStream_t s)
{
	const int sp = rowpanel_split(ncol, nnz_hint);
	const int L = PT_LEAVES / sp;
	const dim3 grid((unsigned) ((ncol + L - 1) / L));
	const size_t lds = (size_t) (npan + 1) * L * 4;
	if (sp == 4)
		hipLaunchKernelGGL((rowpanel_table16_kernel<SCAN, 4>), grid, dim3(PT_LEAVES * 64), lds, s, col_ptr, row_idx,
				   ncol, npan, ps, pt, val, skip, flag);
	else if (sp == 2)
		hipLaunchKernelGGL((rowpanel_table16_kernel<SCAN, 2>), grid, dim3(PT_LEAVES * 64), lds, s, col_ptr, row_idx,
				   ncol, npan, ps, pt, val, skip, flag);
	else
		hipLaunchKernelGGL((rowpanel_table16_kernel<SCAN, 1>), grid, dim3(PT_LEAVES * 64), lds, s, col_ptr, row_idx,
				   ncol, npan, ps, pt, val, skip, flag);
}

// pt[q * ncol + j] = number of offsets of leaf j below q << ps, q = 0 .. npan ((npan + 1) * ncol entries)
void launch_rowpanel_table(const int64_t *col_ptr, const int32_t *row_idx, int64_t ncol, int64_t nnz_hint,
			   int64_t npan, int ps, int32_t *pt, hipStream_t s)
{
	if (ncol > 0 && nnz_hint / ncol < 256 && (size_t) (npan + 1) * PTS_LEAVES * 4 <= 64 * 1024) {
		hipLaunchKernelGGL(rowpanel_table_short_kernel, dim3((unsigned) ((ncol + PTS_LEAVES - 1) / PTS_LEAVES)), dim3(1024),
				   (size_t) (npan + 1) * PTS_LEAVES * 4, s, col_ptr, row_idx, ncol, npan, ps, pt);
	} else if (ncol > 0 && (size_t) (npan + 1) * PT_LEAVES * 4 <= 64 * 1024) {
		launch_table16<0>(col_ptr, row_idx, ncol, nnz_hint, npan, ps, pt, NULL, NULL, NULL, s);
	} else if (ncol > 0) {                      // very tall arrays: the table rows do not fit LDS
		// (grid == ncol selects the workgroup-per-leaf form; never equal to (ncol+3)/4 for ncol > 1)
		const bool wide = nnz_hint / ncol >= 1024 && ncol > 1;
		hipLaunchKernelGGL(rowpanel_table_kernel, dim3((unsigned) (wide ? ncol : (ncol + 3) / 4)),
				   dim3(256), 0, s, col_ptr, row_idx, ncol, npan, ps, pt);
	}
}

// The same pass, looking at the values of the leaves with skip[j] == 0 (all leaves: skip == NULL) on its way:
// *flag = 1 at a non-finite double / an NA.  Returns false when the shape takes the plain table kernel (the
// caller then scans the values in a pass of its own).
bool launch_rowpanel_table_scan(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
				int64_t ncol, int64_t nnz_hint, int64_t npan, int ps, int32_t *pt,
				const uint8_t *skip, int *flag, hipStream_t s)
{
	if (ncol > 0 && (size_t) (npan + 1) * PT_LEAVES * 4 <= 64 * 1024) {
		if (Rtype == SVT_REALSXP)
			launch_table16<1>(col_ptr, row_idx, ncol, nnz_hint, npan, ps, pt, val, skip, flag, s);
		else
			launch_table16<2>(col_ptr, row_idx, ncol, nnz_hint, npan, ps, pt, val, skip, flag, s);
		return true;
	}
	launch_rowpanel_table(col_ptr, row_idx, ncol, nnz_hint, npan, ps, pt, s);
	return false;
}

// `ws`: rowstats_panel_ws_bytes() bytes.
int launch_rowstats_panel(const RowStatsArgs &a, void *ws, hipStream_t s)
{
	if (a.out_len <= 0)
		return 0;
	const int oc = a.opcode;
	const bool sumlike = oc == SVT_OP_SUM || oc == SVT_OP_COUNTNAS || oc == SVT_OP_CENTERED_X2_SUM ||
		oc == SVT_OP_ANYNA;
	// output columns of MANY short leaves, all rows in LDS (rowSums(x, dims = 1) of an N-d array, a 2-d operand of at most
	// 20480 rows and short columns): the persistent whole-column kernel over (column, chunk of 64 leaves) units, cells added
	// to a zeroed `out` when a workgroup's column changes
	{
		const size_t lds_w = (size_t) a.nrow * 8;
		const double leaf_len = a.ncol > 0 ? (double) a.nnz_hint / (double) a.ncol : 0.0;
		const int64_t spu = 4 * (ROWPANEL_NT / 64), nchunks = (a.nstrata + spu - 1) / spu;
		if ((oc == SVT_OP_SUM || oc == SVT_OP_COUNTNAS) && !a.na_bg && a.nnz_hint > 0 && a.nstrata > spu &&
		    a.nrow > (1 << ROWPANEL_BIG_SHIFT) && lds_w <= 160 * 1024 && leaf_len <= 112.0 && leaf_len >= 8.0 &&
		    a.inner * nchunks >= 2048) {
			if (a.table_mode == 1)
				return 0;                       // (this form needs no table)
			HIP_TRY(hipMemsetAsync(a.out, 0, (size_t) a.out_len * 8, s));
			const void *fp = a.Rtype == SVT_REALSXP ? (const void *) rowstats_whole_pipe_kernel<double, true>
								: (const void *) rowstats_whole_pipe_kernel<int, true>;
			(void) hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_w);
			if (a.Rtype == SVT_REALSXP)
				hipLaunchKernelGGL((rowstats_whole_pipe_kernel<double, true>), dim3(256), dim3(ROWPANEL_NT), lds_w, s, a, nchunks);
			else
				hipLaunchKernelGGL((rowstats_whole_pipe_kernel<int, true>), dim3(256), dim3(ROWPANEL_NT), lds_w, s, a, nchunks);
			HIP_TRY(hipGetLastError());
			return 0;
		}
	}
	// many output columns of few short leaves, all rows in LDS: the whole-column kernel
	{
		const size_t lds_whole = (size_t) a.nrow * (oc == SVT_OP_CENTERED_X2_SUM ? 16 : 8);
		const double leaf_len = a.ncol > 0 ? (double) a.nnz_hint / (double) a.ncol : 0.0;
		if (sumlike && !a.na_bg && a.nnz_hint > 0 && a.inner >= 1024 && a.nrow > (1 << ROWPANEL_BIG_SHIFT) &&
		    lds_whole <= 160 * 1024 && leaf_len <= 512.0 && a.nstrata * leaf_len <= 65536.0) {
			if (a.table_mode == 1)
				return 0;                       // (this form needs no table)
			int G = 64;
			while (G > 8 && leaf_len <= G / 2) G >>= 1;
			if (G == 64 && ((int64_t) (leaf_len + 31.0) / 32) * 32 < ((int64_t) (leaf_len + 63.0) / 64) * 64) G = 32;
			const void *fn = a.Rtype == SVT_REALSXP ? (const void *) rowstats_whole_kernel<double>
								: (const void *) rowstats_whole_kernel<int>;
			(void) hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_whole);
			if ((oc == SVT_OP_SUM || oc == SVT_OP_COUNTNAS) && a.nstrata <= 4 * (ROWPANEL_NT / 64) && leaf_len <= 112.0 &&
			    a.inner >= 1024) {
				const void *fp = a.Rtype == SVT_REALSXP ? (const void *) rowstats_whole_pipe_kernel<double, false>
									: (const void *) rowstats_whole_pipe_kernel<int, false>;
				(void) hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_whole);
				const unsigned ng = (unsigned) (a.inner < 256 ? a.inner : 256);       // one workgroup per CU
				if (a.Rtype == SVT_REALSXP)
					hipLaunchKernelGGL((rowstats_whole_pipe_kernel<double, false>), dim3(ng), dim3(ROWPANEL_NT), lds_whole, s, a, (int64_t) 1);
				else
					hipLaunchKernelGGL((rowstats_whole_pipe_kernel<int, false>), dim3(ng), dim3(ROWPANEL_NT), lds_whole, s, a, (int64_t) 1);
				HIP_TRY(hipGetLastError());
				return 0;
			}
			if (a.Rtype == SVT_REALSXP)
				hipLaunchKernelGGL(rowstats_whole_kernel<double>, dim3((unsigned) a.inner), dim3(ROWPANEL_NT),
						   lds_whole, s, a, G);
			else
				hipLaunchKernelGGL(rowstats_whole_kernel<int>, dim3((unsigned) a.inner), dim3(ROWPANEL_NT),
						   lds_whole, s, a, G);
			HIP_TRY(hipGetLastError());
			return 0;
		}
	}
	const bool big = sumlike && !a.na_bg && a.nrow >= (2LL << ROWPANEL_BIG_SHIFT);
	const int ps = big ? ROWPANEL_BIG_SHIFT : ROWPANEL_SHIFT;
	const int64_t prow = 1LL << ps;
	const int64_t npan = (a.nrow + prow - 1) / prow;
	int32_t *pt = (int32_t *) ws;
	if (a.inner > 65535)
		return svt_set_error("row stats: more than 65535 output columns per panel row");
	if (a.table_mode != 2)
		launch_rowpanel_table(a.col_ptr, a.row_idx, a.ncol, a.nnz_hint, npan, ps, pt, s);
	if (a.table_mode == 1) {
		HIP_TRY(hipGetLastError());
		return 0;
	}
	// lanes per leaf segment ~ mean segment length (nnz unknown here: the
	// caller passes it in a.nnz_hint, 0 = assume long segments)
	int G = 64;
	if (a.nnz_hint > 0 && a.ncol > 0) {
		const double seg = (double) a.nnz_hint / ((double) a.ncol * (double) npan);
		while (G > 8 && seg <= G / 2) G >>= 1;
		// long segments: 32 lanes where they waste fewer load slots than 64 (81 nonzeros: three trips of
		// 32 = 96 slots instead of two of 64 = 128; 0.45 -> 0.39 ms at BASELINE config 2)
		if (G == 64 && ((int64_t) (seg + 31.0) / 32) * 32 < ((int64_t) (seg + 63.0) / 64) * 64) G = 32;
	}
	// strata ranges: aim at two workgroups per CU when the panels alone are fewer
	int64_t nsplit = 1;
	if (big) {
		nsplit = (2 * 256 + npan * a.inner / 2) / (npan * a.inner);
		const int64_t cap = a.nstrata / (4 * (ROWPANEL_NT / G));      // >= 4 segments per lane group
		if (nsplit > cap) nsplit = cap;
		if (nsplit > 1024) nsplit = 1024;
		if (nsplit < 1) nsplit = 1;
	}
	const int NT = ROWPANEL_NT;
	const bool centered = oc == SVT_OP_CENTERED_X2_SUM;
	const size_t lds = sumlike && !a.na_bg ? (size_t) prow * (centered ? 16 : 8) : (size_t) prow * 16;
	if (nsplit > 1)
		HIP_TRY(hipMemsetAsync(a.out, 0, (size_t) a.out_len * (oc == SVT_OP_ANYNA ? 4 : 8), s));
	dim3 grid((unsigned) npan, (unsigned) a.inner, (unsigned) nsplit);
	if (lds > 64 * 1024)
		(void) hipFuncSetAttribute(a.Rtype == SVT_REALSXP ? (const void *) rowstats_panel_kernel<double> :
					   (const void *) rowstats_panel_kernel<int>,
					   hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	if (a.Rtype == SVT_REALSXP)
		hipLaunchKernelGGL(rowstats_panel_kernel<double>, grid, dim3(NT), lds, s, a, pt, npan, G, ps);
	else
		hipLaunchKernelGGL(rowstats_panel_kernel<int>, grid, dim3(NT), lds, s, a, pt, npan, G, ps);
	HIP_TRY(hipGetLastError());
	return 0;
}

int launch_rowstats(const RowStatsArgs &a, int64_t nnz, hipStream_t s)
{
	(void) nnz;
	if (a.out_len <= 0)
		return 0;
	const bool is_dbl = a.Rtype == SVT_REALSXP;
	const unsigned nb_out = (unsigned) ((a.out_len + 255) / 256);
	hipLaunchKernelGGL(rowstats_init_kernel, dim3(nb_out), dim3(256), 0, s, a);
	if (a.ncol > 0 && a.nstrata > 0) {
		const unsigned nb = (unsigned) ((a.ncol + 3) / 4);
		if (is_dbl) hipLaunchKernelGGL(rowstats_scatter_kernel<double>, dim3(nb), dim3(256), 0, s, a);
		else hipLaunchKernelGGL(rowstats_scatter_kernel<int>, dim3(nb), dim3(256), 0, s, a);
	}
	if (a.opcode == SVT_OP_MIN || a.opcode == SVT_OP_MAX) {
		if (is_dbl) hipLaunchKernelGGL(rowstats_minmax_finish_kernel<double>, dim3(nb_out), dim3(256), 0, s, a);
		else hipLaunchKernelGGL(rowstats_minmax_finish_kernel<int>, dim3(nb_out), dim3(256), 0, s, a);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// --------------------------------------------------------------------------
// rowsum / colsum
// --------------------------------------------------------------------------
// int32 results.  The reference adds one value at a time with safe_int_add() /
// add_sparse_vec_to_ints() (src/rowsum_methods.c:66-84, 166-199): a cell becomes NA at the first
// NA value (na.rm = FALSE) or at the first running sum outside [-INT_MAX, INT_MAX] -- only the
// latter raises the overflow warning -- and stays NA.  On the device every cell gets its exact
// total (i64), the sum of |values| and an "NA seen" flag in one parallel pass.  If the sum of
// |values| fits, no running sum can leave the range whatever the order, and the total is the
// reference's result.  Otherwise the cell's output column is redone by one thread in the
// reference's order (groupsum_int_exact_kernel): rare (the data sit at the edge of int32), exact.
struct IntSumScratch {
	long long *sum;
	unsigned long long *abs;
	int *na;
	int *redo;               // [number of output columns]
};
__host__ __device__ inline IntSumScratch split_int_scratch(void *p, int64_t n)
{
	IntSumScratch s;
	s.sum = (long long *) p;
	s.abs = (unsigned long long *) (s.sum + n);
	s.na = (int *) (s.abs + n);
	s.redo = s.na + n;
	return s;
}
size_t groupsum_scratch_bytes(int Rtype, int64_t out_len)
{
	// (at most out_len output columns)
	return Rtype == SVT_REALSXP ? 16 : (size_t) out_len * 24 + 16;
}

__device__ inline int64_t col_beg(const GroupSumArgs &a, int64_t j)
{
	return a.col_ptr64 ? a.col_ptr64[j] : (int64_t) a.col_ptr32[j];
}

// One workgroup per column, group accumulators in LDS (ds_add_f64), one
// coalesced store of the finished column: compute_rowsum_doubles,
// src/rowsum_methods.c:44-64.
__global__ void __launch_bounds__(256)
rowsum_f64_lds_kernel(GroupSumArgs a)
{
	extern __shared__ double acc[];
	const int64_t j = blockIdx.x;
	for (int g = threadIdx.x; g < a.ngroup; g += blockDim.x)
		acc[g] = 0.0;
	__syncthreads();
	const double *__restrict__ val = (const double *) a.val;
	const int64_t beg = col_beg(a, j), end = col_beg(a, j + 1);
	for (int64_t k = beg + threadIdx.x; k < end; k += blockDim.x) {
		const double v = val[k];
		if (a.na_rm && v != v)
			continue;
		int g = a.group[a.row_idx[k]];
		if (g == NA_INT) g = a.ngroup;
		atomicAdd(&acc[g - 1], v);
	}
	__syncthreads();
	double *out = (double *) a.out + j * (int64_t) a.ngroup;
	for (int g = threadIdx.x; g < a.ngroup; g += blockDim.x)
		out[g] = acc[g];
}

// One wavefront per column, memory-side atomics (short columns or too many
// groups for LDS).  TARGET 0: rowsum -> out[g-1 + j*ngroup];
// TARGET 1: colsum -> out[row + (group[j]-1)*nrow]  (:141-199, :204-255)
template <typename T, int TARGET>
__global__ void __launch_bounds__(256)
groupsum_atomic_kernel(GroupSumArgs a, int64_t out_len)
{
	const int lane = threadIdx.x & 63;
	const int64_t j = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (j >= a.ncol)
		return;
	const T *__restrict__ val = (const T *) a.val;
	const bool is_dbl = sizeof(T) == 8;
	const int64_t beg = col_beg(a, j), end = col_beg(a, j + 1);
	int64_t colbase = 0;
	if (TARGET == 1) {
		int g = a.group[j];
		if (g == NA_INT) g = a.ngroup;
		colbase = (int64_t) (g - 1) * a.nrow;
	} else {
		colbase = j * (int64_t) a.ngroup;
	}
	IntSumScratch is = split_int_scratch(a.scratch, out_len);
	for (int64_t k = beg + lane; k < end; k += SVT_WAVE) {
		const T v = val[k];
		const bool miss = is_dbl ? (v != v) : ((int) v == NA_INT);
		if (miss && a.na_rm)
			continue;
		int64_t i;
		if (TARGET == 1) {
			i = colbase + a.row_idx[k];
		} else {
			int g = a.group[a.row_idx[k]];
			if (g == NA_INT) g = a.ngroup;
			i = colbase + g - 1;
		}
		if (is_dbl) {
			atomicAdd((double *) a.out + i, (double) v);
		} else if (miss) {
			atomicOr(is.na + i, 1);
		} else {
			const long long x = (long long) (int) v;
			atomicAdd((unsigned long long *) is.sum + i, (unsigned long long) x);
			atomicAdd(is.abs + i, (unsigned long long) (x < 0 ? -x : x));
		}
	}
}

// int32 results from the parallel pass; cells whose running sums could have left the range
// flag their output column for groupsum_int_exact_kernel.
__global__ void groupsum_int_finish_kernel(GroupSumArgs a, int64_t out_len, int64_t col_len)
{
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= out_len)
		return;
	IntSumScratch is = split_int_scratch(a.scratch, out_len);
	if (is.abs[i] > 2147483647ULL) {
		is.redo[i / col_len] = 1;
		return;
	}
	((int *) a.out)[i] = is.na[i] ? NA_INT : (int) is.sum[i];
}

// One thread per flagged output column, the reference's loops as they are:
// TARGET 0 rowsum (compute_rowsum_ints, src/rowsum_methods.c:66-84): output column j = leaf j;
// TARGET 1 colsum (add_sparse_vec_to_ints, :166-199): output column g = the leaves of group g in
// ascending order.
template <int TARGET>
__global__ void groupsum_int_exact_kernel(GroupSumArgs a, int64_t out_len, int64_t ncols_out,
					  int64_t col_len)
{
	const int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= ncols_out)
		return;
	IntSumScratch is = split_int_scratch(a.scratch, out_len);
	if (!is.redo[c])
		return;
	int *__restrict__ out = (int *) a.out + c * col_len;
	const int *__restrict__ val = (const int *) a.val;
	for (int64_t r = 0; r < col_len; r++) out[r] = 0;
	int ov = 0;
	const int64_t j0 = TARGET == 0 ? c : 0, j1 = TARGET == 0 ? c + 1 : a.ncol;
	for (int64_t j = j0; j < j1; j++) {
		if (TARGET == 1) {
			int g = a.group[j];
			if (g == NA_INT) g = a.ngroup;
			if (g - 1 != c) continue;
		}
		const int64_t beg = col_beg(a, j), end = col_beg(a, j + 1);
		for (int64_t k = beg; k < end; k++) {
			int64_t r;
			if (TARGET == 0) {
				int g = a.group[a.row_idx[k]];
				if (g == NA_INT) g = a.ngroup;
				r = g - 1;
			} else {
				r = a.row_idx[k];
			}
			const int cur = out[r], v = val[k];
			if (v == NA_INT) {
				if (!a.na_rm) out[r] = NA_INT;
				continue;
			}
			if (cur == NA_INT)
				continue;
			const long long y = (long long) cur + v;
			if (y > 2147483647LL || y < -2147483647LL) { out[r] = NA_INT; ov = 1; }
			else out[r] = (int) y;
		}
	}
	if (ov && a.ovflow_flag) *a.ovflow_flag = 1;
}

static int groupsum_common(const GroupSumArgs &a, int64_t out_len, bool colsum,
			   hipStream_t s)
{
	const bool is_dbl = a.Rtype == SVT_REALSXP;
	if (out_len <= 0)
		return 0;
	if (is_dbl)
		HIP_TRY(hipMemsetAsync(a.out, 0, (size_t) out_len * 8, s));
	else
		HIP_TRY(hipMemsetAsync(a.scratch, 0, groupsum_scratch_bytes(a.Rtype, out_len), s));
	if (a.ncol > 0) {
		const unsigned nb = (unsigned) ((a.ncol + 3) / 4);
		if (colsum) {
			if (is_dbl) hipLaunchKernelGGL((groupsum_atomic_kernel<double, 1>), dim3(nb), dim3(256), 0, s, a, out_len);
			else hipLaunchKernelGGL((groupsum_atomic_kernel<int, 1>), dim3(nb), dim3(256), 0, s, a, out_len);
		} else {
			if (is_dbl) hipLaunchKernelGGL((groupsum_atomic_kernel<double, 0>), dim3(nb), dim3(256), 0, s, a, out_len);
			else hipLaunchKernelGGL((groupsum_atomic_kernel<int, 0>), dim3(nb), dim3(256), 0, s, a, out_len);
		}
	}
	if (!is_dbl) {
		const int64_t col_len = colsum ? a.nrow : a.ngroup;
		const int64_t ncols_out = colsum ? a.ngroup : a.ncol;
		const unsigned nbo = (unsigned) ((out_len + 255) / 256);
		const unsigned nbc = (unsigned) ((ncols_out + 63) / 64);
		hipLaunchKernelGGL(groupsum_int_finish_kernel, dim3(nbo), dim3(256), 0, s, a, out_len, col_len);
		if (colsum) hipLaunchKernelGGL(groupsum_int_exact_kernel<1>, dim3(nbc), dim3(64), 0, s, a, out_len, ncols_out, col_len);
		else hipLaunchKernelGGL(groupsum_int_exact_kernel<0>, dim3(nbc), dim3(64), 0, s, a, out_len, ncols_out, col_len);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

int launch_rowsum(const GroupSumArgs &a, hipStream_t s)
{
	const int64_t out_len = (int64_t) a.ngroup * a.ncol;
	return groupsum_common(a, out_len, false, s);
}

// The gather of group[row] is what bounds rowsum (one 64-byte sector from L2 per nonzero): a
// 16-bit, zero-based copy of the table halves its footprint in L2 (4 MB -> 2 MB at 1e6 rows;
// 0.657 -> 0.593 ms at BASELINE config 3, copy included; four nonzeros per thread in flight
// instead of one change nothing: 1e8 distinct L2 requests over 128 channels are 0.33 ms by
// themselves).  NA groups take the last slot as in
// src/rowsum_methods.c:44-64.
__global__ void group16_kernel(const int *__restrict__ g, int64_t n, int ngroup, uint16_t *__restrict__ g16)
{
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	// (check_group, src/rowsum_methods.c:15-37, has refused ids outside 1 .. ngroup at the entry points; a caller of the
	// device level that did not gets them folded into the last group: the kernels index LDS cells with the id)
	if (i < n) {
		int v = g[i];
		if (v == NA_INT) v = ngroup;
		unsigned u = (unsigned) (v - 1);
		if (u > (unsigned) (ngroup - 1)) u = (unsigned) (ngroup - 1);
		g16[i] = (uint16_t) u;
	}
}
__global__ void __launch_bounds__(256)
rowsum_f64_lds16_kernel(GroupSumArgs a, const uint16_t *__restrict__ g16)
{
	extern __shared__ double acc[];
	const int64_t j = blockIdx.x;
	for (int g = threadIdx.x; g < a.ngroup; g += blockDim.x)
		acc[g] = 0.0;
	__syncthreads();
	const double *__restrict__ val = (const double *) a.val;
	const int64_t beg = col_beg(a, j), end = col_beg(a, j + 1);
	for (int64_t k = beg + threadIdx.x; k < end; k += blockDim.x) {
		const double v = val[k];
		if (a.na_rm && v != v)
			continue;
		atomicAdd(&acc[g16[a.row_idx[k]]], v);
	}
	__syncthreads();
	double *out = (double *) a.out + j * (int64_t) a.ngroup;
	for (int g = threadIdx.x; g < a.ngroup; g += blockDim.x)
		out[g] = acc[g];
}

// Round 3: a workgroup takes C columns, one wavefront each, and walks them through the SAME window of
// ROWSUM_WIN rows at a time (a barrier per window; the next 64 (row, value) pairs of a column are loaded before
// the current ones are looked up).  The wavefronts of a workgroup -- and, since all workgroups of a round start
// together, of the whole chip -- then ask for group ids from one band of the table at a time instead of
// from all over it while 1.2 GB of offsets and values stream through the same L2.  tools/micro/rowsum_probe.hip
// (uniformly random columns, 1e6 x 1e4 @ 1 %, 1000 groups): one workgroup per column 0.62 ms; this form 0.48 ms
// with windows of 25600-102400 rows and 14 columns (0.52 with 16: the last round of workgroups is emptier;
// 0.56-0.59 with windows of 12800 rows; no windows 0.72; two chunks loaded ahead 0.55; the window's ids staged
// in LDS and looked up there 0.57-0.78: two more barriers per window than the lookups save; every wavefront
// on the same rows -- what a perfect L1 would give -- 0.35).  In the library: 0.58 -> 0.51 ms for the kernel.
// C is chosen so that the last round of workgroups is as full as possible (14 at 1e4 columns: 715 workgroups
// in 3 rounds).
#define ROWSUM_WIN 49152
template <bool NARM>
__global__ void __launch_bounds__(1024)
rowsum_f64_cols_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
		       const double *__restrict__ val, int64_t ncol, int64_t nrow, int ngroup,
		       const uint16_t *__restrict__ g16, double *__restrict__ out, int C)
{
	extern __shared__ double acc[];                 // [C][ngroup]
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t j = (int64_t) blockIdx.x * C + w;
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) acc[g] = 0.0;
	__syncthreads();
	double *mine = acc + w * ngroup;
	const bool have = j < ncol;
	const int64_t beg = have ? col_ptr[j] : 0, end = have ? col_ptr[j + 1] : 0;
	int64_t k = beg;
	int32_t r = k + lane < end ? row_idx[k + lane] : 0x7FFFFFFF;
	double v = k + lane < end ? val[k + lane] : 0.0;
	for (int64_t R = ROWSUM_WIN; ; R += ROWSUM_WIN) {
		const int32_t Rc = R < 0x7FFFFFFF ? (int32_t) R : 0x7FFFFFFF;
		for (;;) {
			const bool in = r < Rc;
			const int cnt = __popcll(__ballot(in));
			const int64_t kn = k + cnt;             // (the rows of a column ascend: the lanes inside the window are the first cnt)
			const int32_t rn = kn + lane < end ? row_idx[kn + lane] : 0x7FFFFFFF;
			const double vn = kn + lane < end ? val[kn + lane] : 0.0;
			if (in && !(NARM && v != v)) atomicAdd(&mine[g16[r]], v);
			k = kn; r = rn; v = vn;
			if (cnt < 64) break;
		}
		if (R >= nrow) break;
		__syncthreads();
	}
	__syncthreads();
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) {
		const int64_t jj = (int64_t) blockIdx.x * C + g / ngroup;
		if (jj < ncol) out[jj * (int64_t) ngroup + g % ngroup] = acc[g];
	}
}

// rowsum(x, group) with the group of every NONZERO known in advance (svt_dev_rowsum_prepare, include/svt_hip.h):
// what a call streams is 10 bytes per nonzero -- the value and a 16-bit 0-based group id beside it -- instead of
// 12 bytes plus one 64-byte L2 sector for the lookup group[row] that bounds the kernel above (the lookups of a
// wavefront go to ~50 different lines of the table; tools/micro/rowsum_probe.hip).  One wavefront per column, C
// columns per workgroup, accumulators in LDS, no window and no barrier inside the walk.  Same rules as
// compute_rowsum_doubles (src/rowsum_methods.c:44-64): NA group -> last group (folded into the ids), na.rm skips
// NaN and NA values.
__global__ void __launch_bounds__(256)
rowsum_gid_kernel(const int32_t *__restrict__ row_idx, int64_t nnz, const int *__restrict__ group, int ngroup,
		  uint16_t *__restrict__ gid)
{
	for (int64_t k = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * 2; k < nnz;
	     k += (int64_t) gridDim.x * blockDim.x * 2) {
		// two ids per thread: one 4-byte store
		// (the entry points have checked 1 <= group <= ngroup or NA, check_group, src/rowsum_methods.c:258-281; a caller of the
		// device-level function that did not gets its stray ids folded into the last group, never an id past the LDS cells)
		const uint32_t top = (uint32_t) (ngroup - 1);
		const int g0 = group[row_idx[k]];
		uint32_t a = (uint32_t) ((g0 == NA_INT ? ngroup : g0) - 1);
		a = a > top ? top : a;
		if (k + 1 < nnz) {
			const int g1 = group[row_idx[k + 1]];
			uint32_t b = (uint32_t) ((g1 == NA_INT ? ngroup : g1) - 1);
			b = b > top ? top : b;
			*(uint32_t *) (gid + k) = (a & 0xFFFFu) | (b << 16);
		} else
			gid[k] = (uint16_t) a;
	}
}

template <bool NARM>
__global__ void __launch_bounds__(1024)
rowsum_f64_gid_kernel(const int64_t *__restrict__ col_ptr, const double *__restrict__ val,
		      const uint16_t *__restrict__ gid, int64_t ncol, int ngroup, double *__restrict__ out, int C)
{
	extern __shared__ double acc[];                 // [C][ngroup]
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t j = (int64_t) blockIdx.x * C + w;
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) acc[g] = 0.0;
	__syncthreads();
	double *mine = acc + w * ngroup;
	if (j < ncol) {
		const int64_t beg = col_ptr[j], end = col_ptr[j + 1];
		// four chunks of 64 nonzeros in flight per wavefront
		for (int64_t k = beg + lane; k < end; k += 256) {
			double v[4];
			uint16_t g[4];
#pragma unroll
			for (int u = 0; u < 4; u++)
				if (k + 64 * u < end) { v[u] = val[k + 64 * u]; g[u] = gid[k + 64 * u]; }
#pragma unroll
			for (int u = 0; u < 4; u++)
				if (k + 64 * u < end && !(NARM && v[u] != v[u])) atomicAdd(&mine[g[u]], v[u]);
		}
	}
	__syncthreads();
	for (int g = threadIdx.x; g < C * ngroup; g += C * 64) {
		const int64_t jj = (int64_t) blockIdx.x * C + g / ngroup;
		if (jj < ncol) out[jj * (int64_t) ngroup + g % ngroup] = acc[g];
	}
}

// columns per workgroup of rowsum_f64_cols_kernel (0: the operand does not suit it)
static int rowsum_cols_per_wg(const GroupSumArgs &a)
{
	const int64_t cap = (int64_t) (160 * 1024) / ((int64_t) a.ngroup * 8);
	int cmax = cap > 16 ? 16 : (int) cap;
	if (cmax < 4 || a.ncol < 64 || a.col_ptr64 == NULL)
		return 0;
	int best = 0;
	int64_t best_cost = 0;
	for (int c = cmax; c >= 4; c--) {
		const int64_t nwg = (a.ncol + c - 1) / c, rounds = (nwg + 255) / 256;
		const int64_t cost = rounds * c;
		if (best == 0 || cost < best_cost) { best = c; best_cost = cost; }
	}
	return best;
}

// Long f64 columns with few groups: LDS accumulators, no memory atomics.
int launch_rowsum_lds(const GroupSumArgs &a, hipStream_t s)
{
	if (a.ncol <= 0 || a.ngroup <= 0)
		return 0;
	if (a.ngroup < 65535 && a.nrow >= 65536) {
		uint16_t *g16 = NULL;
		HIP_TRY(hipMallocAsync((void **) &g16, (size_t) a.nrow * 2 + 16, s));
		hipLaunchKernelGGL(group16_kernel, dim3((unsigned) ((a.nrow + 255) / 256)), dim3(256), 0, s,
				   a.group, a.nrow, a.ngroup, g16);
		const int C = rowsum_cols_per_wg(a);
		if (C > 0) {
			const size_t lds = (size_t) C * a.ngroup * 8;
			const dim3 grid((unsigned) ((a.ncol + C - 1) / C));
			if (a.na_rm) {
				(void) hipFuncSetAttribute((const void *) rowsum_f64_cols_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
				hipLaunchKernelGGL(rowsum_f64_cols_kernel<true>, grid, dim3(C * 64), lds, s, a.col_ptr64, a.row_idx,
						   (const double *) a.val, a.ncol, a.nrow, a.ngroup, g16, (double *) a.out, C);
			} else {
				(void) hipFuncSetAttribute((const void *) rowsum_f64_cols_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
				hipLaunchKernelGGL(rowsum_f64_cols_kernel<false>, grid, dim3(C * 64), lds, s, a.col_ptr64, a.row_idx,
						   (const double *) a.val, a.ncol, a.nrow, a.ngroup, g16, (double *) a.out, C);
			}
		} else
		hipLaunchKernelGGL(rowsum_f64_lds16_kernel, dim3((unsigned) a.ncol), dim3(256),
				   (size_t) a.ngroup * 8, s, a, g16);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipFreeAsync(g16, s));
		return 0;
	}
	hipLaunchKernelGGL(rowsum_f64_lds_kernel, dim3((unsigned) a.ncol), dim3(256),
			   (size_t) a.ngroup * 8, s, a);
	HIP_TRY(hipGetLastError());
	return 0;
}

// The same ids by the walk of rowsum_f64_cols_kernel (round 5): a wavefront per column, 16 columns per workgroup, all of
// them inside the same window of ROWSUM_WIN rows at a time, looking up the 16-bit copy of the table -- the lookups of
// the whole chip then go to one band of 96 KB of it instead of all over 4 MB of int32 while the offsets stream through
// the same L2.  The flat kernel above took longer than an unprepared rowsum() call (0.62 against 0.52 ms at config 3).
__global__ void __launch_bounds__(1024)
rowsum_gid_cols_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx, int64_t ncol,
		       int64_t nrow, const uint16_t *__restrict__ g16, uint16_t *__restrict__ gid, int C)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t j = (int64_t) blockIdx.x * C + w;
	const bool have = j < ncol;
	const int64_t beg = have ? col_ptr[j] : 0, end = have ? col_ptr[j + 1] : 0;
	int64_t k = beg;
	int32_t r = k + lane < end ? row_idx[k + lane] : 0x7FFFFFFF;
	for (int64_t R = ROWSUM_WIN; ; R += ROWSUM_WIN) {
		const int32_t Rc = R < 0x7FFFFFFF ? (int32_t) R : 0x7FFFFFFF;
		for (;;) {
			const bool in = r < Rc;
			const int cnt = __popcll(__ballot(in));
			const int64_t kn = k + cnt;             // (the rows of a column ascend: the lanes inside the window are the first cnt)
			const int32_t rn = kn + lane < end ? row_idx[kn + lane] : 0x7FFFFFFF;
			if (in) gid[k + lane] = g16[r];
			k = kn; r = rn;
			if (cnt < 64) break;
		}
		if (R >= nrow) break;
		__syncthreads();
	}
}

// The 16-bit group id of every nonzero (gid: nnz ids); needs ngroup <= 65535.
int launch_rowsum_gid(const GroupSumArgs &a, int64_t nnz, uint16_t *gid, hipStream_t s)
{
	if (nnz <= 0)
		return 0;
	if (a.col_ptr64 != NULL && a.ncol >= 64 && a.nrow >= 65536 && a.ngroup >= 1 && a.ngroup < 65535) {
		// (group16_kernel folds NA into the last group as the flat kernel does; stray ids are the caller's: check_group)
		uint16_t *g16 = NULL;
		HIP_TRY(hipMallocAsync((void **) &g16, (size_t) a.nrow * 2 + 16, s));
		hipLaunchKernelGGL(group16_kernel, dim3((unsigned) ((a.nrow + 255) / 256)), dim3(256), 0, s,
				   a.group, a.nrow, a.ngroup, g16);
		int C = 16; int64_t best = -1;
		for (int c = 16; c >= 8; c--) {             // the fullest last round of workgroups
			const int64_t nwg = (a.ncol + c - 1) / c, cost = (nwg + 255) / 256 * c;
			if (best < 0 || cost < best) { best = cost; C = c; }
		}
		hipLaunchKernelGGL(rowsum_gid_cols_kernel, dim3((unsigned) ((a.ncol + C - 1) / C)), dim3(C * 64), 0, s,
				   a.col_ptr64, a.row_idx, a.ncol, a.nrow, g16, gid, C);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipFreeAsync(g16, s));
		return 0;
	}
	int64_t nb = ((nnz + 1) / 2 + 255) / 256;
	if (nb > 256 * 32) nb = 256 * 32;
	if (nb < 1) nb = 1;
	hipLaunchKernelGGL(rowsum_gid_kernel, dim3((unsigned) nb), dim3(256), 0, s, a.row_idx, nnz, a.group, a.ngroup, gid);
	HIP_TRY(hipGetLastError());
	return 0;
}

// rowsum on prepared ids; every cell of out (ngroup x ncol) is written.  Returns 1 when the shape does not
// suit the kernel (the caller then runs the unprepared product).
int launch_rowsum_prepared(const GroupSumArgs &a, const uint16_t *gid, hipStream_t s)
{
	if (a.ncol <= 0 || a.ngroup <= 0)
		return 0;
	const int64_t cap = (int64_t) (160 * 1024) / ((int64_t) a.ngroup * 8);
	if (cap < 1 || a.col_ptr64 == NULL)
		return 1;
	int C = cap > 16 ? 16 : (int) cap;
	{       // as rowsum_cols_per_wg(): the fullest last round of workgroups
		int best = C; int64_t best_cost = -1;
		for (int c = C; c >= (C >= 4 ? 4 : 1); c--) {
			const int64_t nwg = (a.ncol + c - 1) / c, rounds = (nwg + 255) / 256, cost = rounds * c;
			if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
		}
		C = best;
	}
	const size_t lds = (size_t) C * a.ngroup * 8;
	const dim3 grid((unsigned) ((a.ncol + C - 1) / C));
	if (a.na_rm) {
		(void) hipFuncSetAttribute((const void *) rowsum_f64_gid_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
		hipLaunchKernelGGL(rowsum_f64_gid_kernel<true>, grid, dim3(C * 64), lds, s, a.col_ptr64, (const double *) a.val,
				   gid, a.ncol, a.ngroup, (double *) a.out, C);
	} else {
		(void) hipFuncSetAttribute((const void *) rowsum_f64_gid_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
		hipLaunchKernelGGL(rowsum_f64_gid_kernel<false>, grid, dim3(C * 64), lds, s, a.col_ptr64, (const double *) a.val,
				   gid, a.ncol, a.ngroup, (double *) a.out, C);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

int launch_colsum(const GroupSumArgs &a, hipStream_t s)
{
	const int64_t out_len = (int64_t) a.ngroup * a.nrow;
	return groupsum_common(a, out_len, true, s);
}
