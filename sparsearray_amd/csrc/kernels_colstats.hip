// col* matrixStats and whole-array summarization on the CSC device layout.
//
// Reference: C_colStats_SVT / REC_colStats_SVT (src/SparseArray_matrixStats.c:
// 200-284) call _summarize_SVT() (src/SparseArray_summarization.c:89-109) once
// per "generalized column".  On the device a generalized column is a run of
// `inner` consecutive leaves, i.e. one contiguous slice of val[]; the op is a
// streaming segmented reduction over that slice (row_idx is never read).
//
// Two launch shapes share one body:
//   NT = 16  : 16 lanes per segment, 16 segments per 256-thread workgroup
//              (short leaves: config 1 / config 5, ~100 nz per leaf)
//   NT = 64  : one wavefront per segment, 4 segments per workgroup
//   NT = 256 : one workgroup per segment (long leaves: config 2, ~1e4 nz)
//
// Sequential NA/NaN rules of src/Rvector_summarization.c:177-734 restated as
// order-independent predicates (valid because every rule is of the form "any
// NA anywhere wins, else any NaN wins, else the plain IEEE reduction"):
//   flags & F_NA   some value is R's NA (payload 1954 / INT_MIN)
//   flags & F_NAN  some value is a NaN that is not NA
// Roofline: HBM; algorithmic bytes = 8 (f64) or 4 (i32) per nonzero.
#include "svt_common.h"

#define F_NA    1
#define F_NAN   2
#define F_TRUE  4   // a non-NA value != 0
#define F_ZERO  8   // a stored value == 0 (not expected in a valid SVT)
#define F_HAVE 16   // at least one non-NA value went into min/max

// 16-lane groups (NT = 16): xor butterflies stay inside an aligned group and leave
// the result in every lane of it
template <typename V, typename F>
__device__ inline V grp16(V v, F f)
{
#pragma unroll
	for (int m = 8; m >= 1; m >>= 1) v = f(v, __shfl_xor(v, m, SVT_WAVE));
	return v;
}

template <int NT>
__device__ inline double red_sum(double v, double *sm)
{
	if (NT == 16) return grp16(v, [](double a, double b) { return a + b; });
	v = wave_sum(v);
	if (NT == SVT_WAVE)
		return __shfl(v, 0, SVT_WAVE);
	const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
	__syncthreads();
	if (l == 0) sm[w] = v;
	__syncthreads();
	double t = 0.0;
	for (int i = 0; i < NT / SVT_WAVE; i++) t += sm[i];
	return t;
}
template <int NT>
__device__ inline double red_prod(double v, double *sm)
{
	if (NT == 16) return grp16(v, [](double a, double b) { return a * b; });
	v = wave_prod(v);
	if (NT == SVT_WAVE)
		return __shfl(v, 0, SVT_WAVE);
	const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
	__syncthreads();
	if (l == 0) sm[w] = v;
	__syncthreads();
	double t = 1.0;
	for (int i = 0; i < NT / SVT_WAVE; i++) t *= sm[i];
	return t;
}
template <int NT>
__device__ inline double red_min(double v, double *sm, bool is_min)
{
	if (NT == 16)
		return is_min ? grp16(v, [](double a, double b) { return b < a ? b : a; })
			      : grp16(v, [](double a, double b) { return b > a ? b : a; });
	v = is_min ? wave_min(v) : wave_max(v);
	if (NT == SVT_WAVE)
		return __shfl(v, 0, SVT_WAVE);
	const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
	__syncthreads();
	if (l == 0) sm[w] = v;
	__syncthreads();
	double t = sm[0];
	for (int i = 1; i < NT / SVT_WAVE; i++)
		t = is_min ? (sm[i] < t ? sm[i] : t) : (sm[i] > t ? sm[i] : t);
	return t;
}
template <int NT>
__device__ inline long long red_sum_ll(long long v, double *sm)
{
	if (NT == 16) return grp16(v, [](long long a, long long b) { return a + b; });
	v = wave_sum_ll(v);
	if (NT == SVT_WAVE)
		return __shfl(v, 0, SVT_WAVE);
	long long *s = (long long *) sm;
	const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
	__syncthreads();
	if (l == 0) s[w] = v;
	__syncthreads();
	long long t = 0;
	for (int i = 0; i < NT / SVT_WAVE; i++) t += s[i];
	return t;
}
template <int NT>
__device__ inline int red_or(int v, double *sm)
{
	if (NT == 16) return grp16(v, [](int a, int b) { return a | b; });
	v = wave_or(v);
	if (NT == SVT_WAVE)
		return __shfl(v, 0, SVT_WAVE);
	int *s = (int *) sm;
	const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
	__syncthreads();
	if (l == 0) s[w] = v;
	__syncthreads();
	int t = 0;
	for (int i = 0; i < NT / SVT_WAVE; i++) t |= s[i];
	return t;
}

template <typename T> struct ValTraits;
template <> struct ValTraits<double> {
	static __device__ inline bool is_missing(double v) { return v != v; }
	static __device__ inline bool is_na(double v) { return svt_is_na(v); }
	static __device__ inline double as_double(double v) { return v; }
};
template <> struct ValTraits<int> {
	static __device__ inline bool is_missing(int v) { return v == NA_INT; }
	static __device__ inline bool is_na(int v) { return v == NA_INT; }
	static __device__ inline double as_double(int v) { return (double) v; }
};

// CAP > 0: the first pass keeps the
// column in registers (CAP values per thread, all loads in flight at once) and
// the centred second pass runs from there instead of re-reading it -- the
// two-pass arithmetic of the reference (src/SparseArray_summarization.c:70-109)
// with one trip to memory.  Columns longer than NT*CAP take the re-read path.
template <typename T, int NT, int CAP>
__global__ void __launch_bounds__(256)
colstats_kernel(StatsArgs a)
{
	__shared__ double sm[4 * (256 / SVT_WAVE)];
	const int per_block = 256 / NT;
	const int sub = NT == 256 ? 0 : (NT == 16 ? (threadIdx.x >> 4) : (threadIdx.x >> 6));
	const int tid = NT == 256 ? threadIdx.x : (threadIdx.x & (NT - 1));
	const int64_t g = (int64_t) blockIdx.x * per_block + sub;
	if (NT <= SVT_WAVE && g >= a.nseg)
		return;   // whole wave / 16-lane group exits together; no barriers on this path
	double *my_sm = sm + (NT == 16 ? 0 : sub * 4);
	(void) my_sm;
	const T *__restrict__ val = (const T *) a.val;
	const int64_t beg = a.col_ptr[g * a.inner];
	const int64_t end = a.col_ptr[(g + 1) * a.inner];
	const int64_t nz = end - beg;
	const int64_t zerocount = a.seg_len - nz;
	const int oc = a.opcode;
	const bool narm = a.na_rm != 0;
	const bool is_dbl = sizeof(T) == 8;
	typedef ValTraits<T> VT;

	int flags = 0;
	long long nacnt = 0;
	double acc = (oc == SVT_OP_PROD) ? 1.0 : 0.0;
	const bool is_minmax = oc == SVT_OP_MIN || oc == SVT_OP_MAX;
	const bool is_min = oc == SVT_OP_MIN;
	double mm = is_min ? INFINITY : -INFINITY;

	// ---- pass 1: flags, NA count, sum / product / extremum -------------
	auto step1 = [&](const T v) {
		if (VT::is_missing(v)) {
			nacnt++;
			flags |= VT::is_na(v) ? F_NA : F_NAN;
			if (narm)
				return;
			// fall through: NaN/NA take part in the IEEE reduction
			// (ints: the value is never used once F_NA is set)
			if (!is_dbl)
				return;
		} else {
			if (v != (T) 0) flags |= F_TRUE; else flags |= F_ZERO;
			flags |= F_HAVE;
		}
		const double d = VT::as_double(v);
		if (oc == SVT_OP_PROD) acc *= d;
		else if (is_minmax) { if (d == d) mm = is_min ? (d < mm ? d : mm) : (d > mm ? d : mm); }
		else acc += d;
	};
	T cache[CAP > 0 ? CAP : 1];
	const bool cached = CAP > 0 && nz <= (int64_t) NT * CAP;   // uniform per column
	if (cached) {
#pragma unroll
		for (int i = 0; i < CAP; i++) {
			const int64_t k = beg + tid + (int64_t) i * NT;
			cache[i] = k < end ? val[k] : (T) 0;
		}
#pragma unroll
		for (int i = 0; i < CAP; i++)
			if (beg + tid + (int64_t) i * NT < end) step1(cache[i]);
	} else {
		for (int64_t k = beg + tid; k < end; k += NT)
			step1(val[k]);
	}
	flags = red_or<NT>(flags, my_sm);
	nacnt = red_sum_ll<NT>(nacnt, my_sm);
	if (oc == SVT_OP_PROD) acc = red_prod<NT>(acc, my_sm);
	else if (is_minmax) mm = red_min<NT>(mm, my_sm, is_min);
	else acc = red_sum<NT>(acc, my_sm);

	// NaArray (a.na_bg): the zerocount implicit values are NAs -- they count as
	// NAs, one NA is fed to the op unless na.rm, and no implicit zero takes part
	// (Rvector_summarization.c:1086-1106)
	const bool nabg = a.na_bg != 0;
	const int64_t implicit_na = nabg ? zerocount : 0;
	const int64_t zeros = nabg ? 0 : zerocount;            // implicit zeros
	if (implicit_na > 0 && !narm) flags |= F_NA;
	const bool brk_na = (flags & F_NA) && !narm;   // "breaking value" NA
	const double n_eff = narm ? (double) ((nabg ? nz : a.seg_len) - nacnt) : (double) a.seg_len;
	const double NAr = svt_na_real();
	double rd = 0.0;
	int ri = 0;
	int warn = 0;

	switch (oc) {
	case SVT_OP_ANYNA:
		ri = ((flags & (F_NA | F_NAN)) || implicit_na > 0) ? 1 : 0;
		break;
	case SVT_OP_COUNTNAS:
		rd = (double) (nacnt + implicit_na);
		break;
	case SVT_OP_ANY:      // src/Rvector_summarization.c:260-284
		ri = (flags & F_TRUE) ? 1 : (brk_na ? NA_INT : 0);
		break;
	case SVT_OP_ALL:      // :289-313 plus the implicit zero of :1100-1106
		ri = ((flags & F_ZERO) || zeros > 0) ? 0 : (brk_na ? NA_INT : 1);
		break;
	case SVT_OP_SUM:
		rd = brk_na ? NAr : acc;
		break;
	case SVT_OP_MEAN:
		rd = brk_na ? NAr : acc / n_eff;
		break;
	case SVT_OP_PROD:
		if (brk_na) rd = NAr;
		else rd = zeros > 0 ? acc * 0.0 : acc;
		break;
	case SVT_OP_MIN: case SVT_OP_MAX:
		if (is_dbl) {
			if (brk_na) { rd = NAr; break; }
			if ((flags & F_NAN) && !narm) { rd = NAN; break; }
			if (zeros > 0)
				mm = is_min ? (0.0 < mm ? 0.0 : mm) : (0.0 > mm ? 0.0 : mm);
			rd = mm;
		} else {
			if (brk_na) { ri = NA_INT; break; }
			bool have = (flags & F_HAVE) != 0;
			if (zeros > 0) {
				mm = have ? (is_min ? (0.0 < mm ? 0.0 : mm) : (0.0 > mm ? 0.0 : mm)) : 0.0;
				have = true;
			}
			if (!have) { ri = NA_INT; warn = 1; }   // :1108-1128
			else ri = (int) mm;
		}
		break;
	case SVT_OP_CENTERED_X2_SUM: case SVT_OP_VAR1: case SVT_OP_SD1: {
		// src/SparseArray_summarization.c:70-109: without a center the
		// mean comes from a first full pass (done above: acc, nacnt).
		double c = a.center;
		if (c != c)
			c = (brk_na && !a.dgc) ? NAr : acc / n_eff;
		double acc2 = 0.0;
		auto step2 = [&](const T v) {
			if (VT::is_missing(v) && (narm || !is_dbl))
				return;
			const double d = VT::as_double(v) - c;
			acc2 += d * d;
		};
		if (cached) {
#pragma unroll
			for (int i = 0; i < CAP; i++)
				if (beg + tid + (int64_t) i * NT < end) step2(cache[i]);
		} else {
			for (int64_t k = beg + tid; k < end; k += NT)
				step2(val[k]);
		}
		acc2 = red_sum<NT>(acc2, my_sm);
		if (a.dgc) {      // col_var(), src/sparseMatrix_utils.c:190-203: IEEE all the way
			rd = (c * c * (double) zeros + acc2) / (n_eff - 1.0);
			break;
		}
		if (brk_na) { rd = NAr; break; }
		rd = acc2 + c * c * (double) zeros;
		if (oc == SVT_OP_CENTERED_X2_SUM) break;
		if (n_eff <= 1.0) { rd = NAr; break; }
		rd /= (n_eff - 1.0);
		if (oc == SVT_OP_SD1) rd = sqrt(rd);
		break;
	}
	default:
		break;
	}
	if (tid == 0) {
		const bool out_is_int = oc == SVT_OP_ANYNA || oc == SVT_OP_ANY ||
			oc == SVT_OP_ALL || (is_minmax && !is_dbl);
		if (out_is_int) ((int *) a.out)[g] = ri;
		else ((double *) a.out)[g] = rd;
		if (warn && a.warn_flag) *a.warn_flag = 1;
	}
}

// --------------------------------------------------------------------------
// Few, very long segments (whole-array sum(), colSums(dims = ndim - 1) of a 3-d
// array, ...): one workgroup per segment would leave the chip idle, so every
// segment is cut into NCHUNK element ranges, one workgroup each, whose partial
// states are combined in chunk order by a small second kernel.  var1 / sd1 /
// centered sums keep the reference's two passes (mean first,
// src/SparseArray_summarization.c:70-109): state + mean, then the centred
// squares, 4 launches.  colstats_final() is the same post-processing as in
// colstats_kernel (Rvector_summarization.c:1078-1177).
// --------------------------------------------------------------------------
struct ColState {
	int flags;
	int pad;
	long long nacnt;
	double acc;       // sum or product
	double mm;        // extremum
};

__device__ inline double colstats_neff(const StatsArgs &a, const ColState &st, int64_t nz)
{
	const bool narm = a.na_rm != 0, nabg = a.na_bg != 0;
	return narm ? (double) ((nabg ? nz : a.seg_len) - st.nacnt) : (double) a.seg_len;
}

__device__ inline bool colstats_brk(const StatsArgs &a, const ColState &st, int64_t nz)
{
	int flags = st.flags;
	if (a.na_bg && a.seg_len - nz > 0 && !a.na_rm) flags |= F_NA;
	return (flags & F_NA) && !a.na_rm;
}

// centre used by the two-pass ops
__device__ inline double colstats_center(const StatsArgs &a, const ColState &st, int64_t nz)
{
	double c = a.center;
	if (c != c)
		c = (colstats_brk(a, st, nz) && !a.dgc) ? svt_na_real() : st.acc / colstats_neff(a, st, nz);
	return c;
}

__device__ inline void colstats_final(const StatsArgs &a, int64_t g, const ColState &st, int64_t nz,
				      double c, double acc2, bool is_dbl)
{
	const int oc = a.opcode;
	const bool narm = a.na_rm != 0, nabg = a.na_bg != 0;
	const int64_t zerocount = a.seg_len - nz;
	const int64_t implicit_na = nabg ? zerocount : 0;
	const int64_t zeros = nabg ? 0 : zerocount;
	int flags = st.flags;
	if (implicit_na > 0 && !narm) flags |= F_NA;
	const bool brk_na = (flags & F_NA) && !narm;
	const double n_eff = colstats_neff(a, st, nz);
	const double NAr = svt_na_real();
	const bool is_minmax = oc == SVT_OP_MIN || oc == SVT_OP_MAX, is_min = oc == SVT_OP_MIN;
	double rd = 0.0, mm = st.mm;
	int ri = 0, warn = 0;
	switch (oc) {
	case SVT_OP_ANYNA: ri = ((flags & (F_NA | F_NAN)) || implicit_na > 0) ? 1 : 0; break;
	case SVT_OP_COUNTNAS: rd = (double) (st.nacnt + implicit_na); break;
	case SVT_OP_ANY: ri = (flags & F_TRUE) ? 1 : (brk_na ? NA_INT : 0); break;
	case SVT_OP_ALL: ri = ((flags & F_ZERO) || zeros > 0) ? 0 : (brk_na ? NA_INT : 1); break;
	case SVT_OP_SUM: rd = brk_na ? NAr : st.acc; break;
	case SVT_OP_MEAN: rd = brk_na ? NAr : st.acc / n_eff; break;
	case SVT_OP_PROD: rd = brk_na ? NAr : (zeros > 0 ? st.acc * 0.0 : st.acc); break;
	case SVT_OP_MIN: case SVT_OP_MAX:
		if (is_dbl) {
			if (brk_na) { rd = NAr; break; }
			if ((flags & F_NAN) && !narm) { rd = NAN; break; }
			if (zeros > 0) mm = is_min ? (0.0 < mm ? 0.0 : mm) : (0.0 > mm ? 0.0 : mm);
			rd = mm;
		} else {
			if (brk_na) { ri = NA_INT; break; }
			bool have = (flags & F_HAVE) != 0;
			if (zeros > 0) {
				mm = have ? (is_min ? (0.0 < mm ? 0.0 : mm) : (0.0 > mm ? 0.0 : mm)) : 0.0;
				have = true;
			}
			if (!have) { ri = NA_INT; warn = 1; }
			else ri = (int) mm;
		}
		break;
	case SVT_OP_CENTERED_X2_SUM: case SVT_OP_VAR1: case SVT_OP_SD1:
		if (a.dgc) { rd = (c * c * (double) zeros + acc2) / (n_eff - 1.0); break; }
		if (brk_na) { rd = NAr; break; }
		rd = acc2 + c * c * (double) zeros;
		if (oc == SVT_OP_CENTERED_X2_SUM) break;
		if (n_eff <= 1.0) { rd = NAr; break; }
		rd /= (n_eff - 1.0);
		if (oc == SVT_OP_SD1) rd = sqrt(rd);
		break;
	default: break;
	}
	const bool out_is_int = oc == SVT_OP_ANYNA || oc == SVT_OP_ANY || oc == SVT_OP_ALL ||
		(is_minmax && !is_dbl);
	if (out_is_int) ((int *) a.out)[g] = ri;
	else ((double *) a.out)[g] = rd;
	if (warn && a.warn_flag) *a.warn_flag = 1;
}

// PASS 1: state of one chunk;  PASS 2: centred squares of one chunk
template <typename T, int PASS>
__global__ void __launch_bounds__(256)
colstats_chunk_kernel(StatsArgs a, int nchunk, ColState *__restrict__ st, const double *__restrict__ centers,
		      double *__restrict__ acc2_out)
{
	__shared__ double sm[4 * (256 / SVT_WAVE)];
	typedef ValTraits<T> VT;
	const bool is_dbl = sizeof(T) == 8;
	const int64_t g = blockIdx.y;
	const int ch = blockIdx.x, tid = threadIdx.x;
	const T *__restrict__ val = (const T *) a.val;
	const int64_t sbeg = a.col_ptr[g * a.inner], send = a.col_ptr[(g + 1) * a.inner];
	const int64_t nz = send - sbeg;
	const int64_t beg = sbeg + nz * ch / nchunk, end = sbeg + nz * (ch + 1) / nchunk;
	const int oc = a.opcode;
	const bool narm = a.na_rm != 0;
	if (PASS == 1) {
		const bool is_minmax = oc == SVT_OP_MIN || oc == SVT_OP_MAX, is_min = oc == SVT_OP_MIN;
		int flags = 0;
		long long nacnt = 0;
		double acc = (oc == SVT_OP_PROD) ? 1.0 : 0.0, mm = is_min ? INFINITY : -INFINITY;
		auto step1 = [&](const T v) {
			if (VT::is_missing(v)) {
				nacnt++;
				flags |= VT::is_na(v) ? F_NA : F_NAN;
				if (narm || !is_dbl) return;
			} else {
				flags |= (v != (T) 0) ? F_TRUE : F_ZERO;
				flags |= F_HAVE;
			}
			const double d = VT::as_double(v);
			if (oc == SVT_OP_PROD) acc *= d;
			else if (is_minmax) { if (d == d) mm = is_min ? (d < mm ? d : mm) : (d > mm ? d : mm); }
			else acc += d;
		};
		int64_t k = beg + tid;
		for (; k + 7 * 256 < end; k += 8 * 256) {      // 8 loads in flight per thread
			T v[8];
#pragma unroll
			for (int u = 0; u < 8; u++) v[u] = val[k + u * 256];
#pragma unroll
			for (int u = 0; u < 8; u++) step1(v[u]);
		}
		for (; k < end; k += 256) step1(val[k]);
		flags = red_or<256>(flags, sm);
		nacnt = red_sum_ll<256>(nacnt, sm);
		if (oc == SVT_OP_PROD) acc = red_prod<256>(acc, sm);
		else if (is_minmax) mm = red_min<256>(mm, sm, is_min);
		else acc = red_sum<256>(acc, sm);
		if (tid == 0) {
			ColState o;
			o.flags = flags; o.pad = 0; o.nacnt = nacnt; o.acc = acc; o.mm = mm;
			st[g * nchunk + ch] = o;
		}
	} else {
		const double c = centers[g];
		double acc2 = 0.0;
		auto step2 = [&](const T v) {
			if (VT::is_missing(v) && (narm || !is_dbl)) return;
			const double d = VT::as_double(v) - c;
			acc2 += d * d;
		};
		int64_t k = beg + tid;
		for (; k + 7 * 256 < end; k += 8 * 256) {
			T v[8];
#pragma unroll
			for (int u = 0; u < 8; u++) v[u] = val[k + u * 256];
#pragma unroll
			for (int u = 0; u < 8; u++) step2(v[u]);
		}
		for (; k < end; k += 256) step2(val[k]);
		acc2 = red_sum<256>(acc2, sm);
		if (tid == 0) acc2_out[g * nchunk + ch] = acc2;
	}
}

// STAGE 1: combine the chunk states (chunk order); final result, or the centre for
// pass 2.  STAGE 2: combine the centred squares, final result.
template <int STAGE>
__global__ void colstats_combine_kernel(StatsArgs a, int nchunk, ColState *__restrict__ st,
					double *__restrict__ centers, const double *__restrict__ acc2_in,
					int two_pass, int is_dbl)
{
	const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= a.nseg) return;
	const int64_t nz = a.col_ptr[(g + 1) * a.inner] - a.col_ptr[g * a.inner];
	if (STAGE == 1) {
		const int oc = a.opcode;
		const bool is_min = oc == SVT_OP_MIN;
		ColState t = st[g * nchunk];
		for (int c = 1; c < nchunk; c++) {
			const ColState u = st[g * nchunk + c];
			t.flags |= u.flags;
			t.nacnt += u.nacnt;
			if (oc == SVT_OP_PROD) t.acc *= u.acc; else t.acc += u.acc;
			t.mm = is_min ? (u.mm < t.mm ? u.mm : t.mm) : (u.mm > t.mm ? u.mm : t.mm);
		}
		if (two_pass) {
			st[g * nchunk] = t;
			centers[g] = colstats_center(a, t, nz);
		} else {
			colstats_final(a, g, t, nz, 0.0, 0.0, is_dbl != 0);
		}
	} else {
		double acc2 = 0.0;
		for (int c = 0; c < nchunk; c++) acc2 += acc2_in[g * nchunk + c];
		colstats_final(a, g, st[g * nchunk], nz, centers[g], acc2, is_dbl != 0);
	}
}

// Very short segments (fewer than ~4 nonzeros on average: arrays with a short first extent,
// leaf-shattering permutations): one THREAD per segment, sequential like the reference; what
// it streams is mostly col_ptr.  Same post-processing (colstats_final).
template <typename T>
__global__ void __launch_bounds__(256)
colstats_thread_kernel(StatsArgs a)
{
	typedef ValTraits<T> VT;
	const bool is_dbl = sizeof(T) == 8;
	const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= a.nseg) return;
	const T *__restrict__ val = (const T *) a.val;
	const int64_t beg = a.col_ptr[g * a.inner], end = a.col_ptr[(g + 1) * a.inner];
	const int oc = a.opcode;
	const bool narm = a.na_rm != 0;
	const bool is_minmax = oc == SVT_OP_MIN || oc == SVT_OP_MAX, is_min = oc == SVT_OP_MIN;
	ColState st;
	st.flags = 0; st.pad = 0; st.nacnt = 0;
	st.acc = (oc == SVT_OP_PROD) ? 1.0 : 0.0;
	st.mm = is_min ? INFINITY : -INFINITY;
	for (int64_t k = beg; k < end; k++) {
		const T v = val[k];
		if (VT::is_missing(v)) {
			st.nacnt++;
			st.flags |= VT::is_na(v) ? F_NA : F_NAN;
			if (narm || !is_dbl) continue;
		} else {
			st.flags |= (v != (T) 0) ? F_TRUE : F_ZERO;
			st.flags |= F_HAVE;
		}
		const double d = VT::as_double(v);
		if (oc == SVT_OP_PROD) st.acc *= d;
		else if (is_minmax) { if (d == d) st.mm = is_min ? (d < st.mm ? d : st.mm) : (d > st.mm ? d : st.mm); }
		else st.acc += d;
	}
	double c = 0.0, acc2 = 0.0;
	if (oc == SVT_OP_CENTERED_X2_SUM || oc == SVT_OP_VAR1 || oc == SVT_OP_SD1) {
		c = colstats_center(a, st, end - beg);
		for (int64_t k = beg; k < end; k++) {
			const T v = val[k];
			if (VT::is_missing(v) && (narm || !is_dbl)) continue;
			const double d = VT::as_double(v) - c;
			acc2 += d * d;
		}
	}
	colstats_final(a, g, st, end - beg, c, acc2, is_dbl);
}

static int launch_colstats_split(const StatsArgs &a, int nchunk, hipStream_t s)
{
	const bool is_dbl = a.Rtype == SVT_REALSXP;
	const bool two_pass = a.opcode == SVT_OP_CENTERED_X2_SUM || a.opcode == SVT_OP_VAR1 ||
			      a.opcode == SVT_OP_SD1;
	const size_t n = (size_t) a.nseg * nchunk;
	const size_t b_st = (n * sizeof(ColState) + 255) / 256 * 256, b_c = ((size_t) a.nseg * 8 + 255) / 256 * 256;
	char *scr = NULL;
	HIP_TRY(hipMallocAsync((void **) &scr, b_st + b_c + n * 8 + 256, s));   // stream-ordered scratch
	ColState *st = (ColState *) scr;
	double *centers = (double *) (scr + b_st), *acc2 = (double *) (scr + b_st + b_c);
	dim3 grid((unsigned) nchunk, (unsigned) a.nseg), cgrid((unsigned) ((a.nseg + 255) / 256));
	if (is_dbl) hipLaunchKernelGGL((colstats_chunk_kernel<double, 1>), grid, dim3(256), 0, s, a, nchunk, st, centers, acc2);
	else hipLaunchKernelGGL((colstats_chunk_kernel<int, 1>), grid, dim3(256), 0, s, a, nchunk, st, centers, acc2);
	hipLaunchKernelGGL(colstats_combine_kernel<1>, cgrid, dim3(256), 0, s, a, nchunk, st, centers, acc2,
			   (int) two_pass, (int) is_dbl);
	if (two_pass) {
		if (is_dbl) hipLaunchKernelGGL((colstats_chunk_kernel<double, 2>), grid, dim3(256), 0, s, a, nchunk, st, centers, acc2);
		else hipLaunchKernelGGL((colstats_chunk_kernel<int, 2>), grid, dim3(256), 0, s, a, nchunk, st, centers, acc2);
		hipLaunchKernelGGL(colstats_combine_kernel<2>, cgrid, dim3(256), 0, s, a, nchunk, st, centers, acc2,
				   (int) two_pass, (int) is_dbl);
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipFreeAsync(scr, s));
	return 0;
}

int launch_colstats(const StatsArgs &a, int64_t nnz, hipStream_t s)
{
	if (a.nseg <= 0)
		return 0;
	if (a.nseg > 0x7FFFFFFFLL)
		return svt_set_error("too many generalized columns");
	const int64_t avg = nnz / a.nseg;
	const bool is_dbl = a.Rtype == SVT_REALSXP;
	if (a.nseg < 512 && avg >= 65536 && a.nseg <= 65535) {
		// few long segments: ~1024 workgroups in all, >= 16K elements each
		int64_t nchunk = (1024 + a.nseg - 1) / a.nseg;
		if (nchunk > avg / 16384) nchunk = avg / 16384;
		if (nchunk >= 2)
			return launch_colstats_split(a, (int) nchunk, s);
	}
	if (avg >= 1024) {
		dim3 grid((unsigned) a.nseg), block(256);
		if (avg <= 256 * 40) {               // (also faster for one-pass ops: all loads in flight)
			if (is_dbl) hipLaunchKernelGGL((colstats_kernel<double, 256, 48>), grid, block, 0, s, a);
			else hipLaunchKernelGGL((colstats_kernel<int, 256, 48>), grid, block, 0, s, a);
		} else {
			if (is_dbl) hipLaunchKernelGGL((colstats_kernel<double, 256, 0>), grid, block, 0, s, a);
			else hipLaunchKernelGGL((colstats_kernel<int, 256, 0>), grid, block, 0, s, a);
		}
	} else {
		dim3 grid((unsigned) ((a.nseg + 3) / 4)), block(256);
		if (avg < 4 && a.nseg >= 4096) {
			dim3 gridt((unsigned) ((a.nseg + 255) / 256));
			if (is_dbl) hipLaunchKernelGGL(colstats_thread_kernel<double>, gridt, block, 0, s, a);
			else hipLaunchKernelGGL(colstats_thread_kernel<int>, gridt, block, 0, s, a);
		} else if (avg < 160) {
			// short leaves (config 1 / config 5, ~100 nonzeros): 16 lanes per segment
			dim3 grid16((unsigned) ((a.nseg + 15) / 16));
			if (is_dbl) hipLaunchKernelGGL((colstats_kernel<double, 16, 16>), grid16, block, 0, s, a);
			else hipLaunchKernelGGL((colstats_kernel<int, 16, 16>), grid16, block, 0, s, a);
		} else if (avg >= 16) {
			if (is_dbl) hipLaunchKernelGGL((colstats_kernel<double, 64, 16>), grid, block, 0, s, a);
			else hipLaunchKernelGGL((colstats_kernel<int, 64, 16>), grid, block, 0, s, a);
		} else {
			if (is_dbl) hipLaunchKernelGGL((colstats_kernel<double, 64, 0>), grid, block, 0, s, a);
			else hipLaunchKernelGGL((colstats_kernel<int, 64, 0>), grid, block, 0, s, a);
		}
	}
	HIP_TRY(hipGetLastError());
	return 0;
}
