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
//   slab       = 16 consecutive columns = one register-indexable vector of
//                partial sums; a wavefront owns NV = CBW/16 consecutive slabs
//   tile(s, p) = the nonzeros of slab s in row panel p, in CSC order
//                (column-major, rows ascending), stored contiguously and
//                zero-padded to a multiple of PBC_BATCH records:
//                rc[i] = (local column << 16) | (row - p*R),  v[i] = value
//   tile order = (wavefront, panel, slab-within-wavefront): everything one
//                wavefront reads is one sequential stream
//   tile_ptr[t] = first record of tile t
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

#include <hipcub/hipcub.hpp>

typedef double d16 __attribute__((ext_vector_type(16)));

struct svt_dev_pbc {
	int64_t nrow, ncol, nnz, nrec;
	int CBW, WPB, logR;
	int64_t ngroups, nblocks, npanels;
	uint32_t *rc;
	double *v;
	int64_t *tile_ptr;     // [ngroups*npanels + 1]
	int *col_has_na;       // [ncol]
};

#define PCH 256            // panels per build chunk

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

#define PBC_BATCH 8        // records per scalar-load batch; tiles are padded to it

// One wavefront per (slab of 16 columns, chunk of PCH panels).  MODE 0: count
// the records of each tile (rounded up to PBC_BATCH).  MODE 1: write records to
// their final position and zero-fill the padding.
template <int MODE>
__global__ void __launch_bounds__(64)
pbc_pass_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
		const double *__restrict__ val, int64_t ncol, int NV, int logR,
		int64_t npanels, int64_t *__restrict__ counts_or_ptr,
		uint32_t *__restrict__ rc, double *__restrict__ v, int *__restrict__ col_has_na)
{
	__shared__ int64_t fill[PCH];
	const int lane = threadIdx.x;
	const int64_t slab = blockIdx.x;
	const int64_t wv = slab / NV;
	const int j = (int) (slab % NV);
	const int64_t p0 = (int64_t) blockIdx.y * PCH;
	const int64_t p1 = p0 + PCH < npanels ? p0 + PCH : npanels;
#define TILE_OF(P) ((wv * npanels + (P)) * NV + j)
	for (int i = lane; i < PCH; i += 64)
		fill[i] = (MODE == 1 && p0 + i < npanels) ? counts_or_ptr[TILE_OF(p0 + i)] : 0;
	__syncthreads();
	const int64_t c0 = slab * 16;
	const int64_t c1 = c0 + 16 < ncol ? c0 + 16 : ncol;
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
				rc[pos] = ((uint32_t) (c - c0) << 16) | (uint32_t) (r - (int32_t) ((p + p0) << logR));
				v[pos] = x;
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
			counts_or_ptr[TILE_OF(p0 + i)] = (fill[i] + PBC_BATCH - 1) / PBC_BATCH * PBC_BATCH;
		} else {
			// fill[i] = one past the last real record; pad up to the next tile
			const int64_t stop = counts_or_ptr[TILE_OF(p0 + i) + 1];
			for (int64_t q = fill[i]; q < stop; q++) { rc[q] = 0; v[q] = 0.0; }
		}
	}
#undef TILE_OF
}

extern "C" void svt_dev_pbc_release(svt_dev_pbc *h)
{
	if (h == NULL) return;
	if (h->rc) (void) hipFree(h->rc);
	if (h->v) (void) hipFree(h->v);
	if (h->tile_ptr) (void) hipFree(h->tile_ptr);
	if (h->col_has_na) (void) hipFree(h->col_has_na);
	free(h);
}

// Not on the launch path: allocates, synchronises.
extern "C" svt_dev_pbc *svt_dev_pbc_build(const svt_dev_csc *A, int CBW, int WPB, int logR)
{
	if (A->Rtype != SVT_REALSXP) {
		svt_set_error("svt_dev_pbc_build: f64 operands only");
		return NULL;
	}
	if (CBW <= 0 || CBW > 64 || (CBW & 15) || WPB <= 0 || WPB > 16 || logR < 4 || logR > 15) {
		svt_set_error("svt_dev_pbc_build: bad parameters");
		return NULL;
	}
	svt_dev_pbc *h = (svt_dev_pbc *) calloc(1, sizeof(*h));
	h->nrow = A->nrow; h->ncol = A->ncol; h->nnz = A->nnz;
	h->CBW = CBW; h->WPB = WPB; h->logR = logR;
	const int64_t CB = (int64_t) CBW * WPB;
	h->nblocks = (A->ncol + CB - 1) / CB;
	h->ngroups = h->nblocks * WPB * (CBW / 16);   // slabs of 16 columns
	h->npanels = (A->nrow + (1LL << logR) - 1) >> logR;
	if (h->npanels < 1) h->npanels = 1;
	const int64_t ntiles = h->ngroups * h->npanels;
	void *tmp = NULL;
	size_t tmp_bytes = 0;
	bool ok = hipMalloc((void **) &h->tile_ptr, (size_t) (ntiles + 1) * 8) == hipSuccess &&
		  hipMalloc((void **) &h->col_has_na, (size_t) (A->ncol > 0 ? A->ncol : 1) * 4) == hipSuccess;
	if (ok) ok = hipMemset(h->tile_ptr, 0, (size_t) (ntiles + 1) * 8) == hipSuccess &&
		     hipMemset(h->col_has_na, 0, (size_t) (A->ncol > 0 ? A->ncol : 1) * 4) == hipSuccess;
	if (ok && A->ncol > 0 && A->nnz > 0) {
		dim3 grid((unsigned) h->ngroups, (unsigned) ((h->npanels + PCH - 1) / PCH));
		hipLaunchKernelGGL(pbc_pass_kernel<0>, grid, dim3(64), 0, 0, A->col_ptr, A->row_idx,
				   (const double *) A->val, A->ncol, CBW / 16, logR, h->npanels,
				   h->tile_ptr, (uint32_t *) NULL, (double *) NULL, h->col_has_na);
		// exclusive scan in place over ntiles+1 entries (last entry = total)
		ok = hipcub::DeviceScan::ExclusiveSum(NULL, tmp_bytes, h->tile_ptr, h->tile_ptr,
						      (int) (ntiles + 1)) == hipSuccess &&
		     hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16) == hipSuccess &&
		     hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, h->tile_ptr, h->tile_ptr,
						      (int) (ntiles + 1)) == hipSuccess;
		int64_t nrec = 0;
		if (ok) ok = hipMemcpy(&nrec, h->tile_ptr + ntiles, 8, hipMemcpyDeviceToHost) == hipSuccess;
		h->nrec = nrec;
		if (ok) ok = hipMalloc((void **) &h->rc, (size_t) (nrec + PBC_BATCH) * 4) == hipSuccess &&
			     hipMalloc((void **) &h->v, (size_t) (nrec + PBC_BATCH) * 8) == hipSuccess;
		if (ok) {
			hipLaunchKernelGGL(pbc_pass_kernel<1>, grid, dim3(64), 0, 0, A->col_ptr,
					   A->row_idx, (const double *) A->val, A->ncol, CBW / 16, logR,
					   h->npanels, h->tile_ptr, h->rc, h->v, h->col_has_na);
			ok = hipDeviceSynchronize() == hipSuccess;
		}
	}
	if (tmp) (void) hipFree(tmp);
	if (!ok) {
		svt_set_error("svt_dev_pbc_build failed: %s", hipGetErrorString(hipGetLastError()));
		svt_dev_pbc_release(h);
		return NULL;
	}
	return h;
}

// ---------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------
struct PbcFlags {
	int *y_nonfinite;    // [1] any NaN/Inf/NA in the dense operand
};

// One batch of PBC_BATCH records applied to one slab's partial sums.  All
// record fields are wave-uniform (SGPRs); `acc[c]` with a uniform c compiles to
// s_set_gpr_idx + v_mov (register-indexed access, no scratch).
__device__ inline void apply_batch(d16 &acc, const uint32_t (&r)[PBC_BATCH],
				   const double (&a)[PBC_BATCH],
				   const double *__restrict__ ycol)
{
	double y[PBC_BATCH];
#pragma unroll
	for (int q = 0; q < PBC_BATCH; q++)
		y[q] = ycol[r[q] & 0xFFFFu];
#pragma unroll
	for (int q = 0; q < PBC_BATCH; q++) {
		const int c = (int) (r[q] >> 16);
		acc[c] = __builtin_fma(a[q], y[q], acc[c]);
	}
}

template <int NV, int WPB>
__global__ void __launch_bounds__(WPB * 64)
crossprod_pbc_kernel(const uint32_t *__restrict__ rc, const double *__restrict__ v,
		     const int64_t *__restrict__ tile_ptr, int64_t npanels, int logR,
		     const double *__restrict__ Y, int64_t ldY, int tr_y, int64_t nrow, int K,
		     int64_t ncol, int64_t panels_per_split, double *__restrict__ part,
		     int64_t Kp, PbcFlags fl)
{
	extern __shared__ double ylds[];            // [64][R + 1]
	const int R = 1 << logR;
	const int RS = R + 1;                       // odd stride (in doubles): conflict-free lane=k reads
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

	d16 acc[NV];
#pragma unroll
	for (int i = 0; i < NV; i++) acc[i] = 0.0;
	int bad = 0;
	const double *__restrict__ ycol = ylds + lane * RS;

	for (int64_t p = pa; p < pb; p++) {
		// tile bounds of this wavefront for panel p: NV+1 consecutive words
		const int64_t *__restrict__ tb = tile_ptr + (wv * npanels + p) * NV;
		int64_t bounds[NV + 1];
#pragma unroll
		for (int j = 0; j <= NV; j++) bounds[j] = tb[j];

		__syncthreads();                        // previous panel fully consumed
		// ---- stage Y[p*R .. p*R+R) x [k0 .. k0+64) into LDS -----------
		const int64_t r0 = p << logR;
		if (!tr_y) {
			// column-major Y: for one k, R consecutive rows are contiguous
			for (int idx = tid; idx < 64 * R; idx += WPB * 64) {
				const int kk = idx >> logR, rr = idx & (R - 1);
				const int64_t r = r0 + rr;
				double y = 0.0;
				if (r < nrow && k0 + kk < K) y = Y[r + (int64_t) (k0 + kk) * ldY];
				if (!svt_is_finite(y)) bad = 1;
				ylds[kk * RS + rr] = y;
			}
		} else {
			// Y given as K x nrow (rows of the product's dense operand contiguous)
			for (int idx = tid; idx < 64 * R; idx += WPB * 64) {
				const int rr = idx >> 6, kk = idx & 63;
				const int64_t r = r0 + rr;
				double y = 0.0;
				if (r < nrow && k0 + kk < K) y = Y[(k0 + kk) + r * ldY];
				if (!svt_is_finite(y)) bad = 1;
				ylds[kk * RS + rr] = y;
			}
		}
		__syncthreads();
		// ---- this wavefront's records, slab by slab ----------------------
#pragma unroll
		for (int j = 0; j < NV; j++) {
			for (int64_t i = bounds[j]; i < bounds[j + 1]; i += PBC_BATCH) {
				uint32_t r[PBC_BATCH];
				double a[PBC_BATCH];
#pragma unroll
				for (int q = 0; q < PBC_BATCH; q++) { r[q] = rc[i + q]; a[q] = v[i + q]; }
				apply_batch(acc[j], r, a, ycol);
			}
		}
	}
	if (b == 0 && __any(bad) && lane == 0)
		*fl.y_nonfinite = 1;
	// ---- partial results: part[(split*Kp + k) * ncol + c] -----------------
	const int64_t c0 = wv * (16 * NV);
	double *__restrict__ dst = part + ((int64_t) split * Kp + k0 + lane) * ncol;
#pragma unroll
	for (int i = 0; i < NV; i++)
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const int64_t c = c0 + i * 16 + j;
			if (c < ncol) dst[c] = acc[i][j];
		}
}

// out[c, k] = sum over splits (fixed order) ; NA_real_ for leaves holding an NA
__global__ void pbc_reduce_kernel(const double *__restrict__ part, int nsplit, int64_t Kp,
				  int K, int64_t ncol, const int *__restrict__ col_has_na,
				  double *__restrict__ out, int64_t sc, int64_t sk)
{
	const int64_t c = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const int k = blockIdx.y;
	if (c >= ncol || k >= K) return;
	double s = 0.0;
	for (int t = 0; t < nsplit; t++)
		s += part[((int64_t) t * Kp + k) * ncol + c];
	if (col_has_na[c]) s = svt_na_real();
	out[c * sc + (int64_t) k * sk] = s;
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
static int pick_nsplit(const svt_dev_pbc *P, int K)
{
	const int64_t kt = ((int64_t) K + 63) / 64;
	int64_t s = (512 + P->nblocks * kt - 1) / (P->nblocks * kt);   // aim for >= 512 workgroups
	s = (s + 7) / 8 * 8;                                           // whole XCD rounds
	if (s > P->npanels) s = P->npanels;
	if (s < 1) s = 1;
	return (int) s;
}

extern "C" size_t svt_dev_crossprod_pbc_ws_bytes(const svt_dev_pbc *P, int K)
{
	const int64_t Kp = ((int64_t) K + 63) / 64 * 64;
	const int ns = pick_nsplit(P, K);
	// [flags 256 B][partials][general-path workspace]
	return 256 + (size_t) ns * Kp * (P->ncol > 0 ? P->ncol : 1) * 8 +
	       crossprod_ws_bytes(P->nrow, P->ncol, K);
}

template <int NV, int WPB>
static void launch_main(const svt_dev_pbc *P, const double *Y, int64_t ldY, int tr_y, int K,
			int nsplit, int64_t pps, double *part, int64_t Kp, PbcFlags fl,
			hipStream_t s)
{
	const int R = 1 << P->logR;
	const size_t lds = (size_t) 64 * (R + 1) * 8;
	dim3 grid((unsigned) nsplit, (unsigned) (Kp / 64), (unsigned) P->nblocks);
	auto kern = crossprod_pbc_kernel<NV, WPB>;
	(void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	hipLaunchKernelGGL(kern, grid, dim3(WPB * 64), lds, s, P->rc, P->v, P->tile_ptr, P->npanels,
			   P->logR, Y, ldY, tr_y, P->nrow, K, P->ncol, pps, part, Kp, fl);
}

int launch_crossprod_general_if(const CrossprodArgs &a, const int *flag, hipStream_t s);

extern "C" int svt_dev_crossprod_pbc(const svt_dev_pbc *P, const svt_dev_csc *A, const double *Y,
				     int64_t ldY, int K, int tr_y, double *out,
				     int64_t out_stride_c, int64_t out_stride_k, void *ws,
				     size_t ws_bytes, void *stream)
{
	hipStream_t s = (hipStream_t) stream;
	if (P->ncol <= 0 || K <= 0)
		return 0;
	if (ws_bytes < svt_dev_crossprod_pbc_ws_bytes(P, K))
		return svt_set_error("svt_dev_crossprod_pbc: workspace too small");
	const int64_t Kp = ((int64_t) K + 63) / 64 * 64;
	const int nsplit = pick_nsplit(P, K);
	const int64_t pps = (P->npanels + nsplit - 1) / nsplit;
	PbcFlags fl;
	fl.y_nonfinite = (int *) ws;
	double *part = (double *) ((char *) ws + 256);
	void *gen_ws = (char *) part + (size_t) nsplit * Kp * P->ncol * 8;
	HIP_TRY(hipMemsetAsync(ws, 0, 256, s));
	const int key = P->CBW / 16 * 100 + P->WPB;
	switch (key) {
	case 116: launch_main<1, 16>(P, Y, ldY, tr_y, K, nsplit, pps, part, Kp, fl, s); break;
	case 216: launch_main<2, 16>(P, Y, ldY, tr_y, K, nsplit, pps, part, Kp, fl, s); break;
	case 316: launch_main<3, 16>(P, Y, ldY, tr_y, K, nsplit, pps, part, Kp, fl, s); break;
	case 208: launch_main<2, 8>(P, Y, ldY, tr_y, K, nsplit, pps, part, Kp, fl, s); break;
	case 308: launch_main<3, 8>(P, Y, ldY, tr_y, K, nsplit, pps, part, Kp, fl, s); break;
	case 408: launch_main<4, 8>(P, Y, ldY, tr_y, K, nsplit, pps, part, Kp, fl, s); break;
	case 404: launch_main<4, 4>(P, Y, ldY, tr_y, K, nsplit, pps, part, Kp, fl, s); break;
	default:
		return svt_set_error("svt_dev_crossprod_pbc: unsupported (CBW=%d, WPB=%d)", P->CBW, P->WPB);
	}
	dim3 rgrid((unsigned) ((P->ncol + 255) / 256), (unsigned) K);
	hipLaunchKernelGGL(pbc_reduce_kernel, rgrid, dim3(256), 0, s, part, nsplit, Kp, K, P->ncol,
			   P->col_has_na, out, out_stride_c, out_stride_k);
	HIP_TRY(hipGetLastError());
	// General (slow-path) semantics if the dense operand is not finite.
	CrossprodArgs a;
	memset(&a, 0, sizeof(a));
	a.col_ptr = A->col_ptr; a.row_idx = A->row_idx; a.val = A->val; a.Rtype = SVT_REALSXP;
	a.nrow = A->nrow; a.ncol = A->ncol; a.Y = Y; a.ldY = ldY; a.K = K; a.tr_y = tr_y;
	a.out = out; a.out_stride_c = out_stride_c; a.out_stride_k = out_stride_k;
	a.ws = gen_ws; a.ws_bytes = crossprod_ws_bytes(P->nrow, P->ncol, K);
	return launch_crossprod_general_if(a, fl.y_nonfinite, s);
}
