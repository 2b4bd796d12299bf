// Device transpose of a 2-d SVT in its CSC device layout (CSC -> CSC of t(A)),
// the device counterpart of transpose_2D_SVT (src/SparseArray_aperm.c:148-423):
// `%*%`, tcrossprod() and the non-native row* stats all start with t()
// (R/SparseMatrix-mult.R:165-206).  The reference makes 3 serial passes (count
// per row, allocate, scatter in column order); here: a stable LSD radix sort of
// (row index -> position) pairs, which keeps the entries of every output leaf in
// ascending column order as the SVT format requires (src/leaf_utils.h:12-15),
// then one gather pass.  Traffic ~ 3 sort passes x 16 B/nz + 28 B/nz.
// (Round 2 also built a two-pass form -- sort on row >> 4 with 16-bit keys, the low 4 bits riding in
// the payload, the order inside every 16-row bucket finished by one wavefront inside the gather
// (ballot ranks + running counters): correct, but 4.4 ms against 4.0: the two passes of 6-byte pairs
// take as long per pass as 8-byte ones (0.65 ms), and the finishing gather stores to 16 row segments
// at once instead of one coalesced run: 2.3 ms against 1.5 + 0.12.)
#include "svt_common.h"

#include <string.h>

#include "svt_scan.h"

#include <rocprim/rocprim.hpp>

// ---------------------------------------------------------------------------
// Exclusive scan of an int64 array (svt_scan.h)
// ---------------------------------------------------------------------------
__device__ inline int64_t wave_incl_scan_i64(int64_t v)
{
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int64_t t = __shfl_up(v, o, SVT_WAVE);
		if (lane >= o) v += t;
	}
	return v;
}

// tile of SVT_SCAN_TILE elements per workgroup: exclusive scan in place, tile total to sums[block]
__global__ void __launch_bounds__(SVT_SCAN_NT)
scan_tile_kernel(int64_t *__restrict__ data, int64_t n, int64_t *__restrict__ sums)
{
	__shared__ int64_t wtot[SVT_SCAN_NT / SVT_WAVE];
	const int64_t base = (int64_t) blockIdx.x * SVT_SCAN_TILE + (int64_t) threadIdx.x * SVT_SCAN_ITEMS;
	int64_t v[SVT_SCAN_ITEMS], run = 0;
#pragma unroll
	for (int u = 0; u < SVT_SCAN_ITEMS; u++) {
		v[u] = base + u < n ? data[base + u] : 0;
		const int64_t t = v[u];
		v[u] = run;                                     // exclusive within the thread
		run += t;
	}
	const int64_t incl = wave_incl_scan_i64(run);
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (lane == 63) wtot[w] = incl;
	__syncthreads();
	int64_t off = incl - run;
	for (int i = 0; i < w; i++) off += wtot[i];
#pragma unroll
	for (int u = 0; u < SVT_SCAN_ITEMS; u++)
		if (base + u < n) data[base + u] = v[u] + off;
	if (threadIdx.x == SVT_SCAN_NT - 1 && sums != NULL) sums[blockIdx.x] = off + run;
}

__global__ void __launch_bounds__(SVT_SCAN_NT)
scan_add_kernel(int64_t *__restrict__ data, int64_t n, const int64_t *__restrict__ sums)
{
	const int64_t add = sums[blockIdx.x];
	const int64_t base = (int64_t) blockIdx.x * SVT_SCAN_TILE + (int64_t) threadIdx.x * SVT_SCAN_ITEMS;
#pragma unroll
	for (int u = 0; u < SVT_SCAN_ITEMS; u++)
		if (base + u < n) data[base + u] += add;
}

int launch_exclusive_scan_i64(int64_t *data, int64_t n, void *ws, hipStream_t s)
{
	if (n <= 0)
		return 0;
	const int64_t nb = (n + SVT_SCAN_TILE - 1) / SVT_SCAN_TILE;
	if (nb > 0x7FFFFFFFLL)
		return svt_set_error("exclusive scan: too many elements");
	int64_t *sums = (int64_t *) ws;
	hipLaunchKernelGGL(scan_tile_kernel, dim3((unsigned) nb), dim3(SVT_SCAN_NT), 0, s, data, n, nb > 1 ? sums : (int64_t *) NULL);
	if (nb > 1) {
		if (launch_exclusive_scan_i64(sums, nb, (char *) ws + ((size_t) nb * 8 + 255) / 256 * 256, s))
			return -1;
		hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned) nb), dim3(SVT_SCAN_NT), 0, s, data, n, sums);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// ---------------------------------------------------------------------------
// t(A): bucketed transposition (round 3; replaces the radix sort of (row, position) pairs of rounds 1-2).
//
// What the reference does serially -- count per row, allocate, scatter column by column
// (src/SparseArray_aperm.c:148-393) -- is stable because it visits the columns in order.  The parallel
// form below keeps that order without a general-purpose sort by using what a CSC operand gives for free:
// the rows inside a column ascend (src/leaf_utils.h:12-15), so the nonzeros of one column that fall into
// a given range of rows are one contiguous run, found by two binary searches.
//
//   fine bucket   = F consecutive rows (F = 64, 128 or 256: ~10^4 nonzeros), the unit of the last pass
//   coarse bucket = 16 fine buckets
//   group         = 512 consecutive columns, one thread per column
//
//   pass 1 (count)   workgroup (group g, coarse bucket i): every thread finds its column's run of rows in
//                    the coarse bucket and counts it by fine bucket; table[fine bucket][g] = the group's
//                    total (16 numbers per workgroup).
//   scan             exclusive scan of the table, fine-bucket-major: where the piece (fine bucket, g)
//                    starts in an intermediate array ordered (fine bucket, group, column, row).
//   pass 2 (scatter) the same workgroups copy their runs to those pieces: position = piece start + the
//                    nonzeros of the same fine bucket in the group's earlier columns (a scan over the 512
//                    threads, per fine bucket) + the index inside the column's run.  Every piece is a few
//                    KB written by one workgroup within microseconds.
//   pass 3 (finish)  one workgroup per fine bucket: its ~10^4 nonzeros are contiguous and already in column
//                    order; a stable counting sort on the F rows (ranks inside a wavefront from ballots of
//                    the row bits, across wavefronts from a small LDS table) puts them in (row, column) order
//                    at their final place, and gives the row pointers.
//
// Traffic at 1e8 nonzeros (f64): pass 1 reads the offsets (0.4 GB), pass 2 reads 1.2 GB and writes 1.3 GB
// (column 4 B + row-in-bucket 1 B + value 8 B), pass 3 reads 1.3 GB and writes 1.2 GB: 5.4 GB against
// 7.6 GB for the sort-and-gather form, and no pass scatters single elements across the whole array.
// Shapes it does not fit (rows with more than ~500 nonzeros on average, or less than one nonzero per
// column and coarse bucket) take the key sort of launch_transpose_sorted() below.
// ---------------------------------------------------------------------------
#define T2_NT 512                 // columns per group = threads per workgroup of passes 1-2
#define T2_NFINE 16               // fine buckets per coarse bucket

struct T2Shape {
	int fbits;                // log2(F)
	int64_t nfb;              // fine buckets
	int64_t ncoarse;
	int64_t ngroups;
};

// first position in [lo, hi) whose row is >= r
__device__ inline int64_t t2_lower_bound(const int32_t *__restrict__ row_idx, int64_t lo, int64_t hi, int64_t r)
{
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if ((int64_t) row_idx[mid] < r) lo = mid + 1; else hi = mid;
	}
	return lo;
}

template <typename T, bool SCATTER>
__global__ void __launch_bounds__(T2_NT)
transpose_bucket_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
			const T *__restrict__ val, int64_t nrow, int64_t ncol, T2Shape sh,
			int64_t *__restrict__ table, int32_t *__restrict__ col1, uint8_t *__restrict__ rlow1,
			T *__restrict__ val1)
{
	__shared__ uint32_t cnt[T2_NFINE][T2_NT];       // [fine bucket][thread]: count, then exclusive prefix over the threads
	__shared__ int64_t piece[T2_NFINE];
	const int t = threadIdx.x, lane = t & 63, w = t >> 6;
	const int64_t g = (int64_t) blockIdx.x % sh.ngroups, i = (int64_t) blockIdx.x / sh.ngroups;
	const int64_t r_lo = (i * T2_NFINE) << sh.fbits;
	int64_t r_hi = ((i + 1) * T2_NFINE) << sh.fbits;
	if (r_hi > nrow) r_hi = nrow;
	const int64_t c = g * T2_NT + t;
	int64_t a = 0, b = 0;
	if (c < ncol) {
		const int64_t beg = col_ptr[c], end = col_ptr[c + 1];
		a = t2_lower_bound(row_idx, beg, end, r_lo);
		b = t2_lower_bound(row_idx, a, end, r_hi);
	}
#pragma unroll
	for (int s = 0; s < T2_NFINE; s++) cnt[s][t] = 0;
	for (int64_t k = a; k < b; k++)                 // (only this thread touches column t of cnt)
		cnt[((int64_t) row_idx[k] - r_lo) >> sh.fbits][t]++;
	__syncthreads();
	// exclusive prefix over the 512 threads, two fine buckets per wavefront
	for (int s = 2 * w; s < 2 * w + 2; s++) {
		uint32_t carry = 0;
		for (int j = 0; j < T2_NT / 64; j++) {
			const uint32_t v = cnt[s][j * 64 + lane];
			uint32_t incl = v;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const uint32_t u = __shfl_up(incl, o, SVT_WAVE);
				if (lane >= o) incl += u;
			}
			cnt[s][j * 64 + lane] = carry + incl - v;
			carry += __shfl(incl, 63, SVT_WAVE);
		}
		const int64_t fb = i * T2_NFINE + s;
		if (lane == 0 && fb < sh.nfb) {
			if (!SCATTER) table[fb * sh.ngroups + g] = carry;
			else piece[s] = table[fb * sh.ngroups + g];
		}
	}
	if (!SCATTER)
		return;
	__syncthreads();
	for (int64_t k = a; k < b; k++) {
		const int64_t rr = (int64_t) row_idx[k] - r_lo;
		const int s = (int) (rr >> sh.fbits);
		const int64_t dst = piece[s] + cnt[s][t]++;
		col1[dst] = (int32_t) c;
		rlow1[dst] = (uint8_t) (rr & (((int64_t) 1 << sh.fbits) - 1));
		val1[dst] = val[k];
	}
}

#define T3_NT 1024
template <typename T>
__global__ void __launch_bounds__(T3_NT)
transpose_finish_kernel(const int64_t *__restrict__ table, T2Shape sh, int64_t nrow, int64_t nnz,
			const int32_t *__restrict__ col1, const uint8_t *__restrict__ rlow1,
			const T *__restrict__ val1, int64_t *__restrict__ out_ptr,
			int32_t *__restrict__ out_idx, T *__restrict__ out_val)
{
	__shared__ int32_t start[256], run[2][256];     // per row of the bucket: first output slot, slots used so far
	__shared__ int32_t wcnt[2][T3_NT / 64][256];    // per wavefront and row: entries of the current chunk
	// (run and wcnt alternate between two copies from chunk to chunk: two barriers per chunk)
	const int t = threadIdx.x, lane = t & 63, w = t >> 6;
	const int F = 1 << sh.fbits;
	const int64_t fb = blockIdx.x;
	const int64_t b0 = table[fb * sh.ngroups], b1 = fb + 1 < sh.nfb ? table[(fb + 1) * sh.ngroups] : nnz;
	const int64_t n = b1 - b0;
	if (t < F) { start[t] = 0; run[0][t] = 0; }
	for (int x = t; x < 2 * (T3_NT / 64) * 256; x += T3_NT) (&wcnt[0][0][0])[x] = 0;
	__syncthreads();
	for (int64_t e = t; e < n; e += T3_NT) atomicAdd(&start[rlow1[b0 + e]], 1);
	__syncthreads();
	if (w == 0) {                                   // exclusive scan of the F (<= 256) counts, 4 per lane
		int32_t v[4], tot = 0;
#pragma unroll
		for (int u = 0; u < 4; u++) { v[u] = lane * 4 + u < F ? start[lane * 4 + u] : 0; tot += v[u]; }
		int32_t incl = tot;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int32_t x = __shfl_up(incl, o, SVT_WAVE);
			if (lane >= o) incl += x;
		}
		int32_t off = incl - tot;
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int r = lane * 4 + u;
			if (r < F) {
				start[r] = off;
				const int64_t row = (fb << sh.fbits) + r;
				if (row < nrow) out_ptr[row] = b0 + off;
			}
			off += v[u];
		}
	}
	if (fb == sh.nfb - 1 && t == 0) out_ptr[nrow] = nnz;
	__syncthreads();
	const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	int p = 0;
	for (int64_t c0 = 0; c0 < n; c0 += T3_NT, p ^= 1) {
		const int64_t e = c0 + t;
		const bool valid = e < n;
		const int r = valid ? (int) rlow1[b0 + e] : 0;
		// lanes of this wavefront that hold the same row (and are valid): one ballot per row bit
		uint64_t peers = __ballot(valid);
		for (int bit = 0; bit < sh.fbits; bit++) {
			const uint64_t m = __ballot((r >> bit) & 1);
			peers &= ((r >> bit) & 1) ? m : ~m;
		}
		const int rank = __popcll(peers & lt);
		if (valid && rank == 0) wcnt[p][w][r] = __popcll(peers);
		__syncthreads();
		if (valid) {
			int32_t before = 0;
			for (int ww = 0; ww < w; ww++) before += wcnt[p][ww][r];
			const int64_t dst = b0 + start[r] + run[p][r] + before + rank;
			out_idx[dst] = col1[b0 + e];
			out_val[dst] = val1[b0 + e];
		}
		if (t < F) {
			int32_t tot = 0;
			for (int ww = 0; ww < T3_NT / 64; ww++) tot += wcnt[p][ww][t];
			run[p ^ 1][t] = run[p][t] + tot;
		}
		// the other copy of wcnt (read in the previous round, before the barrier above) is clear again for the next
		for (int x = t; x < (T3_NT / 64) * 256; x += T3_NT) (&wcnt[p ^ 1][0][0])[x] = 0;
		__syncthreads();
	}
}

// the bucketed form applies to this operand: fills *sh
static bool t2_shape(int64_t nrow, int64_t ncol, int64_t nnz, T2Shape *sh)
{
	if (nrow <= 0 || ncol <= 0 || nnz <= 0)
		return false;
	const double per_row = (double) nnz / (double) nrow, per_col = (double) nnz / (double) ncol;
	int fbits = -1;
	for (int fb = 8; fb >= 6; fb--)                 // the largest bucket of <= 32768 nonzeros
		if (ldexp(per_row, fb) <= 32768.0) { fbits = fb; break; }
	if (fbits < 0)
		return false;
	const double coarse_rows = ldexp((double) T2_NFINE, fbits);
	if (per_col * coarse_rows / (double) nrow < 1.0)        // less than one nonzero per thread of passes 1-2
		return false;
	sh->fbits = fbits;
	sh->nfb = (nrow + ((int64_t) 1 << fbits) - 1) >> fbits;
	sh->ncoarse = (sh->nfb + T2_NFINE - 1) / T2_NFINE;
	sh->ngroups = (ncol + T2_NT - 1) / T2_NT;
	const double ntab = (double) sh->nfb * (double) sh->ngroups, nwg = (double) sh->ncoarse * (double) sh->ngroups;
	return ntab < 1.0e8 && nwg < 2.0e9;
}

static size_t t2_a(size_t n, size_t esz) { return (n * esz + 255) / 256 * 256; }

// ---- fallback: stable LSD radix sort of (row -> position) pairs, then a gather (rounds 1-2; any shape) ----
__global__ void iota_u32_kernel(uint32_t *__restrict__ p, int64_t n)
{
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = (uint32_t) i;
}

// out_ptr[r] = first position of row r in the sorted order (lower bound)
__global__ void row_bounds_kernel(const int32_t *__restrict__ sorted_rows, int64_t nnz,
				  int64_t nrow, int64_t *__restrict__ out_ptr)
{
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (r > nrow) return;
	int64_t lo = 0, hi = nnz;
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if ((int64_t) sorted_rows[mid] < r) lo = mid + 1; else hi = mid;
	}
	out_ptr[r] = lo;
}

// Which chunk of 256 outputs a workgroup takes.  Workgroups are dealt round-robin over the 8 XCDs
// (MI355X_MICROARCH.md, Workgroup dispatch): with chunk = blockIdx the 8 entries of one 64-byte
// sector of `val` -- 8 consecutive nonzeros of a column, i.e. 8 output rows ~100 rows apart -- are
// gathered by workgroups of 8 different XCDs and every XCD's L2 fetches the sector for itself.
// Here XCD x walks the x-th eighth of the output in order, so a sector is fetched by one L2 and
// its other entries hit it (2.9 -> ~1 ms for the gather of t(A) at 1e8 nonzeros).
__device__ inline int64_t xcd_chunk(int64_t nchunks)
{
	const int64_t b = blockIdx.x, per = (nchunks + 7) / 8;
	return (b & 7) * per + (b >> 3);
}

// Column of source position k without a search over all of col_ptr: hint[b] = the column that
// holds position b << HINT_SHIFT.  A full binary search per output costs ~8 L2 requests on the same
// few col_ptr lines from every CU (the gather ran at the L2's request rate: 5.6e8 requests in 3 ms,
// 97 % hits, profiles/r02_transpose_*); with the hint the search runs over the handful of columns
// that start inside one block of 256 positions.
#define HINT_SHIFT 8
__global__ void col_hint_kernel(const int64_t *__restrict__ col_ptr, int64_t ncol, int64_t nblk,
				uint32_t *__restrict__ hint)
{
	const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (b > nblk) return;
	const int64_t k = b << HINT_SHIFT;
	int64_t lo = 0, hi = ncol;                  // last c with col_ptr[c] <= k (ncol if k is past the end)
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (col_ptr[mid] <= k) lo = mid; else hi = mid - 1;
	}
	hint[b] = (uint32_t) lo;
}

// (4 outputs per thread with streamed loads / stores: 2.0 ms against 1.5 ms for this form)
template <typename T>
__global__ void transpose_gather_kernel(const int64_t *__restrict__ col_ptr, int64_t ncol,
					const T *__restrict__ val, const uint32_t *__restrict__ perm,
					const uint32_t *__restrict__ hint,
					int64_t nnz, int32_t *__restrict__ out_idx, T *__restrict__ out_val)
{
	const int64_t i = xcd_chunk((nnz + blockDim.x - 1) / blockDim.x) * blockDim.x + threadIdx.x;
	if (i >= nnz) return;
	const int64_t k = perm[i];
	// column of position k: last c with col_ptr[c] <= k, between the hints of k's block and the next
	int64_t lo = hint[k >> HINT_SHIFT], hi = hint[(k >> HINT_SHIFT) + 1];
	if (hi > ncol - 1) hi = ncol - 1;
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (col_ptr[mid] <= k) lo = mid; else hi = mid - 1;
	}
	out_idx[i] = (int32_t) lo;
	out_val[i] = val[k];
}

static size_t sort_tmp_bytes(int64_t nnz, int end_bit)
{
	size_t b = 0;
	(void) rocprim::radix_sort_pairs(NULL, b, (const int32_t *) NULL, (int32_t *) NULL,
					 (const uint32_t *) NULL, (uint32_t *) NULL,
					 (size_t) nnz, 0u, (unsigned) end_bit);
	return b;
}

static int key_bits(int64_t nrow)
{
	int b = 1;
	while (b < 31 && ((int64_t) 1 << b) < nrow) b++;
	return b;
}

static size_t hint_bytes(int64_t nnz)
{
	return ((size_t) ((nnz >> HINT_SHIFT) + 2) * 4 + 255) / 256 * 256;
}

// [sorted rows nnz*4][positions nnz*4][sorted positions nnz*4][column hints][radix-sort temp]
static size_t transpose_sorted_ws_bytes(int64_t nrow, int64_t nnz)
{
	const size_t a = ((size_t) (nnz > 0 ? nnz : 1) * 4 + 255) / 256 * 256;
	return 3 * a + hint_bytes(nnz) + sort_tmp_bytes(nnz, key_bits(nrow)) + 256;
}

static int launch_transpose_sorted(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
		     int64_t nrow, int64_t ncol, int64_t nnz, int64_t *out_ptr, int32_t *out_idx,
		     void *out_val, void *ws, hipStream_t s)
{
	if (nnz >= ((int64_t) 1 << 31))
		return svt_set_error("svt_dev_transpose: more than 2^31-1 nonzeros");
	const unsigned nbr = (unsigned) ((nrow + 1 + 255) / 256);
	if (nnz == 0) {
		HIP_TRY(hipMemsetAsync(out_ptr, 0, (size_t) (nrow + 1) * 8, s));
		return 0;
	}
	const size_t a = ((size_t) nnz * 4 + 255) / 256 * 256;
	int32_t *srows = (int32_t *) ws;
	uint32_t *pos = (uint32_t *) ((char *) ws + a);
	uint32_t *perm = (uint32_t *) ((char *) ws + 2 * a);
	uint32_t *hint = (uint32_t *) ((char *) ws + 3 * a);
	void *tmp = (char *) ws + 3 * a + hint_bytes(nnz);
	const int bits = key_bits(nrow);
	size_t tb = sort_tmp_bytes(nnz, bits);
	const unsigned nb = (unsigned) ((nnz + 255) / 256);
	const unsigned nb8 = (unsigned) (((nnz + 255) / 256 + 7) / 8 * 8);      // whole rounds over the 8 XCDs (xcd_chunk)
	hipLaunchKernelGGL(iota_u32_kernel, dim3(nb), dim3(256), 0, s, pos, nnz);
	const int64_t nblk = (nnz >> HINT_SHIFT) + 1;
	hipLaunchKernelGGL(col_hint_kernel, dim3((unsigned) ((nblk + 1 + 255) / 256)), dim3(256), 0, s, col_ptr, ncol, nblk, hint);
	HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, row_idx, srows, pos, perm, (size_t) nnz, 0u, (unsigned) bits, s));
	hipLaunchKernelGGL(row_bounds_kernel, dim3(nbr), dim3(256), 0, s, srows, nnz, nrow, out_ptr);
	if (Rtype == SVT_REALSXP)
		hipLaunchKernelGGL(transpose_gather_kernel<double>, dim3(nb8), dim3(256), 0, s, col_ptr, ncol,
				   (const double *) val, perm, hint, nnz, out_idx, (double *) out_val);
	else
		hipLaunchKernelGGL(transpose_gather_kernel<int32_t>, dim3(nb8), dim3(256), 0, s, col_ptr, ncol,
				   (const int32_t *) val, perm, hint, nnz, out_idx, (int32_t *) out_val);
	HIP_TRY(hipGetLastError());
	return 0;
}


// workspace of the bucketed form: [reserve: table (nfb * ngroups + 1) * 8, scan scratch][columns nnz*4][rows in
// bucket nnz][values nnz*8]
// Head of the workspace kept for the bucket table and its scan scratch.  The caller sizes the workspace
// from (nrow, nnz) alone; the shapes the bucketed form accepts have at most ~nnz / 8 + nrow / 64 table
// entries (at least one nonzero per column and coarse bucket), and never more than 64 MiB are set aside:
// a larger table sends the operand through the key sort.
static size_t t2_reserve(int64_t nrow, int64_t nnz)
{
	const double per_row = nrow > 0 ? (double) nnz / (double) nrow : 0.0;
	const double ntab = (double) nnz / 8.0 + (double) nrow / 64.0 + 8.0 * per_row + 16.0;
	const double b = 2.0 * 8.0 * ntab + 8192.0;
	const size_t cap = (size_t) 64 << 20;
	return b >= (double) cap ? cap : ((size_t) b + 255) / 256 * 256;
}

size_t transpose_ws_bytes(int64_t nrow, int64_t nnz)
{
	const size_t n = (size_t) (nnz > 0 ? nnz : 1);
	const size_t sorted = transpose_sorted_ws_bytes(nrow, nnz);
	const size_t b2 = t2_reserve(nrow, nnz) + t2_a(n, 4) + t2_a(n, 1) + t2_a(n, 8) + 512;
	return sorted > b2 ? sorted : b2;
}

int launch_transpose(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
		     int64_t nrow, int64_t ncol, int64_t nnz, int64_t *out_ptr, int32_t *out_idx,
		     void *out_val, void *ws, hipStream_t s)
{
	if (nnz >= ((int64_t) 1 << 31))
		return svt_set_error("svt_dev_transpose: more than 2^31-1 nonzeros");
	if (nnz == 0) {
		HIP_TRY(hipMemsetAsync(out_ptr, 0, (size_t) (nrow + 1) * 8, s));
		return 0;
	}
	T2Shape sh;
	const size_t reserve = t2_reserve(nrow, nnz);
	if (!t2_shape(nrow, ncol, nnz, &sh) ||
	    t2_a((size_t) (sh.nfb * sh.ngroups + 1), 8) + exclusive_scan_ws_bytes(sh.nfb * sh.ngroups + 1) > reserve)
		return launch_transpose_sorted(col_ptr, row_idx, val, Rtype, nrow, ncol, nnz, out_ptr, out_idx, out_val, ws, s);
	const int64_t ntab = sh.nfb * sh.ngroups + 1;
	char *p = (char *) ws;
	int64_t *table = (int64_t *) p;            p += t2_a((size_t) ntab, 8);
	void *scan_ws = p;                         p += exclusive_scan_ws_bytes(ntab);
	p = (char *) ws + reserve;
	int32_t *col1 = (int32_t *) p;             p += t2_a((size_t) nnz, 4);
	uint8_t *rlow1 = (uint8_t *) p;            p += t2_a((size_t) nnz, 1);
	void *val1 = p;
	const unsigned nwg = (unsigned) (sh.ngroups * sh.ncoarse);
	// (fine buckets past the last row of the last coarse bucket are never counted: clear the table)
	HIP_TRY(hipMemsetAsync(table, 0, (size_t) ntab * 8, s));
	if (Rtype == SVT_REALSXP) {
		hipLaunchKernelGGL((transpose_bucket_kernel<double, false>), dim3(nwg), dim3(T2_NT), 0, s, col_ptr, row_idx,
				   (const double *) val, nrow, ncol, sh, table, col1, rlow1, (double *) val1);
	} else {
		hipLaunchKernelGGL((transpose_bucket_kernel<int32_t, false>), dim3(nwg), dim3(T2_NT), 0, s, col_ptr, row_idx,
				   (const int32_t *) val, nrow, ncol, sh, table, col1, rlow1, (int32_t *) val1);
	}
	if (launch_exclusive_scan_i64(table, ntab, scan_ws, s))
		return -1;
	if (Rtype == SVT_REALSXP) {
		hipLaunchKernelGGL((transpose_bucket_kernel<double, true>), dim3(nwg), dim3(T2_NT), 0, s, col_ptr, row_idx,
				   (const double *) val, nrow, ncol, sh, table, col1, rlow1, (double *) val1);
		hipLaunchKernelGGL(transpose_finish_kernel<double>, dim3((unsigned) sh.nfb), dim3(T3_NT), 0, s, table, sh, nrow,
				   nnz, col1, rlow1, (const double *) val1, out_ptr, out_idx, (double *) out_val);
	} else {
		hipLaunchKernelGGL((transpose_bucket_kernel<int32_t, true>), dim3(nwg), dim3(T2_NT), 0, s, col_ptr, row_idx,
				   (const int32_t *) val, nrow, ncol, sh, table, col1, rlow1, (int32_t *) val1);
		hipLaunchKernelGGL(transpose_finish_kernel<int32_t>, dim3((unsigned) sh.nfb), dim3(T3_NT), 0, s, table, sh, nrow,
				   nnz, col1, rlow1, (const int32_t *) val1, out_ptr, out_idx, (int32_t *) out_val);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// ---------------------------------------------------------------------------
// N-d aperm (C_aperm_SVT, src/SparseArray_aperm.c:148-930).  The device layout
// knows leaves only, so the caller passes the array's dims.  Every nonzero gets
// the 64-bit key  new_leaf * new_dim0 + new_row  (its linear index in the
// permuted array), one radix sort over ceil(log2(prod(dim))) bits orders them
// the way the permuted SVT stores them, one pass gathers.  The reference
// distinguishes leaf-preserving permutations (perm[1] == 1: pointer shuffle,
// :949-957) from those that shatter leaves (counting sort, :892-929); the key
// sort covers both.
// ---------------------------------------------------------------------------
struct ApermDims {
	int ndim;
	int64_t dim[8];      // old dims
	int perm[8];         // 0-based: new axis a takes old axis perm[a]
	int64_t mul[8];      // multiplier of old axis a in the new linear index
};

__global__ void aperm_key_kernel(const int64_t *__restrict__ col_ptr,
				 const int32_t *__restrict__ row_idx, int64_t ncol, int64_t nnz,
				 ApermDims d, unsigned long long *__restrict__ keys,
				 uint32_t *__restrict__ pos)
{
	const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nnz) return;
	int64_t lo = 0, hi = ncol;                  // leaf of position k
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (col_ptr[mid] <= k) lo = mid; else hi = mid - 1;
	}
	unsigned long long key = (unsigned long long) row_idx[k] * (unsigned long long) d.mul[0];
	int64_t rest = lo;
	for (int a = 1; a < d.ndim; a++) {
		const int64_t ia = rest % d.dim[a];
		rest /= d.dim[a];
		key += (unsigned long long) ia * (unsigned long long) d.mul[a];
	}
	keys[k] = key;
	pos[k] = (uint32_t) k;
}

__global__ void aperm_bounds_kernel(const unsigned long long *__restrict__ skeys, int64_t nnz,
				    int64_t nleaves, int64_t dim0, int64_t *__restrict__ out_ptr)
{
	const int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (j > nleaves) return;
	const unsigned long long key = (unsigned long long) j * (unsigned long long) dim0;
	int64_t lo = 0, hi = nnz;
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if (skeys[mid] < key) lo = mid + 1; else hi = mid;
	}
	out_ptr[j] = lo;
}

template <typename T>
__global__ void aperm_gather_kernel(const unsigned long long *__restrict__ skeys,
				    const uint32_t *__restrict__ perm, const T *__restrict__ val,
				    int64_t nnz, int64_t dim0, int32_t *__restrict__ out_idx,
				    T *__restrict__ out_val)
{
	const int64_t i = xcd_chunk((nnz + blockDim.x - 1) / blockDim.x) * blockDim.x + threadIdx.x;
	if (i >= nnz) return;
	out_idx[i] = (int32_t) (skeys[i] % (unsigned long long) dim0);
	out_val[i] = val[perm[i]];
}

// ---- leaf-preserving permutations (perm[1] == 1 in R's numbering): the leaves stay what they are and
// only change places -- the reference shuffles pointers (src/SparseArray_aperm.c:949-957); here: the
// length of every new leaf from the old col_ptr, a scan, one streaming copy.  No keys, no sort.
struct LeafMap {
	int ndim;
	int64_t new_dim[8];      // extents of the new axes 1 .. ndim-1
	int64_t old_stride[8];   // leaf stride, in the OLD layout, of the old axis that new axis a takes
};

__device__ inline int64_t old_leaf_of(const LeafMap &m, int64_t jn)
{
	int64_t j = 0;
	for (int a = 1; a < m.ndim; a++) {
		const int64_t ia = jn % m.new_dim[a];
		jn /= m.new_dim[a];
		j += ia * m.old_stride[a];
	}
	return j;
}

__global__ void aperm_leaf_count_kernel(const int64_t *__restrict__ col_ptr, int64_t nleaves, LeafMap m,
					int64_t *__restrict__ cnt)
{
	const int64_t jn = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (jn > nleaves) return;
	if (jn == nleaves) { cnt[jn] = 0; return; }
	const int64_t j = old_leaf_of(m, jn);
	cnt[jn] = col_ptr[j + 1] - col_ptr[j];
}

// One wavefront per four new leaves, their bounds and their runs in flight together (a leaf at a time is
// a chain of three memory latencies per ~100 nonzeros: 1.1 ms for the 3 GB of BASELINE config 5).
#define APERM_LU 4
template <typename T>
__global__ void __launch_bounds__(256)
aperm_leaf_copy_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
		       const T *__restrict__ val, int64_t nleaves, LeafMap m,
		       const int64_t *__restrict__ out_ptr, int32_t *__restrict__ out_idx, T *__restrict__ out_val)
{
	const int lane = threadIdx.x & 63;
	const int64_t j0 = ((int64_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * APERM_LU;
	if (j0 >= nleaves) return;
	int64_t src[APERM_LU], dst[APERM_LU], n[APERM_LU];
#pragma unroll
	for (int u = 0; u < APERM_LU; u++) {
		const int64_t jn = j0 + u;
		src[u] = dst[u] = n[u] = 0;
		if (jn < nleaves) {
			src[u] = col_ptr[old_leaf_of(m, jn)];
			dst[u] = out_ptr[jn];
			n[u] = out_ptr[jn + 1] - dst[u];
		}
	}
	bool more = true;
	for (int64_t k = lane; more; k += 64) {
		int32_t r[APERM_LU];
		T v[APERM_LU];
#pragma unroll
		for (int u = 0; u < APERM_LU; u++)
			if (k < n[u]) { r[u] = row_idx[src[u] + k]; v[u] = val[src[u] + k]; }
		more = false;
#pragma unroll
		for (int u = 0; u < APERM_LU; u++)
			if (k < n[u]) {
				out_idx[dst[u] + k] = r[u];
				out_val[dst[u] + k] = v[u];
				more |= k + 64 < n[u];
			}
		more = __any(more);
	}
}

// ---- leaf-shattering permutations, 32-bit keys.  A stable sort by the NEW LEAF index alone is enough:
// two nonzeros of one new leaf differ only in the coordinate that becomes the new row, and the input order
// (old leaves in order, offsets ascending) already ascends in that coordinate for fixed other ones.  So the
// key is ceil(log2(new leaves)) bits instead of ceil(log2(prod(dim))): 29 instead of 35 at BASELINE config 5,
// 32-bit keys, one radix pass less, and the new row travels beside the sort instead of through it.
__global__ void aperm_key32_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
				   const uint32_t *__restrict__ hint, int64_t ncol, int64_t nnz, ApermDims d,
				   uint32_t *__restrict__ keys, uint32_t *__restrict__ pos, int32_t *__restrict__ newrow)
{
	const int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nnz) return;
	int64_t lo = hint[k >> HINT_SHIFT], hi = hint[(k >> HINT_SHIFT) + 1];      // old leaf of position k
	if (hi > ncol - 1) hi = ncol - 1;
	while (lo < hi) {
		const int64_t mid = (lo + hi + 1) >> 1;
		if (col_ptr[mid] <= k) lo = mid; else hi = mid - 1;
	}
	// d.mul[a] = multiplier of old axis a in the new linear index; new leaf = (linear - new row) / new dim0
	const int64_t r = row_idx[k];
	unsigned long long lin = (unsigned long long) r * (unsigned long long) d.mul[0];
	int64_t nr = d.perm[0] == 0 ? r : 0, rest = lo;
	for (int a = 1; a < d.ndim; a++) {
		const int64_t ia = rest % d.dim[a];
		rest /= d.dim[a];
		lin += (unsigned long long) ia * (unsigned long long) d.mul[a];
		if (d.perm[0] == a) nr = ia;
	}
	keys[k] = (uint32_t) ((lin - (unsigned long long) nr) / (unsigned long long) d.dim[d.perm[0]]);
	pos[k] = (uint32_t) k;
	newrow[k] = (int32_t) nr;
}

// out_ptr[j] = first sorted position whose leaf is >= j: every sorted element fills the leaves between its
// predecessor's and its own (most new leaves are empty when leaves shatter: a search per leaf costs more)
__global__ void aperm_ptr_fill_kernel(const uint32_t *__restrict__ skeys, int64_t nnz, int64_t nleaves,
				      int64_t *__restrict__ out_ptr)
{
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i > nnz) return;
	const int64_t from = i == 0 ? 0 : (int64_t) skeys[i - 1] + 1;
	const int64_t to = i == nnz ? nleaves : (int64_t) skeys[i];
	for (int64_t j = from; j <= to; j++) out_ptr[j] = i;
}

template <typename T>
__global__ void aperm_gather32_kernel(const uint32_t *__restrict__ spos, const int32_t *__restrict__ newrow,
				      const T *__restrict__ val, int64_t nnz, int32_t *__restrict__ out_idx,
				      T *__restrict__ out_val)
{
	const int64_t i = xcd_chunk((nnz + blockDim.x - 1) / blockDim.x) * blockDim.x + threadIdx.x;
	if (i >= nnz) return;
	const uint32_t k = spos[i];
	out_idx[i] = newrow[k];
	out_val[i] = val[k];
}

static int aperm_bits(const int64_t *dim, int ndim)
{
	double tot = 1.0;
	for (int a = 0; a < ndim; a++) tot *= (double) (dim[a] > 0 ? dim[a] : 1);
	int b = 1;
	while (b < 64 && ldexp(1.0, b) < tot) b++;
	return b;
}

static size_t aperm_sort_tmp(int64_t nnz, int bits)
{
	size_t b = 0;
	(void) rocprim::radix_sort_pairs(NULL, b, (const unsigned long long *) NULL,
					 (unsigned long long *) NULL, (const uint32_t *) NULL,
					 (uint32_t *) NULL, (size_t) nnz, 0u, (unsigned) bits);
	return b;
}

// [keys nnz*8][sorted keys nnz*8][pos nnz*4][sorted pos nnz*4][sort temp]
// ---- slab form: the new leading axis is an old outer axis q of small extent, the old rows become new
// axis 1 (aperm(x, c(3, 1, 2)) of a 2e4 x 2e4 x 64 array).  For every index of the remaining axes (a slab)
// the sub-array is a d0 x dq matrix held in dq whole old leaves, to be transposed into d0 new leaves of at
// most dq entries each, which follow one another in the output.  One workgroup per slab: the slab's
// nonzeros (a few thousand) are sorted by old row in LDS -- a stable block radix sort on the row alone; the
// input order, leaf after leaf, already ascends in the index that becomes the new row -- and leave as one
// coalesced run; the leaf pointers of the slab's d0 new leaves are filled from the sorted rows.  Traffic:
// the nonzeros once in, once out, and the new leaf pointers (most new leaves are empty: 3.2 GB of
// pointers for 1.5 GB of nonzeros at BASELINE config 5).  7.8 ms through the global key sort.
// ---------------------------------------------------------------------------
#define SLAB_NT 512
#define SLAB_ITEMS 16
#define SLAB_CAP (SLAB_NT * SLAB_ITEMS)
struct SlabMap {
	int nother;              // axes besides old 0 and old q
	int64_t new_ext[8];      // their extents, in the order of the new axes 2 .. ndim-1
	int64_t old_stride[8];   // their leaf strides in the OLD layout
};

__device__ inline int64_t slab_base_leaf(const SlabMap &m, int64_t g)
{
	int64_t j = 0;
	for (int t = 0; t < m.nother; t++) {
		const int64_t it = g % m.new_ext[t];
		g /= m.new_ext[t];
		j += it * m.old_stride[t];
	}
	return j;
}

// cnt[g] = nonzeros of slab g (cnt[nslab] = 0), *maxcnt = the largest
__global__ void aperm_slab_count_kernel(const int64_t *__restrict__ col_ptr, SlabMap m, int64_t nslab,
					int64_t dq, int64_t osq, int64_t *__restrict__ cnt,
					unsigned long long *__restrict__ maxcnt)
{
	const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (g > nslab) return;
	if (g == nslab) { cnt[g] = 0; return; }
	const int64_t j0 = slab_base_leaf(m, g);
	int64_t n = 0;
	for (int64_t k = 0; k < dq; k++) {
		const int64_t j = j0 + k * osq;
		n += col_ptr[j + 1] - col_ptr[j];
	}
	cnt[g] = n;
	atomicMax(maxcnt, (unsigned long long) n);
}

template <typename T>
__global__ void __launch_bounds__(SLAB_NT)
aperm_slab_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
		  const T *__restrict__ val, SlabMap m, int64_t nslab, int64_t d0, int dq, int64_t osq,
		  int bits, int64_t nnz, const int64_t *__restrict__ slab_base, int64_t *__restrict__ out_ptr,
		  int32_t *__restrict__ out_idx, T *__restrict__ out_val)
{
	typedef rocprim::block_radix_sort<uint32_t, SLAB_NT, SLAB_ITEMS, uint32_t, 1, 1, 4> Sort;        // (4-bit digits; 5-bit ones, three passes over 15 row bits, need twice the LDS: 3.15 against 2.9 ms)
	__shared__ typename Sort::storage_type sort_tmp;
	__shared__ int32_t off[1025];                   // first slab-local index of every old leaf
	__shared__ int64_t lbeg[1024];                  // its first position in the old arrays
	extern __shared__ uint32_t skey[];              // SLAB_CAP sorted rows
	const int tid = threadIdx.x;
	const int64_t g = blockIdx.x;
	const int64_t sb = slab_base[g];
	const int n = (int) (slab_base[g + 1] - sb);
	const int64_t j0 = slab_base_leaf(m, g);
	for (int k = tid; k < dq; k += SLAB_NT) {
		const int64_t j = j0 + (int64_t) k * osq;
		lbeg[k] = col_ptr[j];
		off[k + 1] = (int32_t) (col_ptr[j + 1] - col_ptr[j]);
	}
	if (tid == 0) off[0] = 0;
	__syncthreads();
	if (tid == 0)                                   // (dq <= 1024 additions; the leaves are few)
		for (int k = 1; k <= dq; k++) off[k] += off[k - 1];
	__syncthreads();
	auto leaf_of = [&](const int e) {               // last k with off[k] <= e
		int lo = 0, hi = dq - 1;
		while (lo < hi) {
			const int mid = (lo + hi + 1) >> 1;
			if (off[mid] <= e) lo = mid; else hi = mid - 1;
		}
		return lo;
	};
	uint32_t key[SLAB_ITEMS], pay[SLAB_ITEMS];
	// padding items sort behind every row: the all-ones value of `bits` bits unless that is a row itself
	const bool spare = d0 < ((int64_t) 1 << bits);
	const uint32_t pad = spare ? (1u << bits) - 1u : 1u << bits;
	const int sort_bits = spare ? bits : bits + 1;
#pragma unroll
	for (int u = 0; u < SLAB_ITEMS; u++) {
		const int e = tid * SLAB_ITEMS + u;             // blocked: the input order is the tie-break
		pay[u] = (uint32_t) e;
		if (e < n) {
			const int k = leaf_of(e);
			pay[u] |= (uint32_t) k << 13;               // (e < 8192, k < 1024: the leaf rides along)
			key[u] = (uint32_t) row_idx[lbeg[k] + (e - off[k])];
		} else {
			key[u] = pad;                               // past every row
		}
	}
	Sort().sort_to_striped(key, pay, sort_tmp, 0u, (unsigned) sort_bits);
#pragma unroll
	for (int u = 0; u < SLAB_ITEMS; u++) skey[u * SLAB_NT + tid] = key[u];
	__syncthreads();
	const int64_t lp0 = g * d0;                     // first new leaf of the slab
#pragma unroll
	for (int u = 0; u < SLAB_ITEMS; u++) {
		const int sidx = u * SLAB_NT + tid;             // striped: consecutive lanes, consecutive outputs
		if (sidx >= n) continue;
		const int e = (int) (pay[u] & 8191u), k = (int) (pay[u] >> 13);
		out_idx[sb + sidx] = k;
		out_val[sb + sidx] = val[lbeg[k] + (e - off[k])];
		// leaf pointers: every new leaf from the previous entry's row (exclusive) up to this one's starts here
		const int64_t i = key[u], prev = sidx > 0 ? (int64_t) skey[sidx - 1] : -1;
		for (int64_t ii = prev + 1; ii <= i; ii++) out_ptr[lp0 + ii] = sb + sidx;
	}
	const int64_t last = n > 0 ? (int64_t) skey[n - 1] : -1;
	for (int64_t ii = last + 1 + tid; ii < d0; ii += SLAB_NT) out_ptr[lp0 + ii] = sb + n;
	if (g == nslab - 1 && tid == 0) out_ptr[nslab * d0] = nnz;
}

size_t aperm_ws_bytes(int64_t nnz, const int64_t *dim, int ndim)
{
	const size_t n = (size_t) (nnz > 0 ? nnz : 1);
	const size_t a8 = (n * 8 + 255) / 256 * 256, a4 = (n * 4 + 255) / 256 * 256;
	// (leaf-preserving permutations: only the scratch of a scan over the leaf counts)
	double nl = 1.0;
	for (int a = 1; a < ndim; a++) nl *= (double) (dim[a] > 0 ? dim[a] : 1);
	size_t scan_b = 0;
	if (nl < 2147483646.0)
		scan_b = exclusive_scan_ws_bytes((int64_t) nl + 1);
	size_t t32 = 0;
	(void) rocprim::radix_sort_pairs(NULL, t32, (const uint32_t *) NULL, (uint32_t *) NULL,
					 (const uint32_t *) NULL, (uint32_t *) NULL, (size_t) nnz, 0u, 32u);
	const size_t need64 = 2 * a8 + 2 * a4 + aperm_sort_tmp(nnz, aperm_bits(dim, ndim));
	const size_t need32 = 5 * a4 + hint_bytes(nnz) + t32;
	return (need64 > need32 ? need64 : need32) + scan_b + 256;
}

int launch_aperm(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
		 int64_t ncol, int64_t nnz, const int64_t *dim, int ndim, const int *perm,
		 int64_t *out_ptr, int32_t *out_idx, void *out_val, void *ws, hipStream_t s)
{
	if (ndim < 1 || ndim > 8)
		return svt_set_error("aperm: between 1 and 8 dimensions are supported");
	if (nnz >= ((int64_t) 1 << 31))
		return svt_set_error("aperm: more than 2^31-1 nonzeros");
	ApermDims d;
	d.ndim = ndim;
	bool seen[8] = {false, false, false, false, false, false, false, false};
	double total = 1.0;
	for (int a = 0; a < ndim; a++) {
		if (perm[a] < 0 || perm[a] >= ndim || seen[perm[a]])
			return svt_set_error("'perm' must be a permutation of 1:%d", ndim);
		seen[perm[a]] = true;
		d.dim[a] = dim[a];
		d.perm[a] = perm[a];
		total *= (double) (dim[a] > 0 ? dim[a] : 1);
	}
	if (total >= 9.2e18)
		return svt_set_error("aperm: array too large for 64-bit linear indices");
	// multiplier of old axis perm[a] = product of the new dims below new axis a
	int64_t m = 1;
	for (int a = 0; a < ndim; a++) {
		d.mul[perm[a]] = m;
		m *= dim[perm[a]];
	}
	const int64_t new_dim0 = dim[perm[0]];
	int64_t new_nleaves = 1;
	for (int a = 1; a < ndim; a++) new_nleaves *= dim[perm[a]];
	const unsigned nbl = (unsigned) ((new_nleaves + 1 + 255) / 256);
	if (nnz == 0) {
		HIP_TRY(hipMemsetAsync(out_ptr, 0, (size_t) (new_nleaves + 1) * 8, s));
		return 0;
	}
	if (perm[0] == 0 && new_nleaves < ((int64_t) 1 << 31) - 1) {
		LeafMap lm;
		lm.ndim = ndim;
		int64_t os[8], st = 1;
		for (int a = 1; a < ndim; a++) { os[a] = st; st *= dim[a]; }      // leaf strides of the old axes
		for (int a = 1; a < ndim; a++) { lm.new_dim[a] = dim[perm[a]]; lm.old_stride[a] = os[perm[a]]; }
		// counts into out_ptr, exclusive scan in place (the scan's scratch comes from the workspace)
		hipLaunchKernelGGL(aperm_leaf_count_kernel, dim3(nbl), dim3(256), 0, s, col_ptr, new_nleaves, lm, out_ptr);
		if (launch_exclusive_scan_i64(out_ptr, new_nleaves + 1, ws, s))
			return -1;
		const unsigned nbc = (unsigned) ((new_nleaves + 4 * APERM_LU - 1) / (4 * APERM_LU));
		if (Rtype == SVT_REALSXP)
			hipLaunchKernelGGL(aperm_leaf_copy_kernel<double>, dim3(nbc), dim3(256), 0, s, col_ptr, row_idx,
					   (const double *) val, new_nleaves, lm, out_ptr, out_idx, (double *) out_val);
		else
			hipLaunchKernelGGL(aperm_leaf_copy_kernel<int32_t>, dim3(nbc), dim3(256), 0, s, col_ptr, row_idx,
					   (const int32_t *) val, new_nleaves, lm, out_ptr, out_idx, (int32_t *) out_val);
		HIP_TRY(hipGetLastError());
		return 0;
	}
	// slab form (see aperm_slab_kernel): new axis 0 = an old outer axis of <= 1024 entries, new axis 1 = the old
	// rows, slabs of a few thousand nonzeros.  The largest slab is read back (one synchronisation of the
	// stream): a slab over the cap sends the array through the key sort below.
	if (ndim >= 3 && perm[1] == 0 && perm[0] >= 1 && dim[perm[0]] <= 1024 && dim[0] < ((int64_t) 1 << 30)) {
		const int q = perm[0];
		int64_t os[8], st = 1;
		for (int a = 1; a < ndim; a++) { os[a] = st; st *= dim[a]; }
		SlabMap sm;
		sm.nother = ndim - 2;
		int64_t nslab = 1;
		for (int t = 0; t < ndim - 2; t++) {
			sm.new_ext[t] = dim[perm[2 + t]];
			sm.old_stride[t] = os[perm[2 + t]];
			nslab *= dim[perm[2 + t]];
		}
		size_t tb = 0;
		if (nslab < 2147483646LL)
			tb = exclusive_scan_ws_bytes(nslab + 1);
		const size_t a4 = ((size_t) nnz * 4 + 255) / 256 * 256;
		const size_t cnt_b = ((size_t) (nslab + 2) * 8 + 255) / 256 * 256;
		if (nslab >= 1 && nslab < 2147483646LL && nnz / nslab <= SLAB_CAP * 9 / 10 && cnt_b + tb + 256 <= 5 * a4) {
			int64_t *base = (int64_t *) ws;
			unsigned long long *maxcnt = (unsigned long long *) ((char *) ws + cnt_b);
			void *scan_tmp = (char *) ws + cnt_b + 256;
			HIP_TRY(hipMemsetAsync(maxcnt, 0, 8, s));
			hipLaunchKernelGGL(aperm_slab_count_kernel, dim3((unsigned) ((nslab + 1 + 255) / 256)), dim3(256), 0, s,
					   col_ptr, sm, nslab, dim[q], os[q], base, maxcnt);
			unsigned long long mx = 0;
			HIP_TRY(hipMemcpyAsync(&mx, maxcnt, 8, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			if (mx <= SLAB_CAP) {
				if (launch_exclusive_scan_i64(base, nslab + 1, scan_tmp, s))
					return -1;
				int bits = 1;
				while (bits < 31 && ((int64_t) 1 << bits) < dim[0]) bits++;
				const size_t lds = (size_t) SLAB_CAP * 4;
				if (Rtype == SVT_REALSXP)
					hipLaunchKernelGGL(aperm_slab_kernel<double>, dim3((unsigned) nslab), dim3(SLAB_NT), lds, s,
							   col_ptr, row_idx, (const double *) val, sm, nslab, dim[0], (int) dim[q], os[q],
							   bits, nnz, base, out_ptr, out_idx, (double *) out_val);
				else
					hipLaunchKernelGGL(aperm_slab_kernel<int32_t>, dim3((unsigned) nslab), dim3(SLAB_NT), lds, s,
							   col_ptr, row_idx, (const int32_t *) val, sm, nslab, dim[0], (int) dim[q], os[q],
							   bits, nnz, base, out_ptr, out_idx, (int32_t *) out_val);
				HIP_TRY(hipGetLastError());
				return 0;
			}
		}
	}
	if (new_nleaves < ((int64_t) 1 << 31) - 1) {
		const size_t a4 = ((size_t) nnz * 4 + 255) / 256 * 256;
		uint32_t *keys = (uint32_t *) ws, *skeys = (uint32_t *) ((char *) ws + a4);
		uint32_t *pos = (uint32_t *) ((char *) ws + 2 * a4), *spos = (uint32_t *) ((char *) ws + 3 * a4);
		int32_t *newrow = (int32_t *) ((char *) ws + 4 * a4);
		uint32_t *hint = (uint32_t *) ((char *) ws + 5 * a4);
		void *tmp = (char *) ws + 5 * a4 + hint_bytes(nnz);
		int bits = 1;
		while (bits < 32 && ((int64_t) 1 << bits) < new_nleaves) bits++;
		size_t tb = 0;
		(void) rocprim::radix_sort_pairs(NULL, tb, (const uint32_t *) NULL, (uint32_t *) NULL,
						 (const uint32_t *) NULL, (uint32_t *) NULL, (size_t) nnz, 0u, (unsigned) bits);
		const unsigned nb = (unsigned) ((nnz + 255) / 256);
		const unsigned nb8 = (unsigned) (((nnz + 255) / 256 + 7) / 8 * 8);
		const int64_t nblk = (nnz >> HINT_SHIFT) + 1;
		hipLaunchKernelGGL(col_hint_kernel, dim3((unsigned) ((nblk + 1 + 255) / 256)), dim3(256), 0, s, col_ptr, ncol, nblk, hint);
		hipLaunchKernelGGL(aperm_key32_kernel, dim3(nb), dim3(256), 0, s, col_ptr, row_idx, hint, ncol, nnz, d,
				   keys, pos, newrow);
		HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, keys, skeys, pos, spos, (size_t) nnz, 0u, (unsigned) bits, s));
		hipLaunchKernelGGL(aperm_ptr_fill_kernel, dim3((unsigned) ((nnz + 1 + 255) / 256)), dim3(256), 0, s,
				   skeys, nnz, new_nleaves, out_ptr);
		if (Rtype == SVT_REALSXP)
			hipLaunchKernelGGL(aperm_gather32_kernel<double>, dim3(nb8), dim3(256), 0, s, spos, newrow,
					   (const double *) val, nnz, out_idx, (double *) out_val);
		else
			hipLaunchKernelGGL(aperm_gather32_kernel<int32_t>, dim3(nb8), dim3(256), 0, s, spos, newrow,
					   (const int32_t *) val, nnz, out_idx, (int32_t *) out_val);
		HIP_TRY(hipGetLastError());
		return 0;
	}
	const size_t n = (size_t) nnz;
	const size_t a8 = (n * 8 + 255) / 256 * 256, a4 = (n * 4 + 255) / 256 * 256;
	unsigned long long *keys = (unsigned long long *) ws;
	unsigned long long *skeys = (unsigned long long *) ((char *) ws + a8);
	uint32_t *pos = (uint32_t *) ((char *) ws + 2 * a8);
	uint32_t *spos = (uint32_t *) ((char *) ws + 2 * a8 + a4);
	void *tmp = (char *) ws + 2 * a8 + 2 * a4;
	const int bits = aperm_bits(dim, ndim);
	size_t tb = aperm_sort_tmp(nnz, bits);
	const unsigned nb = (unsigned) ((nnz + 255) / 256);
	const unsigned nb8 = (unsigned) (((nnz + 255) / 256 + 7) / 8 * 8);
	hipLaunchKernelGGL(aperm_key_kernel, dim3(nb), dim3(256), 0, s, col_ptr, row_idx, ncol, nnz, d, keys, pos);
	HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, keys, skeys, pos, spos, (size_t) nnz, 0u, (unsigned) bits, s));
	hipLaunchKernelGGL(aperm_bounds_kernel, dim3(nbl), dim3(256), 0, s, skeys, nnz, new_nleaves, new_dim0, out_ptr);
	if (Rtype == SVT_REALSXP)
		hipLaunchKernelGGL(aperm_gather_kernel<double>, dim3(nb8), dim3(256), 0, s, skeys, spos,
				   (const double *) val, nnz, new_dim0, out_idx, (double *) out_val);
	else
		hipLaunchKernelGGL(aperm_gather_kernel<int32_t>, dim3(nb8), dim3(256), 0, s, skeys, spos,
				   (const int32_t *) val, nnz, new_dim0, out_idx, (int32_t *) out_val);
	HIP_TRY(hipGetLastError());
	return 0;
}
