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

#include "svt_sort.h"

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
// a given range of rows are one contiguous run.
//
//   fine bucket   = F consecutive rows (a power of two <= 64: ~3000 nonzeros), the unit of the last pass
//   coarse bucket = 16 fine buckets
//   group         = 256 consecutive columns
//
//   pass 1 (count)   a streaming pass over the offsets: a workgroup takes a stretch of one group's nonzeros,
//                    counts them by fine bucket in LDS and adds its counts to the table
//                    T[coarse bucket][group][fine bucket in the coarse one].
//   scan             exclusive scan of T in that order: where the piece (coarse i, group g, fine s) starts
//                    in an intermediate array ordered (coarse, group, fine, column, row).
//   pass 2 (scatter) workgroup (g, i), one thread per column: the column's run of rows in the coarse bucket
//                    (an interpolated guess, then a search over what is left), the position of each of its
//                    nonzeros among the workgroup's = a scan over the 128 threads per fine bucket; the
//                    workgroup's nonzeros are assembled in LDS in their final intermediate order and leave
//                    as ONE contiguous, coalesced copy (the 16 pieces of a workgroup follow one another).
//   pass 3 (finish)  one workgroup per fine bucket: its pieces (one per group, in group order = column
//                    order) are read in sequence; a stable counting sort on the F rows (ranks inside a
//                    wavefront from ballots of the row bits, across wavefronts from a small LDS table) puts
//                    them in (row, column) order -- in LDS, and from there in one coalesced copy to their
//                    final place -- and gives the row pointers.
//
// Traffic at 1e8 nonzeros (f64): pass 1 reads the offsets (0.4 GB), pass 2 reads 1.2 GB and writes 1.3 GB
// (column 4 B + row-in-bucket 1 B + value 8 B), pass 3 reads 1.3 GB and writes 1.2 GB: 5.4 GB, every
// store coalesced.  (The first version of this round wrote pass 2 straight to memory, fine-bucket-major:
// every lane of a store instruction in another line, 48 open write streams per workgroup -- 7.5 ms for that
// pass alone; and counted with one thread per column and two binary searches: 2.2 ms.)
// Shapes it does not fit (less than one nonzero per column and coarse bucket) take the key sort of
// launch_transpose_sorted() below.
// ---------------------------------------------------------------------------
// which route the transpositions / permutations of this process took (svt_dev_aperm_route_counts; the fuzzers print it:
// evidence for what is hot and what is a fallback): 0 t() bucketed, 1 t() key sort, 2 aperm leaf-preserving,
// 3 first two axes swapped (bucketed), 4 slab form, 5 3-d via an intermediate, 6 general (composed), 7 key sort, 32-bit
// keys, 8 key sort, 64-bit keys, 9 slab form refused at run time
static int64_t g_route[10];
void aperm_route_counts(int64_t *out, int reset)
{
	for (int i = 0; i < 10; i++) { if (out != NULL) out[i] = g_route[i]; if (reset) g_route[i] = 0; }
}

#define T2_NT 256                 // columns per group = threads per workgroup of pass 2
#define T2_NFINE_MAX 32           // fine buckets per coarse bucket: 16 or 32 (T2Shape::cbits = 4 or 5)
#define T2_CAP 2048               // nonzeros a pass-2 workgroup assembles in LDS (more: straight to memory)
#define T2_CPW 2                  // coarse buckets a pass-2 workgroup takes, one after the other (t(A) at config 2: 1 -> 2.06 ms, 2 -> 2.05, 4 -> 2.12, 8 -> 2.20, 16 -> 2.30)
#define T1_NT 1024
#define T1_HIST 65536             // fine buckets counted per sweep of pass 1 (LDS: two 16-bit counters per word --
                                  // a workgroup's 16 columns hold at most 16 * 64 nonzeros of one fine bucket)
#define T3_NT 512
#define T3_CAP 4096               // nonzeros a pass-3 workgroup ranks in one round (more: round by round, straight to memory)
#define T3_STAGE 2048             // of which the LDS image of the output holds this many at a time

// Batched form (round 4): `nslab` independent matrices of srow x scol that follow one another in the operand's
// columns (slab s = columns [s * scol, (s + 1) * scol)), each transposed on its own -- aperm(x, c(2, 1, 3, ...)) of an
// N-d array; nfb / ncoarse / ngroups count per slab, the table is ordered (slab, coarse, group, fine).  One matrix:
// nslab = 1, srow = nrow, scol = ncol.
struct T2Shape {
	int fbits;                // log2(F)
	int64_t nfb;              // fine buckets (per slab)
	int64_t ncoarse;
	int64_t ngroups;
	int64_t nslab, srow, scol;
	int cbits;                // log2(fine buckets per coarse bucket): 4, or 5 where that fills the pass-2 workgroups better
};

__device__ inline int64_t t2_slot(const T2Shape &sh, int64_t sl, int64_t fb, int64_t g)
{
	return ((((sl * sh.ncoarse + (fb >> sh.cbits)) * sh.ngroups + g) << sh.cbits)) + (fb & ((1 << sh.cbits) - 1));
}

// pass 1: workgroup = 16 columns (16 divides the group size), one per wavefront (coalesced along the column); counts by fine bucket in
// LDS, and cstart[c * (ncoarse + 1) + i] = first position of column c whose row lies in coarse bucket i or
// later (i = 0 .. ncoarse), which is where pass 2 finds its runs without a search (a wavefront writes the
// starts of its column next to one another; a pass-2 workgroup reads nine consecutive ones per column).
__global__ void __launch_bounds__(T1_NT)
transpose_count_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
		       int64_t ncol, T2Shape sh, unsigned long long *__restrict__ table,
		       uint32_t *__restrict__ cstart)
{
	extern __shared__ uint32_t hist[];              // [T1_HIST / 2]
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int64_t wps = (sh.scol + T1_NT / 64 - 1) / (T1_NT / 64);            // workgroups per slab
	const int64_t sl = (int64_t) blockIdx.x / wps, bl = (int64_t) blockIdx.x % wps;
	const int64_t cl = bl * (T1_NT / 64) + w;                                  // column inside the slab
	const bool have = cl < sh.scol;
	const int64_t c = sl * sh.scol + cl;
	const int64_t g = (bl * (T1_NT / 64)) / T2_NT;                            // (16 divides 256: one group per workgroup)
	const int cshift = sh.fbits + sh.cbits;         // log2(rows per coarse bucket)
	int64_t beg = 0, end = 0;
	if (have) { beg = col_ptr[c]; end = col_ptr[c + 1]; }
	for (int64_t w0 = 0; w0 < sh.nfb; w0 += T1_HIST) {          // (one sweep unless there are > 32768 fine buckets)
		const int64_t w1 = w0 + T1_HIST < sh.nfb ? w0 + T1_HIST : sh.nfb;
		for (int x = threadIdx.x; x < (int) ((w1 - w0 + 1) >> 1); x += T1_NT) hist[x] = 0;
		__syncthreads();
		for (int64_t kb = beg; kb < end; kb += 256) {           // four batches of 64 in flight
			int32_t r4[4];
#pragma unroll
			for (int u = 0; u < 4; u++) r4[u] = kb + u * 64 + lane < end ? row_idx[kb + u * 64 + lane] : 0x7FFFFFFF;
			const int32_t r_before = kb > beg ? row_idx[kb - 1] : -1;
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int64_t k0 = kb + u * 64, k = k0 + lane;
				if (k0 >= end) break;
				const int32_t r = r4[u];
				if (k < end) {
					const int64_t fb = (int64_t) r >> sh.fbits;
					if (fb >= w0 && fb < w1) atomicAdd(&hist[(fb - w0) >> 1], 1u << (16 * (int) ((fb - w0) & 1)));
				}
				if (w0 == 0 && have) {
					// coarse buckets that start at position k: those after the previous entry's, up to this one's
					int32_t rp = __shfl_up(r, 1, SVT_WAVE);
					const int32_t last_prev = __shfl(r4[u > 0 ? u - 1 : 0], 63, SVT_WAVE);      // (all lanes take part)
					if (lane == 0) rp = u == 0 ? r_before : last_prev;
					const int64_t ip = rp < 0 ? -1 : ((int64_t) rp >> cshift);
					const int64_t ic = k < end ? ((int64_t) r >> cshift) : (k == end ? sh.ncoarse : ip);
					for (int64_t i = ip + 1; i <= ic; i++) cstart[c * (sh.ncoarse + 1) + i] = (uint32_t) k;
				}
			}
		}
		if (w0 == 0 && have && lane == 0) {
			// (a column whose length is a multiple of 64 -- or zero -- has not closed its last buckets)
			if (((end - beg) & 63) == 0) {
				const int64_t ip = end > beg ? ((int64_t) row_idx[end - 1] >> cshift) : -1;
				for (int64_t i = ip + 1; i <= sh.ncoarse; i++) cstart[c * (sh.ncoarse + 1) + i] = (uint32_t) end;
			}
		}
		__syncthreads();
		for (int x = threadIdx.x; x < (int) (w1 - w0); x += T1_NT) {
			const uint32_t n = (hist[x >> 1] >> (16 * (x & 1))) & 0xFFFFu;
			if (n) atomicAdd(table + t2_slot(sh, sl, w0 + x, g), (unsigned long long) n);
		}
		__syncthreads();
	}
}

// first position in [lo, hi) whose row is >= r
__device__ inline int64_t t2_lower_bound(const int32_t *__restrict__ row_idx, int64_t lo, int64_t hi, int64_t r)
{
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if ((int64_t) row_idx[mid] < r) lo = mid + 1; else hi = mid;
	}
	return lo;
}

// Passes 2 and 3 are the same operation: a workgroup holds a SEQUENCE of nonzeros made of pieces (pass 2:
// the runs of its 128 columns inside the coarse bucket; pass 3: the pieces of its fine bucket, one per
// group) and splits it by a small key (pass 2: the fine bucket, 4 bits; pass 3: the row inside the bucket,
// <= 6 bits) keeping the order inside every key.  Element x of the sequence is read by thread x mod NT
// (consecutive lanes = consecutive addresses inside a piece: coalesced), kept in registers, ranked --
// rank inside the wavefront from one ballot per key bit, the wavefronts' counts per key in an LDS table,
// prefixed in element order -- and written to its place in an LDS image of the output, which leaves in one
// linear copy.  Sequences longer than the image (NT * ITEMS elements) go round by round, straight to memory.
template <int NT, int ITEMS, int MAXKEYS>
struct SplitLds {
	int32_t binstart[MAXKEYS];              // first output slot of every key (histogram first, then its prefix)
	int32_t run[MAXKEYS];                   // slots used by earlier rounds
	int32_t rtot[MAXKEYS];                  // elements of the current round
	int32_t totu[ITEMS][MAXKEYS];           // per key: elements of the chunks before chunk u (this round)
	uint16_t wc[ITEMS][NT / 64][MAXKEYS];   // per key: elements of chunk u in the wavefronts before w (<= NT)
};

// lanes of the wavefront whose element is valid and has the same key: one ballot per key bit
__device__ inline uint64_t split_peers(bool valid, int key, int kbits)
{
	uint64_t peers = __ballot(valid);
	for (int bit = 0; bit < kbits; bit++) {
		const uint64_t m = __ballot((key >> bit) & 1);
		peers &= ((key >> bit) & 1) ? m : ~m;
	}
	return peers;
}

// binstart[0 .. nkeys) holds the histogram: turn it into its exclusive prefix (nkeys <= 256, wavefront 0;
// the other wavefronts wait at the caller's barrier).  f(k, start) is called for every key by the lane that
// owns it.
template <typename F>
__device__ inline void split_scan_bins(int32_t *binstart, int nkeys, F f)
{
	const int lane = threadIdx.x & 63;
	if ((threadIdx.x >> 6) != 0) return;
	int32_t v[4], tot = 0;
#pragma unroll
	for (int u = 0; u < 4; u++) { v[u] = lane * 4 + u < nkeys ? binstart[lane * 4 + u] : 0; tot += v[u]; }
	int32_t incl = tot;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int32_t x = __shfl_up(incl, o, SVT_WAVE);
		if (lane >= o) incl += x;
	}
	int32_t off = incl - tot;
#pragma unroll
	for (int u = 0; u < 4; u++) {
		const int k = lane * 4 + u;
		if (k < nkeys) { binstart[k] = off; f(k, off); }
		off += v[u];
	}
}

// Output slots of one round: key[u] (valid[u]) belongs to element u * NT + t of the round.  All threads of
// the workgroup call it; nkeys = 1 << kbits <= MAXKEYS.
template <int NT, int ITEMS, int MAXKEYS>
__device__ inline void split_round(SplitLds<NT, ITEMS, MAXKEYS> &L, const bool (&valid)[ITEMS],
				   const int (&key)[ITEMS], int kbits, int32_t (&at)[ITEMS])
{
	const int t = threadIdx.x, lane = t & 63, w = t >> 6, nkeys = 1 << kbits;
	const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	for (int k = t; k < nkeys; k += NT) { L.run[k] += L.rtot[k]; L.rtot[k] = 0; }      // the previous round's elements
	for (int x = t; x < ITEMS * (NT / 64) * MAXKEYS / 2; x += NT) ((uint32_t *) &L.wc[0][0][0])[x] = 0;
	__syncthreads();
	int rank[ITEMS];
#pragma unroll
	for (int u = 0; u < ITEMS; u++) {
		const uint64_t peers = split_peers(valid[u], key[u], kbits);
		rank[u] = __popcll(peers & lt);
		if (valid[u] && rank[u] == 0) L.wc[u][w][key[u]] = (uint16_t) __popcll(peers);
	}
	__syncthreads();
	// exclusive prefix in element order (chunk u, then wavefront w), per key
	for (int x = t; x < ITEMS * nkeys; x += NT) {
		const int u = x / nkeys, k = x % nkeys;
		int32_t sum = 0;
		for (int ww = 0; ww < NT / 64; ww++) { const int32_t v = L.wc[u][ww][k]; L.wc[u][ww][k] = (uint16_t) sum; sum += v; }
		L.totu[u][k] = sum;
	}
	__syncthreads();
	for (int k = t; k < nkeys; k += NT) {
		int32_t sum = 0;
		for (int u = 0; u < ITEMS; u++) { const int32_t v = L.totu[u][k]; L.totu[u][k] = sum; sum += v; }
		L.rtot[k] = sum;
	}
	__syncthreads();
#pragma unroll
	for (int u = 0; u < ITEMS; u++)
		at[u] = valid[u] ? L.binstart[key[u]] + L.run[key[u]] + L.totu[u][key[u]] + L.wc[u][w][key[u]] + rank[u] : 0;
}

// last j in [0, np) with pre[j] <= x  (pre[0] = 0, pre[np] = length of the sequence > x)
__device__ inline int split_piece_of(const int32_t *pre, int np, int32_t x)
{
	int lo = 0, hi = np - 1;
	while (lo < hi) {
		const int mid = (lo + hi + 1) >> 1;
		if (pre[mid] <= x) lo = mid; else hi = mid - 1;
	}
	return lo;
}

// pass 2
#define T2_ITEMS (T2_CAP / T2_NT)
template <typename T>
__global__ void __launch_bounds__(T2_NT)
transpose_scatter_kernel(const int64_t *__restrict__ col_ptr, const int32_t *__restrict__ row_idx,
			 const T *__restrict__ val, int64_t nrow, int64_t ncol, T2Shape sh,
			 const int64_t *__restrict__ table, const uint32_t *__restrict__ cstart,
			 int32_t *__restrict__ col1, uint8_t *__restrict__ rlow1, T *__restrict__ val1)
{
	__shared__ SplitLds<T2_NT, T2_ITEMS, T2_NFINE_MAX> L;
	const int nfine = 1 << sh.cbits;
	__shared__ int64_t pa[T2_NT];                   // first position of every column's run
	__shared__ int32_t ppre[T2_NT + 1];             // lengths of the runs, then their prefix
	__shared__ uint8_t owner[T2_CAP];               // column (0 .. 255) of every element of the sequence
	__shared__ int32_t s_col[T2_CAP];
	__shared__ T s_val[T2_CAP];
	__shared__ uint8_t s_row[T2_CAP];
	const int t = threadIdx.x, lane = t & 63, w = t >> 6;
	const int64_t wps = sh.ngroups * ((sh.ncoarse + T2_CPW - 1) / T2_CPW);     // workgroups per slab
	const int64_t sl = (int64_t) blockIdx.x / wps, bl = (int64_t) blockIdx.x % wps;
	const int64_t g = bl % sh.ngroups, i0 = (bl / sh.ngroups) * T2_CPW;
	const bool have = g * T2_NT + t < sh.scol;
	const int64_t c = sl * sh.scol + g * T2_NT + t;
	// where this column's runs of the workgroup's T2_CPW coarse buckets start (recorded by pass 1)
	uint32_t cs[T2_CPW + 1];
#pragma unroll
	for (int q = 0; q <= T2_CPW; q++)
		cs[q] = (have && i0 + q <= sh.ncoarse) ? cstart[c * (sh.ncoarse + 1) + i0 + q] : 0u;
#pragma unroll 1
	for (int q = 0; q < T2_CPW; q++) {
	const int64_t i = i0 + q;
	if (i >= sh.ncoarse) break;                     // (uniform)
	if (q > 0) __syncthreads();                     // the previous bucket's image has left
	const int64_t r_lo = (i << sh.cbits) << sh.fbits;
	int64_t a = 0, b = 0;
	if (have) {
#pragma unroll
		for (int qq = 0; qq < T2_CPW; qq++) if (qq == q) { a = cs[qq]; b = cs[qq + 1]; }
	}
	pa[t] = a;
	// prefix of the run lengths over the threads
	{
		__shared__ int32_t wsum[T2_NT / 64];
		const int32_t v = (int32_t) (b - a);
		int32_t incl = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int32_t x = __shfl_up(incl, o, SVT_WAVE);
			if (lane >= o) incl += x;
		}
		if (lane == 63) wsum[w] = incl;
		if (t < nfine) { L.binstart[t] = 0; L.run[t] = 0; L.rtot[t] = 0; }
		__syncthreads();
		int32_t off = 0;
		for (int ww = 0; ww < w; ww++) off += wsum[ww];
		ppre[t + 1] = incl + off;
		if (t == 0) ppre[0] = 0;
		__syncthreads();
	}
	const int32_t n = ppre[T2_NT];
	const int64_t base = table[((sl * sh.ncoarse + i) * sh.ngroups + g) << sh.cbits];     // start of the workgroup's stretch
	if (n <= T2_CAP) {
		for (int32_t x = ppre[t]; x < ppre[t + 1]; x++) owner[x] = (uint8_t) t;
		__syncthreads();
	}
	const int fmask = (1 << sh.fbits) - 1;
	const bool staged = n <= T2_CAP;
	if (!staged) {
		// more than the image holds (rare): the histogram needs a pass of its own
		for (int32_t x = t; x < n; x += T2_NT) {
			const int j = split_piece_of(ppre, T2_NT, x);
			const int q = (int) ((int64_t) row_idx[pa[j] + (x - ppre[j])] - r_lo);
			atomicAdd(&L.binstart[q >> sh.fbits], 1);
		}
		__syncthreads();
		split_scan_bins(L.binstart, nfine, [](int, int32_t) {});
		__syncthreads();
	}
	for (int32_t x0 = 0; x0 < n; x0 += T2_CAP) {
		bool valid[T2_ITEMS];
		int key[T2_ITEMS], rl[T2_ITEMS], cj[T2_ITEMS];
		int32_t rw[T2_ITEMS];
		T vv[T2_ITEMS];
		int32_t at[T2_ITEMS];
		int64_t kk[T2_ITEMS];
#pragma unroll
		for (int u = 0; u < T2_ITEMS; u++) {
			const int32_t x = x0 + u * T2_NT + t;
			valid[u] = x < n;
			cj[u] = !valid[u] ? 0 : (n <= T2_CAP ? (int) owner[x] : split_piece_of(ppre, T2_NT, x));
			kk[u] = valid[u] ? pa[cj[u]] + (x - ppre[cj[u]]) : 0;
		}
		// (all loads of the round in flight together: one memory latency per round, not one per element)
#pragma unroll
		for (int u = 0; u < T2_ITEMS; u++) rw[u] = valid[u] ? row_idx[kk[u]] : 0;
#pragma unroll
		for (int u = 0; u < T2_ITEMS; u++) vv[u] = valid[u] ? val[kk[u]] : (T) 0;
#pragma unroll
		for (int u = 0; u < T2_ITEMS; u++) {
			const int q = valid[u] ? (int) ((int64_t) rw[u] - r_lo) : 0;
			key[u] = q >> sh.fbits; rl[u] = q & fmask;
		}
		if (staged) {                                   // the only round: histogram from the registers
#pragma unroll
			for (int u = 0; u < T2_ITEMS; u++) if (valid[u]) atomicAdd(&L.binstart[key[u]], 1);
			__syncthreads();
			split_scan_bins(L.binstart, nfine, [](int, int32_t) {});
			__syncthreads();
		}
		split_round<T2_NT, T2_ITEMS, T2_NFINE_MAX>(L, valid, key, sh.cbits, at);
#pragma unroll
		for (int u = 0; u < T2_ITEMS; u++) {
			if (!valid[u]) continue;
			const int32_t cc = (int32_t) (g * T2_NT + cj[u]);
			if (staged) { s_col[at[u]] = cc; s_row[at[u]] = (uint8_t) rl[u]; s_val[at[u]] = vv[u]; }
			else { col1[base + at[u]] = cc; rlow1[base + at[u]] = (uint8_t) rl[u]; val1[base + at[u]] = vv[u]; }
		}
	}
	if (!staged)
		continue;
	__syncthreads();
	for (int32_t e = t; e < n; e += T2_NT) {
		col1[base + e] = s_col[e];
		rlow1[base + e] = s_row[e];
		val1[base + e] = s_val[e];
	}
	}
}

// pass 3
#define T3_ITEMS (T3_CAP / T3_NT)
template <typename T>
__global__ void __launch_bounds__(T3_NT)
transpose_finish_kernel(const int64_t *__restrict__ table, T2Shape sh, int64_t nrow, int64_t nnz,
			const int32_t *__restrict__ col1, const uint8_t *__restrict__ rlow1,
			const T *__restrict__ val1, const int64_t *__restrict__ fb_base,
			int64_t *__restrict__ out_ptr, int32_t *__restrict__ out_idx, T *__restrict__ out_val)
{
	extern __shared__ unsigned char t3_lds[];
	// [split tables][piece starts ngroups * 8][piece prefix (ngroups + 1) * 4][sorted columns CAP][sorted values CAP]
	typedef SplitLds<T3_NT, T3_ITEMS, 64> Tables;
	Tables &L = *(Tables *) t3_lds;
	const int ng = (int) sh.ngroups;
	int64_t *pa = (int64_t *) (t3_lds + ((sizeof(Tables) + 15) & ~(size_t) 15));
	int32_t *ppre = (int32_t *) (pa + ng);
	int32_t *s_idx = ppre + ((ng + 1 + 3) & ~3);
	T *s_val = (T *) (s_idx + T3_STAGE);
	const int t = threadIdx.x, lane = t & 63, w = t >> 6;
	const int F = 1 << sh.fbits;
	const int64_t sl = (int64_t) blockIdx.x / sh.nfb, fb = (int64_t) blockIdx.x % sh.nfb, ci = fb >> sh.cbits;
	const int sf = (int) (fb & ((1 << sh.cbits) - 1));
	// the pieces of this fine bucket, one per group: starts, sizes, then the prefix of the sizes
	for (int g = t; g < ng; g += T3_NT) {
		const int64_t slot = (((sl * sh.ncoarse + ci) * ng + g) << sh.cbits) + sf;
		const int64_t p0 = table[slot];
		pa[g] = p0;
		ppre[g + 1] = (int32_t) (table[slot + 1] - p0);
	}
	for (int k = t; k < 64; k += T3_NT) { L.binstart[k] = 0; L.run[k] = 0; L.rtot[k] = 0; }
	__syncthreads();
	if (w == 0) {                                   // inclusive scan of the piece sizes by one wavefront
		int32_t carry = 0;
		for (int g0 = 0; g0 < ng; g0 += 64) {
			const int g = g0 + lane;
			const int32_t v = g < ng ? ppre[g + 1] : 0;
			int32_t incl = v;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const int32_t u = __shfl_up(incl, o, SVT_WAVE);
				if (lane >= o) incl += u;
			}
			if (g < ng) ppre[g + 1] = carry + incl;
			carry += __shfl(incl, 63, SVT_WAVE);
		}
		if (lane == 0) ppre[0] = 0;
	}
	__syncthreads();
	const int32_t n = ppre[ng];
	const int64_t b0 = fb_base[sl * sh.nfb + fb];   // first output position of the bucket's rows
	const bool staged = n <= T3_CAP;
	auto bins_done = [&]() {
		__syncthreads();
		split_scan_bins(L.binstart, F, [&](int r, int32_t off) {
			const int64_t row = (fb << sh.fbits) + r;
			if (row < sh.srow) out_ptr[sl * sh.srow + row] = b0 + off;
		});
		if (sl == sh.nslab - 1 && fb == sh.nfb - 1 && t == 0) out_ptr[sh.nslab * sh.srow] = nnz;
		__syncthreads();
	};
	if (!staged || n == 0) {
		// more than the image holds (rare): the histogram needs a pass of its own
		for (int32_t x = t; x < n; x += T3_NT) {
			const int j = split_piece_of(ppre, ng, x);
			atomicAdd(&L.binstart[rlow1[pa[j] + (x - ppre[j])]], 1);
		}
		bins_done();
	}
	for (int32_t x0 = 0; x0 < n; x0 += T3_CAP) {
		bool valid[T3_ITEMS];
		int key[T3_ITEMS];
		int32_t cc[T3_ITEMS], at[T3_ITEMS];
		T vv[T3_ITEMS];
		int64_t kk[T3_ITEMS];
#pragma unroll
		for (int u = 0; u < T3_ITEMS; u++) {
			const int32_t x = x0 + u * T3_NT + t;
			valid[u] = x < n;
			const int j = valid[u] ? split_piece_of(ppre, ng, x) : 0;
			kk[u] = valid[u] ? pa[j] + (x - ppre[j]) : 0;
		}
		// (all loads of the round in flight together)
#pragma unroll
		for (int u = 0; u < T3_ITEMS; u++) key[u] = valid[u] ? (int) rlow1[kk[u]] : 0;
#pragma unroll
		for (int u = 0; u < T3_ITEMS; u++) cc[u] = valid[u] ? col1[kk[u]] : 0;
#pragma unroll
		for (int u = 0; u < T3_ITEMS; u++) vv[u] = valid[u] ? val1[kk[u]] : (T) 0;
		if (staged) {                                   // the only round: histogram from the registers
#pragma unroll
			for (int u = 0; u < T3_ITEMS; u++) if (valid[u]) atomicAdd(&L.binstart[key[u]], 1);
			bins_done();
		}
		split_round<T3_NT, T3_ITEMS, 64>(L, valid, key, sh.fbits, at);
		if (!staged) {
#pragma unroll
			for (int u = 0; u < T3_ITEMS; u++)
				if (valid[u]) { out_idx[b0 + at[u]] = cc[u]; out_val[b0 + at[u]] = vv[u]; }
			continue;
		}
		// the LDS image holds a quarter of the output at a time (the elements stay in registers): 24 KB
		// instead of 96, three workgroups per CU instead of one
		for (int32_t q0 = 0; q0 < n; q0 += T3_STAGE) {
			const int32_t q1 = q0 + T3_STAGE < n ? q0 + T3_STAGE : n;
			__syncthreads();
#pragma unroll
			for (int u = 0; u < T3_ITEMS; u++)
				if (valid[u] && at[u] >= q0 && at[u] < q1) { s_idx[at[u] - q0] = cc[u]; s_val[at[u] - q0] = vv[u]; }
			__syncthreads();
			for (int32_t e = q0 + t; e < q1; e += T3_NT) { out_idx[b0 + e] = s_idx[e - q0]; out_val[b0 + e] = s_val[e - q0]; }
		}
	}
}

// fb_base[fb] = nonzeros in the fine buckets before fb (= first output position of its rows): the sum of
// the table over (group, fine = fb) ... computed from the scanned table: the pieces of one COARSE bucket are
// contiguous, so base(coarse i) = T[(i * ngroups) * 16]; inside it the fine buckets are interleaved by group.
__global__ void transpose_fb_base_kernel(const int64_t *__restrict__ table, T2Shape sh, int64_t nnz,
					 int64_t *__restrict__ fb_base)
{
	// one wavefront per coarse bucket: sizes of its 16 fine buckets summed over the groups, then a prefix
	const int lane = threadIdx.x & 63;
	// (ci counts over all slabs: the pieces of one (slab, coarse bucket) pair are contiguous in the table)
	const int64_t ci = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (ci >= sh.nslab * sh.ncoarse) return;
	const int64_t sl = ci / sh.ncoarse, cil = ci % sh.ncoarse;
	const int64_t ng = sh.ngroups;
	const int nfine = 1 << sh.cbits;
	int64_t acc[T2_NFINE_MAX];
#pragma unroll
	for (int s = 0; s < T2_NFINE_MAX; s++) acc[s] = 0;
	const int64_t tot = ng * nfine;                   // slots of this coarse bucket
	const int64_t s0 = ci * tot;
	for (int64_t x = lane; x < tot; x += 64) {
		const int64_t sz = table[s0 + x + 1] - table[s0 + x];
		const int s = (int) (x & (nfine - 1));
#pragma unroll
		for (int q = 0; q < T2_NFINE_MAX; q++) if (q == s) acc[q] += sz;
	}
	int64_t run = table[s0];
#pragma unroll
	for (int s = 0; s < T2_NFINE_MAX; s++) {
		if (s >= nfine) break;
		int64_t v = acc[s];
		for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, SVT_WAVE);
		const int64_t fb = cil * nfine + s;
		if (lane == 0 && fb < sh.nfb) fb_base[sl * sh.nfb + fb] = run;
		run += v;
	}
	if (ci == sh.nslab * sh.ncoarse - 1 && lane == 0) fb_base[sh.nslab * sh.nfb] = nnz;
}

// the bucketed form applies to this operand: fills *sh
static bool t2_shape(int64_t nrow, int64_t ncol, int64_t nnz_all, T2Shape *sh, int64_t nslab = 1)
{
	if (nrow <= 0 || ncol <= 0 || nnz_all <= 0 || nslab <= 0)
		return false;
	const int64_t nnz = nnz_all / nslab;            // (per slab; nrow, ncol are a slab's)
	if (nnz <= 0)
		return false;
	const double per_row = (double) nnz / (double) nrow, per_col = (double) nnz / (double) ncol;
	// the largest F <= 64 with ~3000 nonzeros per fine bucket (pass 3 ranks 4096 per round) and ~1500 per
	// pass-2 workgroup (it assembles 2048 in LDS); 32 fine buckets per coarse one where that costs no F (short
	// columns: aperm(x, c(2, 1, 3)) at BASELINE config 5 fills a pass-2 workgroup with 1310 nonzeros instead of 655)
	int fbits = -1, cbits = 4;
	for (int cb = 5; cb >= 4; cb--) {
		int fb = 6;
		while (fb > 0 && (ldexp(per_row, fb) > 3520.0 ||
				  (double) T2_NT * per_col * ldexp((double) (1 << cb), fb) / (double) nrow > 1536.0))
			fb--;
		if (fb > fbits) { fbits = fb; cbits = cb; }
	}
	const int nfine = 1 << cbits;
	if (per_col * ldexp((double) nfine, fbits) / (double) nrow < 1.0)      // less than one nonzero per thread of pass 2
		return false;
	sh->cbits = cbits;
	sh->fbits = fbits;
	sh->nfb = (nrow + ((int64_t) 1 << fbits) - 1) >> fbits;
	sh->ncoarse = (sh->nfb + nfine - 1) / nfine;
	sh->ngroups = (ncol + T2_NT - 1) / T2_NT;
	sh->nslab = nslab; sh->srow = nrow; sh->scol = ncol;
	const double ntab = (double) nslab * (double) sh->ncoarse * (double) sh->ngroups * nfine;
	const double nwg = (double) nslab * (double) sh->ncoarse * (double) sh->ngroups;
	if ((double) nslab * (double) sh->nfb >= 2.0e9 || (double) nslab * (double) ncol >= 2.0e9)
		return false;
	// many small slabs: a pass-3 workgroup per fine bucket needs a few hundred nonzeros to be worth its launch
	// (slabs of 2e4 x 64 with 6400 nonzeros would make 6e6 workgroups of 20) -- the key sort takes those
	if (nslab > 1 && ldexp(per_row, fbits) < 512.0)
		return false;
	if ((double) (sh->ncoarse + 1) * (double) ncol > 2.0 * (double) nnz + 32.0)     // (the table of coarse-bucket starts)
		return false;
	return ntab < 1.0e8 && nwg < 2.0e9 && sh->ngroups <= 6000;      // (pass 3 keeps one int per group in LDS)
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

// [sorted rows nnz*4][positions nnz*4][sorted positions nnz*4][second buffers of the sort's passes 2 * nnz*4][column hints]
// [the sort's histograms]
static size_t transpose_sorted_ws_bytes(int64_t nrow, int64_t nnz)
{
	(void) nrow;
	const size_t a = ((size_t) (nnz > 0 ? nnz : 1) * 4 + 255) / 256 * 256;
	return 5 * a + hint_bytes(nnz) + svt_sort_ws_bytes(nnz) + 256;
}

static int launch_transpose_sorted(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
		     int64_t nrow, int64_t ncol, int64_t nnz, int64_t *out_ptr, int32_t *out_idx,
		     void *out_val, void *ws, hipStream_t s)
{
	if (nnz >= ((int64_t) 1 << 31))
		return svt_set_unsupported("svt_dev_transpose: more than 2^31-1 nonzeros");
	g_route[1]++;
	const unsigned nbr = (unsigned) ((nrow + 1 + 255) / 256);
	if (nnz == 0) {
		HIP_TRY(hipMemsetAsync(out_ptr, 0, (size_t) (nrow + 1) * 8, s));
		return 0;
	}
	const size_t a = ((size_t) nnz * 4 + 255) / 256 * 256;
	int32_t *srows = (int32_t *) ws;
	uint32_t *pos = (uint32_t *) ((char *) ws + a);
	uint32_t *perm = (uint32_t *) ((char *) ws + 2 * a);
	uint32_t *ktmp = (uint32_t *) ((char *) ws + 3 * a), *ptmp = (uint32_t *) ((char *) ws + 4 * a);
	uint32_t *hint = (uint32_t *) ((char *) ws + 5 * a);
	void *tmp = (char *) ws + 5 * a + hint_bytes(nnz);
	const int bits = key_bits(nrow);
	const unsigned nb = (unsigned) ((nnz + 255) / 256);
	const unsigned nb8 = (unsigned) (((nnz + 255) / 256 + 7) / 8 * 8);      // whole rounds over the 8 XCDs (xcd_chunk)
	hipLaunchKernelGGL(iota_u32_kernel, dim3(nb), dim3(256), 0, s, pos, nnz);
	const int64_t nblk = (nnz >> HINT_SHIFT) + 1;
	hipLaunchKernelGGL(col_hint_kernel, dim3((unsigned) ((nblk + 1 + 255) / 256)), dim3(256), 0, s, col_ptr, ncol, nblk, hint);
	// stable sort of (row, position) by row: positions ascend inside a row = (row, column) order
	if (svt_sort_pairs<uint32_t>((const uint32_t *) row_idx, (uint32_t *) srows, ktmp, pos, perm, ptmp, nnz, bits, tmp, s))
		return -1;
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
// Head of the workspace kept for the bucket table, the fine buckets' output positions and the scan scratch.
// The caller sizes the workspace from (nrow, nnz) alone; the shapes the bucketed form accepts (at least one
// nonzero per column and coarse bucket) have at most ~nnz / 8 + 512 * nnz / nrow table entries and at most
// nrow fine buckets, and never more than 96 MiB are set aside: an operand that needs more takes the key sort.
static size_t t2_reserve(int64_t nrow, int64_t nnz)
{
	const double per_row = nrow > 0 ? (double) nnz / (double) nrow : 0.0;
	const double ntab = (double) nnz / 8.0 + 512.0 * per_row + (double) nrow + 4096.0;
	const double b = 8.25 * ntab + 65536.0;
	const size_t cap = (size_t) 96 << 20;
	return b >= (double) cap ? cap : ((size_t) b + 255) / 256 * 256;
}

size_t transpose_ws_bytes(int64_t nrow, int64_t nnz)
{
	const size_t n = (size_t) (nnz > 0 ? nnz : 1);
	const size_t sorted = transpose_sorted_ws_bytes(nrow, nnz);
	// (+ the coarse-bucket starts per column: (ncoarse + 1) * ncol <= 2 * nnz entries for the shapes accepted)
	const size_t b2 = t2_reserve(nrow, nnz) + t2_a(n, 4) + t2_a(n, 1) + t2_a(n, 8) + t2_a(2 * n + 64, 4) + 512;
	return sorted > b2 ? sorted : b2;
}

template <typename T>
static int launch_transpose_bucketed(const int64_t *col_ptr, const int32_t *row_idx, const T *val,
				     int64_t nrow, int64_t ncol, int64_t nnz, const T2Shape &sh, int64_t *out_ptr,
				     int32_t *out_idx, T *out_val, void *ws, size_t reserve, hipStream_t s)
{
	const int64_t ntab = ((sh.nslab * sh.ncoarse * sh.ngroups) << sh.cbits) + 1;
	char *p = (char *) ws;
	int64_t *table = (int64_t *) p;            p += t2_a((size_t) ntab, 8);
	int64_t *fb_base = (int64_t *) p;          p += t2_a((size_t) (sh.nslab * sh.nfb + 1), 8);
	void *scan_ws = p;
	p = (char *) ws + reserve;
	int32_t *col1 = (int32_t *) p;             p += t2_a((size_t) nnz, 4);
	uint8_t *rlow1 = (uint8_t *) p;            p += t2_a((size_t) nnz, 1);
	T *val1 = (T *) p;                         p += t2_a((size_t) nnz, 8);
	uint32_t *cstart = (uint32_t *) p;         // [(ncoarse + 1) * ncol]
	HIP_TRY(hipMemsetAsync(table, 0, (size_t) ntab * 8, s));
	const size_t hist_b = (size_t) ((sh.nfb < T1_HIST ? sh.nfb : T1_HIST) + 1) / 2 * 4;
	(void) hipFuncSetAttribute((const void *) transpose_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
				   T1_HIST * 2);
	hipLaunchKernelGGL(transpose_count_kernel, dim3((unsigned) (sh.nslab * ((sh.scol + T1_NT / 64 - 1) / (T1_NT / 64)))), dim3(T1_NT),
			   hist_b, s, col_ptr, row_idx, ncol, sh, (unsigned long long *) table, cstart);
	if (launch_exclusive_scan_i64(table, ntab, scan_ws, s))
		return -1;
	hipLaunchKernelGGL(transpose_fb_base_kernel, dim3((unsigned) ((sh.nslab * sh.ncoarse + 3) / 4)), dim3(256), 0, s,
			   table, sh, nnz, fb_base);
	hipLaunchKernelGGL(transpose_scatter_kernel<T>, dim3((unsigned) (sh.nslab * sh.ngroups * ((sh.ncoarse + T2_CPW - 1) / T2_CPW))), dim3(T2_NT), 0, s,
			   col_ptr, row_idx, val, nrow, ncol, sh, table, cstart, col1, rlow1, val1);
	const size_t lds = ((sizeof(SplitLds<T3_NT, T3_ITEMS, 64>) + 15) & ~(size_t) 15) + (size_t) sh.ngroups * 8 +
			   (size_t) ((sh.ngroups + 1 + 3) & ~(int64_t) 3) * 4 + (size_t) T3_STAGE * 4 + (size_t) T3_STAGE * sizeof(T);
	(void) hipFuncSetAttribute((const void *) transpose_finish_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
				   (int) lds);
	hipLaunchKernelGGL(transpose_finish_kernel<T>, dim3((unsigned) (sh.nslab * sh.nfb)), dim3(T3_NT), lds, s, table, sh, nrow, nnz,
			   col1, rlow1, val1, fb_base, out_ptr, out_idx, out_val);
	HIP_TRY(hipGetLastError());
	return 0;
}

int launch_transpose(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
		     int64_t nrow, int64_t ncol, int64_t nnz, int64_t *out_ptr, int32_t *out_idx,
		     void *out_val, void *ws, hipStream_t s)
{
	if (nnz >= ((int64_t) 1 << 31))
		return svt_set_unsupported("svt_dev_transpose: more than 2^31-1 nonzeros");
	if (nnz == 0) {
		HIP_TRY(hipMemsetAsync(out_ptr, 0, (size_t) (nrow + 1) * 8, s));
		return 0;
	}
	T2Shape sh;
	const size_t reserve = t2_reserve(nrow, nnz);
	bool bucketed = t2_shape(nrow, ncol, nnz, &sh);
	if (bucketed) {
		const int64_t ntab = ((sh.ncoarse * sh.ngroups) << sh.cbits) + 1;
		bucketed = t2_a((size_t) ntab, 8) + t2_a((size_t) (sh.nfb + 1), 8) + exclusive_scan_ws_bytes(ntab) <= reserve;
	}
	// (one matrix: t2_shape() has left nslab = 1, srow = nrow, scol = ncol)
	if (!bucketed)
		return launch_transpose_sorted(col_ptr, row_idx, val, Rtype, nrow, ncol, nnz, out_ptr, out_idx, out_val, ws, s);
	g_route[0]++;
	if (Rtype == SVT_REALSXP)
		return launch_transpose_bucketed<double>(col_ptr, row_idx, (const double *) val, nrow, ncol, nnz, sh, out_ptr,
							 out_idx, (double *) out_val, ws, reserve, s);
	return launch_transpose_bucketed<int32_t>(col_ptr, row_idx, (const int32_t *) val, nrow, ncol, nnz, sh, out_ptr,
						  out_idx, (int32_t *) out_val, ws, reserve, s);
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

// out_ptr[j] = first sorted position whose leaf is >= j, j = 0 .. nleaves.  Most new leaves are empty when leaves
// shatter, so the work is WRITING the pointers (20 GB for the 2.56e9 leaves of aperm(x, c(3,2,4,1)) of a 2e4 x 2e3 x
// 10 x 64 array), not finding them.  Rounds 1-5: one binary search over the sorted keys per new leaf on the 64-bit
// route (48 ms there).  Round 6, first form: every sorted element fills the leaves between its predecessor's and its
// own (9.6 ms: a lane's run of leaves is contiguous, a wavefront's store instruction is not).  Now: a workgroup takes
// 256 consecutive sorted elements (plus the sentinel "element" nnz with leaf nleaves), keeps their leaves in LDS and
// fills the range of leaves they cover with consecutive lanes on consecutive leaves -- coalesced stores --, each lane
// finding its leaf's element by a binary search in LDS.  K = uint32_t: the key is the leaf; unsigned long long:
// leaf * dim0 + row.
#define PTRFILL_NT 256
template <typename K>
__global__ void __launch_bounds__(PTRFILL_NT)
aperm_ptr_fill_kernel(const K *__restrict__ skeys, int64_t nnz, int64_t nleaves, unsigned long long dim0,
		      int64_t *__restrict__ out_ptr)
{
	__shared__ int64_t s_leaf[PTRFILL_NT];
	__shared__ int64_t s_first;
	const int tid = threadIdx.x;
	const int64_t i0 = (int64_t) blockIdx.x * PTRFILL_NT, i = i0 + tid;
	// element i covers the leaves (leaf(i - 1), leaf(i)]; past the sentinel: nothing (a leaf no search will stop at)
	s_leaf[tid] = i < nnz ? (int64_t) ((unsigned long long) skeys[i] / dim0) : (i == nnz ? nleaves : nleaves + 1);
	if (tid == 0)
		s_first = i0 == 0 ? 0 : (int64_t) ((unsigned long long) skeys[i0 - 1] / dim0) + 1;
	__syncthreads();
	const int nvalid = (int) (nnz + 1 - i0 < PTRFILL_NT ? nnz + 1 - i0 : PTRFILL_NT);
	const int64_t first = s_first, last = s_leaf[nvalid - 1];
	for (int64_t j = first + tid; j <= last; j += PTRFILL_NT) {
		int lo = 0, hi = nvalid - 1;                    // smallest t with s_leaf[t] >= j (exists: s_leaf[nvalid - 1] = last >= j)
		while (lo < hi) {
			const int mid = (lo + hi) >> 1;
			if (s_leaf[mid] < j) lo = mid + 1; else hi = mid;
		}
		out_ptr[j] = i0 + lo;
	}
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

// aperm(x, c(2, 1, 3, ...)): the first two axes change places inside every slab of the remaining ones -- prod(dim[2..])
// independent transpositions of dim[0] x dim[1] matrices that follow one another in the operand's leaves: the bucketed
// transposition, batched (T2Shape).  Returns the workspace it needs (0: the shape does not suit it) and fills *sh, *reserve.
static size_t aperm_swap01_bytes(int64_t nnz, const int64_t *dim, int ndim, T2Shape *sh, size_t *reserve)
{
	if (ndim < 3 || nnz <= 0 || nnz >= ((int64_t) 1 << 31))
		return 0;
	double ns = 1.0;
	for (int a = 2; a < ndim; a++) ns *= (double) dim[a];
	if (ns < 1.0 || ns > 1.0e9)
		return 0;
	if (!t2_shape(dim[0], dim[1], nnz, sh, (int64_t) ns))
		return 0;
	const int64_t ntab = ((sh->nslab * sh->ncoarse * sh->ngroups) << sh->cbits) + 1;
	const size_t res = t2_a((size_t) ntab, 8) + t2_a((size_t) (sh->nslab * sh->nfb + 1), 8) + exclusive_scan_ws_bytes(ntab) + 256;
	const size_t cst = t2_a((size_t) (sh->ncoarse + 1) * (size_t) (sh->nslab * sh->scol), 4);
	if (reserve) *reserve = (res + 255) / 256 * 256;
	return (res + 255) / 256 * 256 + t2_a((size_t) nnz, 4) + t2_a((size_t) nnz, 1) + t2_a((size_t) nnz, 8) + cst + 512;
}

static size_t aperm_ws_core(int64_t nnz, const int64_t *dim, int ndim);

// 3-d arrays: the two permutations that neither keep axis 1 nor are one of the forms above are two of those in a row --
// c(2,3,1) = c(2,1,3) then c(1,3,2); c(3,2,1) = c(2,1,3) then c(3,1,2) -- through an intermediate array in the workspace
// (9.3 and 11.5 ms through the library key sort at BASELINE config 5).  Bytes of that intermediate (0: not applicable).
static size_t aperm_via_bytes(int64_t nnz, const int64_t *dim, int ndim)
{
	if (ndim != 3 || nnz <= 0)
		return 0;
	const double nly = (double) dim[0] * (double) dim[2];
	if (nly >= 2147483646.0)
		return 0;
	return t2_a((size_t) nly + 1, 8) + t2_a((size_t) nnz, 4) + t2_a((size_t) nnz, 8);
}

// General permutation of an array with three or more axes (round 5; rounds 1-4: a device-wide key sort):
//   A  aperm(x, c(1, q, others))   old axis q = perm[0] next to the rows; axis 1 stays: whole leaves move (no step when q = 2)
//   B  aperm(., c(2, 1, 3, ...))   the first two axes change places: the batched bucketed transposition
//   C  aperm(., c(1, ...))         the remaining axes into their final order: whole leaves move (no step when they are in order)
// through up to two intermediate arrays in the workspace.  aperm_general_plan() fills the steps for a given perm and returns
// false when the shape does not suit (leaf counts past 2^31, or a first-two-axes matrix the bucketed transposition refuses):
// those arrays take the library's own radix sort of (new linear index, position) pairs (svt_sort.h).
struct ApermPlan3 {
	bool a_id, c_id;
	bool swap_ok;             // step B as the batched bucketed transposition
	bool slab_first;          // steps A + B in one: aperm(x, c(q, 1, others)) by the slab form (one workgroup per slab), tried first
	int pa[8], pc[8];
	int64_t dim_a[8], dim_b[8];
	int64_t leaves_a, leaves_b;
};

static size_t aperm_inter_bytes(int64_t leaves, int64_t nnz)
{
	return t2_a((size_t) leaves + 1, 8) + t2_a((size_t) (nnz > 0 ? nnz : 1), 4) + t2_a((size_t) (nnz > 0 ? nnz : 1), 8);
}

static bool aperm_general_plan(int64_t nnz, const int64_t *dim, int ndim, const int *perm, ApermPlan3 *pl)
{
	if (ndim < 3 || perm[0] == 0 || nnz <= 0)
		return false;
	const int q = perm[0];
	pl->pa[0] = 0; pl->pa[1] = q;
	for (int a = 1, i = 2; a < ndim; a++)
		if (a != q) pl->pa[i++] = a;
	pl->a_id = q == 1;
	double la = 1.0, lb = 1.0;
	for (int i = 0; i < ndim; i++) pl->dim_a[i] = dim[pl->pa[i]];
	for (int i = 0; i < ndim; i++) pl->dim_b[i] = i == 0 ? pl->dim_a[1] : i == 1 ? pl->dim_a[0] : pl->dim_a[i];
	for (int i = 1; i < ndim; i++) { la *= (double) pl->dim_a[i]; lb *= (double) pl->dim_b[i]; }
	if (la >= 2147483646.0 || lb >= 2147483646.0 || la < 1.0 || lb < 1.0)
		return false;
	pl->leaves_a = (int64_t) la; pl->leaves_b = (int64_t) lb;
	// a leaf-preserving step reads and writes every leaf pointer a few times: with many more leaves than nonzeros (a short
	// new leading axis: 4e8 leaves of at most 64 entries for aperm(x, c(4,3,2,1)) of a 2e4 x 2e3 x 10 x 64 array) the
	// steps cost more than the sort of the nonzeros (41 against 28 ms there)
	if (la > 2.0 * (double) nnz + 1024.0 || lb > 2.0 * (double) nnz + 1024.0)
		return false;
	// axes of the array after step B, by old axis: l2 = (q, 0, pa[2], pa[3], ...)
	pl->c_id = true;
	pl->pc[0] = 0;
	for (int a = 1; a < ndim; a++) {
		int at = -1;
		for (int i = 1; i < ndim; i++) {
			const int old_axis = i == 1 ? 0 : pl->pa[i];
			if (old_axis == perm[a]) at = i;
		}
		if (at < 0)
			return false;
		pl->pc[a] = at;
		if (at != a) pl->c_id = false;
	}
	T2Shape sh;
	pl->swap_ok = aperm_swap01_bytes(nnz, pl->dim_a, ndim, &sh, NULL) > 0;
	// many small slabs (which the batched transposition refuses) are what the slab form is for; it serves a request
	// that still has a step C to do -- c(q, 1, others) itself has met the slab form on its way here
	pl->slab_first = !pl->c_id && dim[q] <= 1024 && dim[0] < ((int64_t) 1 << 30) &&
			 (double) nnz / (lb / (double) dim[0]) <= 0.9 * 8192.0;
	return pl->swap_ok || pl->slab_first;
}

// Workspace of the three-step form over all permutations of `dim`: the two intermediates + the largest step.
static size_t aperm_general_bytes(int64_t nnz, const int64_t *dim, int ndim, size_t *inter_max)
{
	*inter_max = 0;
	if (ndim < 3 || nnz <= 0)
		return 0;
	size_t need = 0;
	for (int q = 1; q < ndim; q++) {
		int perm[8];
		perm[0] = q; perm[1] = 0;
		for (int a = 1, i = 2; a < ndim; a++)
			if (a != q) perm[i++] = a;
		ApermPlan3 pl;
		if (!aperm_general_plan(nnz, dim, ndim, perm, &pl))
			continue;
		T2Shape sh;
		size_t step = aperm_swap01_bytes(nnz, pl.dim_a, ndim, &sh, NULL);
		const size_t sa = exclusive_scan_ws_bytes(pl.leaves_a + 1) + 256, sb = exclusive_scan_ws_bytes(pl.leaves_b + 1) + 256;
		if (sa > step) step = sa;
		if (sb > step) step = sb;
		const size_t inter = aperm_inter_bytes(pl.leaves_a, nnz) + aperm_inter_bytes(pl.leaves_b, nnz) + 512;
		if (inter > *inter_max) *inter_max = inter;
		if (inter + step > need) need = inter + step;
	}
	return need;
}

size_t aperm_ws_bytes(int64_t nnz, const int64_t *dim, int ndim)
{
	size_t need = aperm_ws_core(nnz, dim, ndim);
	{
		// (a step of the general form may itself fall back to the forms that aperm_ws_core() sizes: the intermediates
		// come on top of the larger of the two)
		size_t inter = 0;
		const size_t gen = aperm_general_bytes(nnz, dim, ndim, &inter);
		if (gen > need + inter) need = gen; else need += inter;
	}
	const size_t via = aperm_via_bytes(nnz, dim, ndim);
	if (via > 0) {
		const int64_t dimy[3] = {dim[1], dim[0], dim[2]};
		const size_t second = aperm_ws_core(nnz, dimy, 3);
		need = via + (need > second ? need : second);
	}
	return need;
}

static size_t aperm_ws_core(int64_t nnz, const int64_t *dim, int ndim)
{
	const size_t n = (size_t) (nnz > 0 ? nnz : 1);
	const size_t a8 = (n * 8 + 255) / 256 * 256, a4 = (n * 4 + 255) / 256 * 256;
	// (leaf-preserving permutations: only the scratch of a scan over the leaf counts)
	double nl = 1.0;
	for (int a = 1; a < ndim; a++) nl *= (double) (dim[a] > 0 ? dim[a] : 1);
	size_t scan_b = 0;
	if (nl < 2147483646.0)
		scan_b = exclusive_scan_ws_bytes((int64_t) nl + 1);
	const size_t t32 = svt_sort_ws_bytes(nnz);
	const size_t need64 = 3 * a8 + 3 * a4 + t32;
	const size_t need32 = 7 * a4 + hint_bytes(nnz) + t32;
	T2Shape sh;
	const size_t swap01 = aperm_swap01_bytes(nnz, dim, ndim, &sh, NULL);
	size_t need = need64 > need32 ? need64 : need32;
	if (swap01 > need) need = swap01;
	return need + scan_b + 256;
}

// nested != 0: a step of a composed route (the 3-d "via" form, the general form).  Such a call gets what is left of
// the workspace behind its caller's intermediates, which aperm_ws_bytes() sizes for the forms of aperm_ws_core() only:
// it must not carve intermediates of its own (a slab form that refuses at run time -- one slab over
// SLAB_CAP -- used to re-enter the general form inside `sub` and write past the workspace), so it goes from the
// direct forms (leaf-preserving, first two axes swapped, slab) straight to the key sort.
static int launch_aperm_n(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
			  int64_t ncol, int64_t nnz, const int64_t *dim, int ndim, const int *perm,
			  int64_t *out_ptr, int32_t *out_idx, void *out_val, void *ws, hipStream_t s, int nested);

int launch_aperm(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
		 int64_t ncol, int64_t nnz, const int64_t *dim, int ndim, const int *perm,
		 int64_t *out_ptr, int32_t *out_idx, void *out_val, void *ws, hipStream_t s)
{
	return launch_aperm_n(col_ptr, row_idx, val, Rtype, ncol, nnz, dim, ndim, perm, out_ptr, out_idx, out_val, ws, s, 0);
}

static int launch_aperm_n(const int64_t *col_ptr, const int32_t *row_idx, const void *val, int Rtype,
			  int64_t ncol, int64_t nnz, const int64_t *dim, int ndim, const int *perm,
			  int64_t *out_ptr, int32_t *out_idx, void *out_val, void *ws, hipStream_t s, int nested)
{
	if (ndim < 1 || ndim > 8)
		return svt_set_error("aperm: between 1 and 8 dimensions are supported");
	if (nnz >= ((int64_t) 1 << 31))
		return svt_set_unsupported("aperm: more than 2^31-1 nonzeros");
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
		g_route[2]++;
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
	// 3-d: c(2,3,1) and c(3,2,1) as c(2,1,3) followed by c(1,3,2) / c(3,1,2) (see aperm_via_bytes)
	if (!nested && ndim == 3 && ((perm[0] == 1 && perm[1] == 2 && perm[2] == 0) ||
			  (perm[0] == 2 && perm[1] == 1 && perm[2] == 0 && dim[2] <= 1024))) {
		T2Shape sh;
		size_t reserve = 0;
		const size_t via = aperm_via_bytes(nnz, dim, 3);
		if (via > 0 && aperm_swap01_bytes(nnz, dim, 3, &sh, &reserve) > 0) {
			g_route[5]++;
			const int64_t nly = dim[0] * dim[2];
			char *p = (char *) ws;
			int64_t *ycp = (int64_t *) p;         p += t2_a((size_t) nly + 1, 8);
			int32_t *yri = (int32_t *) p;         p += t2_a((size_t) nnz, 4);
			void *yv = p;
			void *sub = (char *) ws + via;
			int rc;
			if (Rtype == SVT_REALSXP)
				rc = launch_transpose_bucketed<double>(col_ptr, row_idx, (const double *) val, dim[0], ncol, nnz, sh,
								       ycp, yri, (double *) yv, sub, reserve, s);
			else
				rc = launch_transpose_bucketed<int32_t>(col_ptr, row_idx, (const int32_t *) val, dim[0], ncol, nnz, sh,
									ycp, yri, (int32_t *) yv, sub, reserve, s);
			if (rc)
				return rc;
			const int64_t dimy[3] = {dim[1], dim[0], dim[2]};
			const int p231[3] = {0, 2, 1}, p321[3] = {2, 0, 1};
			return launch_aperm_n(ycp, yri, yv, Rtype, nly, nnz, dimy, 3, perm[0] == 1 ? p231 : p321,
					      out_ptr, out_idx, out_val, sub, s, 1);
		}
	}
	// the first two axes change places, the others stay: one batched bucketed transposition, no sort
	{
		bool swap01 = ndim >= 3 && perm[0] == 1 && perm[1] == 0;
		for (int a = 2; a < ndim && swap01; a++) swap01 = perm[a] == a;
		T2Shape sh;
		size_t reserve = 0;
		if (swap01 && aperm_swap01_bytes(nnz, dim, ndim, &sh, &reserve) > 0) {
			g_route[3]++;
			if (Rtype == SVT_REALSXP)
				return launch_transpose_bucketed<double>(col_ptr, row_idx, (const double *) val, dim[0], ncol, nnz, sh,
									 out_ptr, out_idx, (double *) out_val, ws, reserve, s);
			return launch_transpose_bucketed<int32_t>(col_ptr, row_idx, (const int32_t *) val, dim[0], ncol, nnz, sh,
								  out_ptr, out_idx, (int32_t *) out_val, ws, reserve, s);
		}
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
			if (mx > SLAB_CAP) g_route[9]++;
			if (mx <= SLAB_CAP) {
				g_route[4]++;
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
	// the general form: leaf-preserving step, first two axes swapped, leaf-preserving step (aperm_general_plan)
	if (!nested) {
		ApermPlan3 pl;
		if (aperm_general_plan(nnz, dim, ndim, perm, &pl)) {
			g_route[6]++;
			char *p = (char *) ws;
			const int64_t *cp_a = col_ptr; const int32_t *ri_a = row_idx; const void *v_a = val;
			int64_t ncol_a = ncol;
			int64_t *xcp = NULL; int32_t *xri = NULL; void *xv = NULL;
			if (!pl.a_id && !pl.slab_first) {       // (the slab form goes from the operand to y in one step)
				xcp = (int64_t *) p; p += t2_a((size_t) pl.leaves_a + 1, 8);
				xri = (int32_t *) p; p += t2_a((size_t) nnz, 4);
				xv = p;              p += t2_a((size_t) nnz, 8);
			}
			int64_t *ycp = out_ptr; int32_t *yri = out_idx; void *yv = out_val;
			if (!pl.c_id) {
				ycp = (int64_t *) p; p += t2_a((size_t) pl.leaves_b + 1, 8);
				yri = (int32_t *) p; p += t2_a((size_t) nnz, 4);
				yv = p;              p += t2_a((size_t) nnz, 8);
			}
			void *sub = p;
			if (pl.slab_first) {
				// c(q, 1, others) in one go (slab form; if a slab turns out too long the call takes the two steps, or the
				// sort, itself), then the leaf-preserving step
				int p1[8];
				p1[0] = perm[0]; p1[1] = 0;
				for (int a = 2; a < ndim; a++) p1[a] = pl.pa[a];
				int rc = launch_aperm_n(col_ptr, row_idx, val, Rtype, ncol, nnz, dim, ndim, p1, ycp, yri, yv, sub, s, 1);
				if (rc) return rc;
				return launch_aperm_n(ycp, yri, yv, Rtype, pl.leaves_b, nnz, pl.dim_b, ndim, pl.pc, out_ptr, out_idx, out_val, sub, s, 1);
			}
			if (!pl.a_id) {
				const int rc = launch_aperm_n(col_ptr, row_idx, val, Rtype, ncol, nnz, dim, ndim, pl.pa, xcp, xri, xv, sub, s, 1);
				if (rc) return rc;
				cp_a = xcp; ri_a = xri; v_a = xv; ncol_a = pl.leaves_a;
			}
			int pb[8];
			for (int a = 0; a < ndim; a++) pb[a] = a == 0 ? 1 : a == 1 ? 0 : a;
			int rc = launch_aperm_n(cp_a, ri_a, v_a, Rtype, ncol_a, nnz, pl.dim_a, ndim, pb, ycp, yri, yv, sub, s, 1);
			if (rc) return rc;
			if (!pl.c_id)
				rc = launch_aperm_n(ycp, yri, yv, Rtype, pl.leaves_b, nnz, pl.dim_b, ndim, pl.pc, out_ptr, out_idx, out_val, sub, s, 1);
			return rc;
		}
	}
	if (new_nleaves < ((int64_t) 1 << 31) - 1) {
		const size_t a4 = ((size_t) nnz * 4 + 255) / 256 * 256;
		uint32_t *keys = (uint32_t *) ws, *skeys = (uint32_t *) ((char *) ws + a4);
		uint32_t *pos = (uint32_t *) ((char *) ws + 2 * a4), *spos = (uint32_t *) ((char *) ws + 3 * a4);
		int32_t *newrow = (int32_t *) ((char *) ws + 4 * a4);
		uint32_t *ktmp = (uint32_t *) ((char *) ws + 5 * a4), *ptmp = (uint32_t *) ((char *) ws + 6 * a4);
		uint32_t *hint = (uint32_t *) ((char *) ws + 7 * a4);
		void *tmp = (char *) ws + 7 * a4 + hint_bytes(nnz);
		int bits = 1;
		while (bits < 32 && ((int64_t) 1 << bits) < new_nleaves) bits++;
		const unsigned nb = (unsigned) ((nnz + 255) / 256);
		const unsigned nb8 = (unsigned) (((nnz + 255) / 256 + 7) / 8 * 8);
		const int64_t nblk = (nnz >> HINT_SHIFT) + 1;
		g_route[7]++;
		hipLaunchKernelGGL(col_hint_kernel, dim3((unsigned) ((nblk + 1 + 255) / 256)), dim3(256), 0, s, col_ptr, ncol, nblk, hint);
		hipLaunchKernelGGL(aperm_key32_kernel, dim3(nb), dim3(256), 0, s, col_ptr, row_idx, hint, ncol, nnz, d,
				   keys, pos, newrow);
		if (svt_sort_pairs<uint32_t>(keys, skeys, ktmp, pos, spos, ptmp, nnz, bits, tmp, s))
			return -1;
		hipLaunchKernelGGL(aperm_ptr_fill_kernel<uint32_t>, dim3((unsigned) ((nnz + 1 + PTRFILL_NT - 1) / PTRFILL_NT)), dim3(PTRFILL_NT), 0, s,
				   skeys, nnz, new_nleaves, 1ULL, out_ptr);
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
	unsigned long long *ktmp = (unsigned long long *) ((char *) ws + 2 * a8 + 2 * a4);
	uint32_t *ptmp = (uint32_t *) ((char *) ws + 3 * a8 + 2 * a4);
	void *tmp = (char *) ws + 3 * a8 + 3 * a4;
	const int bits = aperm_bits(dim, ndim);
	const unsigned nb = (unsigned) ((nnz + 255) / 256);
	const unsigned nb8 = (unsigned) (((nnz + 255) / 256 + 7) / 8 * 8);
	g_route[8]++;
	hipLaunchKernelGGL(aperm_key_kernel, dim3(nb), dim3(256), 0, s, col_ptr, row_idx, ncol, nnz, d, keys, pos);
	if (svt_sort_pairs<unsigned long long>(keys, skeys, ktmp, pos, spos, ptmp, nnz, bits, tmp, s))
		return -1;
	hipLaunchKernelGGL(aperm_ptr_fill_kernel<unsigned long long>, dim3((unsigned) ((nnz + 1 + PTRFILL_NT - 1) / PTRFILL_NT)), dim3(PTRFILL_NT), 0, s,
			   skeys, nnz, new_nleaves, (unsigned long long) (new_dim0 > 0 ? new_dim0 : 1), out_ptr);
	if (Rtype == SVT_REALSXP)
		hipLaunchKernelGGL(aperm_gather_kernel<double>, dim3(nb8), dim3(256), 0, s, skeys, spos,
				   (const double *) val, nnz, new_dim0, out_idx, (double *) out_val);
	else
		hipLaunchKernelGGL(aperm_gather_kernel<int32_t>, dim3(nb8), dim3(256), 0, s, skeys, spos,
				   (const int32_t *) val, nnz, new_dim0, out_idx, (int32_t *) out_val);
	HIP_TRY(hipGetLastError());
	return 0;
}
