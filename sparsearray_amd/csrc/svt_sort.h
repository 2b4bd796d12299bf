// Stable sort of (key, 32-bit payload) pairs on the device, least significant digit first, 8 bits per pass -- the
// library's own, for the shapes its sort-free passes do not take (the transposition of an operand with less than one
// nonzero per column and coarse bucket; permutations of N-d arrays whose composition into leaf-preserving steps and a
// batched transposition does not fit).  Rounds 1-4 called rocprim's device-wide radix_sort_pairs here.
//
// A pass over digit d:
//   (1) sort_hist_kernel     a workgroup counts the digit values of its tile of 2048 keys:  hist[value][tile];
//   (2) exclusive scan of hist in (value, tile) order (svt_scan.h) = where the tile's keys of each value start;
//   (3) sort_scatter_kernel  the workgroup orders its tile by the digit, stably (the block-level radix sort of
//                            rocprim, the one library primitive left in this directory, on (digit, index in tile)),
//                            and every element goes to start[value][tile] + its rank among the tile's keys of that
//                            value.
// Traffic per pass: keys read twice, keys and payloads read and written once (28 B per pair for 32-bit keys, 36 B for
// 64-bit ones); passes = ceil(bits / 8).  Not a fast path: every shape that matters takes the bucketed passes.
#pragma once

#include "svt_common.h"
#include "svt_scan.h"

#include <rocprim/block/block_radix_sort.hpp>

#define SVT_SORT_NT 256
#define SVT_SORT_ITEMS 8
#define SVT_SORT_TILE (SVT_SORT_NT * SVT_SORT_ITEMS)

// [hist 256 * ntiles * 8][scan scratch]
static inline size_t svt_sort_ws_bytes(int64_t n)
{
	const int64_t nt = (n + SVT_SORT_TILE - 1) / SVT_SORT_TILE;
	return ((size_t) 256 * (size_t) (nt > 0 ? nt : 1) * 8 + 255) / 256 * 256 + exclusive_scan_ws_bytes(256 * (nt > 0 ? nt : 1)) + 512;
}

template <typename K>
__global__ void __launch_bounds__(SVT_SORT_NT)
svt_sort_hist_kernel(const K *__restrict__ keys, int64_t n, int shift, int64_t ntiles, int64_t *__restrict__ hist)
{
	__shared__ unsigned cnt[256];
	cnt[threadIdx.x] = 0;
	__syncthreads();
	const int64_t base = (int64_t) blockIdx.x * SVT_SORT_TILE;
#pragma unroll
	for (int u = 0; u < SVT_SORT_ITEMS; u++) {
		const int64_t i = base + u * SVT_SORT_NT + threadIdx.x;
		if (i < n) atomicAdd(&cnt[(unsigned) (keys[i] >> shift) & 255u], 1u);
	}
	__syncthreads();
	hist[(int64_t) threadIdx.x * ntiles + blockIdx.x] = cnt[threadIdx.x];
}

template <typename K>
__global__ void __launch_bounds__(SVT_SORT_NT)
svt_sort_scatter_kernel(const K *__restrict__ keys, const uint32_t *__restrict__ pay, int64_t n, int shift,
			int64_t ntiles, const int64_t *__restrict__ start, K *__restrict__ keys_out,
			uint32_t *__restrict__ pay_out)
{
	typedef rocprim::block_radix_sort<uint32_t, SVT_SORT_NT, SVT_SORT_ITEMS, uint32_t> Sort;
	__shared__ typename Sort::storage_type tmp;
	__shared__ unsigned first[257];
	__shared__ uint32_t sdig[SVT_SORT_TILE];
	const int64_t base = (int64_t) blockIdx.x * SVT_SORT_TILE;
	const int tile_n = (int) (n - base < SVT_SORT_TILE ? n - base : SVT_SORT_TILE);
	// blocked arrangement: thread t holds the tile's elements t * ITEMS .. t * ITEMS + ITEMS - 1 (the order that counts)
	uint32_t dig[SVT_SORT_ITEMS], idx[SVT_SORT_ITEMS];
#pragma unroll
	for (int u = 0; u < SVT_SORT_ITEMS; u++) {
		const int e = threadIdx.x * SVT_SORT_ITEMS + u;
		idx[u] = (uint32_t) e;
		dig[u] = e < tile_n ? (uint32_t) (keys[base + e] >> shift) & 255u : 256u;    // (past the end: after every digit value)
	}
	Sort().sort(dig, idx, tmp, 0, 9);                          // stable, by the 9 bits of (digit | padding)
	__syncthreads();
#pragma unroll
	for (int u = 0; u < SVT_SORT_ITEMS; u++) sdig[threadIdx.x * SVT_SORT_ITEMS + u] = dig[u];
	__syncthreads();
	// first[v] = sorted position of the tile's first key with digit value >= v
	for (int v = threadIdx.x; v <= 256; v += SVT_SORT_NT) {
		int lo = 0, hi = SVT_SORT_TILE;
		while (lo < hi) {
			const int mid = (lo + hi) >> 1;
			if (sdig[mid] < (uint32_t) v) lo = mid + 1; else hi = mid;
		}
		first[v] = (unsigned) lo;
	}
	__syncthreads();
#pragma unroll
	for (int u = 0; u < SVT_SORT_ITEMS; u++) {
		const int p = threadIdx.x * SVT_SORT_ITEMS + u;    // sorted position inside the tile
		const uint32_t d = dig[u];
		if (d > 255u)
			continue;
		const int64_t to = start[(int64_t) d * ntiles + blockIdx.x] + (p - (int) first[d]);
		const int64_t from = base + idx[u];
		keys_out[to] = keys[from];
		pay_out[to] = pay[from];
	}
}

// Sorts (keys, pay)[0 .. n) by the low `bits` bits of the keys into (keys_out, pay_out); the inputs are only read, the
// passes alternate between the out arrays and (keys_tmp, pay_tmp) of the same sizes.  ws: svt_sort_ws_bytes(n).
// Stream-ordered.
template <typename K>
static inline int svt_sort_pairs(const K *keys, K *keys_out, K *keys_tmp, const uint32_t *pay, uint32_t *pay_out,
				 uint32_t *pay_tmp, int64_t n, int bits, void *ws, hipStream_t s)
{
	if (n <= 0)
		return 0;
	const int64_t nt = (n + SVT_SORT_TILE - 1) / SVT_SORT_TILE;
	if (nt > 0x7FFFFFFFLL / 256)
		return svt_set_error("sort: too many elements");
	int64_t *hist = (int64_t *) ws;
	void *scan_ws = (char *) ws + ((size_t) 256 * (size_t) nt * 8 + 255) / 256 * 256;
	int passes = (bits + 7) / 8;
	if (passes < 1) passes = 1;
	const K *src_k = keys;
	const uint32_t *src_p = pay;
	for (int ps = 0; ps < passes; ps++) {
		// the last pass lands in the out arrays, the one before it in the tmp arrays, ...
		const bool to_out = ((passes - 1 - ps) & 1) == 0;
		K *dst_k = to_out ? keys_out : keys_tmp;
		uint32_t *dst_p = to_out ? pay_out : pay_tmp;
		const int shift = ps * 8;
		hipLaunchKernelGGL((svt_sort_hist_kernel<K>), dim3((unsigned) nt), dim3(SVT_SORT_NT), 0, s, src_k, n, shift, nt, hist);
		if (launch_exclusive_scan_i64(hist, 256 * nt, scan_ws, s))
			return -1;
		hipLaunchKernelGGL((svt_sort_scatter_kernel<K>), dim3((unsigned) nt), dim3(SVT_SORT_NT), 0, s, src_k, src_p, n, shift,
				   nt, hist, dst_k, dst_p);
		src_k = dst_k; src_p = dst_p;
	}
	HIP_TRY(hipGetLastError());
	return 0;
}
