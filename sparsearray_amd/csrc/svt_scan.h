// Exclusive prefix sum of an int64 array in place (device): count / scan / scatter passes all over the
// library need one (tile tables of the panel-blocked layout, leaf pointers of aperm(), bucket tables of
// the transposition).  Three small kernels, no library: (1) every 256-thread workgroup scans 4096
// elements and leaves its total, (2) the totals are scanned the same way (recursively: two levels reach
// 2^24 elements, three reach 2^36), (3) the totals are added back.  Reads and writes every element
// twice; the arrays it is used on are a few MB.
#pragma once

#include "svt_common.h"

#define SVT_SCAN_NT 256
#define SVT_SCAN_ITEMS 16
#define SVT_SCAN_TILE (SVT_SCAN_NT * SVT_SCAN_ITEMS)

// scratch bytes for n elements (the block totals of every level)
static inline size_t exclusive_scan_ws_bytes(int64_t n)
{
	size_t b = 256;
	while (n > SVT_SCAN_TILE) {
		n = (n + SVT_SCAN_TILE - 1) / SVT_SCAN_TILE;
		b += ((size_t) n * 8 + 255) / 256 * 256;
	}
	return b;
}

// data[0 .. n) <- exclusive prefix sums; asynchronous on `s`
int launch_exclusive_scan_i64(int64_t *data, int64_t n, void *ws, hipStream_t s);
