// Host side of libsvt_hip.so: the C ABI of include/svt_hip.h.
//
// Host-level functions restate the argument checks and dispatch of the
// reference's .Call entry points (file:line at each function), marshal the
// SVT leaves into the CSC device layout (model:
// dump_SVT_to_CsparseMatrix_slots, src/SVT_SparseArray_class.c:598-633),
// launch the kernels and copy the result back.  No arithmetic of the hot path
// happens on the host.
#include "svt_common.h"
#include <unistd.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

static thread_local char g_err[1024];
static int g_device = -1;
static char g_arch[64] = "";

int svt_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return -1;
}

// "Not supported here" (include/svt_hip.h: status > 0): inside the library it unwinds like an error (-1, with the
// message), and the entry point that hands a status to the caller turns it into 1 (svt_status).  A caller inside
// the library that recovers from it (another route) clears the mark.
static thread_local int g_unsupported = 0;
int svt_set_unsupported(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	g_unsupported = 1;
	return -1;
}
void svt_clear_unsupported(void) { g_unsupported = 0; }
static inline int svt_status(int rc)
{
	return rc < 0 && g_unsupported ? 1 : rc;
}

extern "C" const char *svt_last_error(void) { return g_err; }
extern "C" const char *svt_device_arch(void) { return g_arch; }

extern "C" int svt_init(int device)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		return svt_set_error("libsvt_hip: no HIP device visible -- the SVT "
				     "backend has no CPU fallback");
	if (device < 0 || device >= n)
		return svt_set_error("libsvt_hip: device %d out of range (0..%d)",
				     device, n - 1);
	HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	snprintf(g_arch, sizeof(g_arch), "%s", prop.gcnArchName);
	if (strncmp(g_arch, "gfx950", 6) != 0)
		return svt_set_error("libsvt_hip: device %d is %s; this library "
				     "carries gfx950 (MI355X) code only", device, g_arch);
	g_device = device;
	return 0;
}

// ---- thread control (src/thread_control.c:47-66) ---------------------------------
static int g_max_threads = 0;
extern "C" int svt_get_num_procs(void)
{
	const long n = sysconf(_SC_NPROCESSORS_ONLN);
	return n > 0 ? (int) n : 0;
}
extern "C" int svt_get_max_threads(void)
{
	return g_max_threads > 0 ? g_max_threads : svt_get_num_procs();
}
extern "C" int svt_set_max_threads(int nthread)
{
	const int prev = svt_get_max_threads();
	if (nthread > 0) g_max_threads = nthread;
	return prev;
}

static int ensure_init()
{
	if (g_device >= 0)
		return 0;
	return svt_init(0);
}

// ---- host -> device staging ----------------------------------------------------------
// .Call hands over pageable host memory.  hipMemcpy() from pageable memory runs at a
// few GB/s; instead the bytes go through two pinned buffers: a small thread team
// gathers the next chunk (for an SVT: the leaves' nzoffs / nzvals, scattered over
// the R heap) into one buffer while the previous one is in flight on a copy stream
// (the reference's counterpart is the leaf walk of
// src/SVT_SparseArray_class.c:598-633, which never leaves the host).
#include <algorithm>
#include <functional>
#include <mutex>
#include <vector>
#include <thread>

struct Stager {
	static const size_t CHUNK = (size_t) 48 << 20;      // bytes per pinned buffer
	char *buf[2] = {NULL, NULL};
	hipEvent_t done[2];
	hipStream_t stream = NULL;
	bool ok = false;
	int next = 0;
	bool busy[2] = {false, false};

	int init()
	{
		if (ok) return 0;
		for (int i = 0; i < 2; i++) {
			HIP_TRY(hipHostMalloc((void **) &buf[i], CHUNK, hipHostMallocDefault));
			HIP_TRY(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
		}
		HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
		ok = true;
		return 0;
	}
	// a free pinned buffer (waits for the copy that last used it)
	int acquire(char **p, int *slot)
	{
		if (init()) return -1;
		const int i = next;
		next ^= 1;
		if (busy[i]) { HIP_TRY(hipEventSynchronize(done[i])); busy[i] = false; }
		*p = buf[i]; *slot = i;
		return 0;
	}
	int send(int slot, void *dst, size_t off_in_buf, size_t n)
	{
		if (n) HIP_TRY(hipMemcpyAsync(dst, buf[slot] + off_in_buf, n, hipMemcpyHostToDevice, stream));
		return 0;
	}
	int commit(int slot)
	{
		HIP_TRY(hipEventRecord(done[slot], stream));
		busy[slot] = true;
		return 0;
	}
	int drain()
	{
		if (!ok) return 0;
		HIP_TRY(hipStreamSynchronize(stream));
		busy[0] = busy[1] = false;
		return 0;
	}
};
static Stager g_stager;

// Bytes of a pinned buffer used per trip.  SVT_STAGING_CHUNK (bytes, read once) lowers it so
// that a test can drive the multi-chunk paths -- a leaf longer than a chunk among them --
// with small inputs.
static size_t g_stager_chunk()
{
	static size_t v = 0;
	if (v == 0) {
		v = Stager::CHUNK;
		const char *e = getenv("SVT_STAGING_CHUNK");
		if (e != NULL) {
			const long long t = atoll(e);
			if (t >= 4096 && (size_t) t < Stager::CHUNK) v = (size_t) t / 4096 * 4096;
		}
	}
	return v;
}

// run fn(t, nt) on a small team (the calling thread is one of them)
static void team_run(int nt, const std::function<void(int, int)> &fn)
{
	std::vector<std::thread> th;
	for (int t = 1; t < nt; t++) th.emplace_back(fn, t, nt);
	fn(0, nt);
	for (auto &x : th) x.join();
}

static int team_size(size_t bytes)
{
	int nt = svt_get_max_threads();
	if (nt > 8) nt = 8;
	if (bytes < ((size_t) 4 << 20) || nt < 1) nt = 1;
	return nt;
}

// contiguous host array -> device, through the pinned buffers
static int staged_copy(void *dst, const void *src, size_t n)
{
	if (n < ((size_t) 1 << 20)) {
		if (n) HIP_TRY(hipMemcpy(dst, src, n, hipMemcpyHostToDevice));
		return 0;
	}
	for (size_t off = 0; off < n; off += Stager::CHUNK) {
		const size_t len = n - off < Stager::CHUNK ? n - off : Stager::CHUNK;
		char *b; int slot;
		if (g_stager.acquire(&b, &slot)) return -1;
		const char *s0 = (const char *) src + off;
		team_run(team_size(len), [&](int t, int nt) {
			const size_t a = len * t / nt, e = len * (t + 1) / nt;
			memcpy(b + a, s0 + a, e - a);
		});
		if (g_stager.send(slot, (char *) dst + off, 0, len) || g_stager.commit(slot)) return -1;
	}
	return g_stager.drain();
}

// device -> contiguous host array: D2H into one pinned buffer while the thread team
// copies the previous chunk out of the other (results can be large: the 1e6 x 128
// product of BASELINE config 3 is 1 GB)
static int staged_download(void *dst, const void *src, size_t n)
{
	if (n < ((size_t) 4 << 20)) {
		if (n) HIP_TRY(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
		return 0;
	}
	if (g_stager.init() || g_stager.drain()) return -1;
	HIP_TRY(hipDeviceSynchronize());            // the producer kernels ran on other streams
	const size_t C = Stager::CHUNK;
	const size_t nchunk = (n + C - 1) / C;
	for (size_t c = 0; c <= nchunk; c++) {
		if (c < nchunk) {                   // start chunk c into buffer c & 1
			const size_t off = c * C, len = n - off < C ? n - off : C;
			HIP_TRY(hipMemcpyAsync(g_stager.buf[c & 1], (const char *) src + off, len,
					       hipMemcpyDeviceToHost, g_stager.stream));
			HIP_TRY(hipEventRecord(g_stager.done[c & 1], g_stager.stream));
		}
		if (c > 0) {                        // chunk c - 1 has landed: copy it out
			const size_t off = (c - 1) * C, len = n - off < C ? n - off : C;
			HIP_TRY(hipEventSynchronize(g_stager.done[(c - 1) & 1]));
			const char *b = g_stager.buf[(c - 1) & 1];
			char *d0 = (char *) dst + off;
			team_run(team_size(len), [&](int t, int nt) {
				const size_t a = len * t / nt, e = len * (t + 1) / nt;
				memcpy(d0 + a, b + a, e - a);
			});
		}
	}
	return 0;
}

// ---- small RAII device buffer --------------------------------------------------
struct DevBuf {
	void *p = nullptr;
	size_t bytes = 0;
	DevBuf() {}
	DevBuf(const DevBuf &) = delete;
	~DevBuf() { if (p) (void) hipFree(p); }
	int alloc(size_t n)
	{
		if (n == 0) n = 16;
		HIP_TRY(hipMalloc(&p, n));
		bytes = n;
		return 0;
	}
	int upload(const void *src, size_t n)
	{
		if (alloc(n)) return -1;
		return staged_copy(p, src, n);
	}
	int zero()
	{
		HIP_TRY(hipMemset(p, 0, bytes));
		return 0;
	}
	template <typename T> T *as() { return (T *) p; }
};

// ---- SVT -> CSC marshal ----------------------------------------------------------
static size_t elt_size(int Rtype) { return Rtype == SVT_REALSXP ? 8 : 4; }

static int check_view(const svt_view *x)
{
	if (x == NULL || x->ndim < 1 || x->dim == NULL)
		return svt_set_error("invalid svt_view");
	if (x->Rtype != SVT_REALSXP && x->Rtype != SVT_INTSXP &&
	    x->Rtype != SVT_LGLSXP)
		return svt_set_error("does not support SparseArray objects of "
				     "type code %d", x->Rtype);
	int64_t n = 1;
	for (int a = 1; a < x->ndim; a++) n *= x->dim[a];
	if (n != x->nleaves)
		return svt_set_error("svt_view: nleaves does not match dim");
	return 0;
}

// counts within [0, dim0], no NULL offsets behind a positive count
static int check_leaves(const svt_view *x)
{
	if (x->svt_is_null) return 0;
	const int dim0 = x->dim[0];
	for (int64_t j = 0; j < x->nleaves; j++) {
		const int c = x->nzcount[j];
		if (c < 0 || c > dim0)
			return svt_set_error("invalid SVT leaf (nzcount %d, dim %d)", c, dim0);
		if (c > 0 && x->nzoffs[j] == NULL)
			return svt_set_error("invalid SVT leaf (NULL nzoffs)");
	}
	return 0;
}

extern "C" svt_dev_csc *svt_upload(const svt_view *x)
{
	if (ensure_init() || check_view(x))
		return NULL;
	const int64_t n = x->nleaves;
	const size_t esz = elt_size(x->Rtype);
	const int dim0 = x->dim[0];
	std::vector<int64_t> col_ptr((size_t) n + 1, 0);
	for (int64_t j = 0; j < n; j++) {
		const int c = x->svt_is_null ? 0 : x->nzcount[j];
		if (c < 0 || c > dim0) {
			svt_set_error("invalid SVT leaf (nzcount %d, dim %d)", c, dim0);
			return NULL;
		}
		if (c > 0 && x->nzoffs[j] == NULL) {
			svt_set_error("invalid SVT leaf (NULL nzoffs)");
			return NULL;
		}
		col_ptr[j + 1] = col_ptr[j] + c;
	}
	svt_dev_csc *d = (svt_dev_csc *) calloc(1, sizeof(*d));
	d->Rtype = x->Rtype;
	d->owned = 1;
	d->na_background = x->na_background != 0;
	d->nrow = dim0;
	d->ncol = n;
	d->nnz = col_ptr[(size_t) n];
	const size_t nn = (size_t) (d->nnz > 0 ? d->nnz : 1);
	if (hipMalloc((void **) &d->col_ptr, (size_t) (d->ncol + 1) * 8) != hipSuccess ||
	    hipMalloc((void **) &d->row_idx, nn * 4) != hipSuccess ||
	    hipMalloc(&d->val, nn * esz) != hipSuccess) {
		svt_set_error("hipMalloc failed while uploading an SVT (%lld nnz)",
			      (long long) d->nnz);
		svt_release(d);
		return NULL;
	}
	bool ok = hipMemcpy(d->col_ptr, col_ptr.data(), (size_t) (d->ncol + 1) * 8,
			    hipMemcpyHostToDevice) == hipSuccess;
	// nonzeros k0 .. k0+cnt-1 per trip: as many as fit one pinned buffer
	// ([offsets of the chunk][values of the chunk], 4 + esz bytes per nonzero).  Chunks are
	// cut in nonzeros, not in leaves: one leaf may be longer than a buffer (dim0 only has to
	// exceed CHUNK / 12), and the team splits a chunk evenly whatever the leaf lengths are.
	const int64_t cap = (int64_t) (g_stager_chunk() / (4 + esz));
	const int64_t nnz = d->nnz;
	for (int64_t k0 = 0; ok && k0 < nnz; k0 += cap) {
		const int64_t cnt = nnz - k0 < cap ? nnz - k0 : cap;
		char *b; int slot;
		if (g_stager.acquire(&b, &slot)) { ok = false; break; }
		int32_t *so = (int32_t *) b;
		char *sv = b + (size_t) cnt * 4;
		team_run(team_size((size_t) cnt * (4 + esz)), [&](int t, int nt) {
			const int64_t ka = k0 + cnt * t / nt, kb = k0 + cnt * (t + 1) / nt;
			if (ka >= kb) return;
			// first leaf that reaches past ka
			int64_t j = std::upper_bound(col_ptr.begin(), col_ptr.end(), ka) - col_ptr.begin() - 1;
			for (; j < n && col_ptr[j] < kb; j++) {
				const int64_t a = col_ptr[j] > ka ? col_ptr[j] : ka;
				const int64_t e = col_ptr[j + 1] < kb ? col_ptr[j + 1] : kb;
				if (e <= a) continue;
				const int64_t in_leaf = a - col_ptr[j], s = a - k0, c = e - a;
				memcpy(so + s, x->nzoffs[j] + in_leaf, (size_t) c * 4);
				const void *v = x->nzvals[j];
				if (v != NULL) {
					memcpy(sv + (size_t) s * esz, (const char *) v + (size_t) in_leaf * esz,
					       (size_t) c * esz);
				} else if (esz == 8) {      // lacunar leaf: all ones
					double *o = (double *) sv + s;
					for (int64_t k = 0; k < c; k++) o[k] = 1.0;
				} else {
					int *o = (int *) sv + s;
					for (int64_t k = 0; k < c; k++) o[k] = 1;
				}
			}
		});
		ok = g_stager.send(slot, d->row_idx + k0, 0, (size_t) cnt * 4) == 0 &&
		     g_stager.send(slot, (char *) d->val + (size_t) k0 * esz, (size_t) cnt * 4,
				   (size_t) cnt * esz) == 0 &&
		     g_stager.commit(slot) == 0;
	}
	if (ok) ok = g_stager.drain() == 0;
	if (!ok) {
		if (svt_last_error()[0] == '\0') svt_set_error("H2D copy failed");
		svt_release(d);
		return NULL;
	}
	return d;
}

extern "C" svt_dev_csc *svt_wrap_device_csc(int Rtype, int64_t nrow, int64_t ncol,
					    int64_t nnz, int64_t *col_ptr,
					    int32_t *row_idx, void *val)
{
	svt_dev_csc *d = (svt_dev_csc *) calloc(1, sizeof(*d));
	d->Rtype = Rtype;
	d->owned = 0;
	d->nrow = nrow;
	d->ncol = ncol;
	d->nnz = nnz;
	d->col_ptr = col_ptr;
	d->row_idx = row_idx;
	d->val = val;
	return d;
}

extern "C" void svt_release(svt_dev_csc *h)
{
	if (h == NULL)
		return;
	if (h->owned) {
		if (h->col_ptr) (void) hipFree(h->col_ptr);
		if (h->row_idx) (void) hipFree(h->row_idx);
		if (h->val) (void) hipFree(h->val);
	}
	free(h);
}

// ---- resident operands (opt-in) -------------------------------------------------------
// R code keeps calling the entry points on the same object (colSums(x); colVars(x);
// crossprod(x, y1); crossprod(x, y2) ...), and every call marshals and uploads the whole
// tree again (26 ms of a 54 ms crossprod at BASELINE config 2).  With a byte limit set
// (svt_resident_set_limit), the host-level entry points keep the device copy of an
// operand -- and the layouts derived from it: panel-blocked records, t(x) -- and find it
// again through a fingerprint of the view: dims, type, and per leaf the two host
// pointers, the count and eight evenly spread (offset, value) samples.  R vectors are
// immutable once shared, so equal pointers + counts + samples mean equal contents for
// well-behaved callers; code that overwrites leaves in place must call
// svt_resident_clear().  Off by default.  
struct Resident {
	uint64_t key;
	svt_dev_csc *csc;
	svt_dev_pbc *pbc;        // panel-blocked layout of csc, built on first use
	svt_dev_csc *tr;         // t(csc), built on first use
	svt_dev_pbc *tr_pbc;     // layout of t(csc)
	size_t bytes;
	uint64_t stamp;
	int pins;
};
static std::vector<Resident> g_res;
static std::mutex g_res_mu;
static size_t g_res_limit = 0, g_res_bytes = 0;
static uint64_t g_res_clock = 0, g_res_hits = 0, g_res_misses = 0;

static size_t csc_bytes(const svt_dev_csc *c)
{
	return (size_t) (c->ncol + 1) * 8 + (size_t) c->nnz * (4 + elt_size(c->Rtype));
}

static size_t pbc_bytes(const svt_dev_pbc *P) { return svt_dev_pbc_bytes(P); }

static void resident_free(Resident &r)
{
	if (r.pbc) svt_dev_pbc_release(r.pbc);
	if (r.tr_pbc) svt_dev_pbc_release(r.tr_pbc);
	svt_release(r.tr);
	svt_release(r.csc);
}

// drop least-recently-used unpinned entries until `need` more bytes fit
static void resident_make_room(size_t need)
{
	while (g_res_bytes + need > g_res_limit) {
		int victim = -1;
		for (size_t i = 0; i < g_res.size(); i++)
			if (g_res[i].pins == 0 && (victim < 0 || g_res[i].stamp < g_res[victim].stamp))
				victim = (int) i;
		if (victim < 0) return;
		g_res_bytes -= g_res[victim].bytes;
		resident_free(g_res[victim]);
		g_res.erase(g_res.begin() + victim);
	}
}

extern "C" int svt_resident_set_limit(size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_res_mu);
	g_res_limit = bytes;
	resident_make_room(0);
	return 0;
}

extern "C" void svt_resident_clear(void)
{
	std::lock_guard<std::mutex> lk(g_res_mu);
	const size_t keep = g_res_limit;
	g_res_limit = 0;
	resident_make_room(0);
	g_res_limit = keep;
}

extern "C" void svt_resident_stats(size_t *bytes, int64_t *entries, int64_t *hits, int64_t *misses)
{
	std::lock_guard<std::mutex> lk(g_res_mu);
	if (bytes) *bytes = g_res_bytes;
	if (entries) *entries = (int64_t) g_res.size();
	if (hits) *hits = (int64_t) g_res_hits;
	if (misses) *misses = (int64_t) g_res_misses;
}

static inline uint64_t fp_mix(uint64_t h, uint64_t v)
{
	h ^= v + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
	h *= 0xFF51AFD7ED558CCDULL;
	return h ^ (h >> 33);
}

static uint64_t view_fingerprint(const svt_view *x)
{
	uint64_t h = fp_mix(0x5356545F48495031ULL, (uint64_t) x->Rtype);
	h = fp_mix(h, (uint64_t) x->ndim);
	for (int a = 0; a < x->ndim; a++) h = fp_mix(h, (uint64_t) x->dim[a]);
	h = fp_mix(h, (uint64_t) x->svt_is_null * 2 + (uint64_t) (x->na_background != 0));
	h = fp_mix(h, (uint64_t) x->nleaves);
	if (x->svt_is_null) return h;
	const size_t esz = elt_size(x->Rtype);
	for (int64_t j = 0; j < x->nleaves; j++) {
		const int n = x->nzcount[j];
		h = fp_mix(h, (uint64_t) n);
		if (n <= 0) continue;
		h = fp_mix(h, (uint64_t) (uintptr_t) x->nzoffs[j]);
		h = fp_mix(h, (uint64_t) (uintptr_t) x->nzvals[j]);
		int at[8];
		for (int t = 0; t < 8; t++) at[t] = (int) ((int64_t) (n - 1) * t / 7);
		for (int t = 0; t < 8; t++) {
			uint64_t v = (uint64_t) (uint32_t) x->nzoffs[j][at[t]];
			if (x->nzvals[j] != NULL) {               // (NULL: lacunar leaf, all ones)
				uint64_t bits = 0;
				memcpy(&bits, (const char *) x->nzvals[j] + (size_t) at[t] * esz, esz);
				v ^= bits * 0x9E3779B97F4A7C15ULL;
			}
			h = fp_mix(h, v);
		}
	}
	return h;
}

// An operand of a host-level call: resident if the cache is on (found or inserted, pinned
// for the lifetime of the guard), else uploaded for this call and released with the guard.
struct CscGuard {
	svt_dev_csc *h;
	uint64_t key;            // != 0: h belongs to the resident set
	explicit CscGuard(svt_dev_csc *p) : h(p), key(0) {}
	explicit CscGuard(const svt_view *x) : h(NULL), key(0)
	{
		if (g_res_limit == 0) { h = svt_upload(x); return; }
		if (check_view(x) || check_leaves(x)) return;      // (the fingerprint reads the leaves)
		const uint64_t k = view_fingerprint(x) | 1;
		{
			std::lock_guard<std::mutex> lk(g_res_mu);
			for (Resident &r : g_res)
				if (r.key == k) {
					r.pins++; r.stamp = ++g_res_clock; g_res_hits++;
					h = r.csc; key = k;
					return;
				}
			g_res_misses++;
		}
		h = svt_upload(x);
		if (h == NULL) return;
		std::lock_guard<std::mutex> lk(g_res_mu);
		const size_t nb = csc_bytes(h);
		resident_make_room(nb);
		if (g_res_bytes + nb > g_res_limit) return;        // does not fit: one-call operand
		Resident r = { k, h, NULL, NULL, NULL, nb, ++g_res_clock, 1 };
		g_res.push_back(r);
		g_res_bytes += nb;
		key = k;
	}
	~CscGuard()
	{
		if (key == 0) { svt_release(h); return; }
		std::lock_guard<std::mutex> lk(g_res_mu);
		for (Resident &r : g_res)
			if (r.key == key) { r.pins--; return; }
	}
	void drop()              // a one-call operand that is no longer needed
	{
		if (key == 0) { svt_release(h); h = NULL; }
	}
	CscGuard(const CscGuard &) = delete;
	CscGuard &operator=(const CscGuard &) = delete;
};

// index of the resident entry that owns the device handle A (as its operand or as its
// transposed copy), or -1; call with g_res_mu held
static int resident_find(const svt_dev_csc *A)
{
	for (size_t i = 0; i < g_res.size(); i++)
		if (g_res[i].csc == A || g_res[i].tr == A) return (int) i;
	return -1;
}

// The panel-blocked layout of a resident operand lives with it; for a one-call operand it
// is built and released by the caller.  *owned tells which.
static svt_dev_pbc *pbc_for(const svt_dev_csc *A, int *owned)
{
	*owned = 1;
	{
		std::lock_guard<std::mutex> lk(g_res_mu);
		for (Resident &r : g_res)
			if (r.csc == A || r.tr == A) {
				svt_dev_pbc *&slot = r.csc == A ? r.pbc : r.tr_pbc;
				if (slot != NULL) { *owned = 0; return slot; }
				break;
			}
	}
	svt_dev_pbc *P = svt_dev_pbc_build(A, 0, 0, 0);        // layout by density (pbc_auto_layout)
	if (P == NULL) return NULL;
	std::lock_guard<std::mutex> lk(g_res_mu);
	if (resident_find(A) >= 0) {
		const size_t nb = pbc_bytes(P);
		resident_make_room(nb);                   // (the entry itself is pinned by its guard)
		// make_room() erases entries: look the operand up again, never keep a reference across it
		const int i = resident_find(A);
		if (i >= 0 && g_res_bytes + nb <= g_res_limit) {
			Resident &r = g_res[(size_t) i];
			(r.csc == A ? r.pbc : r.tr_pbc) = P;
			r.bytes += nb; g_res_bytes += nb;
			*owned = 0;
		}
	}
	return P;
}

// ==================================================================================
// Device level
// ==================================================================================
extern "C" size_t svt_dev_crossprod_ws_bytes(int64_t nrow, int64_t ncol, int K)
{
	return crossprod_ws_bytes(nrow, ncol, K);
}

extern "C" int svt_dev_crossprod_csc_dense(const svt_dev_csc *A, const void *Y,
					   int64_t ldY, int K, int tr_y, double *out,
					   int64_t out_stride_c, int64_t out_stride_k,
					   void *ws, size_t ws_bytes, void *stream)
{
	CrossprodArgs a;
	a.col_ptr = A->col_ptr; a.row_idx = A->row_idx; a.val = A->val;
	a.Rtype = A->Rtype == SVT_REALSXP ? SVT_REALSXP : SVT_INTSXP;
	a.nrow = A->nrow; a.ncol = A->ncol;
	a.Y = Y; a.ldY = ldY; a.K = K; a.tr_y = tr_y;
	a.out = out; a.out_stride_c = out_stride_c; a.out_stride_k = out_stride_k;
	a.ws = ws; a.ws_bytes = ws_bytes;
	return launch_crossprod_csc_dense(a, (hipStream_t) stream);
}

extern "C" int svt_dev_dense_prepare(const void *Y, int64_t ldY, int64_t nrow, int K,
				     int tr_y, int Rtype, void *ws, size_t ws_bytes,
				     void *stream)
{
	CrossprodArgs a;
	memset(&a, 0, sizeof(a));
	a.Rtype = Rtype == SVT_REALSXP ? SVT_REALSXP : SVT_INTSXP;
	a.nrow = nrow; a.Y = Y; a.ldY = ldY; a.K = K; a.tr_y = tr_y;
	a.ws = ws; a.ws_bytes = ws_bytes;
	return launch_dense_prepare(a, (hipStream_t) stream);
}

extern "C" int svt_dev_crossprod_prepared(const svt_dev_csc *A, const void *ws, int K,
					  double *out, int64_t out_stride_c,
					  int64_t out_stride_k, void *stream)
{
	CrossprodArgs a;
	memset(&a, 0, sizeof(a));
	a.col_ptr = A->col_ptr; a.row_idx = A->row_idx; a.val = A->val;
	a.Rtype = A->Rtype == SVT_REALSXP ? SVT_REALSXP : SVT_INTSXP;
	a.nrow = A->nrow; a.ncol = A->ncol; a.K = K;
	a.out = out; a.out_stride_c = out_stride_c; a.out_stride_k = out_stride_k;
	a.ws = (void *) ws;
	return launch_crossprod_prepared(a, (hipStream_t) stream);
}

static int check_stat_op(int opcode, int Rtype)
{
	// _get_summarize_opcode(), src/Rvector_summarization.c:19-78
	if (opcode < SVT_OP_ANYNA || opcode > SVT_OP_SD2)
		return svt_set_error("'op' must be one of: \"anyNA\", \"countNAs\", "
				     "\"any\", \"all\", \"min\", \"max\", \"range\", \"sum\", "
				     "\"prod\", \"mean\", \"centered_X2_sum\", \"sum_X_X2\", "
				     "\"var1\", \"var2\", \"sd1\", \"sd2\"");
	if ((opcode == SVT_OP_ANY || opcode == SVT_OP_ALL) && Rtype == SVT_REALSXP)
		return svt_set_error("%s() does not support SparseArray objects of "
				     "type() \"double\"", opcode == SVT_OP_ANY ? "any" : "all");
	return 0;
}

extern "C" int svt_colStats_out_Rtype(int opcode, int in_Rtype)
{
	// _init_SummarizeResult(), src/Rvector_summarization.c:97-165
	switch (opcode) {
	case SVT_OP_ANYNA: case SVT_OP_ANY: case SVT_OP_ALL:
		return SVT_LGLSXP;
	case SVT_OP_MIN: case SVT_OP_MAX: case SVT_OP_RANGE:
		return in_Rtype == SVT_REALSXP ? SVT_REALSXP : SVT_INTSXP;
	default:
		if (opcode < SVT_OP_ANYNA || opcode > SVT_OP_SD2)
			return svt_set_error("unknown opcode %d", opcode);
		return SVT_REALSXP;
	}
}

static int device_op_supported(int opcode)
{
	if (opcode == SVT_OP_RANGE || opcode == SVT_OP_SUM_X_X2 ||
	    opcode == SVT_OP_VAR2 || opcode == SVT_OP_SD2)
		return svt_set_unsupported("op code %d is not reachable from the R API for "
					   "col/row stats and is not implemented on the device",
					   opcode);
	return 0;
}

static int dev_colstats_ex(const svt_dev_csc *A, int opcode, int na_rm, double center,
			   int64_t inner, void *out, int *warn_flag, void *stream, int dgc)
{
	if (check_stat_op(opcode, A->Rtype) || device_op_supported(opcode))
		return -1;
	if (inner <= 0 || A->ncol % inner != 0)
		return svt_set_error("'inner' must divide the number of leaves");
	StatsArgs a;
	a.col_ptr = A->col_ptr; a.val = A->val; a.Rtype = A->Rtype;
	a.nseg = A->ncol / inner; a.inner = inner; a.seg_len = inner * A->nrow;
	a.opcode = opcode; a.na_rm = na_rm; a.center = center;
	a.out = out; a.warn_flag = warn_flag; a.na_bg = A->na_background;
	a.dgc = dgc;
	return launch_colstats(a, A->nnz, (hipStream_t) stream);
}

extern "C" int svt_dev_colstats(const svt_dev_csc *A, int opcode, int na_rm,
				double center, int64_t inner, void *out,
				int *warn_flag, void *stream)
{
	g_unsupported = 0;
	return svt_status(dev_colstats_ex(A, opcode, na_rm, center, inner, out, warn_flag, stream, 0));
}

extern "C" size_t svt_dev_colmedians_ws_bytes(int64_t nnz, int64_t ncol)
{
	return colmedians_ws_bytes(nnz, ncol);
}

static int dev_colmedians_impl(const svt_dev_csc *A, int na_rm, double *out, void *ws,
				  size_t ws_bytes, void *stream)
{
	if (A->na_background)
		return svt_set_error("colMedians() is not supported on NaArray objects");
	if (ws_bytes < colmedians_ws_bytes(A->nnz, A->ncol))
		return svt_set_error("svt_dev_colmedians: workspace too small");
	return launch_colmedians(A->col_ptr, A->val, A->Rtype, A->nrow, A->ncol, A->nnz, na_rm, out, ws,
				 (hipStream_t) stream);
}
extern "C" int svt_dev_colmedians(const svt_dev_csc *A, int na_rm, double *out, void *ws,
				  size_t ws_bytes, void *stream)
{
	g_unsupported = 0;
	return svt_status(dev_colmedians_impl(A, na_rm, out, ws, ws_bytes, stream));
}

extern "C" size_t svt_dev_rowstats_ws_bytes(int64_t nrow, int64_t ncol)
{
	return rowstats_panel_ws_bytes(nrow, ncol);
}

static int dev_rowsums(const svt_dev_csc *A, int na_rm, int64_t inner, double *out, void *ws, size_t ws_bytes,
		       void *stream, int table_mode)
{
	if (inner <= 0 || A->ncol % inner != 0)
		return svt_set_error("'inner' must divide the number of leaves");
	if (ws_bytes < rowstats_panel_ws_bytes(A->nrow, A->ncol))
		return svt_set_error("svt_dev_rowsums: workspace too small");
	RowStatsArgs a;
	memset(&a, 0, sizeof(a));
	a.col_ptr = A->col_ptr; a.row_idx = A->row_idx; a.val = A->val;
	a.Rtype = A->Rtype; a.ncol = A->ncol; a.nrow = A->nrow;
	a.inner = inner; a.nstrata = A->ncol / inner;
	a.out_len = inner * A->nrow;
	a.opcode = SVT_OP_SUM; a.na_rm = na_rm; a.out = out; a.nnz_hint = A->nnz;
	a.na_bg = A->na_background;
	a.table_mode = table_mode;
	return launch_rowstats_panel(a, ws, (hipStream_t) stream);
}

extern "C" int svt_dev_rowsums(const svt_dev_csc *A, int na_rm, int64_t inner,
			       double *out, void *ws, size_t ws_bytes, void *stream)
{
	return dev_rowsums(A, na_rm, inner, out, ws, ws_bytes, stream, 0);
}

// The table of run bounds per row panel depends on the operand alone (one pass over its offsets, a quarter of
// a rowSums at BASELINE config 2): built once into `ws`, then any number of svt_dev_rowsums_prepared() calls on
// the same operand with the same `inner` read it.
extern "C" int svt_dev_rowsums_prepare(const svt_dev_csc *A, int64_t inner, void *ws, size_t ws_bytes, void *stream)
{
	return dev_rowsums(A, 0, inner, NULL, ws, ws_bytes, stream, 1);
}

extern "C" int svt_dev_rowsums_prepared(const svt_dev_csc *A, int na_rm, int64_t inner,
					double *out, void *ws, size_t ws_bytes, void *stream)
{
	return dev_rowsums(A, na_rm, inner, out, ws, ws_bytes, stream, 2);
}

extern "C" size_t svt_dev_transpose_ws_bytes(int64_t nrow, int64_t nnz)
{
	return transpose_ws_bytes(nrow, nnz);
}

static int dev_transpose_impl(const svt_dev_csc *A, int64_t *out_col_ptr, int32_t *out_row_idx,
				 void *out_val, void *ws, size_t ws_bytes, void *stream)
{
	if (ws_bytes < transpose_ws_bytes(A->nrow, A->nnz))
		return svt_set_error("svt_dev_transpose: workspace too small");
	return launch_transpose(A->col_ptr, A->row_idx, A->val, A->Rtype, A->nrow, A->ncol, A->nnz,
				out_col_ptr, out_row_idx, out_val, ws, (hipStream_t) stream);
}
extern "C" int svt_dev_transpose(const svt_dev_csc *A, int64_t *out_col_ptr, int32_t *out_row_idx,
				 void *out_val, void *ws, size_t ws_bytes, void *stream)
{
	g_unsupported = 0;
	return svt_status(dev_transpose_impl(A, out_col_ptr, out_row_idx, out_val, ws, ws_bytes, stream));
}

// A %*% B, both sparse (kernels_spmm.hip): out[r + k * ldo], r < A->nrow, k < B->ncol.
extern "C" size_t svt_dev_matmul_csc_csc_ws_bytes(const svt_dev_csc *A)
{
	return spmm_ws_bytes(A->nrow, A->ncol) + 256;       // [flag][table of run bounds]
}

static SpmmArgs spmm_args(const svt_dev_csc *A, const svt_dev_csc *B, double *out, int64_t ldo, int *flag)
{
	SpmmArgs a;
	memset(&a, 0, sizeof(a));
	a.a_ptr = A->col_ptr; a.a_idx = A->row_idx; a.a_val = A->val; a.a_type = A->Rtype;
	a.nrow = A->nrow; a.ninner = A->ncol;
	a.b_ptr = B->col_ptr; a.b_idx = B->row_idx; a.b_val = B->val; a.b_type = B->Rtype; a.K = B->ncol;
	a.out = out; a.ldo = ldo; a.flag = flag;
	return a;
}

// ws: [int: A holds a non-finite value / an NA][int: the same for the last product, B included][...][table at 256]
extern "C" int svt_dev_matmul_csc_csc_prepare(const svt_dev_csc *A, void *ws, size_t ws_bytes, void *stream)
{
	if (ws_bytes < svt_dev_matmul_csc_csc_ws_bytes(A))
		return svt_set_error("svt_dev_matmul_csc_csc: workspace too small");
	hipStream_t s = (hipStream_t) stream;
	HIP_TRY(hipMemsetAsync(ws, 0, 8, s));
	svt_dev_csc none;
	memset(&none, 0, sizeof(none));
	return launch_spmm_prepare(spmm_args(A, &none, NULL, 0, (int *) ws), A->nnz, (char *) ws + 256, s);
}

static int dev_matmul_csc_csc_prepared_impl(const svt_dev_csc *A, const svt_dev_csc *B, double *out, int64_t ldo,
					       void *ws, size_t ws_bytes, int *not_finite, void *stream)
{
	if (A->ncol != B->nrow)
		return svt_set_error("svt_dev_matmul_csc_csc: non-conformable operands");
	if (ws_bytes < svt_dev_matmul_csc_csc_ws_bytes(A))
		return svt_set_error("svt_dev_matmul_csc_csc: workspace too small");
	if (ldo < A->nrow)
		return svt_set_error("svt_dev_matmul_csc_csc: leading dimension of the result too small");
	hipStream_t s = (hipStream_t) stream;
	int *flag = (int *) ws + 1;
	HIP_TRY(hipMemcpyAsync(flag, ws, 4, hipMemcpyDeviceToDevice, s));
	if (launch_spmm_product(spmm_args(A, B, out, ldo, flag), A->nnz, B->nnz, (char *) ws + 256, s))
		return -1;
	if (not_finite != NULL)
		HIP_TRY(hipMemcpyAsync(not_finite, flag, 4, hipMemcpyDeviceToDevice, s));
	return 0;
}
extern "C" int svt_dev_matmul_csc_csc_prepared(const svt_dev_csc *A, const svt_dev_csc *B, double *out, int64_t ldo,
					       void *ws, size_t ws_bytes, int *not_finite, void *stream)
{
	g_unsupported = 0;
	return svt_status(dev_matmul_csc_csc_prepared_impl(A, B, out, ldo, ws, ws_bytes, not_finite, stream));
}

static int dev_matmul_csc_csc_impl(const svt_dev_csc *A, const svt_dev_csc *B, double *out, int64_t ldo,
				      void *ws, size_t ws_bytes, int *not_finite, void *stream)
{
	if (A->ncol != B->nrow)
		return svt_set_error("svt_dev_matmul_csc_csc: non-conformable operands");
	if (ws_bytes < svt_dev_matmul_csc_csc_ws_bytes(A))
		return svt_set_error("svt_dev_matmul_csc_csc: workspace too small");
	if (ldo < A->nrow)
		return svt_set_error("svt_dev_matmul_csc_csc: leading dimension of the result too small");
	// one product: the table pass looks only at the leaves of A that B does not refer to, the product kernel at
	// the values it reads (kernels_spmm.hip); ws[0] is left as "not known for A alone" -- this ws is not a
	// prepared one afterwards
	hipStream_t s = (hipStream_t) stream;
	int *flag = (int *) ws + 1;
	const SpmmArgs a = spmm_args(A, B, out, ldo, flag);
	// (the two flag words are zeroed by the first kernel of the pass: a memset in front of it is a blit kernel of its own)
	if (launch_spmm_prepare_for(a, A->nnz, B->nnz, (char *) ws + 256, s, (int *) ws))
		return -1;
	if (launch_spmm_product(a, A->nnz, B->nnz, (char *) ws + 256, s))
		return -1;
	if (not_finite != NULL)
		HIP_TRY(hipMemcpyAsync(not_finite, flag, 4, hipMemcpyDeviceToDevice, s));
	return 0;
}
extern "C" int svt_dev_matmul_csc_csc(const svt_dev_csc *A, const svt_dev_csc *B, double *out, int64_t ldo,
				      void *ws, size_t ws_bytes, int *not_finite, void *stream)
{
	g_unsupported = 0;
	return svt_status(dev_matmul_csc_csc_impl(A, B, out, ldo, ws, ws_bytes, not_finite, stream));
}

// crossprod(X, Y) of two sparse operands on t(X) and Y (kernels_gram.hip)
extern "C" size_t svt_dev_crossprod_csc_csc_ws_bytes(const svt_dev_csc *Xt)
{
	return gram_ws_bytes(Xt->nrow, Xt->ncol, Xt->nnz);
}

extern "C" void svt_dev_crossprod_csc_csc_set_panel(int one_block_max, int log2_panel)
{
	gram_set_panel(one_block_max, log2_panel);
}

static int dev_crossprod_csc_csc_impl(const svt_dev_csc *Xt, const svt_dev_csc *Y, int sym, double *out,
					 int64_t ldo, void *ws, size_t ws_bytes, int *not_finite, void *stream)
{
	if (Xt->ncol != Y->nrow)
		return svt_set_error("svt_dev_crossprod_csc_csc: non-conformable operands");
	if (sym && Xt->nrow != Y->ncol)
		return svt_set_error("svt_dev_crossprod_csc_csc: the symmetric form needs t(Y) and Y");
	if (ws_bytes < svt_dev_crossprod_csc_csc_ws_bytes(Xt))
		return svt_set_error("svt_dev_crossprod_csc_csc: workspace too small");
	if (ldo < Xt->nrow)
		return svt_set_error("svt_dev_crossprod_csc_csc: leading dimension of the result too small");
	if ((Xt->Rtype != SVT_REALSXP && Xt->Rtype != SVT_INTSXP) || (Y->Rtype != SVT_REALSXP && Y->Rtype != SVT_INTSXP))
		return svt_set_error("svt_dev_crossprod_csc_csc: double or integer operands");
	hipStream_t s = (hipStream_t) stream;
	GramArgs a;
	memset(&a, 0, sizeof(a));
	a.a_ptr = Xt->col_ptr; a.a_idx = Xt->row_idx; a.a_val = Xt->val; a.a_type = Xt->Rtype;
	a.nx = Xt->nrow; a.nrow = Xt->ncol;
	a.b_ptr = Y->col_ptr; a.b_idx = Y->row_idx; a.b_val = Y->val; a.b_type = Y->Rtype; a.ny = Y->ncol;
	a.out = out; a.ldo = ldo; a.sym = sym != 0;
	if (launch_gram(a, Xt->nnz, Y->nnz, ws, s))
		return -1;
	if (not_finite != NULL)
		HIP_TRY(hipMemcpyAsync(not_finite, ws, 4, hipMemcpyDeviceToDevice, s));
	return 0;
}
extern "C" int svt_dev_crossprod_csc_csc(const svt_dev_csc *Xt, const svt_dev_csc *Y, int sym, double *out,
					 int64_t ldo, void *ws, size_t ws_bytes, int *not_finite, void *stream)
{
	g_unsupported = 0;
	return svt_status(dev_crossprod_csc_csc_impl(Xt, Y, sym, out, ldo, ws, ws_bytes, not_finite, stream));
}

extern "C" void svt_dev_aperm_route_counts(int64_t *counts, int reset)
{
	aperm_route_counts(counts, reset);
}

static int aperm_args(int ndim, const int *perm, int *perm0)
{
	if (ndim < 1 || ndim > 8)
		return svt_set_error("aperm: between 1 and 8 dimensions are supported");
	for (int a = 0; a < ndim; a++) {
		if (perm[a] < 1 || perm[a] > ndim)
			return svt_set_error("'perm' must be a permutation of 1:%d", ndim);
		perm0[a] = perm[a] - 1;
	}
	return 0;
}

extern "C" size_t svt_dev_aperm_ws_bytes(int64_t nnz, int ndim, const int64_t *dim)
{
	return aperm_ws_bytes(nnz, dim, ndim);
}

static int dev_aperm_impl(const svt_dev_csc *A, int ndim, const int64_t *dim, const int *perm,
			     int64_t *out_col_ptr, int32_t *out_row_idx, void *out_val,
			     void *ws, size_t ws_bytes, void *stream)
{
	int perm0[8];
	if (aperm_args(ndim, perm, perm0))
		return -1;
	int64_t nl = 1;
	for (int a = 1; a < ndim; a++) nl *= dim[a];
	if (dim[0] != A->nrow || nl != A->ncol)
		return svt_set_error("aperm: 'dim' does not match the operand");
	if (ws_bytes < aperm_ws_bytes(A->nnz, dim, ndim))
		return svt_set_error("svt_dev_aperm: workspace too small");
	return launch_aperm(A->col_ptr, A->row_idx, A->val, A->Rtype, A->ncol, A->nnz, dim, ndim,
			    perm0, out_col_ptr, out_row_idx, out_val, ws, (hipStream_t) stream);
}
extern "C" int svt_dev_aperm(const svt_dev_csc *A, int ndim, const int64_t *dim, const int *perm,
			     int64_t *out_col_ptr, int32_t *out_row_idx, void *out_val,
			     void *ws, size_t ws_bytes, void *stream)
{
	g_unsupported = 0;
	return svt_status(dev_aperm_impl(A, ndim, dim, perm, out_col_ptr, out_row_idx, out_val, ws, ws_bytes, stream));
}

// C_aperm_SVT, src/SparseArray_aperm.c:935-970
static int aperm_SVT_impl(const svt_view *x, const int *perm, int64_t *out_col_ptr,
			     int32_t *out_row_idx, void *out_val)
{
	if (ensure_init() || check_view(x))
		return -1;
	int perm0[8];
	if (aperm_args(x->ndim, perm, perm0))
		return -1;
	int64_t dim[8], new_nl = 1;
	for (int a = 0; a < x->ndim; a++) dim[a] = x->dim[a];
	for (int a = 1; a < x->ndim; a++) new_nl *= dim[perm0[a]];
	CscGuard A(x);
	if (A.h == NULL) return -1;
	const size_t esz = elt_size(x->Rtype);
	const size_t nn = (size_t) (A.h->nnz > 0 ? A.h->nnz : 1);
	DevBuf P, I, V, W;
	if (P.alloc((size_t) (new_nl + 1) * 8) || I.alloc(nn * 4) || V.alloc(nn * esz) ||
	    W.alloc(aperm_ws_bytes(A.h->nnz, dim, x->ndim)))
		return -1;
	if (launch_aperm(A.h->col_ptr, A.h->row_idx, A.h->val, A.h->Rtype, A.h->ncol, A.h->nnz, dim,
			 x->ndim, perm0, P.as<int64_t>(), I.as<int32_t>(), V.p, W.p, 0))
		return -1;
	HIP_TRY(hipMemcpy(out_col_ptr, P.p, (size_t) (new_nl + 1) * 8, hipMemcpyDeviceToHost));
	if (A.h->nnz) {
		if (staged_download(out_row_idx, I.p, (size_t) A.h->nnz * 4) ||
		    staged_download(out_val, V.p, (size_t) A.h->nnz * esz))
			return -1;
	}
	return 0;
}
extern "C" int svt_aperm_SVT(const svt_view *x, const int *perm, int64_t *out_col_ptr,
			     int32_t *out_row_idx, void *out_val)
{
	g_unsupported = 0;
	return svt_status(aperm_SVT_impl(x, perm, out_col_ptr, out_row_idx, out_val));
}

extern "C" int svt_dev_rowsum(const svt_dev_csc *A, const int *group, int ngroup,
			      int na_rm, double *out, void *stream)
{
	if (A->Rtype != SVT_REALSXP)
		return svt_set_error("svt_dev_rowsum: f64 input only");
	GroupSumArgs a;
	memset(&a, 0, sizeof(a));
	a.col_ptr64 = A->col_ptr; a.row_idx = A->row_idx; a.val = A->val;
	a.Rtype = A->Rtype; a.nrow = A->nrow; a.ncol = A->ncol;
	a.group = group; a.ngroup = ngroup; a.na_rm = na_rm; a.out = out;
	if (ngroup <= 8192 && A->ncol > 0 && A->nnz / A->ncol >= ngroup / 4)
		return launch_rowsum_lds(a, (hipStream_t) stream);
	return launch_rowsum(a, (hipStream_t) stream);
}

// rowsum(x, group) for a (x, group) pair that is used more than once: the group of every nonzero, as a 16-bit
// 0-based id (NA -> ngroup - 1), is written to `gid` (svt_dev_rowsum_gid_bytes(A) bytes) once; the prepared call
// then streams values and ids -- 10 bytes per nonzero, no lookup in the group table.
extern "C" size_t svt_dev_rowsum_gid_bytes(const svt_dev_csc *A)
{
	return (size_t) (A->nnz > 0 ? A->nnz : 1) * 2 + 16;
}

extern "C" int svt_dev_rowsum_prepare(const svt_dev_csc *A, const int *group, int ngroup, void *gid,
				      size_t gid_bytes, void *stream)
{
	if (ngroup < 1 || ngroup > 65535)
		return svt_set_error("svt_dev_rowsum_prepare: between 1 and 65535 groups");
	if (gid_bytes < svt_dev_rowsum_gid_bytes(A))
		return svt_set_error("svt_dev_rowsum_prepare: id buffer too small");
	GroupSumArgs a;
	memset(&a, 0, sizeof(a));
	a.col_ptr64 = A->col_ptr; a.row_idx = A->row_idx; a.nrow = A->nrow; a.ncol = A->ncol;
	a.group = group; a.ngroup = ngroup;
	return launch_rowsum_gid(a, A->nnz, (uint16_t *) gid, (hipStream_t) stream);
}

extern "C" int svt_dev_rowsum_prepared(const svt_dev_csc *A, const void *gid, int ngroup, int na_rm,
				       double *out, void *stream)
{
	if (A->Rtype != SVT_REALSXP)
		return svt_set_error("svt_dev_rowsum: f64 input only");
	if (ngroup < 1 || ngroup > 65535)
		return svt_set_error("svt_dev_rowsum_prepared: between 1 and 65535 groups");
	if ((int64_t) ngroup * 8 > 160 * 1024)
		return svt_set_error("svt_dev_rowsum_prepared: more groups than a workgroup's LDS holds (20480)");
	GroupSumArgs a;
	memset(&a, 0, sizeof(a));
	a.col_ptr64 = A->col_ptr; a.val = A->val; a.Rtype = A->Rtype; a.nrow = A->nrow; a.ncol = A->ncol;
	a.ngroup = ngroup; a.na_rm = na_rm; a.out = out;
	const int rc = launch_rowsum_prepared(a, (const uint16_t *) gid, (hipStream_t) stream);
	if (rc > 0)
		return svt_set_error("svt_dev_rowsum_prepared: unsupported shape");
	return rc;
}

// ==================================================================================
// Host level: crossprod
// ==================================================================================
static int check_mult_view(const svt_view *x, const char *what)
{
	if (check_view(x))
		return -1;
	if (x->ndim != 2)
		return svt_set_error("%s must have 2 dimensions", what);
	// get_and_check_input_Rtype(), src/SparseMatrix_mult.c:915-928
	if (x->Rtype != SVT_REALSXP && x->Rtype != SVT_INTSXP)
		return svt_set_error("input type is not supported yet");
	if (x->na_background)      // crossprod()/%*% have no NaArray methods (R/SparseMatrix-mult.R)
		return svt_set_error("NaArray objects are not supported by this operation");
	return 0;
}

// Largest number of dense columns handled per launch so that the row-major
// staging copy of the dense operand stays below ~1 GiB.
static int chunk_K(int64_t nrow, int64_t K)
{
	const int64_t budget = (int64_t) 1 << 30;
	int64_t kc = budget / (8 * (nrow > 0 ? nrow : 1));
	kc = kc / 64 * 64;
	if (kc < 64) kc = 64;
	if (kc > K) kc = K;
	return (int) kc;
}

// The panel-blocked layout stores every (column group, 128-row panel) tile as whole batches of
// 8 records (96 bytes, at least one per tile) plus 8 bytes of tile table: a hypersparse operand
// (1e6 x 1e6 with 3e7 nonzeros: 20 GB of records for 0.36 GB of CSC) would be streamed at a
// fraction of the general kernels' speed, and the tile count must fit the 32-bit scan.
static bool pbc_shape_ok(int64_t nrow, int64_t ncol, int64_t nnz)
{
	int cbw, wpb, logr;
	pbc_auto_layout(nrow, ncol, nnz, &cbw, &wpb, &logr);      // (very sparse operands: panels of 1024 rows)
	const int64_t cb = (int64_t) cbw * wpb;
	const double ngroups = (double) ((ncol + cb - 1) / cb) * (double) wpb;
	const double npanels = (double) ((nrow + ((int64_t) 1 << logr) - 1) >> logr);
	const double ntiles = ngroups * npanels;
	return ntiles + 1.0 < 2147483647.0 && (double) nnz >= 4.0 * ntiles;
}

static bool pbc_applies(const svt_dev_csc *A, int64_t K, int tr_y)
{
	(void) tr_y;       // a dense operand given by rows is transposed on the device (kernels_mult_pbc.hip)
	return A->Rtype == SVT_REALSXP && A->nrow >= 256 && A->ncol > 0 &&
	       (double) A->nnz * (double) K >= 268435456.0 && pbc_shape_ok(A->nrow, A->ncol, A->nnz);
}

// The layout build is device work (a few ms at 1e8 nonzeros) and the upload of the dense
// operand is PCIe + host threads: start the build on a helper thread, upload meanwhile.
struct PbcAhead {
	std::thread th;
	svt_dev_pbc *P = NULL;
	int own = 1;
	bool started = false, taken = false;
	void start(const svt_dev_csc *A, int64_t K, int tr_y)
	{
		if (!pbc_applies(A, K, tr_y)) return;
		started = true;
		th = std::thread([this, A] {
			(void) hipSetDevice(g_device);
			P = pbc_for(A, &own);
		});
	}
	svt_dev_pbc *get(int *own_out)
	{
		if (th.joinable()) th.join();
		taken = true;
		*own_out = own;
		return P;
	}
	~PbcAhead()
	{
		if (th.joinable()) th.join();
		if (P && own && !taken) svt_dev_pbc_release(P);
	}
};

// out (device) receives all K columns; the dense operand is already on the device.
static int dev_crossprod_chunked(const svt_dev_csc *A, const void *Y_dev, int64_t ldY,
				 int64_t K, int tr_y, double *out_dev,
				 int64_t sc, int64_t sk, PbcAhead *ahead = NULL)
{
	if (K <= 0 || A->ncol <= 0)
		return 0;
	// Large integer products: the same panel-blocked kernels on f64 copies of the values and
	// of the dense operand (count matrices are integer; see int_to_f64_kernel for why the
	// results, NA rules included, are those of the integer path -- bit for bit while the sums
	// stay below 2^53).
	if (A->Rtype == SVT_INTSXP && !tr_y && ldY == A->nrow && A->nrow >= 256 &&
	    (double) A->nnz * (double) K >= 268435456.0 && pbc_shape_ok(A->nrow, A->ncol, A->nnz)) {
		DevBuf V, Yf;
		if (V.alloc((size_t) (A->nnz > 0 ? A->nnz : 1) * 8) || Yf.alloc((size_t) A->nrow * K * 8) ||
		    launch_int_to_f64((const int *) A->val, A->nnz, V.as<double>(), 0) ||
		    launch_int_to_f64((const int *) Y_dev, A->nrow * K, Yf.as<double>(), 0))
			return -1;
		svt_dev_csc Af = *A;
		Af.Rtype = SVT_REALSXP; Af.val = V.p; Af.owned = 0;
		return dev_crossprod_chunked(&Af, Yf.p, ldY, K, 0, out_dev, sc, sk, NULL);
	}
	// Large double products with a column-major dense operand take the panel-blocked
	// kernels: the one-off layout build (a few ms at 1e8 nonzeros)
	// pays for itself within the call.  Below the threshold the general kernels run,
	// whose sums are bit-identical to the reference's sequential ones; above it the
	// row-split partial sums differ from those in the last bits (parity bar: 1e-6).
	if (pbc_applies(A, K, tr_y)) {
		int own_P = 1;
		svt_dev_pbc *P = (ahead && ahead->started) ? ahead->get(&own_P) : pbc_for(A, &own_P);
		if (P == NULL)                 // e.g. more records than 32-bit stream offsets reach:
			goto general;          // the general kernels take any size
		const int kc = K < 512 ? (int) K : 512;
		DevBuf ws;
		int rc = ws.alloc(svt_dev_crossprod_pbc_ws_bytes(P, kc));
		for (int64_t k0 = 0; rc == 0 && k0 < K; k0 += kc) {
			const int kn = (int) (K - k0 < kc ? K - k0 : kc);
			rc = svt_dev_crossprod_pbc(P, A, (const double *) Y_dev + (tr_y ? k0 : k0 * ldY), ldY, kn,
						   tr_y, out_dev + k0 * sk, sc, sk, ws.p, ws.bytes, 0);
		}
		if (rc == 0 && hipDeviceSynchronize() != hipSuccess)
			rc = svt_set_error("device error in the panel-blocked crossprod");
		if (own_P) svt_dev_pbc_release(P);
		return rc;
	}
general:
	const int kc = chunk_K(A->nrow, K);
	DevBuf ws;
	if (ws.alloc(crossprod_ws_bytes(A->nrow, A->ncol, kc)))
		return -1;
	const size_t esz = elt_size(A->Rtype);
	for (int64_t k0 = 0; k0 < K; k0 += kc) {
		const int kn = (int) (K - k0 < kc ? K - k0 : kc);
		const char *Yc = (const char *) Y_dev +
			(tr_y ? (size_t) k0 : (size_t) k0 * (size_t) ldY) * esz;
		if (svt_dev_crossprod_csc_dense(A, Yc, ldY, kn, tr_y,
						out_dev + k0 * sk, sc, sk,
						ws.p, ws.bytes, 0))
			return -1;
	}
	HIP_TRY(hipDeviceSynchronize());
	return 0;
}

// Mixed integer / double operands.  The reference's entry points refuse them ("not supported
// yet") and its R methods coerce the integer operand on the host first (type(x) <- "double",
// R/SparseMatrix-mult.R:75-120) -- a copy of the whole tree.  Here the integer side is uploaded
// as it is (4 bytes per value over PCIe) and widened on the device; as.double(NA_integer_) is
// NA_real_, which is what int_to_f64_kernel writes.
struct Promoted {
	svt_dev_csc view;
	DevBuf val, dense;
	const svt_dev_csc *A;
	const void *Y;
	int promote(const svt_dev_csc *A_in, const void *Y_dev, int y_Rtype, size_t y_elems)
	{
		A = A_in; Y = Y_dev;
		if (A_in->Rtype == y_Rtype)
			return 0;
		if (A_in->Rtype == SVT_INTSXP) {                 // sparse int, dense double
			if (val.alloc((size_t) (A_in->nnz > 0 ? A_in->nnz : 1) * 8) ||
			    launch_int_to_f64((const int *) A_in->val, A_in->nnz, val.as<double>(), 0))
				return -1;
			view = *A_in;
			view.Rtype = SVT_REALSXP; view.val = val.p; view.owned = 0;
			A = &view;
		} else {                                          // sparse double, dense int
			if (dense.alloc((y_elems > 0 ? y_elems : 1) * 8) ||
			    launch_int_to_f64((const int *) Y_dev, (int64_t) y_elems, dense.as<double>(), 0))
				return -1;
			Y = dense.p;
		}
		return 0;
	}
};

static bool mult_types_ok(int a, int b)
{
	return (a == SVT_REALSXP || a == SVT_INTSXP) && (b == SVT_REALSXP || b == SVT_INTSXP);
}

// C_crossprod2_SVT_mat, src/SparseMatrix_mult.c:931-982
static int crossprod2_SVT_mat_impl(const svt_view *x, const void *y, int y_nrow,
				      int y_ncol, int y_Rtype, int tr_y, double *out)
{
	if (ensure_init() || check_mult_view(x, "input objects"))
		return -1;
	const int in_nrow = x->dim[0], out_nrow = x->dim[1];
	if (in_nrow != (tr_y ? y_ncol : y_nrow))
		return svt_set_error("input objects are non-conformable");
	if (y_Rtype == SVT_LGLSXP) y_Rtype = SVT_INTSXP;
	if (!mult_types_ok(x->Rtype, y_Rtype))
		return svt_set_error("SparseArray internal error in "
				     "C_crossprod2_SVT_mat():\n"
				     "    'x_Rtype != TYPEOF(y)' not supported yet");
	const int out_ncol = tr_y ? y_nrow : y_ncol;
	const size_t out_n = (size_t) out_nrow * out_ncol;
	memset(out, 0, out_n * sizeof(double));
	if (x->svt_is_null || out_n == 0)     // :389-390
		return 0;
	CscGuard A(x);
	if (A.h == NULL) return -1;
	DevBuf Y, O;
	PbcAhead ahead;
	ahead.start(A.h, out_ncol, tr_y);
	if (Y.upload(y, (size_t) y_nrow * y_ncol * elt_size(y_Rtype)) ||
	    O.alloc(out_n * 8) || O.zero())
		return -1;
	Promoted pr;
	if (pr.promote(A.h, Y.p, y_Rtype, (size_t) y_nrow * y_ncol))
		return -1;
	if (dev_crossprod_chunked(pr.A, pr.Y, y_nrow, out_ncol, tr_y, O.as<double>(),
				  1, out_nrow, pr.A == A.h ? &ahead : NULL))
		return -1;
	if (staged_download(out, O.p, out_n * 8)) return -1;
	return 0;
}
extern "C" int svt_crossprod2_SVT_mat(const svt_view *x, const void *y, int y_nrow,
				      int y_ncol, int y_Rtype, int tr_y, double *out)
{
	g_unsupported = 0;
	return svt_status(crossprod2_SVT_mat_impl(x, y, y_nrow, y_ncol, y_Rtype, tr_y, out));
}

// C_crossprod2_mat_SVT, src/SparseMatrix_mult.c:985-1034
static int crossprod2_mat_SVT_impl(const void *x, int x_nrow, int x_ncol,
				      int x_Rtype, const svt_view *y, int tr_x,
				      double *out)
{
	if (ensure_init() || check_mult_view(y, "input objects"))
		return -1;
	const int in_nrow = y->dim[0], out_ncol = y->dim[1];
	if ((tr_x ? x_ncol : x_nrow) != in_nrow)
		return svt_set_error("input objects are non-conformable");
	if (x_Rtype == SVT_LGLSXP) x_Rtype = SVT_INTSXP;
	if (!mult_types_ok(x_Rtype, y->Rtype))
		return svt_set_error("input objects must have the same type() for now");
	const int out_nrow = tr_x ? x_nrow : x_ncol;
	const size_t out_n = (size_t) out_nrow * out_ncol;
	memset(out, 0, out_n * sizeof(double));
	if (y->svt_is_null || out_n == 0)     // :439-440
		return 0;
	CscGuard A(y);
	if (A.h == NULL) return -1;
	DevBuf X, O;
	PbcAhead ahead;
	ahead.start(A.h, out_nrow, tr_x);
	if (X.upload(x, (size_t) x_nrow * x_ncol * elt_size(x_Rtype)) ||
	    O.alloc(out_n * 8) || O.zero())
		return -1;
	// result cell (i = dense vector, j = leaf) lives at out[i + j*out_nrow]
	Promoted pr;
	if (pr.promote(A.h, X.p, x_Rtype, (size_t) x_nrow * x_ncol))
		return -1;
	if (dev_crossprod_chunked(pr.A, pr.Y, x_nrow, out_nrow, tr_x, O.as<double>(),
				  out_nrow, 1, pr.A == A.h ? &ahead : NULL))
		return -1;
	if (staged_download(out, O.p, out_n * 8)) return -1;
	return 0;
}
extern "C" int svt_crossprod2_mat_SVT(const void *x, int x_nrow, int x_ncol,
				      int x_Rtype, const svt_view *y, int tr_x,
				      double *out)
{
	g_unsupported = 0;
	return svt_status(crossprod2_mat_SVT_impl(x, x_nrow, x_ncol, x_Rtype, y, tr_x, out));
}

// Densify columns of `pp` chunk by chunk and multiply every chunk with the
// leaves of `other`: crossprod2_Lpp_* / crossprod2_Rpp_*,
// src/SparseMatrix_mult.c:728-820.
static int dev_crossprod_pp(const svt_dev_csc *other, const svt_dev_csc *pp,
			    double *out_dev, int64_t sc, int64_t sk)
{
	const int64_t K = pp->ncol, nrow = pp->nrow;
	if (K <= 0 || other->ncol <= 0)
		return 0;
	const bool big = nrow >= 256 && (double) other->nnz * (double) K >= 268435456.0 &&
			 pbc_shape_ok(other->nrow, other->ncol, other->nnz);
	if (other->Rtype == SVT_INTSXP && big) {                         // as in dev_crossprod_chunked
		DevBuf V1, V2;
		if (V1.alloc((size_t) (other->nnz > 0 ? other->nnz : 1) * 8) ||
		    V2.alloc((size_t) (pp->nnz > 0 ? pp->nnz : 1) * 8) ||
		    launch_int_to_f64((const int *) other->val, other->nnz, V1.as<double>(), 0) ||
		    launch_int_to_f64((const int *) pp->val, pp->nnz, V2.as<double>(), 0))
			return -1;
		svt_dev_csc of = *other, pf = *pp;
		of.Rtype = pf.Rtype = SVT_REALSXP; of.owned = pf.owned = 0;
		of.val = V1.p; pf.val = V2.p;
		return dev_crossprod_pp(&of, other == pp ? &of : &pf, out_dev, sc, sk);
	}
	int kc = chunk_K(nrow, K);
	const size_t esz = elt_size(pp->Rtype);
	// large double products: panel-blocked layout of `other`, built once, against
	// every densified chunk (same threshold and caveat as dev_crossprod_chunked)
	svt_dev_pbc *P = NULL;
	int own_P = 1;
	if (other->Rtype == SVT_REALSXP && big) {
		P = pbc_for(other, &own_P);
		if (P != NULL && kc > 512) kc = 512;
	}
	DevBuf dense, ws;
	int rc = 0;
	if (dense.alloc((size_t) (nrow > 0 ? nrow : 1) * kc * esz) ||
	    ws.alloc(P ? svt_dev_crossprod_pbc_ws_bytes(P, kc)
		       : crossprod_ws_bytes(nrow, other->ncol, kc)))
		rc = -1;
	// crossprod(x) (other == pp): of the dense chunk [k0, k0 + kn) only the leaves c >= k0 are
	// needed -- the cells with c >= k, which the caller mirrors -- as in compute_sym_dotprods_*
	// (src/SparseMatrix_mult.c:263-296: ncol^2 / 2 dot products).
	const bool sym = other == pp;
	for (int64_t k0 = 0; rc == 0 && k0 < K; k0 += kc) {
		const int kn = (int) (K - k0 < kc ? K - k0 : kc);
		if (launch_densify(pp->col_ptr, pp->row_idx, pp->val, pp->Rtype, nrow,
				   k0, kn, dense.p, 0)) {
			rc = -1;
		} else if (P) {
			rc = svt_dev_crossprod_pbc_from(P, other, (const double *) dense.p, nrow, kn, 0,
							out_dev + k0 * sk, sc, sk, ws.p, ws.bytes, 0,
							sym ? k0 : 0);
		} else if (sym && k0 > 0) {
			svt_dev_csc tail = *other;                 // leaves k0 .. ncol-1 (col_ptr entries stay absolute)
			tail.col_ptr += k0; tail.ncol -= k0; tail.owned = 0;
			rc = svt_dev_crossprod_csc_dense(&tail, dense.p, nrow, kn, 0,
							 out_dev + k0 * sk + k0 * sc, sc, sk,
							 ws.p, ws.bytes, 0);
		} else {
			rc = svt_dev_crossprod_csc_dense(other, dense.p, nrow, kn, 0,
							 out_dev + k0 * sk, sc, sk,
							 ws.p, ws.bytes, 0);
		}
	}
	if (rc == 0 && hipDeviceSynchronize() != hipSuccess)
		rc = svt_set_error("device error in the sparse x sparse crossprod");
	if (P && own_P) svt_dev_pbc_release(P);
	return rc;
}

struct CscGuard;
static svt_dev_csc *transposed_for(const CscGuard &A, int *owned);
struct OwnedCsc {            // releases a handle only if this call built it
	svt_dev_csc *t;
	int own;
	~OwnedCsc() { if (own) svt_release(t); }
};

// The sparse-aware route (kernels_gram.hip) multiplies only the pairs of nonzeros that meet in a row -- about
// nnz(x) * nnz(y) / nrow of them (half that for the unary form), each an LDS atomic behind a gathered 12-byte read --
// where the dense-buffer route below does `dense_ops` multiply-adds (the reference's Lpp_nops / Rpp_nops,
// src/SparseMatrix_mult.c:1077-1078).  Times measured on one MI355X (tools/debug/sparse_crossprod_time.py, round 6):
// gathered pairs at 2.6e11 / s behind t(x) (2.5e-11 s per nonzero) and ~0.2 ms of launches; the dense-buffer route at
// 3e12 multiply-adds / s behind ~1 ms of allocations, layout build and synchronisation.
static double g_gram_cost = 1.0;
extern "C" void svt_sparse_crossprod_set_cost(double factor)
{
	g_gram_cost = factor;      // the sparse-aware route's estimated time is multiplied by it; < 0: never that route; 0: always
}

static double dense_route_seconds(double dense_ops) { return 1.0e-3 + dense_ops / 3.0e12; }
static double sparse_route_seconds(double pairs, int64_t nnz_x) { return 0.2e-3 + pairs / 2.6e11 + (double) nnz_x * 2.5e-11; }

static bool sparse_route_pays(int64_t nnz_x, int64_t nnz_y, int64_t nrow, double dense_ops, bool sym)
{
	if (g_gram_cost < 0.0 || nrow <= 0 || nnz_x <= 0 || nnz_y <= 0)
		return false;
	double pairs = (double) nnz_x * (double) nnz_y / (double) nrow;
	// small products stay with the general kernels of the dense-buffer route: their sums run in the reference's own
	// ascending order, bit for bit (tests/test_hip_vs_oracle.py); from the size on where that route takes the panel
	// kernels (pbc_applies) neither route is ordered like the reference and the faster one is taken
	if (g_gram_cost > 0.0 && dense_ops < 268435456.0)
		return false;
	if (sym) { pairs *= 0.5; dense_ops *= 0.5; }
	return sparse_route_seconds(pairs, nnz_x) * g_gram_cost < dense_route_seconds(dense_ops);
}

// 0: `O` holds the result; 1: a non-finite value or an NA took part (the caller takes the dense-buffer route, whose
// dirty-leaf rules are the reference's); -1: error
static int dev_crossprod_sparse_on(const svt_dev_csc *T, const svt_dev_csc *Y, bool sym, double *O, int64_t ldo, double dense_ops)
{
	DevBuf Ws;
	int bad = 1;
	if (Ws.alloc(svt_dev_crossprod_csc_csc_ws_bytes(T)))
		return -1;
	if (sym && g_gram_cost > 0.0) {
		// The choice was made on nnz^2 / (2 nrow) pairs; with t(x) at hand they can be counted: rows of very unequal
		// length hold more (sum of len^2), and an operand whose count says the other route is clearly faster goes there.
		double pairs = 0.0;
		if (launch_gram_pairs(T->col_ptr, T->ncol, (double *) Ws.p, 0))
			return -1;
		HIP_TRY(hipMemcpy(&pairs, Ws.p, 8, hipMemcpyDeviceToHost));
		if (sparse_route_seconds(pairs, 0) * g_gram_cost > 1.5 * dense_route_seconds(0.5 * dense_ops))
			return 1;
	}
	const int rc = svt_dev_crossprod_csc_csc(T, Y, sym ? 1 : 0, O, ldo, Ws.p, Ws.bytes, NULL, 0);
	if (rc > 0) { g_unsupported = 0; return 1; }      // a shape this kernel refuses: the other route
	if (rc < 0) return -1;
	HIP_TRY(hipMemcpy(&bad, Ws.p, 4, hipMemcpyDeviceToHost));
	return bad ? 1 : 0;
}

static int dev_crossprod_sparse(const CscGuard &X, const svt_dev_csc *Y, bool sym, double *O, int64_t ldo, double dense_ops)
{
	int own_T = 1;
	svt_dev_csc *T = transposed_for(X, &own_T);
	OwnedCsc TX = { T, own_T };
	if (T == NULL) {
		// an operand the transposition does not take (2^31 nonzeros or more): the dense-buffer route needs no t(x)
		if (g_unsupported) { g_unsupported = 0; return 1; }
		return -1;
	}
	return dev_crossprod_sparse_on(T, Y, sym, O, ldo, dense_ops);
}

// The dense-buffer route on resident operands (what the entry points below fall back to, and the yardstick of
// tools/debug/sparse_crossprod_time.py): out = ncol(X) x ncol(Y), column-major, zeroed here.  Y == X (the same
// handle): the unary form, half the dot products + mirror.  Allocates and synchronises.
extern "C" int svt_dev_crossprod_csc_csc_dense_buffer(const svt_dev_csc *X, const svt_dev_csc *Y, double *out)
{
	if (X->nrow != Y->nrow)
		return svt_set_error("svt_dev_crossprod_csc_csc_dense_buffer: non-conformable operands");
	const int64_t nx = X->ncol, ny = Y->ncol;
	if (nx == 0 || ny == 0) return 0;
	HIP_TRY(hipMemset(out, 0, (size_t) nx * ny * 8));
	if (X == Y) {
		if (dev_crossprod_pp(X, X, out, 1, nx)) return -1;
		if (launch_mirror_lower(out, nx, 0)) return -1;
		HIP_TRY(hipDeviceSynchronize());
		return 0;
	}
	const double Lpp = (double) Y->nnz * (double) nx, Rpp = (double) X->nnz * (double) ny;
	return Lpp < Rpp ? dev_crossprod_pp(Y, X, out, nx, 1) : dev_crossprod_pp(X, Y, out, 1, nx);
}

static int64_t view_nzcount(const svt_view *x)   // _REC_nzcount_SVT, SVT_SparseArray_class.c:200-218
{
	int64_t t = 0;
	if (x->svt_is_null) return 0;
	for (int64_t j = 0; j < x->nleaves; j++) t += x->nzcount[j];
	return t;
}

// C_crossprod2_SVT_SVT, src/SparseMatrix_mult.c:1037-1101
static int crossprod2_SVT_SVT_impl(const svt_view *x, const svt_view *y, double *out)
{
	if (ensure_init() || check_mult_view(x, "input objects") ||
	    check_mult_view(y, "input objects"))
		return -1;
	const int in_nrow = x->dim[0];
	if (in_nrow != y->dim[0])
		return svt_set_error("input SVT_SparseMatrix objects are non-conformable");
	if (x->Rtype != y->Rtype)
		return svt_set_error("input SVT_SparseMatrix objects must have the "
				     "same type() for now");
	const int out_nrow = x->dim[1], out_ncol = y->dim[1];
	const size_t out_n = (size_t) out_nrow * out_ncol;
	memset(out, 0, out_n * sizeof(double));
	if (out_n == 0)
		return 0;
	const int64_t Lpp_nops = view_nzcount(y) * out_nrow;   // :1077-1078
	const int64_t Rpp_nops = view_nzcount(x) * out_ncol;
	CscGuard X(x), Y(y);
	if (X.h == NULL || Y.h == NULL) return -1;
	DevBuf O;
	if (O.alloc(out_n * 8))
		return -1;
	// few pairs of nonzeros meet in a row: multiply only those (kernels_gram.hip); a non-finite value or an NA
	// anywhere sends the product down the reference's route below
	if (!x->svt_is_null && !y->svt_is_null &&
	    sparse_route_pays(X.h->nnz, Y.h->nnz, in_nrow, (double) (Lpp_nops < Rpp_nops ? Lpp_nops : Rpp_nops), false)) {
		const int st = dev_crossprod_sparse(X, Y.h, false, O.as<double>(), out_nrow, (double) (Lpp_nops < Rpp_nops ? Lpp_nops : Rpp_nops));
		if (st < 0) return -1;
		if (st == 0)
			return staged_download(out, O.p, out_n * 8) ? -1 : 0;
	}
	if (O.zero())
		return -1;
	int rc;
	if (Lpp_nops < Rpp_nops)   // expand the columns of x, walk the leaves of y
		rc = dev_crossprod_pp(Y.h, X.h, O.as<double>(), out_nrow, 1);
	else                       // expand the columns of y, walk the leaves of x
		rc = dev_crossprod_pp(X.h, Y.h, O.as<double>(), 1, out_nrow);
	if (rc) return -1;
	if (staged_download(out, O.p, out_n * 8)) return -1;
	return 0;
}
extern "C" int svt_crossprod2_SVT_SVT(const svt_view *x, const svt_view *y, double *out)
{
	g_unsupported = 0;
	return svt_status(crossprod2_SVT_SVT_impl(x, y, out));
}

// t(A) on the device, as a handle that owns its buffers (A may be released afterwards).
static svt_dev_csc *dev_transposed(const svt_dev_csc *A)
{
	svt_dev_csc *T = (svt_dev_csc *) calloc(1, sizeof(*T));
	if (T == NULL) { svt_set_error("out of memory"); return NULL; }
	T->Rtype = A->Rtype; T->owned = 1; T->na_background = A->na_background;
	T->nrow = A->ncol; T->ncol = A->nrow; T->nnz = A->nnz;
	const size_t n = A->nnz > 0 ? (size_t) A->nnz : 1;
	DevBuf ws;
	if (hipMalloc((void **) &T->col_ptr, ((size_t) T->ncol + 1) * 8) != hipSuccess ||
	    hipMalloc((void **) &T->row_idx, n * 4) != hipSuccess ||
	    hipMalloc(&T->val, n * elt_size(A->Rtype)) != hipSuccess ||
	    ws.alloc(transpose_ws_bytes(A->nrow, A->nnz))) {
		svt_set_error("device allocation failed (transposed operand)");
		svt_release(T);
		return NULL;
	}
	if (launch_transpose(A->col_ptr, A->row_idx, A->val, A->Rtype, A->nrow, A->ncol, A->nnz,
			     T->col_ptr, T->row_idx, T->val, ws.p, 0) ||
	    hipDeviceSynchronize() != hipSuccess) {
		if (svt_last_error()[0] == '\0') svt_set_error("device transposition failed");
		svt_release(T);
		return NULL;
	}
	return T;
}

// t(A) of an operand: kept with a resident operand, else built for this call (*owned = 1).
static svt_dev_csc *transposed_for(const CscGuard &A, int *owned)
{
	*owned = 1;
	if (A.key != 0) {
		std::lock_guard<std::mutex> lk(g_res_mu);
		for (Resident &r : g_res)
			if (r.key == A.key && r.tr != NULL) { *owned = 0; return r.tr; }
	}
	svt_dev_csc *T = dev_transposed(A.h);
	if (T == NULL || A.key == 0) return T;
	std::lock_guard<std::mutex> lk(g_res_mu);
	const size_t nb = csc_bytes(T);
	resident_make_room(nb);                           // may erase entries: search afterwards
	if (g_res_bytes + nb <= g_res_limit)
		for (Resident &r : g_res)
			if (r.key == A.key) {
				r.tr = T; r.bytes += nb; g_res_bytes += nb;
				*owned = 0;
				break;
			}
	return T;
}

// C_transpose_2D_SVT, src/SparseArray_aperm.c:395-423
static int transpose_2D_SVT_impl(const svt_view *x, int64_t *out_col_ptr,
				    int32_t *out_row_idx, void *out_val)
{
	if (ensure_init() || check_view(x))
		return -1;
	if (x->ndim != 2)
		return svt_set_error("object to transpose must have exactly 2 dimensions");
	const int64_t nrow = x->dim[0];
	if (x->svt_is_null || x->dim[1] == 0 || nrow == 0) {     // :405-406: nothing to move
		for (int64_t i = 0; i <= nrow; i++) out_col_ptr[i] = 0;
		return 0;
	}
	CscGuard A(x);
	if (A.h == NULL) return -1;
	int own_T = 1;
	svt_dev_csc *T = transposed_for(A, &own_T);
	OwnedCsc TA = { T, own_T };
	if (T == NULL) return -1;
	HIP_TRY(hipMemcpy(out_col_ptr, T->col_ptr, (size_t) (nrow + 1) * 8, hipMemcpyDeviceToHost));
	if (T->nnz > 0) {
		if (staged_download(out_row_idx, T->row_idx, (size_t) T->nnz * 4) ||
		    staged_download(out_val, T->val, (size_t) T->nnz * elt_size(T->Rtype)))
			return -1;
	}
	return 0;
}
extern "C" int svt_transpose_2D_SVT(const svt_view *x, int64_t *out_col_ptr,
				    int32_t *out_row_idx, void *out_val)
{
	g_unsupported = 0;
	return svt_status(transpose_2D_SVT_impl(x, out_col_ptr, out_row_idx, out_val));
}

// x %*% y, y an ordinary matrix: the R method (R/SparseMatrix-mult.R:195-215) is
// .crossprod2_SparseMatrix_matrix(t(x), y), i.e. C_transpose_2D_SVT on the host
// followed by C_crossprod2_SVT_mat.  Here the transposition happens on the device,
// between the upload and the product (no second marshalling of a 1e8-nonzero tree).
static int matmul_SVT_mat_impl(const svt_view *x, const void *y, int y_nrow,
				  int y_ncol, int y_Rtype, double *out)
{
	if (ensure_init() || check_mult_view(x, "input objects"))
		return -1;
	const int out_nrow = x->dim[0], in_nrow = x->dim[1];
	if (in_nrow != y_nrow)
		return svt_set_error("input objects are non-conformable");
	if (y_Rtype == SVT_LGLSXP) y_Rtype = SVT_INTSXP;
	if (!mult_types_ok(x->Rtype, y_Rtype))
		return svt_set_error("SparseArray internal error in "
				     "C_crossprod2_SVT_mat():\n"
				     "    'x_Rtype != TYPEOF(y)' not supported yet");
	const size_t out_n = (size_t) out_nrow * y_ncol;
	memset(out, 0, out_n * sizeof(double));
	if (x->svt_is_null || out_n == 0)
		return 0;
	CscGuard A(x);
	if (A.h == NULL) return -1;
	int own_T = 1;
	svt_dev_csc *T = transposed_for(A, &own_T);
	OwnedCsc TA = { T, own_T };
	if (T == NULL) return -1;
	A.drop();                           // a one-call operand: its untransposed copy can go now
	DevBuf Y, O;
	PbcAhead ahead;
	ahead.start(T, y_ncol, 0);
	if (Y.upload(y, (size_t) y_nrow * y_ncol * elt_size(y_Rtype)) ||
	    O.alloc(out_n * 8) || O.zero())
		return -1;
	Promoted pr;
	if (pr.promote(T, Y.p, y_Rtype, (size_t) y_nrow * y_ncol))
		return -1;
	if (dev_crossprod_chunked(pr.A, pr.Y, y_nrow, y_ncol, 0, O.as<double>(), 1, out_nrow,
				  pr.A == T ? &ahead : NULL))
		return -1;
	if (staged_download(out, O.p, out_n * 8)) return -1;
	return 0;
}
extern "C" int svt_matmul_SVT_mat(const svt_view *x, const void *y, int y_nrow,
				  int y_ncol, int y_Rtype, double *out)
{
	g_unsupported = 0;
	return svt_status(matmul_SVT_mat_impl(x, y, y_nrow, y_ncol, y_Rtype, out));
}

// x %*% y, both SVT_SparseMatrix: .crossprod2_SparseMatrix_SparseMatrix(t(x), y) with
// the transposition on the device; operand to expand chosen as C_crossprod2_SVT_SVT
// does (src/SparseMatrix_mult.c:1075-1097; nzcount(t(x)) == nzcount(x)).
static int matmul_SVT_SVT_impl(const svt_view *x, const svt_view *y, double *out)
{
	if (ensure_init() || check_mult_view(x, "input objects") ||
	    check_mult_view(y, "input objects"))
		return -1;
	if (x->dim[1] != y->dim[0])
		return svt_set_error("input SVT_SparseMatrix objects are non-conformable");
	if (x->Rtype != y->Rtype)
		return svt_set_error("input SVT_SparseMatrix objects must have the "
				     "same type() for now");
	const int out_nrow = x->dim[0], out_ncol = y->dim[1];
	const size_t out_n = (size_t) out_nrow * out_ncol;
	memset(out, 0, out_n * sizeof(double));
	if (out_n == 0)
		return 0;
	const int64_t Lpp_nops = view_nzcount(y) * out_nrow;
	const int64_t Rpp_nops = view_nzcount(x) * out_ncol;
	CscGuard X(x);
	if (X.h == NULL) return -1;
	CscGuard Y(y);                                  // (one upload of y for both routes)
	if (Y.h == NULL) return -1;
	// y much sparser than a dense matrix (<= 5 % filled), finite operands: the row-panel kernel on x itself,
	// no transposition, no dense operand (kernels_spmm.hip); a non-finite value or an NA anywhere sends the
	// product down the reference's route below.  The kernel adds the products of a cell in the order its lane
	// groups get to them: doubles can differ in the last bits from run to run and from the reference's
	// ascending-index order (inside the 1e-6 the contract allows; integer operands are exact below 2^53).
	if (view_nzcount(y) * 20 <= (int64_t) y->dim[0] * y->dim[1]) {
		DevBuf Os, Ws;
		int bad = 1;
		if (Os.alloc(out_n * 8) || Ws.alloc(svt_dev_matmul_csc_csc_ws_bytes(X.h)))
			return -1;
		const int rc_s = svt_dev_matmul_csc_csc(X.h, Y.h, Os.as<double>(), out_nrow, Ws.p, Ws.bytes, NULL, 0);
		if (rc_s < 0)
			return -1;
		if (rc_s == 0) {
			HIP_TRY(hipMemcpy(&bad, (char *) Ws.p + 4, 4, hipMemcpyDeviceToHost));
			if (!bad)
				return staged_download(out, Os.p, out_n * 8) ? -1 : 0;
		} else
			g_unsupported = 0;                // (a shape the row-panel kernel refuses: the route below)
	}
	int own_T = 1;
	svt_dev_csc *T = transposed_for(X, &own_T);
	OwnedCsc TX = { T, own_T };
	if (T == NULL) return -1;
	X.drop();
	DevBuf O;
	if (O.alloc(out_n * 8) || O.zero())
		return -1;
	int rc;
	if (Lpp_nops < Rpp_nops)
		rc = dev_crossprod_pp(Y.h, T, O.as<double>(), out_nrow, 1);
	else
		rc = dev_crossprod_pp(T, Y.h, O.as<double>(), 1, out_nrow);
	if (rc) return -1;
	if (staged_download(out, O.p, out_n * 8)) return -1;
	return 0;
}
extern "C" int svt_matmul_SVT_SVT(const svt_view *x, const svt_view *y, double *out)
{
	g_unsupported = 0;
	return svt_status(matmul_SVT_SVT_impl(x, y, out));
}

// C_crossprod1_SVT, src/SparseMatrix_mult.c:1104-1140
static int crossprod1_SVT_impl(const svt_view *x, double *out)
{
	if (ensure_init() || check_mult_view(x, "'x'"))
		return -1;
	const int n = x->dim[1];
	const size_t out_n = (size_t) n * n;
	memset(out, 0, out_n * sizeof(double));
	if (x->svt_is_null || out_n == 0)      // :880-881
		return 0;
	CscGuard X(x);
	if (X.h == NULL) return -1;
	DevBuf O;
	if (O.alloc(out_n * 8))
		return -1;
	if (sparse_route_pays(X.h->nnz, X.h->nnz, x->dim[0], (double) X.h->nnz * (double) n, true)) {
		const int st = dev_crossprod_sparse(X, X.h, true, O.as<double>(), n, (double) X.h->nnz * (double) n);
		if (st < 0) return -1;
		if (st == 0)
			return staged_download(out, O.p, out_n * 8) ? -1 : 0;
	}
	if (O.zero())
		return -1;
	if (dev_crossprod_pp(X.h, X.h, O.as<double>(), 1, n))
		return -1;
	if (launch_mirror_lower(O.as<double>(), n, 0))
		return -1;
	if (staged_download(out, O.p, out_n * 8)) return -1;
	return 0;
}
extern "C" int svt_crossprod1_SVT(const svt_view *x, double *out)
{
	g_unsupported = 0;
	return svt_status(crossprod1_SVT_impl(x, out));
}

// tcrossprod(x) = crossprod(t(x)) and tcrossprod(x, y) = crossprod(t(x), t(y)) of SVT_SparseMatrix objects in one call.
// The R methods (R/SparseMatrix-mult.R:165-193) take t() on the host first (C_transpose_2D_SVT: the transposed tree comes
// back as R leaves and is marshalled again by C_crossprod1_SVT / C_crossprod2_SVT_SVT).  Here the operands are uploaded as
// they are and transposed on the device; the sparse-aware kernel needs the ROWS of its first operand t(x), i.e. x itself --
// no transposition at all on that side.  Checks, messages and the route choice are those of the crossprod entry points
// applied to the transposed operands; a non-finite value or an NA anywhere takes the dense-buffer route (the reference's
// dirty-leaf rules).  out: nrow(x) x nrow(x) / nrow(x) x nrow(y) doubles, column-major.
static int tcrossprod1_SVT_impl(const svt_view *x, double *out)
{
	if (ensure_init() || check_mult_view(x, "'x'"))
		return -1;
	const int n = x->dim[0];
	const size_t out_n = (size_t) n * n;
	memset(out, 0, out_n * sizeof(double));
	if (x->svt_is_null || out_n == 0 || x->dim[1] == 0)      // (t(x)@SVT is NULL: src/SparseMatrix_mult.c:880-881)
		return 0;
	CscGuard X(x);
	if (X.h == NULL) return -1;
	int own_T = 1;
	svt_dev_csc *M = transposed_for(X, &own_T);             // M = t(x): ncol(x) rows, nrow(x) leaves
	OwnedCsc TM = { M, own_T };
	if (M == NULL) return -1;
	DevBuf O;
	if (O.alloc(out_n * 8))
		return -1;
	const double dense_ops = (double) M->nnz * (double) n;
	if (sparse_route_pays(M->nnz, M->nnz, M->nrow, dense_ops, true)) {
		const int st = dev_crossprod_sparse_on(X.h, M, true, O.as<double>(), n, dense_ops);     // t(M) is x
		if (st < 0) return -1;
		if (st == 0)
			return staged_download(out, O.p, out_n * 8) ? -1 : 0;
	}
	if (O.zero())
		return -1;
	if (dev_crossprod_pp(M, M, O.as<double>(), 1, n))
		return -1;
	if (launch_mirror_lower(O.as<double>(), n, 0))
		return -1;
	if (staged_download(out, O.p, out_n * 8)) return -1;
	return 0;
}
extern "C" int svt_tcrossprod1_SVT(const svt_view *x, double *out)
{
	g_unsupported = 0;
	return svt_status(tcrossprod1_SVT_impl(x, out));
}

static int tcrossprod2_SVT_SVT_impl(const svt_view *x, const svt_view *y, double *out)
{
	if (ensure_init() || check_mult_view(x, "input objects") ||
	    check_mult_view(y, "input objects"))
		return -1;
	if (x->dim[1] != y->dim[1])
		return svt_set_error("input SVT_SparseMatrix objects are non-conformable");
	if (x->Rtype != y->Rtype)
		return svt_set_error("input SVT_SparseMatrix objects must have the "
				     "same type() for now");
	const int out_nrow = x->dim[0], out_ncol = y->dim[0];
	const size_t out_n = (size_t) out_nrow * out_ncol;
	memset(out, 0, out_n * sizeof(double));
	if (out_n == 0)
		return 0;
	const int64_t Lpp_nops = view_nzcount(y) * out_nrow;
	const int64_t Rpp_nops = view_nzcount(x) * out_ncol;
	CscGuard X(x), Y(y);
	if (X.h == NULL || Y.h == NULL) return -1;
	int own_Ty = 1;
	svt_dev_csc *Ty = transposed_for(Y, &own_Ty);
	OwnedCsc TY = { Ty, own_Ty };
	if (Ty == NULL) return -1;
	DevBuf O;
	if (O.alloc(out_n * 8))
		return -1;
	const double dense_ops = (double) (Lpp_nops < Rpp_nops ? Lpp_nops : Rpp_nops);
	if (!x->svt_is_null && !y->svt_is_null && x->dim[1] > 0 &&
	    sparse_route_pays(X.h->nnz, Y.h->nnz, x->dim[1], dense_ops, false)) {
		const int st = dev_crossprod_sparse_on(X.h, Ty, false, O.as<double>(), out_nrow, dense_ops);   // t(t(x)) is x
		if (st < 0) return -1;
		if (st == 0)
			return staged_download(out, O.p, out_n * 8) ? -1 : 0;
	}
	int own_Tx = 1;
	svt_dev_csc *Tx = transposed_for(X, &own_Tx);
	OwnedCsc TXg = { Tx, own_Tx };
	if (Tx == NULL) return -1;
	if (O.zero())
		return -1;
	int rc;
	if (Lpp_nops < Rpp_nops)
		rc = dev_crossprod_pp(Ty, Tx, O.as<double>(), out_nrow, 1);
	else
		rc = dev_crossprod_pp(Tx, Ty, O.as<double>(), 1, out_nrow);
	if (rc) return -1;
	if (staged_download(out, O.p, out_n * 8)) return -1;
	return 0;
}
extern "C" int svt_tcrossprod2_SVT_SVT(const svt_view *x, const svt_view *y, double *out)
{
	g_unsupported = 0;
	return svt_status(tcrossprod2_SVT_SVT_impl(x, y, out));
}

// ==================================================================================
// Host level: stats
// ==================================================================================
static int run_colstats(const svt_dev_csc *A, int opcode, int na_rm, double center,
			int64_t inner, void *out_host, int out_Rtype, int *warn)
{
	const int64_t nseg = A->ncol / inner;
	const size_t osz = out_Rtype == SVT_REALSXP ? 8 : 4;
	DevBuf O, W;
	if (O.alloc((size_t) nseg * osz) || W.alloc(16) || W.zero())
		return -1;
	if (svt_dev_colstats(A, opcode, na_rm, center, inner, O.p, W.as<int>(), 0))
		return -1;
	int w = 0;
	HIP_TRY(hipMemcpy(out_host, O.p, (size_t) nseg * osz, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(&w, W.p, 4, hipMemcpyDeviceToHost));
	if (w) *warn = 1;
	return 0;
}

// C_colStats_SVT, src/SparseArray_matrixStats.c:234-284
static int colStats_SVT_impl(const svt_view *x, int opcode, int na_rm, double center,
				int dims, void *out, int *warn)
{
	*warn = 0;
	if (ensure_init() || check_view(x) || check_stat_op(opcode, x->Rtype))
		return -1;
	if (dims < 1 || dims > x->ndim)
		return svt_set_error("'dims' must be >= 1 and <= %d", x->ndim);
	if (device_op_supported(opcode))
		return -1;
	int64_t inner = 1, nout = 1;
	for (int a = 1; a < dims; a++) inner *= x->dim[a];
	for (int a = dims; a < x->ndim; a++) nout *= x->dim[a];
	if (nout == 0)
		return 0;
	const int out_Rtype = svt_colStats_out_Rtype(opcode, x->Rtype);
	CscGuard A(x);
	if (A.h == NULL) return -1;
	if (inner == 0) {
		// zero-extent inner dims: every result summarizes an empty vector.
		// One empty leaf per result gives exactly that.
		std::vector<int64_t> cp((size_t) nout + 1, 0);
		DevBuf P;
		if (P.upload(cp.data(), cp.size() * 8)) return -1;
		svt_dev_csc E = *A.h;
		E.owned = 0; E.ncol = nout; E.nnz = 0; E.nrow = 0; E.col_ptr = P.as<int64_t>();
		return run_colstats(&E, opcode, na_rm, center, 1, out, out_Rtype, warn);
	}
	return run_colstats(A.h, opcode, na_rm, center, inner, out, out_Rtype, warn);
}
extern "C" int svt_colStats_SVT(const svt_view *x, int opcode, int na_rm, double center,
				int dims, void *out, int *warn)
{
	g_unsupported = 0;
	return svt_status(colStats_SVT_impl(x, opcode, na_rm, center, dims, out, warn));
}

// colMedians(): .colMedians_SVT_SparseMatrix, R/SparseArray-matrixStats.R:761-784 (pure R in the
// reference, with a TODO asking for a .Call version).  out: ncol(x) doubles.
static int medians_SVT(const svt_view *x, int na_rm, int by_row, double *out)
{
	if (ensure_init() || check_view(x))
		return -1;
	if (x->ndim != 2)       // stopifnot_2D_object(), R/SparseArray-matrixStats.R:51-57
		return svt_set_error("the %s() method for SparseArray objects only supports 2D "
				     "objects (i.e. SparseMatrix objects) at the moment",
				     by_row ? "rowMedians" : "colMedians");
	if (x->Rtype != SVT_REALSXP && x->Rtype != SVT_INTSXP && x->Rtype != SVT_LGLSXP)
		return svt_set_error("colMedians(): unsupported type");
	if (x->na_background)
		return svt_set_error("colMedians() is not supported on NaArray objects");
	const int64_t nout = x->dim[by_row ? 0 : 1];
	if (nout == 0)
		return 0;
	CscGuard A(x);
	if (A.h == NULL) return -1;
	const svt_dev_csc *M = A.h;
	int own_T = 0;
	svt_dev_csc *T = NULL;
	if (by_row) {           // rowMedians(x) = colMedians(t(x)), :802-815; t() on the device
		T = transposed_for(A, &own_T);
		if (T == NULL) return -1;
		M = T;
	}
	OwnedCsc TG = { T, own_T };
	DevBuf O, W;
	if (O.alloc((size_t) nout * 8) || W.alloc(colmedians_ws_bytes(M->nnz, nout)))
		return -1;
	if (launch_colmedians(M->col_ptr, M->val, M->Rtype, M->nrow, nout, M->nnz, na_rm,
			      O.as<double>(), W.p, 0))
		return -1;
	HIP_TRY(hipDeviceSynchronize());
	return staged_download(out, O.p, (size_t) nout * 8);
}

static int colMedians_SVT_impl(const svt_view *x, int na_rm, double *out)
{
	return medians_SVT(x, na_rm, 0, out);
}
extern "C" int svt_colMedians_SVT(const svt_view *x, int na_rm, double *out)
{
	g_unsupported = 0;
	return svt_status(colMedians_SVT_impl(x, na_rm, out));
}

static int rowMedians_SVT_impl(const svt_view *x, int na_rm, double *out)
{
	return medians_SVT(x, na_rm, 1, out);
}
extern "C" int svt_rowMedians_SVT(const svt_view *x, int na_rm, double *out)
{
	g_unsupported = 0;
	return svt_status(rowMedians_SVT_impl(x, na_rm, out));
}

// C_summarize_SVT, src/SparseArray_summarization.c:112-142
static int summarize_SVT_impl(const svt_view *x, int opcode, int na_rm, double center,
				 double *out_d, int *out_i, int *out_Rtype, int *warn)
{
	*warn = 0;
	if (ensure_init() || check_view(x) || check_stat_op(opcode, x->Rtype))
		return -1;
	if (opcode == SVT_OP_SUM_X_X2 || opcode == SVT_OP_VAR2 || opcode == SVT_OP_SD2)
		return device_op_supported(opcode);
	const int rt = svt_colStats_out_Rtype(opcode, x->Rtype);
	*out_Rtype = rt;
	out_d[0] = out_d[1] = 0.0;
	out_i[0] = out_i[1] = 0;
	CscGuard A(x);
	if (A.h == NULL) return -1;
	svt_dev_csc V = *A.h;      // the whole array as one generalized column
	V.owned = 0;
	int64_t inner = V.ncol;
	DevBuf P;
	if (inner == 0) {          // some outer dim is 0: one empty segment
		int64_t cp[2] = {0, 0};
		if (P.upload(cp, sizeof(cp))) return -1;
		V.ncol = 1; V.nrow = 0; V.nnz = 0; V.col_ptr = P.as<int64_t>();
		inner = 1;
	}
	const int ops[2] = { opcode == SVT_OP_RANGE ? SVT_OP_MIN : opcode, SVT_OP_MAX };
	const int nops = opcode == SVT_OP_RANGE ? 2 : 1;
	for (int t = 0; t < nops; t++) {
		double d = 0.0;
		int i = 0;
		void *dst = rt == SVT_REALSXP ? (void *) &d : (void *) &i;
		if (run_colstats(&V, ops[t], na_rm, center, inner, dst, rt, warn))
			return -1;
		out_d[t] = d;
		out_i[t] = i;
	}
	return 0;
}
extern "C" int svt_summarize_SVT(const svt_view *x, int opcode, int na_rm, double center,
				 double *out_d, int *out_i, int *out_Rtype, int *warn)
{
	g_unsupported = 0;
	return svt_status(summarize_SVT_impl(x, opcode, na_rm, center, out_d, out_i, out_Rtype, warn));
}

// C_rowStats_SVT, src/SparseArray_matrixStats.c:1121-1205
static int rowStats_SVT_impl(const svt_view *x, int opcode, int na_rm,
				const double *center, int dims, void *out, int *warn)
{
	*warn = 0;
	if (ensure_init() || check_view(x) || check_stat_op(opcode, x->Rtype))
		return -1;
	if (dims < 1 || dims > x->ndim - 1)
		return svt_set_error("'dims' must be >= 1 and <= %d", x->ndim - 1);
	if (x->na_background && opcode == SVT_OP_CENTERED_X2_SUM)   // :639-642
		return svt_set_error("operation not yet supported on NaArray objects");
	if (opcode != SVT_OP_COUNTNAS && opcode != SVT_OP_ANYNA &&
	    opcode != SVT_OP_MIN && opcode != SVT_OP_MAX &&
	    opcode != SVT_OP_SUM && opcode != SVT_OP_CENTERED_X2_SUM)
		return svt_set_error("SparseArray internal error in C_rowStats_SVT():\n"
				     "    operation not supported");
	const int out_Rtype = svt_colStats_out_Rtype(opcode, x->Rtype);
	const size_t osz = out_Rtype == SVT_REALSXP ? 8 : 4;
	int64_t inner = 1, nstrata = 1;
	for (int a = 1; a < dims; a++) inner *= x->dim[a];
	for (int a = dims; a < x->ndim; a++) nstrata *= x->dim[a];
	const int64_t out_len = inner * x->dim[0];
	if (out_len == 0)
		return 0;
	if ((opcode == SVT_OP_MIN || opcode == SVT_OP_MAX) && nstrata == 0) {
		// constant fill, :970-982
		for (int64_t i = 0; i < out_len; i++) {
			if (out_Rtype == SVT_REALSXP)
				((double *) out)[i] = opcode == SVT_OP_MIN ? INFINITY : -INFINITY;
			else
				((int *) out)[i] = NA_INT;
		}
		if (out_Rtype != SVT_REALSXP) *warn = 1;
		return 0;
	}
	if (nstrata > 0xFFFFFFFFLL)
		return svt_set_unsupported("too many strata for the device coverage counters");
	CscGuard A(x);
	if (A.h == NULL) return -1;
	DevBuf O, C, S, W;
	if (O.alloc((size_t) out_len * osz) ||
	    S.alloc(rowstats_scratch_bytes(opcode, out_Rtype, out_len)) ||
	    W.alloc(16) || W.zero())
		return -1;
	if (center != NULL && C.upload(center, (size_t) out_len * 8))
		return -1;
	RowStatsArgs a;
	a.col_ptr = A.h->col_ptr; a.row_idx = A.h->row_idx; a.val = A.h->val;
	a.Rtype = A.h->Rtype; a.ncol = A.h->ncol; a.nrow = A.h->nrow;
	a.inner = inner; a.nstrata = nstrata; a.out_len = out_len;
	a.opcode = opcode; a.na_rm = na_rm;
	a.center = center ? C.as<double>() : NULL;
	a.out = O.p; a.scratch = S.p; a.warn_flag = W.as<int>(); a.nnz_hint = A.h->nnz;
	a.na_bg = x->na_background != 0;
	if (a.na_bg && inner > 65535)
		return svt_set_unsupported("row statistics of NaArray objects: more than 65535 output columns");
	if (inner <= 65535) {
		DevBuf T;
		if (T.alloc(rowstats_panel_ws_bytes(a.nrow, a.ncol)) || launch_rowstats_panel(a, T.p, 0))
			return -1;
		HIP_TRY(hipDeviceSynchronize());
	} else if (launch_rowstats(a, A.h->nnz, 0)) {
		return -1;
	}
	int w = 0;
	if (staged_download(out, O.p, (size_t) out_len * osz)) return -1;
	HIP_TRY(hipMemcpy(&w, W.p, 4, hipMemcpyDeviceToHost));
	if (w) *warn = 1;
	return 0;
}
extern "C" int svt_rowStats_SVT(const svt_view *x, int opcode, int na_rm,
				const double *center, int dims, void *out, int *warn)
{
	g_unsupported = 0;
	return svt_status(rowStats_SVT_impl(x, opcode, na_rm, center, dims, out, warn));
}

// ==================================================================================
// Host level: rowsum / colsum
// ==================================================================================
static int check_group(const int *group, int n, int ngroup)   // rowsum_methods.c:15-37
{
	for (int i = 0; i < n; i++) {
		const int g = group[i];
		if (g == NA_INT) {
			if (ngroup < 1)
				return svt_set_error("'ngroup' must be >= 1 when 'group' "
						     "contains missing values");
		} else if (g < 1 || g > ngroup) {
			return svt_set_error("all non-NA values in 'group' must "
					     "be >= 1 and <= 'ngroup'");
		}
	}
	return 0;
}

static int groupsum_host(const svt_dev_csc *A, const int32_t *col_ptr32,
			 const int *group, int ngroup, int na_rm, bool colsum,
			 void *out, int *ovflow)
{
	const int64_t glen = colsum ? A->ncol : A->nrow;
	const int64_t out_len = colsum ? A->nrow * (int64_t) ngroup
				       : (int64_t) ngroup * A->ncol;
	if (out_len > 0x7FFFFFFFLL)   // safe_int_mult() guard, :296-301
		return svt_set_error("too many groups (matrix of sums will be too big)");
	const size_t osz = elt_size(A->Rtype);
	if (out_len == 0)
		return 0;
	DevBuf G, O, S, W;
	if (G.upload(group, (size_t) glen * 4) || O.alloc((size_t) out_len * osz) ||
	    S.alloc(groupsum_scratch_bytes(A->Rtype, out_len)) || W.alloc(16) || W.zero())
		return -1;
	GroupSumArgs a;
	memset(&a, 0, sizeof(a));
	a.col_ptr64 = col_ptr32 ? NULL : A->col_ptr;
	a.col_ptr32 = col_ptr32;
	a.row_idx = A->row_idx; a.val = A->val; a.Rtype = A->Rtype;
	a.nrow = A->nrow; a.ncol = A->ncol;
	a.group = G.as<int>(); a.ngroup = ngroup; a.na_rm = na_rm;
	a.out = O.p; a.scratch = S.p; a.ovflow_flag = W.as<int>();
	int rc;
	if (colsum)
		rc = launch_colsum(a, 0);
	else if (A->Rtype == SVT_REALSXP && ngroup <= 8192 && A->ncol > 0 &&
		 A->nnz / A->ncol >= ngroup / 4)
		rc = launch_rowsum_lds(a, 0);
	else
		rc = launch_rowsum(a, 0);
	if (rc) return -1;
	int w = 0;
	if (staged_download(out, O.p, (size_t) out_len * osz)) return -1;
	HIP_TRY(hipMemcpy(&w, W.p, 4, hipMemcpyDeviceToHost));
	if (ovflow && w) *ovflow = 1;
	return 0;
}

static int xsum_SVT(const svt_view *x, const int *group, int ngroup, int na_rm,
		    bool colsum, void *out, int *ovflow)
{
	*ovflow = 0;
	if (ensure_init() || check_view(x))
		return -1;
	if (x->ndim != 2)
		return svt_set_error("input object must have 2 dimensions");
	if (x->na_background)      // rowsum()/colsum() have no NaArray methods (R/rowsum-methods.R)
		return svt_set_error("NaArray objects are not supported by this operation");
	if (x->Rtype != SVT_REALSXP && x->Rtype != SVT_INTSXP)
		return svt_set_error("rowsum() and colsum() do not support "
				     "SVT_SparseMatrix objects of this type at the moment");
	if (check_group(group, colsum ? x->dim[1] : x->dim[0], ngroup))
		return -1;
	CscGuard A(x);
	if (A.h == NULL) return -1;
	return groupsum_host(A.h, NULL, group, ngroup, na_rm, colsum, out, ovflow);
}

// C_rowsum_SVT, src/rowsum_methods.c:281-325
static int rowsum_SVT_impl(const svt_view *x, const int *group, int ngroup,
			      int na_rm, void *out, int *ovflow)
{
	return xsum_SVT(x, group, ngroup, na_rm, false, out, ovflow);
}
extern "C" int svt_rowsum_SVT(const svt_view *x, const int *group, int ngroup,
			      int na_rm, void *out, int *ovflow)
{
	g_unsupported = 0;
	return svt_status(rowsum_SVT_impl(x, group, ngroup, na_rm, out, ovflow));
}

// C_colsum_SVT, src/rowsum_methods.c:363-401
static int colsum_SVT_impl(const svt_view *x, const int *group, int ngroup,
			      int na_rm, void *out, int *ovflow)
{
	return xsum_SVT(x, group, ngroup, na_rm, true, out, ovflow);
}
extern "C" int svt_colsum_SVT(const svt_view *x, const int *group, int ngroup,
			      int na_rm, void *out, int *ovflow)
{
	g_unsupported = 0;
	return svt_status(colsum_SVT_impl(x, group, ngroup, na_rm, out, ovflow));
}

static int xsum_dgC(int nrow, int ncol, const double *xx, const int *xi, const int *xp,
		    const int *group, int ngroup, int na_rm, bool colsum, double *out)
{
	if (ensure_init() || check_group(group, colsum ? ncol : nrow, ngroup))
		return -1;
	const int64_t nnz = ncol > 0 ? xp[ncol] : 0;
	DevBuf P, I, X;
	if (P.upload(xp, (size_t) (ncol + 1) * 4) || I.upload(xi, (size_t) nnz * 4) ||
	    X.upload(xx, (size_t) nnz * 8))
		return -1;
	svt_dev_csc A;
	memset(&A, 0, sizeof(A));
	A.Rtype = SVT_REALSXP; A.nrow = nrow; A.ncol = ncol; A.nnz = nnz;
	A.row_idx = I.as<int32_t>(); A.val = X.p;
	int ov = 0;
	// the LDS path reads col_ptr64; keep the int32 'p' slot on the atomic path
	const int64_t out_len = colsum ? (int64_t) nrow * ngroup : (int64_t) ngroup * ncol;
	if (out_len > 0x7FFFFFFFLL)
		return svt_set_error("too many groups (matrix of sums will be too big)");
	if (out_len == 0)
		return 0;
	DevBuf G, O, S;
	if (G.upload(group, (size_t) (colsum ? ncol : nrow) * 4) ||
	    O.alloc((size_t) out_len * 8) || S.alloc(16))
		return -1;
	GroupSumArgs a;
	memset(&a, 0, sizeof(a));
	a.col_ptr32 = P.as<int32_t>();
	a.row_idx = A.row_idx; a.val = A.val; a.Rtype = SVT_REALSXP;
	a.nrow = nrow; a.ncol = ncol; a.group = G.as<int>(); a.ngroup = ngroup;
	a.na_rm = na_rm; a.out = O.p; a.scratch = S.p;
	if (colsum ? launch_colsum(a, 0) : launch_rowsum(a, 0))
		return -1;
	(void) ov;
	if (staged_download(out, O.p, (size_t) out_len * 8)) return -1;
	return 0;
}

// C_rowsum_dgCMatrix / C_colsum_dgCMatrix, src/rowsum_methods.c:328-356, 404-439
static int rowsum_dgCMatrix_impl(int nrow, int ncol, const double *xx, const int *xi,
				    const int *xp, const int *group, int ngroup,
				    int na_rm, double *out)
{
	return xsum_dgC(nrow, ncol, xx, xi, xp, group, ngroup, na_rm, false, out);
}
extern "C" int svt_rowsum_dgCMatrix(int nrow, int ncol, const double *xx, const int *xi,
				    const int *xp, const int *group, int ngroup,
				    int na_rm, double *out)
{
	g_unsupported = 0;
	return svt_status(rowsum_dgCMatrix_impl(nrow, ncol, xx, xi, xp, group, ngroup, na_rm, out));
}
static int colsum_dgCMatrix_impl(int nrow, int ncol, const double *xx, const int *xi,
				    const int *xp, const int *group, int ngroup,
				    int na_rm, double *out)
{
	return xsum_dgC(nrow, ncol, xx, xi, xp, group, ngroup, na_rm, true, out);
}
extern "C" int svt_colsum_dgCMatrix(int nrow, int ncol, const double *xx, const int *xi,
				    const int *xp, const int *group, int ngroup,
				    int na_rm, double *out)
{
	g_unsupported = 0;
	return svt_status(colsum_dgCMatrix_impl(nrow, ncol, xx, xi, xp, group, ngroup, na_rm, out));
}

// ==================================================================================
// Host level: column statistics of a dgCMatrix (src/sparseMatrix_utils.c:106-223)
// ==================================================================================
// The (x, p) slots of a dgCMatrix are the CSC layout itself (the 'i' slot is not read, as in
// the reference: :115-116, :152-153, :213-214).  which: 0 = colMins, 1 = colMaxs,
// 2 = colRanges (out: ncol x 2, mins then maxs, :155), 3 = colVars.  The extrema follow the
// SVT rules for doubles (NA wins, then NaN, one implicit zero when nzcount < nrow:
// min_double/max_double/minmax_double, :15-103); colVars is col_var() (:173-203): plain IEEE
// arithmetic, no NA rule and no "NA when fewer than two values" rule.
static int colstat_dgC(int nrow, int ncol, const double *xx, const int *xp, int na_rm,
		       int which, double *out)
{
	if (ensure_init())
		return -1;
	if (nrow < 0 || ncol < 0 || xp == NULL)
		return svt_set_error("invalid dgCMatrix slots");
	if (ncol == 0)
		return 0;
	std::vector<int64_t> cp((size_t) ncol + 1);
	for (int j = 0; j <= ncol; j++) {
		if (xp[j] < 0 || (j > 0 && (xp[j] < xp[j - 1] || xp[j] - xp[j - 1] > nrow)))
			return svt_set_error("invalid dgCMatrix 'p' slot");
		cp[(size_t) j] = xp[j];
	}
	const int64_t nnz = cp[(size_t) ncol];
	if (nnz > 0 && xx == NULL)
		return svt_set_error("invalid dgCMatrix slots");
	DevBuf P, X, O;
	if (P.upload(cp.data(), cp.size() * 8) || X.upload(xx, (size_t) nnz * 8) ||
	    O.alloc((size_t) ncol * 8 * (which == 2 ? 2 : 1)))
		return -1;
	svt_dev_csc A;
	memset(&A, 0, sizeof(A));
	A.Rtype = SVT_REALSXP; A.nrow = nrow; A.ncol = ncol; A.nnz = nnz;
	A.col_ptr = P.as<int64_t>(); A.val = X.p;
	const double NA = svt_na_real();
	int rc;
	switch (which) {
	case 0: rc = dev_colstats_ex(&A, SVT_OP_MIN, na_rm, NA, 1, O.p, NULL, 0, 1); break;
	case 1: rc = dev_colstats_ex(&A, SVT_OP_MAX, na_rm, NA, 1, O.p, NULL, 0, 1); break;
	case 2:
		rc = dev_colstats_ex(&A, SVT_OP_MIN, na_rm, NA, 1, O.p, NULL, 0, 1);
		if (rc == 0)
			rc = dev_colstats_ex(&A, SVT_OP_MAX, na_rm, NA, 1, O.as<double>() + ncol, NULL, 0, 1);
		break;
	default: rc = dev_colstats_ex(&A, SVT_OP_VAR1, na_rm, NA, 1, O.p, NULL, 0, 1); break;
	}
	if (rc) return -1;
	HIP_TRY(hipDeviceSynchronize());
	return staged_download(out, O.p, (size_t) ncol * 8 * (which == 2 ? 2 : 1));
}

// C_colMins_dgCMatrix / C_colMaxs_dgCMatrix, src/sparseMatrix_utils.c:128-138
static int colMins_dgCMatrix_impl(int nrow, int ncol, const double *xx, const int *xp,
				     int na_rm, double *out)
{
	return colstat_dgC(nrow, ncol, xx, xp, na_rm, 0, out);
}
extern "C" int svt_colMins_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
				     int na_rm, double *out)
{
	g_unsupported = 0;
	return svt_status(colMins_dgCMatrix_impl(nrow, ncol, xx, xp, na_rm, out));
}
static int colMaxs_dgCMatrix_impl(int nrow, int ncol, const double *xx, const int *xp,
				     int na_rm, double *out)
{
	return colstat_dgC(nrow, ncol, xx, xp, na_rm, 1, out);
}
extern "C" int svt_colMaxs_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
				     int na_rm, double *out)
{
	g_unsupported = 0;
	return svt_status(colMaxs_dgCMatrix_impl(nrow, ncol, xx, xp, na_rm, out));
}
// C_colRanges_dgCMatrix, src/sparseMatrix_utils.c:143-166
static int colRanges_dgCMatrix_impl(int nrow, int ncol, const double *xx, const int *xp,
				       int na_rm, double *out)
{
	return colstat_dgC(nrow, ncol, xx, xp, na_rm, 2, out);
}
extern "C" int svt_colRanges_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
				       int na_rm, double *out)
{
	g_unsupported = 0;
	return svt_status(colRanges_dgCMatrix_impl(nrow, ncol, xx, xp, na_rm, out));
}
// C_colVars_dgCMatrix, src/sparseMatrix_utils.c:205-223
static int colVars_dgCMatrix_impl(int nrow, int ncol, const double *xx, const int *xp,
				     int na_rm, double *out)
{
	return colstat_dgC(nrow, ncol, xx, xp, na_rm, 3, out);
}
extern "C" int svt_colVars_dgCMatrix(int nrow, int ncol, const double *xx, const int *xp,
				     int na_rm, double *out)
{
	g_unsupported = 0;
	return svt_status(colVars_dgCMatrix_impl(nrow, ncol, xx, xp, na_rm, out));
}
