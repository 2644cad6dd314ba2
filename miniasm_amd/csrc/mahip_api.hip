// mahip_api.hip -- context lifetime, device memory, counters and event-based kernel timing behind include/mahip.h
#include "mahip_internal.hpp"
#include <stdarg.h>

static thread_local char g_err[1024] = "";

void mahip_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

extern "C" const char *mahip_strerror(void) { return g_err; }

extern "C" int mahip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// ---- device memory of a context ----
// Freed buffers are kept and handed out again (first the smallest piece that fits, split when much is left over; neighbours of one allocation grow together
// again) instead of going back to the driver: on part of the MI355X pool an allocation that reuses memory freed a moment ago waits for the driver to clear it at
// ~8 GB/s -- the CLI releases the text and the parser's columns (60 GB at BASELINE configs[4]) and allocates the pipeline's buffers right after: 2 s in the first
// coverage pass there, 0.2 s at 50 M overlaps, half of cfg4's 1.2 s end-to-end (profiles/r03_tiewalk.txt, tools/probes/alloc_probe.hip).  Everything runs on the
// context's one stream, so a piece that changes hands is safe in stream order.  MA_DEV_POOL=0: plain hipMalloc / hipFree (the guard-page run of the CPU build).
static bool pool_on() { static int v = -1; if (v < 0) { const char *e = getenv("MA_DEV_POOL"); v = !(e && atoi(e) == 0); } return v != 0; }

// The contexts of this process (bench.py's second context for the tail, several batches in flight): a context whose hipMalloc fails asks the
// others to let go of what their pools hold idle.  Other PROCESSES on the same GPU (the shared-memory ranks of a one-GPU test) cannot be asked;
// for them the orchestration trims at the phase boundaries (mahip_mem_trim).
static std::mutex g_reg_mu;
static std::vector<mahip_ctx*> g_ctxs;
static void ctx_register(mahip_ctx *c) { std::lock_guard<std::mutex> g(g_reg_mu); g_ctxs.push_back(c); }
static void ctx_unregister(mahip_ctx *c)
{
	std::lock_guard<std::mutex> g(g_reg_mu);
	for (size_t i = 0; i < g_ctxs.size(); ++i) if (g_ctxs[i] == c) { g_ctxs[i] = g_ctxs.back(); g_ctxs.pop_back(); break; }
}

static DevPool::Base *pool_base_of(DevPool &P, const char *p)
{
	for (DevPool::Base &b : P.bases) if (p >= b.p && p < b.p + b.bytes) return &b;
	return nullptr;
}

static void pool_give_back(mahip_ctx *c, char *p, size_t cap)
{
	DevPool &P = c->pool;
	std::lock_guard<std::mutex> g(P.mu);
	DevPool::Base *base = pool_base_of(P, p);
	if (!base) { (void)hipFree(p); return; } // (not from the pool: cannot happen)
	DevPool::Piece nw = { p, cap, base->p };
	for (size_t i = 0; i < P.free_pieces.size();) { // grow together with free neighbours of the same allocation
		DevPool::Piece &f = P.free_pieces[i];
		if (f.base == nw.base && (f.p + f.cap == nw.p || nw.p + nw.cap == f.p)) {
			if (f.p < nw.p) nw.p = f.p;
			nw.cap += f.cap;
			P.free_pieces[i] = P.free_pieces.back(); P.free_pieces.pop_back();
			i = 0; // the grown piece may now touch another one
		} else ++i;
	}
	P.free_pieces.push_back(nw);
}

// Whole allocations that are free go back to the driver.  A piece may still be read by kernels queued before it was given up (stream order is
// all that protects a piece inside the pool); hipFree waits for the device, so handing it to the driver is safe from any thread.
// Returns the bytes released.  A partly used allocation cannot be returned -- which is why pool_take never splits an allocation it makes
// for a request (a base = one buffer's worth) and the pool therefore fragments only inside recycled bases.
static size_t pool_trim_locked(DevPool &P)
{
	size_t freed = 0;
	for (size_t i = 0; i < P.free_pieces.size();) {
		const DevPool::Piece f = P.free_pieces[i];
		DevPool::Base *b = pool_base_of(P, f.p);
		if (b && f.p == b->p && f.cap == b->bytes) {
			(void)hipFree(b->p);
			freed += b->bytes;
			*b = P.bases.back(); P.bases.pop_back();
			P.free_pieces[i] = P.free_pieces.back(); P.free_pieces.pop_back();
		} else ++i;
	}
	return freed;
}
static size_t pool_trim(mahip_ctx *c) { std::lock_guard<std::mutex> g(c->pool.mu); return pool_trim_locked(c->pool); }
static void pool_trim_siblings(mahip_ctx *c)
{
	std::lock_guard<std::mutex> g(g_reg_mu);
	for (mahip_ctx *o : g_ctxs) if (o != c && o->dev == c->dev) (void)pool_trim(o);
}
static size_t pool_idle_bytes(mahip_ctx *c)
{
	std::lock_guard<std::mutex> g(c->pool.mu);
	size_t n = 0;
	for (const DevPool::Piece &f : c->pool.free_pieces) n += f.cap;
	return n;
}

static int pool_take(mahip_ctx *c, size_t want, void **out, size_t *cap)
{
	DevPool &P = c->pool;
	{
		std::lock_guard<std::mutex> g(P.mu);
		size_t best = (size_t)-1;
		for (size_t i = 0; i < P.free_pieces.size(); ++i)
			if (P.free_pieces[i].cap >= want && (best == (size_t)-1 || P.free_pieces[i].cap < P.free_pieces[best].cap)) best = i;
		if (best != (size_t)-1) {
			DevPool::Piece &f = P.free_pieces[best];
			*out = f.p;
			if (f.cap - want >= ((size_t)1 << 20)) { *cap = want; f.p += want; f.cap -= want; }
			else { *cap = f.cap; P.free_pieces[best] = P.free_pieces.back(); P.free_pieces.pop_back(); }
			return 0;
		}
	}
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, want);
	if (e != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(c->st); (void)pool_trim(c); e = hipMalloc(&p, want); }   // what this context holds idle
	if (e != hipSuccess) { (void)hipGetLastError(); pool_trim_siblings(c); e = hipMalloc(&p, want); }                                    // what the process's other contexts on this GPU hold idle
	if (e != hipSuccess) { (void)hipGetLastError(); mahip_set_error("hipMalloc(%zu bytes) failed: %s (%zu bytes in use by this context)", want, hipGetErrorString(e), c->mem_bytes); return -1; }
	std::lock_guard<std::mutex> g(P.mu);
	P.bases.push_back({ (char*)p, want });
	*out = p; *cap = want;
	return 0;
}

void dev_pool_destroy(mahip_ctx *c)
{
	std::lock_guard<std::mutex> g(c->pool.mu);
	for (DevPool::Base &b : c->pool.bases) (void)hipFree(b.p);
	c->pool.bases.clear(); c->pool.free_pieces.clear();
}

int dev_reserve(mahip_ctx *c, DevBuf &b, size_t bytes)
{
	if (bytes <= b.cap) return 0;
	if (b.p) { HIPCHK(hipStreamSynchronize(c->st)); dev_free(c, b); }
	size_t want = (bytes + 255) & ~(size_t)255;
	if (pool_on()) {
		if (pool_take(c, want, &b.p, &b.cap) != 0) { b.p = nullptr; b.cap = 0; return -1; }
	} else {
		hipError_t e = hipMalloc(&b.p, want);
		if (e != hipSuccess) { b.p = nullptr; mahip_set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e)); return -1; }
		b.cap = want;
	}
	c->mem_bytes += b.cap;
	return 0;
}

void dev_free(mahip_ctx *c, DevBuf &b)
{
	if (b.p) {
		c->mem_bytes -= b.cap;
		if (pool_on()) {
			// a buffer whose address left the library may be in use on a stream this context knows nothing about (collectives run by the
			// caller): stream order does not cover it, so it changes hands only after the device is idle -- what hipFree did implicitly
			if (b.ext) (void)hipDeviceSynchronize();
			pool_give_back(c, (char*)b.p, b.cap);
		} else (void)hipFree(b.p);
	}
	b.p = nullptr; b.cap = 0; b.ext = false;
}

// give the idle part of the pool back to the driver (phase boundaries of a run that shares its GPU with other processes; callers that are done for now)
extern "C" int mahip_mem_trim(mahip_ctx_t *c, size_t *released)
{
	HIPCHK(hipSetDevice(c->dev));
	HIPCHK(hipStreamSynchronize(c->st));
	const size_t n = pool_trim(c);
	if (released) *released = n;
	return 0;
}
extern "C" size_t mahip_mem_pool_bytes(mahip_ctx_t *c) { return pool_idle_bytes(c); }

int ctr_zero(mahip_ctx *c)
{
	HIPCHK(hipMemsetAsync(c->ctr.p, 0, CT_STICKY * 8, c->st));
	return 0;
}

// The counters come back through a mailbox: a one-wave kernel behind the pass copies the 64 words into host-coherent pinned memory and then
// raises a sequence number there; the host spins on that word.  A copy + hipStreamSynchronize costs a blit launch plus the runtime's wake-up
// (20-30 us of idle GPU per fetch in the round-3 kernel trace, about 50 fetches per input: 1.3 ms of a 21 ms pass at cfg4, 40 % of one at cfg2);
// the spin sees the word a microsecond or two after the kernel wrote it.  h_ctr[64] = the sequence word.  Measured (round 3, profiles/
// r03_experiments.txt): 19.47 -> 19.02 ms per pass at cfg4, 2.81 -> 2.74 at cfg2; making the runtime's own waits spin (hipDeviceScheduleSpin) instead: nothing.
__global__ __launch_bounds__(64) void k_ctr_publish(const unsigned long long *__restrict__ ctr, volatile unsigned long long *h, unsigned long long seq)
{
	h[threadIdx.x] = ctr[threadIdx.x];
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x == 0) { h[64] = seq; __threadfence_system(); }
}

static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
	__builtin_ia32_pause();
#elif defined(__aarch64__)
	__asm__ __volatile__("yield");
#endif
}

int ctr_fetch(mahip_ctx *c)
{
	const unsigned long long seq = ++c->ctr_seq;
	hipLaunchKernelGGL(k_ctr_publish, dim3(1), dim3(64), 0, c->st, (const unsigned long long*)c->ctr.p, (volatile unsigned long long*)c->h_ctr, seq);
	{ hipError_t le = hipGetLastError(); if (le != hipSuccess) { mahip_set_error("ctr_fetch: launch failed: %s", hipGetErrorString(le)); return -1; } } // (a launch that never ran would leave the spin below waiting for the stream query)
	volatile unsigned long long *flag = (volatile unsigned long long*)c->h_ctr + 64;
	for (unsigned long spins = 0;; ++spins) {
		if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
		cpu_relax();
		if ((spins & 0xffff) == 0xffff) { // every few hundred microseconds: has the stream ended without the word (a fault)?
			hipError_t e = hipStreamQuery(c->st);
			if (e == hipSuccess) { if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0; HIPCHK(hipStreamSynchronize(c->st)); if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0; mahip_set_error("ctr_fetch: the stream is idle but the counters never arrived"); return -1; }
			if (e != hipErrorNotReady) { mahip_set_error("ctr_fetch: %s", hipGetErrorString(e)); return -1; }
		}
	}
}

// Keep the process on the NUMA node the GPU hangs off: staging copies (page cache -> pinned slots -> DMA) and the driver's own
// page-table work run at half speed from the far socket of a two-socket host (measured: 35 vs 18 GB/s file -> HBM).  The calling
// thread's mask is narrowed once; threads created later inherit it.  MA_NO_NUMA_PIN=1 leaves the affinity alone.
#include <sched.h>
#include <ctype.h>
#include <dirent.h>
static void pin_to_gpu_node(int device)
{
	static int done = 0;
	if (done || getenv("MA_NO_NUMA_PIN")) return;
	done = 1;
	char bdf[64], path[256], buf[4096];
	if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return;
	for (char *p = bdf; *p; ++p) *p = (char)tolower((unsigned char)*p);
	snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
	FILE *f = fopen(path, "r");
	int node = -1;
	if (!f) return;
	if (fscanf(f, "%d", &node) != 1) node = -1;
	fclose(f);
	if (node < 0) return;
	snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
	f = fopen(path, "r");
	if (!f) return;
	if (!fgets(buf, sizeof(buf), f)) { fclose(f); return; }
	fclose(f);
	cpu_set_t want, have;
	CPU_ZERO(&want);
	int n_set = 0;
	for (char *p = buf; *p;) { // "0-63,128-191"
		char *e;
		long a = strtol(p, &e, 10), b = a;
		if (e == p) break;
		if (*e == '-') b = strtol(e + 1, &e, 10);
		for (long k = a; k <= b && k < CPU_SETSIZE; ++k) { CPU_SET((int)k, &want); ++n_set; }
		p = *e == ',' ? e + 1 : e;
		if (*e != ',' ) break;
	}
	if (n_set == 0 || sched_getaffinity(0, sizeof(have), &have) != 0) return;
	CPU_AND(&want, &want, &have); // never widen what the launcher allowed
	if (CPU_COUNT(&want) == 0) return;
	DIR *dp = opendir("/proc/self/task"); // every thread that exists now (the context may be created on a helper thread); later ones inherit
	if (!dp) { (void)sched_setaffinity(0, sizeof(want), &want); return; }
	for (struct dirent *de; (de = readdir(dp)) != nullptr;) {
		const long tid = atol(de->d_name);
		if (tid > 0) (void)sched_setaffinity((pid_t)tid, sizeof(want), &want);
	}
	closedir(dp);
}

extern "C" mahip_ctx_t *mahip_create(int device, void *stream)
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		mahip_set_error("mahip_create: no HIP device available (%s); this library has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
		return nullptr;
	}
	if (device < 0 || device >= n) { mahip_set_error("mahip_create: device %d out of range (0..%d)", device, n - 1); return nullptr; }
	if (hipSetDevice(device) != hipSuccess) { mahip_set_error("mahip_create: hipSetDevice(%d) failed", device); return nullptr; }
	pin_to_gpu_node(device);
	mahip_ctx *c = new mahip_ctx();
	c->dev = device;
	ctx_register(c);
	if (stream) c->st = (hipStream_t)stream, c->own_stream = false;
	else {
		if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) { mahip_set_error("mahip_create: hipStreamCreate failed"); ctx_unregister(c); delete c; return nullptr; }
		c->own_stream = true;
	}
	if (dev_reserve(c, c->ctr, CT_ALLOC_WORDS * 8) != 0) { ctx_unregister(c); delete c; return nullptr; }
	if (hipHostMalloc((void**)&c->h_ctr, 72 * 8, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { mahip_set_error("mahip_create: hipHostMalloc failed"); ctx_unregister(c); delete c; return nullptr; }
	memset(c->h_ctr, 0, 72 * 8);
	if (hipMemsetAsync(c->ctr.p, 0, CT_ALLOC_WORDS * 8, c->st) != hipSuccess) { mahip_set_error("mahip_create: memset failed"); ctx_unregister(c); delete c; return nullptr; }
	{ const char *s = getenv("MA_EXACT_TIES"); c->tie_mode = s == 0 || *s == 0 ? 2 : atoi(s) != 0 ? 1 : 0; } // unset: automatic
	return c;
}

extern "C" void mahip_destroy(mahip_ctx_t *c)
{
	if (!c) return;
	ctx_unregister(c); // nobody trims a pool that is being taken down
	(void)hipSetDevice(c->dev);
	(void)hipStreamSynchronize(c->st);
	for (auto &e : c->pev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
	for (hipEvent_t e : c->mark_ev) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->sub_ev) if (e) (void)hipEventDestroy(e);
	for (hipStream_t t : c->sub_side) if (t) (void)hipStreamDestroy(t);
	DevBuf *all[] = { &c->aos_own, &c->goff, &c->sub[0], &c->sub[1], &c->r_cont, &c->r_used, &c->r_del, &c->r_live, &c->map, &c->surv,
		&c->au[0], &c->au[1], &c->av[0], &c->av[1], &c->alen[0], &c->alen[1], &c->aol[0], &c->aol[1], &c->idx, &c->sdel, &c->slen,
		&c->keep, &c->pos, &c->gs_tmp, &c->key[0], &c->key[1], &c->val[0], &c->val[1], &c->hist, &c->scan_tmp[0], &c->scan_tmp[1], &c->scan_tmp[2],
		&c->ctr, &c->ovf, &c->big0, &c->big1, &c->marks, &c->sgmask, &c->sidx, &c->hrank, &c->orank, &c->aslot, &c->apos, &c->pushrows[0], &c->pushrows[1], &c->xb[0], &c->xb[1], &c->gpos, &c->tdig, &c->wantb, &c->wseg };
	for (DevBuf *b : all) dev_free(c, *b);
	for (int k = 0; k < 8; ++k) dev_free(c, c->col[k]);
	c->hwalk.drop(); c->hdig.drop();
	mahip_comm_destroy(c);
	paf_free(c);
	clean_free(c);
	ug_free(c);
	useq_free(c);
	xfer_pool_free(c);
	dev_pool_destroy(c); // after everybody gave its buffers back
	if (c->h_ctr) (void)hipHostFree(c->h_ctr);
	if (c->own_stream) (void)hipStreamDestroy(c->st);
	delete c;
}

// exchange buffer `slot` (0/1) of at least `bytes` bytes (sharded mode: send / receive side of a collective)
extern "C" int mahip_xbuf(mahip_ctx_t *c, int slot, size_t bytes, void **d_ptr)
{
	if (slot < 0 || slot > 1) { mahip_set_error("mahip_xbuf: bad slot"); return -1; }
	HIPCHK(hipSetDevice(c->dev));
	CHK(dev_reserve(c, c->xb[slot], bytes + 256));
	c->xb[slot].ext = true;
	*d_ptr = c->xb[slot].p;
	return 0;
}

// one trivial launch + wait: the first launch of a process loads the library's code object onto the device; callers that time their start-up make that cost visible with this
extern "C" int mahip_first_launch(mahip_ctx_t *c)
{
	HIPCHK(hipSetDevice(c->dev));
	return ctr_fetch(c);
}

extern "C" int mahip_sync(mahip_ctx_t *c)
{
	HIPCHK(hipSetDevice(c->dev));
	HIPCHK(hipStreamSynchronize(c->st));
	return 0;
}

extern "C" size_t mahip_mem_bytes(mahip_ctx_t *c) { return c->mem_bytes + pool_idle_bytes(c); } // in use + kept idle by the pool: what the driver sees

extern "C" void *mahip_devptr(mahip_ctx_t *c, int which, size_t *bytes)
{
	DevBuf *b = which == MAHIP_PTR_SUB0 ? &c->sub[0] : which == MAHIP_PTR_SUB1 ? &c->sub[1] : which == MAHIP_PTR_RDFLAG ? &c->r_del : nullptr;
	if (!b) return nullptr;
	b->ext = true;
	if (bytes) *bytes = which == MAHIP_PTR_RDFLAG ? (size_t)c->n_seq : (size_t)c->n_seq * 8;
	return b->p;
}

// ---- profiling: an event pair per instrumented launch, resolved lazily ----
void prof_begin(mahip_ctx *c, const char *name, double alg_bytes)
{
	ProfEvent e;
	e.name = name; e.alg_bytes = alg_bytes;
	if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
	(void)hipEventRecord(e.a, c->st);
	c->pstack.push_back(c->pev.size());
	c->pev.push_back(e);
}

void prof_end(mahip_ctx *c)
{
	if (c->pstack.empty()) return; // scopes may nest (a compaction contains a scan)
	size_t k = c->pstack.back();
	c->pstack.pop_back();
	(void)hipEventRecord(c->pev[k].b, c->st);
}

// correct the algorithmic byte count of the most recent launch recorded under `name` (when the number of units a
// kernel really processed is only known after a later counter fetch)
void prof_patch_last(mahip_ctx *c, const char *name, double alg_bytes)
{
	for (size_t k = c->pev.size(); k-- > 0;)
		if (strcmp(c->pev[k].name, name) == 0) { c->pev[k].alg_bytes = alg_bytes; return; }
}

int prof_collect(mahip_ctx *c)
{
	HIPCHK(hipStreamSynchronize(c->st));
	for (auto &e : c->pev) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) ms = 0;
		size_t k;
		for (k = 0; k < c->pacc.size(); ++k) if (strcmp(c->pacc[k].name, e.name) == 0) break;
		if (k == c->pacc.size()) { ProfAcc a; a.name = e.name; a.launches = 0; a.ms = 0; a.alg_bytes = 0; c->pacc.push_back(a); }
		c->pacc[k].launches += 1; c->pacc[k].ms += ms; c->pacc[k].alg_bytes += e.alg_bytes;
		(void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
	}
	c->pev.clear();
	return 0;
}

extern "C" int mahip_mark(mahip_ctx_t *c, int slot)
{
	if (slot < 0 || slot >= 64) { mahip_set_error("mahip_mark: bad slot"); return -1; }
	HIPCHK(hipSetDevice(c->dev));
	if (!c->mark_ev[slot]) HIPCHK(hipEventCreate(&c->mark_ev[slot]));
	HIPCHK(hipEventRecord(c->mark_ev[slot], c->st));
	c->mark_set |= 1ull << slot;
	return 0;
}

extern "C" int mahip_marks_ms(mahip_ctx_t *c, int first, int n, float *ms)
{
	for (int i = 0; i < n; ++i) {
		const int a = first + i, b = a + 1;
		ms[i] = 0;
		if (a < 0 || b >= 64 || !(c->mark_set >> a & 1) || !(c->mark_set >> b & 1)) continue;
		if (hipEventElapsedTime(&ms[i], c->mark_ev[a], c->mark_ev[b]) != hipSuccess) ms[i] = 0;
	}
	c->mark_set = 0;
	return 0;
}

extern "C" int mahip_prof_enable(mahip_ctx_t *c, int enable) { c->prof = enable != 0; return 0; }

extern "C" int mahip_prof_reset(mahip_ctx_t *c)
{
	CHK(prof_collect(c));
	c->pacc.clear();
	return 0;
}

extern "C" int mahip_prof_get(mahip_ctx_t *c, mahip_prof_t *out, int max)
{
	if (prof_collect(c) != 0) return -1;
	int n = (int)c->pacc.size();
	for (int i = 0; i < n && i < max; ++i) {
		out[i].name = c->pacc[i].name; out[i].launches = c->pacc[i].launches;
		out[i].total_ms = c->pacc[i].ms; out[i].alg_bytes = c->pacc[i].alg_bytes;
	}
	return n;
}
