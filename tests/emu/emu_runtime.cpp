// tests/emu/emu_runtime.cpp -- fiber scheduler and host API of the CPU stand-in for the HIP runtime.
// TEST INFRASTRUCTURE ONLY (see include/hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include <errno.h>
#include <time.h>
#include <unistd.h>
#include <sys/mman.h>
#include <pthread.h>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <vector>
#include <map>
#include <string>
#include <execinfo.h>

namespace emu {

thread_local Fiber *t_fiber = nullptr;
thread_local Block *t_block = nullptr;

enum { F_RUNNABLE = 0, F_WAVE, F_BLOCK, F_DONE };

void fatal(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	fprintf(stderr, "[hip-emu] fatal: ");
	vfprintf(stderr, fmt, ap);
	fprintf(stderr, "\n");
	va_end(ap);
	abort();
}

// ---- context switch (x86-64 SysV: callee-saved registers + stack pointer) ----
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
	.text
	.globl emu_switch
	.type emu_switch,@function
emu_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq %rsi, %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
	.size emu_switch,.-emu_switch
)");

static size_t g_stack_bytes = 64 << 10;
static const uint64_t CANARY = 0x5afe57ac6b1d0ull;

struct Worker {
	char *stacks = nullptr; // 1024 fiber stacks
	size_t stack_bytes = 0;
	void *sched_sp = nullptr;
	Fiber fibers[1024];
	Wave waves[16];
	LaunchFn fn;
};
static thread_local Worker *t_worker = nullptr;

static Worker *worker()
{
	if (!t_worker) {
		Worker *w = new Worker;
		w->stack_bytes = g_stack_bytes;
		w->stacks = (char *)mmap(nullptr, w->stack_bytes * 1024, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (w->stacks == MAP_FAILED) fatal("mmap of the fiber stacks failed");
		t_worker = w;
	}
	return t_worker;
}

static inline void to_scheduler()
{
	Worker *w = t_worker;
	emu_switch(&t_fiber->sp, w->sched_sp);
}

static void fiber_entry()
{
	Worker *w = t_worker;
	w->fn.call(w->fn.arg);
	t_fiber->state = F_DONE;
	to_scheduler();
	fatal("a finished fiber was resumed");
}

static inline void resume(Worker *w, Fiber *f)
{
	t_fiber = f;
	emu_switch(&w->sched_sp, f->sp);
}

Wave *wave_op(uint64_t val, unsigned op, uintptr_t site)
{
	Fiber *f = t_fiber;
	f->val = val;
	f->site = site;
	f->op = op;
	f->state = F_WAVE;
	to_scheduler();
	return &t_worker->waves[f->wave];
}

void block_barrier()
{
	t_fiber->state = F_BLOCK;
	to_scheduler();
}

static std::atomic<uint64_t> g_divergent_ops{0}, g_wave_ops{0}, g_launches{0};
}
static std::atomic<uint64_t> g_syncs{0}, g_d2h{0}, g_h2d{0};
namespace emu {
static int g_verbose = 0;
static int g_reverse = 0; // EMU_ORDER=reverse: lanes, waves and blocks run in descending order (shakes out code that leans on lock-step or launch order)
// EMU_VERBOSE=1: per-kernel table at exit -- launches, blocks, cross-lane operations, and how many of those were resolved for only part of
// the wave's live lanes (divergent call sites: the emulator orders those by code address, which is a heuristic -- audit them)
struct KStat { uint64_t launches = 0, blocks = 0, ops = 0, div = 0; };
static std::mutex g_kstat_mu;
static std::map<std::string, KStat> g_kstat;
static thread_local uint64_t t_ops = 0, t_div = 0;
static void kstat_dump()
{
	fprintf(stderr, "[hip-emu] %llu stream syncs, %llu D2H and %llu H2D copies\n", (unsigned long long)g_syncs.load(), (unsigned long long)g_d2h.load(), (unsigned long long)g_h2d.load());
	fprintf(stderr, "[hip-emu] %-56s %9s %10s %12s %10s\n", "kernel", "launches", "blocks", "wave-ops", "partial");
	for (auto &kv : g_kstat)
		fprintf(stderr, "[hip-emu] %-56.56s %9llu %10llu %12llu %10llu\n", kv.first.c_str(), (unsigned long long)kv.second.launches,
			(unsigned long long)kv.second.blocks, (unsigned long long)kv.second.ops, (unsigned long long)kv.second.div);
}

static void run_block(Worker *w, dim3 bid, dim3 bdim, dim3 gdim)
{
	const unsigned n = bdim.x * bdim.y * bdim.z, nw = (n + 63) / 64;
	Block blk;
	blk.bid = bid, blk.bdim = bdim, blk.gdim = gdim, blk.waves = w->waves;
	t_block = &blk;
	for (unsigned i = 0; i < n; ++i) {
		Fiber *f = &w->fibers[i];
		char *lo = w->stacks + (size_t)i * w->stack_bytes;
		*(uint64_t *)lo = CANARY;
		uint64_t *top = (uint64_t *)(lo + w->stack_bytes);
		top[-1] = 0;
		top[-2] = (uint64_t)(uintptr_t)&fiber_entry;
		for (int k = 3; k <= 8; ++k) top[-k] = 0;
		f->sp = (void *)(top - 8);
		f->state = F_RUNNABLE;
		f->lane = i & 63, f->wave = i >> 6;
		f->tid.x = i % bdim.x, f->tid.y = i / bdim.x % bdim.y, f->tid.z = i / (bdim.x * bdim.y);
	}
	uint64_t n_wave_ops = 0, n_div = 0;
	for (;;) {
		unsigned done = 0;
		for (unsigned wi = 0; wi < nw; ++wi) {
			const unsigned wv = g_reverse ? nw - 1 - wi : wi;
			Fiber *fb = &w->fibers[wv * 64];
			const unsigned nl = (wv + 1) * 64 <= n ? 64 : n - wv * 64;
			for (unsigned li = 0; li < nl; ++li) {
				const unsigned l = g_reverse ? nl - 1 - li : li;
				if (fb[l].state == F_RUNNABLE) resume(w, &fb[l]);
			}
			for (;;) { // resolve the wave's pending cross-lane operations: the earliest call site first
				uintptr_t site = 0;
				unsigned op = 0, waiting = 0, live = 0;
				for (unsigned l = 0; l < nl; ++l) {
					if (fb[l].state != F_DONE) ++live;
					if (fb[l].state == F_WAVE) {
						++waiting;
						if (site == 0 || fb[l].site < site) site = fb[l].site, op = fb[l].op;
					}
				}
				if (!waiting) break;
				Wave *W = &w->waves[wv];
				uint64_t mask = 0;
				for (unsigned l = 0; l < nl; ++l)
					if (fb[l].state == F_WAVE && fb[l].op == op) mask |= 1ull << l, W->slot[l] = fb[l].val;
				W->mask = mask;
				++n_wave_ops;
				if ((unsigned)__builtin_popcountll(mask) != live) ++n_div;
				for (unsigned l = 0; l < nl; ++l)
					if (mask >> l & 1) fb[l].state = F_RUNNABLE;
				for (unsigned li = 0; li < nl; ++li) {
					const unsigned l = g_reverse ? nl - 1 - li : li;
					if (mask >> l & 1) resume(w, &fb[l]);
				}
			}
			for (unsigned l = 0; l < nl; ++l)
				if (fb[l].state == F_DONE) ++done;
		}
		if (done == n) break;
		for (unsigned i = 0; i < n; ++i)
			if (w->fibers[i].state == F_BLOCK) w->fibers[i].state = F_RUNNABLE;
	}
	for (unsigned i = 0; i < n; ++i)
		if (*(uint64_t *)(w->stacks + (size_t)i * w->stack_bytes) != CANARY) fatal("fiber stack overflow (raise EMU_STACK_KB)");
	g_wave_ops += n_wave_ops, g_divergent_ops += n_div;
	t_ops += n_wave_ops, t_div += n_div;
	t_block = nullptr, t_fiber = nullptr;
}

// ---- block distribution over OS threads ----
struct Job {
	dim3 grid, block;
	LaunchFn fn;
	std::atomic<uint64_t> next{0}, ops{0}, div{0};
	uint64_t total = 0;
};

static void run_job(Job *j)
{
	Worker *w = worker();
	w->fn = j->fn;
	t_ops = t_div = 0;
	const uint64_t chunk = j->total > 4096 ? 16 : 1;
	for (;;) {
		uint64_t b0 = j->next.fetch_add(chunk);
		if (b0 >= j->total) break;
		uint64_t b1 = b0 + chunk < j->total ? b0 + chunk : j->total;
		for (uint64_t bi = b0; bi < b1; ++bi) {
			const uint64_t b = g_reverse ? j->total - 1 - bi : bi;
			dim3 bid((unsigned)(b % j->grid.x), (unsigned)(b / j->grid.x % j->grid.y), (unsigned)(b / ((uint64_t)j->grid.x * j->grid.y)));
			run_block(w, bid, j->block, j->grid);
		}
	}
	j->ops += t_ops, j->div += t_div;
}

struct Pool {
	std::mutex mu, launch_mu;
	std::condition_variable cv_go, cv_done;
	std::vector<std::thread> th;
	Job *job = nullptr;
	uint64_t gen = 0;
	int busy = 0;
	bool stop = false;
	int nthreads = 1;
	Pool()
	{
		const char *e = getenv("EMU_THREADS");
		long nc = sysconf(_SC_NPROCESSORS_ONLN);
		nthreads = e ? atoi(e) : (int)(nc > 8 ? 8 : nc);
		if (nthreads < 1) nthreads = 1;
		if ((e = getenv("EMU_STACK_KB")) != nullptr && atoi(e) >= 16) g_stack_bytes = (size_t)atoi(e) << 10;
		if ((e = getenv("EMU_VERBOSE")) != nullptr) g_verbose = atoi(e);
		if ((e = getenv("EMU_ORDER")) != nullptr) g_reverse = !strcmp(e, "reverse");
	}
	void start()
	{
		for (int i = 1; i < nthreads; ++i)
			th.emplace_back([this] {
				uint64_t seen = 0;
				for (;;) {
					Job *j;
					{
						std::unique_lock<std::mutex> lk(mu);
						cv_go.wait(lk, [&] { return stop || gen != seen; });
						if (stop) return;
						seen = gen, j = job;
					}
					run_job(j);
					{
						std::lock_guard<std::mutex> lk(mu);
						if (--busy == 0) cv_done.notify_all();
					}
				}
			});
	}
	~Pool()
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			stop = true;
		}
		cv_go.notify_all();
		for (auto &t : th) t.join();
	}
};
static Pool *g_pool = nullptr;
static std::once_flag g_pool_once;
static pid_t g_pool_pid = 0;

static Pool *pool()
{
	// a forked child (MA_GPUS=N forks its ranks) must not inherit a pool whose threads do not exist in it
	if (g_pool && g_pool_pid != getpid()) g_pool = nullptr;
	if (!g_pool) {
		g_pool = new Pool;
		g_pool_pid = getpid();
		g_pool->start();
	}
	return g_pool;
}

void launch(const char *name, dim3 grid, dim3 block, size_t shmem, LaunchFn fn)
{
	(void)shmem;
	const unsigned n = block.x * block.y * block.z;
	if (n == 0 || n > 1024) fatal("block of %u threads", n);
	if (shmem > (160u << 10)) fatal("dynamic LDS of %zu bytes", shmem);
	Pool *p = pool();
	std::lock_guard<std::mutex> serial(p->launch_mu);
	Job j;
	j.grid = grid, j.block = block, j.fn = fn;
	j.total = (uint64_t)grid.x * grid.y * grid.z;
	++g_launches;
	if (j.total == 0) return;
	if (p->nthreads > 1 && j.total > 1) {
		{
			std::lock_guard<std::mutex> lk(p->mu);
			p->job = &j, p->busy = p->nthreads - 1, ++p->gen;
		}
		p->cv_go.notify_all();
		run_job(&j);
		std::unique_lock<std::mutex> lk(p->mu);
		p->cv_done.wait(lk, [&] { return p->busy == 0; });
	} else run_job(&j);
	if (g_verbose) {
		std::lock_guard<std::mutex> lk(g_kstat_mu);
		if (g_kstat.empty()) atexit(kstat_dump);
		KStat &k = g_kstat[name];
		k.launches++, k.blocks += j.total, k.ops += j.ops, k.div += j.div;
	}
}

// ---- cross-lane operations that are not templates ----
uint64_t ballot(unsigned op, int p)
{
	Wave *w = wave_op((uint64_t)(p != 0), op, EMU_SITE);
	uint64_t m = 0;
	for (unsigned l = 0; l < 64; ++l)
		if ((w->mask >> l & 1) && w->slot[l]) m |= 1ull << l;
	return m;
}

uint32_t readfirstlane(unsigned op, uint32_t v)
{
	Wave *w = wave_op(v, op, EMU_SITE);
	return (uint32_t)w->slot[__builtin_ctzll(w->mask)];
}

void wave_barrier(unsigned op) { wave_op(0, op, EMU_SITE); }

int ds_bpermute(unsigned op, int addr, int v)
{
	Wave *w = wave_op((uint32_t)v, op, EMU_SITE);
	const unsigned s = ((unsigned)addr >> 2) & 63u;
	return (w->mask >> s & 1) ? (int)(uint32_t)w->slot[s] : 0;
}

int ds_permute(unsigned op, int addr, int v)
{ // every participant deposits (destination, value); a lane collects what was sent to it
	Wave *w = wave_op((uint64_t)(((unsigned)addr >> 2) & 63u) << 32 | (uint32_t)v, op, EMU_SITE);
	const unsigned me = t_fiber->lane;
	int r = 0;
	for (unsigned l = 0; l < 64; ++l)
		if ((w->mask >> l & 1) && (unsigned)(w->slot[l] >> 32) == me) r = (int)(uint32_t)w->slot[l];
	return r;
}

int readlane(unsigned op, int v, int lane)
{
	Wave *w = wave_op((uint32_t)v, op, EMU_SITE);
	return (int)(uint32_t)w->slot[(unsigned)lane & 63u];
}

static inline int dpp_resolve(Wave *w, unsigned lane, int old, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
	const unsigned row = lane >> 4, col = lane & 15, bank = col >> 2;
	if (!(row_mask >> row & 1) || !(bank_mask >> bank & 1)) return old;
	int s = -1; // source lane, -1 = out of range
	if (ctrl >= 0 && ctrl <= 0xff) s = (int)((lane & ~3u) | ((unsigned)ctrl >> (2 * (lane & 3)) & 3u));
	else if (ctrl >= 0x101 && ctrl <= 0x10f) { unsigned k = ctrl & 15; s = col + k < 16 ? (int)(lane + k) : -1; }          // row_shl
	else if (ctrl >= 0x111 && ctrl <= 0x11f) { unsigned k = ctrl & 15; s = col >= k ? (int)(lane - k) : -1; }              // row_shr
	else if (ctrl >= 0x121 && ctrl <= 0x12f) { unsigned k = ctrl & 15; s = (int)((lane & ~15u) | ((col + 16 - k) & 15)); } // row_ror
	else if (ctrl == 0x130) s = lane + 1 < 64 ? (int)lane + 1 : -1;  // wave_shl:1
	else if (ctrl == 0x134) s = (int)((lane + 1) & 63);              // wave_rol:1
	else if (ctrl == 0x138) s = lane >= 1 ? (int)lane - 1 : -1;      // wave_shr:1
	else if (ctrl == 0x13c) s = (int)((lane + 63) & 63);             // wave_ror:1
	else if (ctrl == 0x140) s = (int)((lane & ~15u) | (15 - col));   // row_mirror
	else if (ctrl == 0x141) s = (int)((lane & ~7u) | (7 - (lane & 7))); // row_half_mirror
	else if (ctrl == 0x142) s = row == 0 ? -1 : (int)(row * 16 - 1);                                  // row_bcast:15: every lane of a row reads lane 15 of the row before it
	else if (ctrl == 0x143) s = lane < 32 ? -1 : 31;                                                  // row_bcast:31: rows 2 and 3 read lane 31
	else fatal("DPP control 0x%x is not modelled", ctrl);
	if (s < 0 || !(w->mask >> s & 1)) return bound_ctrl ? 0 : old;
	return (int)(uint32_t)w->slot[s];
}

int update_dpp(unsigned op, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
	Wave *w = wave_op((uint32_t)src, op, EMU_SITE);
	return dpp_resolve(w, t_fiber->lane, old, ctrl, row_mask, bank_mask, bound_ctrl);
}

int mov_dpp(unsigned op, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
	Wave *w = wave_op((uint32_t)src, op, EMU_SITE);
	return dpp_resolve(w, t_fiber->lane, 0, ctrl, row_mask, bank_mask, bound_ctrl);
}

// ---- device memory ----
struct Alloc { void *map; size_t map_bytes; size_t bytes; };
static std::mutex g_mem_mu;
static std::map<void *, Alloc> g_allocs;
static int g_guard = -1;

static void *dev_alloc(size_t bytes)
{
	if (g_guard < 0) { const char *e = getenv("EMU_GUARD"); g_guard = e ? atoi(e) : 0; }
	if (bytes == 0) bytes = 1;
	Alloc a;
	void *p;
	if (g_guard) { // the allocation ends (to 16 bytes) at a page that faults
		size_t pg = (size_t)sysconf(_SC_PAGESIZE), need = (bytes + 15) & ~(size_t)15, body = (need + pg - 1) / pg * pg;
		a.map_bytes = body + 2 * pg;
		a.map = mmap(nullptr, a.map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (a.map == MAP_FAILED) return nullptr;
		mprotect(a.map, pg, PROT_NONE);
		mprotect((char *)a.map + pg + body, pg, PROT_NONE);
		p = (char *)a.map + pg + body - need;
		memset(p, 0xA5, need);
	} else {
		a.map_bytes = 0;
		if (posix_memalign(&a.map, 256, bytes) != 0) return nullptr;
		p = a.map;
		memset(p, 0xA5, bytes < (64u << 20) ? bytes : (64u << 20)); // fresh device memory is not zero
	}
	a.bytes = bytes;
	std::lock_guard<std::mutex> lk(g_mem_mu);
	g_allocs[p] = a;
	return p;
}

static int dev_release(void *p)
{
	if (!p) return 0;
	Alloc a;
	{
		std::lock_guard<std::mutex> lk(g_mem_mu);
		auto it = g_allocs.find(p);
		if (it == g_allocs.end()) return -1;
		a = it->second;
		g_allocs.erase(it);
	}
	if (a.map_bytes) munmap(a.map, a.map_bytes);
	else free(a.map);
	return 0;
}

} // namespace emu

// ---- host API ----
struct emu_stream { int flags; };
struct emu_event { double t; };

static double now_ms()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

extern "C" {
hipError_t hipGetDeviceCount(int *n)
{
	const char *e = getenv("EMU_DEVICES");
	*n = e ? atoi(e) : 1;
	return hipSuccess;
}
hipError_t hipSetDevice(int dev) { (void)dev; return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int dev) { snprintf(buf, (size_t)len, "0000:e%d:00.0", dev); return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory (emu)" : "hipErrorInvalidValue (emu)"; }
hipError_t hipMalloc(void **p, size_t bytes) { *p = emu::dev_alloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { return emu::dev_release(p) == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags) { (void)flags; *p = emu::dev_alloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { return emu::dev_release(p) == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) { (void)kind; if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st) { (void)st; if (kind == hipMemcpyDeviceToHost) ++g_d2h; else if (kind == hipMemcpyHostToDevice) ++g_h2d; if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemset(void *dst, int v, size_t bytes) { if (bytes) memset(dst, v, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t st) { (void)st; if (bytes) memset(dst, v, bytes); return hipSuccess; }
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int v, size_t count, hipStream_t st) { (void)st; for (size_t i = 0; i < count; ++i) ((int*)dst)[i] = v; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *st) { *st = new emu_stream{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned flags) { *st = new emu_stream{(int)flags}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t st) { delete st; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t st)
{
	(void)st;
	++g_syncs;
	if (emu::g_verbose >= 2) { // who waits for the device: the callers of this sync (build with -rdynamic for names)
		void *bt[6];
		int n = backtrace(bt, 6);
		char **sym = backtrace_symbols(bt, n);
		fprintf(stderr, "[hip-emu] sync:");
		for (int i = 1; i < n && i < 5; ++i) { const char *p = strrchr(sym[i], '/'); fprintf(stderr, " <- %s", p ? p + 1 : sym[i]); }
		fprintf(stderr, "\n");
		free(sym);
	}
	return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags) { (void)flags; *e = new emu_event{0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) { (void)st; e->t = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e; return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

// statistics for the tests: launches, resolved cross-lane operations, and how many of those were issued by only part of
// the live lanes of a wave (divergent call sites are resolved lowest code address first)
void emu_stats(uint64_t *launches, uint64_t *wave_ops, uint64_t *divergent)
{
	*launches = emu::g_launches, *wave_ops = emu::g_wave_ops, *divergent = emu::g_divergent_ops;
}
}
