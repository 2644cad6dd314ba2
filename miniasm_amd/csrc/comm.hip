// comm.hip -- the collectives of the sharded multi-GPU mode, called from C on the context's stream.
//
//   RCCL  : one process per GPU, ncclAllGather / ncclAllReduce over xGMI queued on the context's stream (no host sync between a
//           collective and the kernels around it).  librccl is opened at run time (dlopen): a single-GPU user never loads it, and a
//           process that already carries a copy (PyTorch ships one) is not handed a second one at link time.
//   shm   : the same three operations staged through a POSIX shared-memory segment by the host -- a test double for boxes with
//           ONE GPU (RCCL refuses two ranks on one device): N processes, one context each on the same device, exercise the
//           orchestration code (host/sharded.c) end to end.  Never the production path.
//   ext   : the caller's own transport over HOST buffers (mahip_comm_init_ext): five callbacks.  tests/test_dist_gloo.py runs host/sharded.c over
//           torch.distributed's gloo backend this way -- the product's orchestration on the CPU build of the kernels, no second copy of the sequence.
#include "mahip_internal.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

struct RcclApi {
	void *dl = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)(void) = nullptr;
	ncclResult_t (*GroupEnd)(void) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int rccl_load(void)
{
	if (g_rccl.dl) return 0;
	const char *names[] = { getenv("MA_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	for (const char *n : names) {
		if (!n || !*n) continue;
		g_rccl.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
		if (g_rccl.dl) break;
	}
	if (!g_rccl.dl) { mahip_set_error("cannot open librccl (%s); set MA_RCCL_LIB", dlerror()); return -1; }
#define SYM(field, name) do { *(void**)&g_rccl.field = dlsym(g_rccl.dl, name); if (!g_rccl.field) { mahip_set_error("librccl has no %s", name); g_rccl.dl = nullptr; return -1; } } while (0)
	SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
	SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
	SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
#undef SYM
	return 0;
}

// RCCL writes its version banner and its NCCL_DEBUG lines to STDOUT -- the stream this program's result goes to (the GFA of the command line, the
// JSON line of bench.py).  Found on the GPU box in round 3: the first one-rank run's GFA began with "RCCL version ...".  Its log goes to stderr
// unless the user asked for a file, and whatever the first calls still print is kept away from descriptor 1.
struct StdoutToStderr {
	int saved;
	StdoutToStderr() { setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0); fflush(stdout); saved = dup(1); if (saved >= 0) (void)dup2(2, 1); }
	~StdoutToStderr() { if (saved >= 0) { fflush(stdout); (void)dup2(saved, 1); close(saved); } }
};

#define NCCLCHK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { mahip_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r_)); return -1; } } while (0)

// ---- shared-memory test double ----
struct ShmHeader { volatile unsigned arrive, sense; unsigned world; size_t slot; volatile unsigned turn_lock; };
// per rank, sparse: pages exist only where a collective wrote.  MA_SHM_SLOT_LOG2 (tests): a smaller slot, so that small inputs take the chunked rounds of the personalised exchange
static size_t shm_slot_bytes() { static size_t v = 0; if (!v) { const char *e = getenv("MA_SHM_SLOT_LOG2"); int l2 = e ? atoi(e) : 30; if (l2 < 12) l2 = 12; if (l2 > 34) l2 = 34; v = (size_t)1 << l2; } return v; }
#define SHM_SLOT_BYTES shm_slot_bytes()

struct Comm {
	int kind = 0, rank = 0, world = 1; // kind 1 = RCCL, 2 = shm, 3 = the caller's own transport (host buffers: mahip_comm_init_ext)
	mahip_comm_ext_t ext = {};
	std::vector<char> hbuf[2]; // kind 3: host staging
	ncclComm_t nccl = nullptr;
	ShmHeader *hdr = nullptr;
	char *slots = nullptr;
	size_t map_bytes = 0;
	unsigned my_sense = 0;
	char name[128] = "";
};

// MA_SHM_SERIAL=1 (tools/shard_projection.py): the ranks of a shared-memory run take turns on the one GPU they share -- a rank holds the lock
// while it computes and gives it up for the length of a collective -- so that the device time of a rank's phases (mahip_mark) is what the rank
// would see on a GPU of its own.
static bool shm_serial() { static int v = -1; if (v < 0) { const char *e = getenv("MA_SHM_SERIAL"); v = e && atoi(e) != 0; } return v != 0; }
static void shm_turn_take(Comm *m) { if (shm_serial()) while (__atomic_exchange_n(&m->hdr->turn_lock, 1u, __ATOMIC_ACQUIRE)) sched_yield(); }
static void shm_turn_give(Comm *m) { if (shm_serial()) __atomic_store_n(&m->hdr->turn_lock, 0u, __ATOMIC_RELEASE); }

static void shm_barrier(Comm *m)
{
	ShmHeader *h = m->hdr;
	m->my_sense ^= 1;
	if (__atomic_add_fetch(&h->arrive, 1, __ATOMIC_ACQ_REL) == (unsigned)m->world) {
		__atomic_store_n(&h->arrive, 0, __ATOMIC_RELAXED);
		__atomic_store_n(&h->sense, m->my_sense, __ATOMIC_RELEASE);
	} else while (__atomic_load_n(&h->sense, __ATOMIC_ACQUIRE) != m->my_sense) sched_yield();
}

extern "C" int mahip_comm_unique_id(void *id128)
{
	CHK(rccl_load());
	ncclUniqueId id;
	{ StdoutToStderr quiet; NCCLCHK(g_rccl.GetUniqueId(&id)); }
	memcpy(id128, &id, sizeof(id));
	return 0;
}

extern "C" void mahip_comm_destroy(mahip_ctx_t *c)
{
	Comm *m = (Comm*)c->comm;
	if (!m) return;
	if (m->kind == 1 && m->nccl) (void)g_rccl.CommDestroy(m->nccl);
	if (m->kind == 2 && m->hdr) { shm_turn_give(m); munmap((void*)m->hdr, m->map_bytes); }
	delete m;
	c->comm = nullptr;
}

extern "C" int mahip_comm_init(mahip_ctx_t *c, const void *id128, int rank, int world)
{
	HIPCHK(hipSetDevice(c->dev));
	CHK(rccl_load());
	mahip_comm_destroy(c);
	Comm *m = new Comm();
	m->kind = 1; m->rank = rank; m->world = world;
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	ncclResult_t r;
	{ StdoutToStderr quiet; r = g_rccl.CommInitRank(&m->nccl, world, id, rank); }
	if (r != ncclSuccess) { mahip_set_error("ncclCommInitRank(rank %d of %d on device %d): %s", rank, world, c->dev, g_rccl.GetErrorString(r)); delete m; return -1; }
	c->comm = m;
	return 0;
}

extern "C" int mahip_comm_init_shm(mahip_ctx_t *c, const char *name, int rank, int world)
{
	mahip_comm_destroy(c);
	Comm *m = new Comm();
	m->kind = 2; m->rank = rank; m->world = world;
	snprintf(m->name, sizeof(m->name), "/%s", name);
	m->map_bytes = 4096 + (size_t)world * SHM_SLOT_BYTES;
	int fd = -1;
	for (int tries = 0; tries < 180000 && fd < 0; ++tries) { // rank 0 creates, the others wait for it -- up to three minutes: eight processes that start together on a fresh box (first import of
		// the Python stack, the HIP runtime coming up eight times) have been seen to be more than 20 s apart (round 6: the projection's N = 8 run lost a rank to the old limit)
		fd = rank == 0 ? shm_open(m->name, O_CREAT | O_RDWR, 0600) : shm_open(m->name, O_RDWR, 0600);
		if (fd < 0) usleep(1000);
	}
	if (fd < 0) { mahip_set_error("mahip_comm_init_shm: shm_open(%s) failed", m->name); delete m; return -1; }
	if (rank == 0 && ftruncate(fd, (off_t)m->map_bytes) != 0) { close(fd); mahip_set_error("mahip_comm_init_shm: ftruncate failed"); delete m; return -1; }
	if (rank != 0) for (int tries = 0; tries < 180000; ++tries) { off_t sz = lseek(fd, 0, SEEK_END); if ((size_t)sz >= m->map_bytes) break; usleep(1000); }
	void *p = mmap(nullptr, m->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED) { mahip_set_error("mahip_comm_init_shm: mmap failed"); delete m; return -1; }
	m->hdr = (ShmHeader*)p; m->slots = (char*)p + 4096;
	if (rank == 0) { m->hdr->arrive = 0; m->hdr->sense = 0; m->hdr->slot = SHM_SLOT_BYTES; m->hdr->turn_lock = 0; __atomic_store_n(&m->hdr->world, (unsigned)world, __ATOMIC_RELEASE); }
	else while (__atomic_load_n(&m->hdr->world, __ATOMIC_ACQUIRE) != (unsigned)world) sched_yield();
	c->comm = m;
	shm_barrier(m);
	if (rank == 0) shm_unlink(m->name); // every rank has mapped it: the name can go now, so that no run -- however it ends -- leaves a segment behind
	shm_turn_take(m);
	return 0;
}

extern "C" int mahip_comm_init_ext(mahip_ctx_t *c, int rank, int world, const mahip_comm_ext_t *ext)
{
	if (!ext || !ext->all_gather || !ext->all_reduce_max_u8 || !ext->all_reduce_sum_u64 || !ext->all_reduce_sum_u32 || !ext->all_to_all_v) { mahip_set_error("mahip_comm_init_ext: a callback is missing"); return -1; }
	mahip_comm_destroy(c);
	Comm *m = new Comm();
	m->kind = 3; m->rank = rank; m->world = world; m->ext = *ext;
	c->comm = m;
	return 0;
}
// kind 3: a device buffer through the host and the caller's callback
static int ext_stage_out(mahip_ctx *c, Comm *m, int k, const void *d_src, size_t bytes)
{
	if (m->hbuf[k].size() < bytes + 16) m->hbuf[k].resize(bytes + 16);
	if (bytes) { HIPCHK(hipMemcpyAsync(m->hbuf[k].data(), d_src, bytes, hipMemcpyDeviceToHost, c->st)); }
	HIPCHK(hipStreamSynchronize(c->st));
	return 0;
}
static int ext_stage_in(mahip_ctx *c, Comm *m, int k, void *d_dst, size_t bytes)
{
	if (bytes) { HIPCHK(hipMemcpyAsync(d_dst, m->hbuf[k].data(), bytes, hipMemcpyHostToDevice, c->st)); HIPCHK(hipStreamSynchronize(c->st)); }
	return 0;
}
#define EXTCHK(call, what) do { if ((call) != 0) { mahip_set_error("the caller's %s callback failed", what); return -1; } } while (0)

// MA_RCCL_ONE_RANK=1: a one-rank RCCL communicator still goes through ncclAllGather / ncclAllReduce (symbol binding, datatypes, in-place
// conventions and stream ordering run on hardware wherever a single GPU is all there is); default: one rank = plain copies
static bool one_rank_forced() { static int v = -1; if (v < 0) { const char *e = getenv("MA_RCCL_ONE_RANK"); v = e && atoi(e) != 0; } return v != 0; }
static inline bool comm_live(const Comm *m) { return m && (m->world > 1 || one_rank_forced()); }
extern "C" int mahip_comm_active(mahip_ctx_t *c) { return comm_live((Comm*)c->comm) ? 1 : 0; }
extern "C" int mahip_comm_rank(mahip_ctx_t *c) { return c->comm ? ((Comm*)c->comm)->rank : 0; }
extern "C" int mahip_comm_world(mahip_ctx_t *c) { return c->comm ? ((Comm*)c->comm)->world : 1; }

// every rank contributes `bytes` from d_send; d_recv receives world * bytes, rank-major
extern "C" int mahip_comm_all_gather(mahip_ctx_t *c, const void *d_send, void *d_recv, size_t bytes)
{
	HIPCHK(hipSetDevice(c->dev));
	Comm *m = (Comm*)c->comm;
	if (!comm_live(m)) { if (bytes && d_recv != d_send) HIPCHK(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, c->st)); return 0; }
	if (bytes == 0) return 0;
	if (m->kind == 1) { NCCLCHK(g_rccl.AllGather(d_send, d_recv, bytes, ncclUint8, m->nccl, c->st)); return 0; }
	if (m->kind == 3) {
		CHK(ext_stage_out(c, m, 0, d_send, bytes));
		if (m->hbuf[1].size() < bytes * (size_t)m->world + 16) m->hbuf[1].resize(bytes * (size_t)m->world + 16);
		EXTCHK(m->ext.all_gather(m->ext.user, m->hbuf[0].data(), m->hbuf[1].data(), bytes), "all_gather");
		return ext_stage_in(c, m, 1, d_recv, bytes * (size_t)m->world);
	}
	if (bytes > SHM_SLOT_BYTES) { mahip_set_error("shm all-gather: %zu bytes per rank exceed the slot", bytes); return -1; }
	HIPCHK(hipMemcpyAsync(m->slots + (size_t)m->rank * SHM_SLOT_BYTES, d_send, bytes, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	shm_turn_give(m);
	shm_barrier(m);
	for (int r = 0; r < m->world; ++r) HIPCHK(hipMemcpyAsync((char*)d_recv + (size_t)r * bytes, m->slots + (size_t)r * SHM_SLOT_BYTES, bytes, hipMemcpyHostToDevice, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	shm_barrier(m);
	shm_turn_take(m);
	return 0;
}

// element-wise maximum of n bytes over the ranks, in place (the OR of 0/1 flag arrays)
extern "C" int mahip_comm_all_reduce_max_u8(mahip_ctx_t *c, void *d_buf, size_t n)
{
	HIPCHK(hipSetDevice(c->dev));
	Comm *m = (Comm*)c->comm;
	if (!comm_live(m) || n == 0) return 0;
	if (m->kind == 1) { NCCLCHK(g_rccl.AllReduce(d_buf, d_buf, n, ncclUint8, ncclMax, m->nccl, c->st)); return 0; }
	if (m->kind == 3) {
		CHK(ext_stage_out(c, m, 0, d_buf, n));
		EXTCHK(m->ext.all_reduce_max_u8(m->ext.user, m->hbuf[0].data(), n), "all_reduce_max_u8");
		return ext_stage_in(c, m, 0, d_buf, n);
	}
	if (n > SHM_SLOT_BYTES) { mahip_set_error("shm all-reduce: %zu bytes exceed the slot", n); return -1; }
	uint8_t *mine = (uint8_t*)(m->slots + (size_t)m->rank * SHM_SLOT_BYTES);
	HIPCHK(hipMemcpyAsync(mine, d_buf, n, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	shm_turn_give(m);
	shm_barrier(m);
	uint8_t *acc = (uint8_t*)malloc(n);
	memcpy(acc, m->slots, n);
	for (int r = 1; r < m->world; ++r) { const uint8_t *s = (const uint8_t*)(m->slots + (size_t)r * SHM_SLOT_BYTES); for (size_t i = 0; i < n; ++i) acc[i] = s[i] > acc[i] ? s[i] : acc[i]; }
	shm_barrier(m);
	HIPCHK(hipMemcpyAsync(d_buf, acc, n, hipMemcpyHostToDevice, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	free(acc);
	shm_turn_take(m);
	return 0;
}

// element-wise sum of n u64 counters over the ranks, host values in and out (one device round trip)
extern "C" int mahip_comm_all_reduce_sum_u64(mahip_ctx_t *c, uint64_t *h_vals, size_t n)
{
	HIPCHK(hipSetDevice(c->dev));
	Comm *m = (Comm*)c->comm;
	if (!comm_live(m) || n == 0) return 0;
	if (n > CT_XCHG_WORDS) { mahip_set_error("mahip_comm_all_reduce_sum_u64: at most %d counters", (int)CT_XCHG_WORDS); return -1; }
	if (m->kind == 3) { HIPCHK(hipStreamSynchronize(c->st)); EXTCHK(m->ext.all_reduce_sum_u64(m->ext.user, h_vals, n), "all_reduce_sum_u64"); return 0; }
	if (m->kind == 1) {
		unsigned long long *d = P<unsigned long long>(c->ctr) + CT_XCHG; // the collectives' own words of the counter block (mahip_internal.hpp)
		HIPCHK(hipMemcpyAsync(d, h_vals, n * 8, hipMemcpyHostToDevice, c->st));
		NCCLCHK(g_rccl.AllReduce(d, d, n, ncclUint64, ncclSum, m->nccl, c->st));
		HIPCHK(hipMemcpyAsync(h_vals, d, n * 8, hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		return 0;
	}
	memcpy(m->slots + (size_t)m->rank * SHM_SLOT_BYTES, h_vals, n * 8);
	if (shm_serial()) { HIPCHK(hipStreamSynchronize(c->st)); shm_turn_give(m); }
	shm_barrier(m);
	uint64_t acc[32];
	memset(acc, 0, sizeof(acc));
	for (int r = 0; r < m->world; ++r) { const uint64_t *s = (const uint64_t*)(m->slots + (size_t)r * SHM_SLOT_BYTES); for (size_t i = 0; i < n; ++i) acc[i] += s[i]; }
	shm_barrier(m);
	memcpy(h_vals, acc, n * 8);
	shm_turn_take(m);
	return 0;
}

// element-wise sum of n u32 words over the ranks, in place on the device (per-read hit counts of the ranks' text ranges: host/ingest_sharded.c)
extern "C" int mahip_comm_all_reduce_sum_u32(mahip_ctx_t *c, void *d_buf, size_t n)
{
	HIPCHK(hipSetDevice(c->dev));
	Comm *m = (Comm*)c->comm;
	if (!comm_live(m) || n == 0) return 0;
	if (m->kind == 1) { NCCLCHK(g_rccl.AllReduce(d_buf, d_buf, n, ncclUint32, ncclSum, m->nccl, c->st)); return 0; }
	if (m->kind == 3) {
		CHK(ext_stage_out(c, m, 0, d_buf, n * 4));
		EXTCHK(m->ext.all_reduce_sum_u32(m->ext.user, m->hbuf[0].data(), n), "all_reduce_sum_u32");
		return ext_stage_in(c, m, 0, d_buf, n * 4);
	}
	if (n * 4 > SHM_SLOT_BYTES) { mahip_set_error("shm all-reduce: %zu bytes exceed the slot", n * 4); return -1; }
	uint32_t *mine = (uint32_t*)(m->slots + (size_t)m->rank * SHM_SLOT_BYTES);
	HIPCHK(hipMemcpyAsync(mine, d_buf, n * 4, hipMemcpyDeviceToHost, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	shm_turn_give(m);
	shm_barrier(m);
	uint32_t *acc = (uint32_t*)malloc(n * 4);
	memcpy(acc, m->slots, n * 4);
	for (int r = 1; r < m->world; ++r) { const uint32_t *s = (const uint32_t*)(m->slots + (size_t)r * SHM_SLOT_BYTES); for (size_t i = 0; i < n; ++i) acc[i] += s[i]; }
	shm_barrier(m);
	HIPCHK(hipMemcpyAsync(d_buf, acc, n * 4, hipMemcpyHostToDevice, c->st));
	HIPCHK(hipStreamSynchronize(c->st));
	free(acc);
	shm_turn_take(m);
	return 0;
}

// Personalised exchange: bytes[i * world + j] = what rank i sends to rank j (the same table on every rank: all-gathered counts).  d_send holds this rank's
// pieces back to back in destination order, d_recv receives the pieces meant for this rank back to back in SOURCE order -- for contiguous, ordered text
// ranges that is the order of the input.  RCCL: one group of ncclSend / ncclRecv pairs (every pair its own xGMI link, 7 per GPU); shm: through the segment.
extern "C" int mahip_comm_all_to_all_v(mahip_ctx_t *c, const void *d_send, void *d_recv, const uint64_t *bytes)
{
	HIPCHK(hipSetDevice(c->dev));
	Comm *m = (Comm*)c->comm;
	const int W = m ? m->world : 1, me = m ? m->rank : 0;
	if (!comm_live(m)) { if (bytes[0] && d_recv != d_send) HIPCHK(hipMemcpyAsync(d_recv, d_send, bytes[0], hipMemcpyDeviceToDevice, c->st)); return 0; }
	size_t soff = 0, roff = 0;
	if (m->kind == 1) {
		NCCLCHK(g_rccl.GroupStart());
		int bad = 0; // a call that fails inside the group must not leave it open: close it, then report
		for (int r = 0; r < W && !bad; ++r) {
			const size_t sb = bytes[(size_t)me * W + r], rb = bytes[(size_t)r * W + me];
			if (sb && g_rccl.Send((const char*)d_send + soff, sb, ncclUint8, r, m->nccl, c->st) != 0) bad = 1;
			if (!bad && rb && g_rccl.Recv((char*)d_recv + roff, rb, ncclUint8, r, m->nccl, c->st) != 0) bad = 1;
			soff += sb; roff += rb;
		}
		const int end_rc = (int)g_rccl.GroupEnd();
		if (bad || end_rc != 0) { mahip_set_error("mahip_comm_all_to_all_v: ncclSend / ncclRecv / ncclGroupEnd failed (%d)", end_rc); return -1; }
		return 0;
	}
	size_t mine = 0;
	for (int r = 0; r < W; ++r) mine += bytes[(size_t)me * W + r];
	if (m->kind == 3) {
		size_t in = 0;
		for (int r = 0; r < W; ++r) in += bytes[(size_t)r * W + me];
		CHK(ext_stage_out(c, m, 0, d_send, mine));
		if (m->hbuf[1].size() < in + 16) m->hbuf[1].resize(in + 16);
		EXTCHK(m->ext.all_to_all_v(m->ext.user, m->hbuf[0].data(), m->hbuf[1].data(), bytes), "all_to_all_v");
		return ext_stage_in(c, m, 1, d_recv, in);
	}
	// through the segment, destination by destination, a slot's worth of every rank's piece at a time (a rank's pieces for ALL destinations need not fit its slot:
	// BASELINE configs[3] on two ranks sends 3.2 GB from each)
	HIPCHK(hipStreamSynchronize(c->st));
	shm_turn_give(m);
	for (int dst = 0; dst < W; ++dst) {
		size_t longest = 0, my_off = 0;
		for (int r = 0; r < W; ++r) if (bytes[(size_t)r * W + dst] > longest) longest = bytes[(size_t)r * W + dst];
		for (int j = 0; j < dst; ++j) my_off += bytes[(size_t)me * W + j]; // where my piece for `dst` starts in d_send
		const size_t my_len = bytes[(size_t)me * W + dst];
		for (size_t c0 = 0; c0 < longest; c0 += SHM_SLOT_BYTES) { // every rank runs the same number of rounds
			const size_t part = c0 < my_len ? (my_len - c0 < SHM_SLOT_BYTES ? my_len - c0 : SHM_SLOT_BYTES) : 0;
			if (part) { HIPCHK(hipMemcpyAsync(m->slots + (size_t)me * SHM_SLOT_BYTES, (const char*)d_send + my_off + c0, part, hipMemcpyDeviceToHost, c->st)); HIPCHK(hipStreamSynchronize(c->st)); }
			shm_barrier(m);
			if (me == dst) {
				size_t ro = 0;
				for (int r = 0; r < W; ++r) {
					const size_t len = bytes[(size_t)r * W + me], p2 = c0 < len ? (len - c0 < SHM_SLOT_BYTES ? len - c0 : SHM_SLOT_BYTES) : 0;
					if (p2) HIPCHK(hipMemcpyAsync((char*)d_recv + ro + c0, m->slots + (size_t)r * SHM_SLOT_BYTES, p2, hipMemcpyHostToDevice, c->st));
					ro += len;
				}
				HIPCHK(hipStreamSynchronize(c->st));
			}
			shm_barrier(m);
		}
	}
	shm_turn_take(m);
	(void)roff; (void)mine;
	return 0;
}

// the same u64 words from every rank (host values): out[r * n + k] = rank r's vals[k]
extern "C" int mahip_comm_all_gather_u64(mahip_ctx_t *c, const uint64_t *h_vals, size_t n, uint64_t *h_out)
{
	Comm *m = (Comm*)c->comm;
	const int W = m ? m->world : 1, me = m ? m->rank : 0;
	if (!comm_live(m)) { memcpy(h_out, h_vals, n * 8); return 0; }
	const size_t tot = n * (size_t)W;
	memset(h_out, 0, tot * 8);
	memcpy(h_out + (size_t)me * n, h_vals, n * 8);
	for (size_t k = 0; k < tot; k += 32) CHK(mahip_comm_all_reduce_sum_u64(c, h_out + k, tot - k < 32 ? tot - k : 32)); // every word has one contributor: the sum IS the gather
	return 0;
}

extern "C" int mahip_comm_barrier(mahip_ctx_t *c)
{
	uint64_t x = 1;
	return mahip_comm_all_reduce_sum_u64(c, &x, 1);
}
