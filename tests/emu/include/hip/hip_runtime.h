// tests/emu: a CPU stand-in for <hip/hip_runtime.h>.  TEST INFRASTRUCTURE ONLY.
//
// The product (miniasm_amd/lib/libminiasm_amd.so) is built by hipcc for gfx950 and has no CPU path.  This container has
// no GPU, so to exercise the *same kernel sources* here the CPU test-suite compiles csrc/*.hip a second time with g++
// against this header into tests/emu/_build/libminiasm_amd_emu.so (tests/emu/Makefile).  Nothing under miniasm_amd/,
// bench.py or __graft_entry__.py loads that library; only tests/test_emu_*.py do.
//
// Execution model (emu_runtime.cpp): every thread of a block is a fiber; a wave is 64 consecutive fibers.  A fiber runs
// until it reaches a cross-lane operation (__shfl*, __ballot, DPP, readfirstlane, wave barrier) or __syncthreads(),
// where it parks; when every lane of the wave is parked the operation is resolved for the lanes that wait at the same
// call site (those are the "active" lanes) and they continue.  Blocks are distributed over a few OS threads.  Kernel
// launches, copies and memsets are synchronous, so stream-ordering mistakes are NOT visible here; out-of-bounds device
// accesses are (EMU_GUARD=1 puts every allocation against a guard page).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define __HIPCC__ 1
#define MA_HIP_EMU 1

// ---- qualifiers ----
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local // block scope: implies static; one block at a time per OS thread
#ifndef __restrict__
#define __restrict__ __restrict
#endif

// ---- types ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct emu_stream *hipStream_t;
typedef struct emu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
enum { hipDeviceScheduleSpin = 1 };

struct dim3 {
	unsigned x, y, z;
	dim3() : x(1), y(1), z(1) {}
	template <typename T> dim3(T x_) : x((unsigned)x_), y(1), z(1) {}
	dim3(unsigned x_, unsigned y_, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) int2 { int32_t x, y; };
struct alignas(16) int4 { int32_t x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r = {x, y}; return r; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r = {x, y, z, w}; return r; }
static inline int2 make_int2(int32_t x, int32_t y) { int2 r = {x, y}; return r; }
static inline int4 make_int4(int32_t x, int32_t y, int32_t z, int32_t w) { int4 r = {x, y, z, w}; return r; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { ulonglong2 r = {x, y}; return r; }

// ---- runtime (emu_runtime.cpp) ----
namespace emu {
struct Fiber {
	void *sp;
	int state;
	unsigned lane, wave;
	dim3 tid;
	uint64_t val;  // value handed to the pending cross-lane operation
	uintptr_t site; // code address of the pending operation's call (orders the groups of a diverged wave)
	unsigned op;    // source-level id of the pending operation (groups the lanes: immune to code duplication by the compiler)
};
struct Wave {
	uint64_t slot[64]; // values of the last resolved operation, by lane
	uint64_t mask;     // lanes that took part in it
};
struct Block {
	dim3 bid, bdim, gdim;
	Wave *waves;
};
extern thread_local Fiber *t_fiber;
extern thread_local Block *t_block;

// park the calling lane in a cross-lane operation; returns the wave record holding every participant's value
Wave *wave_op(uint64_t val, unsigned op, uintptr_t site);
void block_barrier();
struct LaunchFn { void (*call)(void *); void *arg; };
void launch(const char *name, dim3 grid, dim3 block, size_t shmem, LaunchFn fn);
[[noreturn]] void fatal(const char *fmt, ...);
} // namespace emu

#define threadIdx (emu::t_fiber->tid)
#define blockIdx (emu::t_block->bid)
#define blockDim (emu::t_block->bdim)
#define gridDim (emu::t_block->gdim)
#define warpSize 64

#define EMU_SITE ((uintptr_t)__builtin_return_address(0))

template <typename F> static inline void emu_launch_(const char *name, dim3 g, dim3 b, size_t sh, F &&f)
{
	emu::LaunchFn fn = {[](void *p) { (*(F *)p)(); }, (void *)&f};
	emu::launch(name, g, b, sh, fn);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
	emu_launch_(#kernel, (grid), (block), (shmem), [&]() { (kernel)(__VA_ARGS__); })

// ---- cross-lane operations ----
namespace emu {
template <typename T> static inline uint64_t to_bits(T v)
{
	static_assert(sizeof(T) <= 8, "cross-lane value wider than 64 bits");
	uint64_t b = 0;
	memcpy(&b, &v, sizeof(T));
	return b;
}
template <typename T> static inline T from_bits(uint64_t b)
{
	T v;
	memcpy(&v, &b, sizeof(T));
	return v;
}
// these are deliberately NOT inlined: EMU_SITE must be the address inside the (inlined) kernel body
template <typename T> __attribute__((noinline)) T shfl_idx(unsigned op, T v, int src, int width = 64)
{
	Wave *w = wave_op(to_bits(v), op, EMU_SITE);
	unsigned lane = t_fiber->lane, base = lane & ~(unsigned)(width - 1), s = base + ((unsigned)src & (unsigned)(width - 1));
	return (w->mask >> s & 1) ? from_bits<T>(w->slot[s]) : T(0);
}
template <typename T> __attribute__((noinline)) T shfl_xor(unsigned op, T v, int m, int width = 64)
{
	Wave *w = wave_op(to_bits(v), op, EMU_SITE);
	unsigned lane = t_fiber->lane, base = lane & ~(unsigned)(width - 1), s = lane ^ (unsigned)m;
	if (s >= base + (unsigned)width || s < base) return v;
	return (w->mask >> s & 1) ? from_bits<T>(w->slot[s]) : T(0);
}
template <typename T> __attribute__((noinline)) T shfl_up(unsigned op, T v, unsigned d, int width = 64)
{
	Wave *w = wave_op(to_bits(v), op, EMU_SITE);
	unsigned lane = t_fiber->lane, base = lane & ~(unsigned)(width - 1);
	if (lane < base + d) return v;
	unsigned s = lane - d;
	return (w->mask >> s & 1) ? from_bits<T>(w->slot[s]) : T(0);
}
template <typename T> __attribute__((noinline)) T shfl_down(unsigned op, T v, unsigned d, int width = 64)
{
	Wave *w = wave_op(to_bits(v), op, EMU_SITE);
	unsigned lane = t_fiber->lane, base = lane & ~(unsigned)(width - 1), s = lane + d;
	if (s >= base + (unsigned)width) return v;
	return (w->mask >> s & 1) ? from_bits<T>(w->slot[s]) : T(0);
}
// ds_bpermute_b32: lane i reads the value of lane (addr_i >> 2) & 63; ds_permute_b32: lane i sends its value to lane (addr_i >> 2) & 63 (the highest
// sending lane wins, a lane nobody sends to reads 0); v_readlane_b32: the value of one lane for all
__attribute__((noinline)) int ds_bpermute(unsigned op, int addr, int v);
__attribute__((noinline)) int ds_permute(unsigned op, int addr, int v);
__attribute__((noinline)) int readlane(unsigned op, int v, int lane);
__attribute__((noinline)) uint64_t ballot(unsigned op, int p);
__attribute__((noinline)) uint32_t readfirstlane(unsigned op, uint32_t v);
__attribute__((noinline)) void wave_barrier(unsigned op);
// v_mov_b32 with a DPP modifier (gfx9 encodings): quad_perm, row_shl/shr/ror, row_mirror, row_half_mirror, row_bcast
__attribute__((noinline)) int update_dpp(unsigned op, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl);
__attribute__((noinline)) int mov_dpp(unsigned op, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl);
struct BufRsrc { char *base; uint32_t num_records; };
} // namespace emu

// Every cross-lane operation carries a source-level id (one per macro expansion): lanes parked at the same id are the
// lanes that execute the operation together, however many copies of the call the compiler made (loop rotation, jump
// threading); the code address only orders the groups of a diverged wave.
#define EMU_OP ((unsigned)__COUNTER__ + 1u)
#define __shfl(...) emu::shfl_idx(EMU_OP, __VA_ARGS__)
#define __shfl_xor(...) emu::shfl_xor(EMU_OP, __VA_ARGS__)
#define __shfl_up(...) emu::shfl_up(EMU_OP, __VA_ARGS__)
#define __shfl_down(...) emu::shfl_down(EMU_OP, __VA_ARGS__)
#define __ballot(p) emu::ballot(EMU_OP, (p) ? 1 : 0)
#define __any(p) (emu::ballot(EMU_OP, (p) ? 1 : 0) != 0)
#define __all(p) (emu::ballot(EMU_OP, (p) ? 0 : 1) == 0)
#define __lane_id() (emu::t_fiber->lane)
#define __syncthreads() emu::block_barrier()
#define __threadfence() ((void)0)
#define __threadfence_block() ((void)0)
#define __threadfence_system() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) emu::mov_dpp(EMU_OP, (int)(src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu::update_dpp(EMU_OP, (int)(old), (int)(src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_ds_bpermute(addr, v) emu::ds_bpermute(EMU_OP, (int)(addr), (int)(v))
#define __builtin_amdgcn_ds_permute(addr, v) emu::ds_permute(EMU_OP, (int)(addr), (int)(v))
#define __builtin_amdgcn_readlane(v, l) emu::readlane(EMU_OP, (int)(v), (int)(l))
#define __builtin_amdgcn_ballot_w64(p) emu::ballot(EMU_OP, (p) ? 1 : 0)
#define __builtin_amdgcn_mbcnt_lo(m, c) ((unsigned)(c) + (unsigned)__builtin_popcount((unsigned)(m) & (unsigned)((emu::t_fiber->lane >= 32 ? ~0ull : (1ull << emu::t_fiber->lane) - 1ull))))
#define __builtin_amdgcn_mbcnt_hi(m, c) ((unsigned)(c) + (unsigned)__builtin_popcount((unsigned)(m) & (unsigned)(emu::t_fiber->lane < 32 ? 0ull : (1ull << (emu::t_fiber->lane - 32)) - 1ull)))
#define __builtin_amdgcn_alignbyte(hi, lo, sh) ((uint32_t)((((uint64_t)(uint32_t)(hi) << 32) | (uint32_t)(lo)) >> (8u * ((unsigned)(sh) & 3u))))
#define __builtin_amdgcn_alignbit(hi, lo, sh) ((uint32_t)((((uint64_t)(uint32_t)(hi) << 32) | (uint32_t)(lo)) >> ((unsigned)(sh) & 31u)))
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_s_waitcnt(n) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) ((void)(*(p) = (v)))
#define __builtin_amdgcn_readfirstlane(v) emu::readfirstlane(EMU_OP, (uint32_t)(v))
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier(EMU_OP)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_s_barrier() emu::block_barrier()
typedef emu::BufRsrc __amdgpu_buffer_rsrc_t;
// raw buffer (stride 0): an access whose offset reaches num_records is dropped by the hardware
static inline emu::BufRsrc __builtin_amdgcn_make_buffer_rsrc(const void *p, short stride, int num, int flags)
{
	(void)stride; (void)flags;
	emu::BufRsrc r = {(char *)p, (uint32_t)num};
	return r;
}
static inline void __builtin_amdgcn_raw_buffer_store_b32(uint32_t v, emu::BufRsrc r, int voff, int soff, int aux)
{
	(void)aux;
	uint32_t off = (uint32_t)voff + (uint32_t)soff;
	if (off < r.num_records && off + 4u <= r.num_records) memcpy(r.base + off, &v, 4);
}
static inline uint32_t __builtin_amdgcn_raw_buffer_load_b32(emu::BufRsrc r, int voff, int soff, int aux)
{
	(void)aux;
	uint32_t off = (uint32_t)voff + (uint32_t)soff, v = 0;
	if (off < r.num_records && off + 4u <= r.num_records) memcpy(&v, r.base + off, 4);
	return v;
}

// ---- bit intrinsics ----
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline unsigned __brev(unsigned x)
{
	unsigned r = 0;
	for (int i = 0; i < 32; ++i) r |= (x >> i & 1u) << (31 - i);
	return r;
}

// ---- atomics (blocks run on several OS threads) ----
template <typename T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicSub(T *p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicXor(T *p, T v) { return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicCAS(T *p, T cmp, T v)
{
	__atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
	return cmp;
}
template <typename T> static inline T atomicMin(T *p, T v)
{
	T o = __atomic_load_n(p, __ATOMIC_RELAXED);
	while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
	return o;
}
template <typename T> static inline T atomicMax(T *p, T v)
{
	T o = __atomic_load_n(p, __ATOMIC_RELAXED);
	while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
	return o;
}

// ---- host API ----
extern "C" {
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int dev);
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int dev);
hipError_t hipDeviceSynchronize(void);
hipError_t hipGetLastError(void);
const char *hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st);
hipError_t hipMemset(void *dst, int v, size_t bytes);
hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t st);
typedef void *hipDeviceptr_t;
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int v, size_t count, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t *st);
hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; } /* launches are synchronous here */
static inline hipError_t hipSetDeviceFlags(unsigned) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; } /* launches are synchronous here */
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
}
template <typename T> static inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc((void **)p, bytes); }
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned flags = 0) { return hipHostMalloc((void **)p, bytes, flags); }
