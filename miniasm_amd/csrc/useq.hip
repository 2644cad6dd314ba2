// useq.hip -- unitig sequences on the device (reference asm.c:216-290 ma_ug_seq): every read placed on a unitig contributes the
// first `len` bases of its kept interval (forward) or the reverse complement of its last `len` bases (reverse strand).
// The host reads the FASTA/FASTQ records (gzip or plain: a byte stream only one thread can inflate) and hands the bases of the
// PLACED reads over in batches; the placement itself is a byte gather: one block per read, coalesced 1-byte loads / stores,
// complement through a 128-entry table in LDS.  HBM-bound by construction (bytes in = bytes out); the unitig arena stays in
// HBM until the last batch and comes back once.
#include "mahip_internal.hpp"

struct UseqBufs { DevBuf arena, seq, jobs; size_t arena_bytes = 0; };

static UseqBufs *useq_bufs(mahip_ctx *c)
{
	if (!c->useq) c->useq = new UseqBufs();
	return (UseqBufs*)c->useq;
}

void useq_free(mahip_ctx *c)
{
	UseqBufs *b = (UseqBufs*)c->useq;
	if (!b) return;
	dev_free(c, b->arena); dev_free(c, b->seq); dev_free(c, b->jobs);
	delete b;
	c->useq = nullptr;
}

// the reference's complement table (asm.c:225-234): IUPAC letters in both cases, everything else maps to itself, 0x60 to 0x40
__device__ __forceinline__ unsigned char comp_of(unsigned c)
{
	const char *from = "ABCDGHKMRTUVYabcdghkmrtuvy", *to = "TVGHCDMKYAABRtvghcdmkyaabr";
	if (c == 0x60) return 0x40;
	for (int k = 0; k < 26; ++k) if ((unsigned char)from[k] == c) return (unsigned char)to[k];
	return (unsigned char)c;
}

__global__ __launch_bounds__(256) void k_useq_gather(const unsigned char *__restrict__ seq, const mahip_useq_job_t *__restrict__ jobs, size_t n_jobs, unsigned char *__restrict__ arena)
{
	__shared__ unsigned char s_comp[128];
	if (threadIdx.x < 128) s_comp[threadIdx.x] = comp_of(threadIdx.x);
	__syncthreads();
	for (size_t j = blockIdx.x; j < n_jobs; j += gridDim.x) {
		const mahip_useq_job_t jb = jobs[j];
		const unsigned char *src = seq + jb.src_off;
		unsigned char *dst = arena + jb.dst_off;
		// a read file that disagrees with the PAF may hold fewer bases than the unitig takes from the read: the reference then reads outside its
		// buffer (asm.c:279-285, undefined); here those positions keep the arena's 'N' and nothing outside the batch is touched
		const uint32_t len = jb.len < jb.src_len ? jb.len : jb.src_len;
		if (!jb.rev) for (uint32_t i = threadIdx.x; i < len; i += 256) dst[i] = src[i];
		else for (uint32_t i = threadIdx.x; i < len; i += 256) { const unsigned ch = src[jb.src_len - 1 - i]; dst[i] = ch >= 128 ? 'N' : s_comp[ch]; } // asm.c:283-285
	}
}

extern "C" int mahip_useq_begin(mahip_ctx_t *c, size_t arena_bytes)
{
	HIPCHK(hipSetDevice(c->dev));
	UseqBufs *b = useq_bufs(c);
	CHK(dev_reserve(c, b->arena, arena_bytes + 64));
	b->arena_bytes = arena_bytes;
	if (arena_bytes) HIPCHK(hipMemsetAsync(b->arena.p, 'N', arena_bytes, c->st)); // asm.c:246: positions no read covers stay 'N'
	return 0;
}

extern "C" int mahip_useq_batch(mahip_ctx_t *c, const char *h_seq, size_t seq_bytes, const mahip_useq_job_t *h_jobs, size_t n_jobs)
{
	HIPCHK(hipSetDevice(c->dev));
	UseqBufs *b = useq_bufs(c);
	if (n_jobs == 0) return 0;
	CHK(dev_reserve(c, b->seq, seq_bytes + 64)); CHK(dev_reserve(c, b->jobs, n_jobs * sizeof(mahip_useq_job_t)));
	CHK(xfer_copy(c, b->seq.p, (void*)h_seq, seq_bytes, 1));
	HIPCHK(hipMemcpyAsync(b->jobs.p, h_jobs, n_jobs * sizeof(mahip_useq_job_t), hipMemcpyHostToDevice, c->st));
	ProfScope ps(c, "k_useq_gather", 0);
	hipLaunchKernelGGL(k_useq_gather, dim3(grid_for(n_jobs, 1, 65536)), dim3(256), 0, c->st, (const unsigned char*)b->seq.p, (const mahip_useq_job_t*)b->jobs.p, n_jobs, (unsigned char*)b->arena.p);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(c->st)); // the host reuses h_seq / h_jobs for the next batch
	return 0;
}

extern "C" int mahip_useq_end(mahip_ctx_t *c, char *h_arena)
{
	HIPCHK(hipSetDevice(c->dev));
	UseqBufs *b = useq_bufs(c);
	if (b->arena_bytes) CHK(xfer_copy(c, b->arena.p, h_arena, b->arena_bytes, 0));
	return 0;
}
