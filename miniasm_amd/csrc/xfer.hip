// xfer.hip -- host<->device bulk copies of pageable memory at PCIe speed.
// hipMemcpy from pageable memory is staged by ONE runtime thread through a pinned bounce buffer (measured
// 3.9 GB/s on the MI355X host); here W worker threads each own two pinned 4 MiB slots: a worker memcpy()s (or
// pread()s) its slices into a slot and queues the DMA on the context's stream, so the CPU copies of all workers
// and the DMA overlap.  No extra streams: creating one costs 8 ms (155 ms for the first of a process) and a
// single stream already moves 57 GB/s (measured, tools/probes/pin_probe.hip).
// Used for the hit records (640 MB per 10 M overlaps), PAF text and the exact-tie key/permutation traffic.
#include "mahip_internal.hpp"
#include <pthread.h>
#include <unistd.h>
#include <time.h>

#define XF_SLOT (4u << 20)
#define XF_MAX_WORKERS 16

struct XferPool {
	int n = 0;
	char *slot[XF_MAX_WORKERS][2] = {};
	bool ready[XF_MAX_WORKERS] = {};
	hipEvent_t ev[XF_MAX_WORKERS][2] = {};
};

struct XferJob { XferPool *p; hipStream_t st; int w, n_workers, dev, to_device, rc, fd; char *dev_ptr; char *host_ptr; size_t bytes, fd_off; }; // fd >= 0: the source is a file (pread)

static void *xfer_worker(void *arg)
{
	XferJob *j = (XferJob*)arg;
	XferPool *p = j->p;
	const int w = j->w;
	if (hipSetDevice(j->dev) != hipSuccess) { j->rc = -1; return 0; }
	if (!p->ready[w]) { // first use of this worker: pin its slots here, in parallel with the other workers
		for (int b = 0; b < 2; ++b) {
			if (hipHostMalloc((void**)&p->slot[w][b], XF_SLOT, hipHostMallocDefault) != hipSuccess) { j->rc = -1; return 0; }
			if (hipEventCreateWithFlags(&p->ev[w][b], hipEventDisableTiming) != hipSuccess) { j->rc = -1; return 0; }
		}
		p->ready[w] = true;
	}
	const hipStream_t st = j->st;
	const size_t n_slices = (j->bytes + XF_SLOT - 1) / XF_SLOT;
	int k = 0;
	for (size_t s = (size_t)w; s < n_slices; s += (size_t)j->n_workers, ++k) {
		const size_t off = s * XF_SLOT, len = j->bytes - off < XF_SLOT ? j->bytes - off : XF_SLOT;
		const int b = k & 1;
		if (j->to_device) {
			if (k >= 2 && hipEventSynchronize(p->ev[w][b]) != hipSuccess) { j->rc = -1; return 0; }
			if (j->fd >= 0) { // file -> pinned slot directly: no pageable intermediate copy
				size_t got = 0;
				while (got < len) {
					ssize_t r = pread(j->fd, p->slot[w][b] + got, len - got, (off_t)(j->fd_off + off + got));
					if (r <= 0) { j->rc = -1; return 0; }
					got += (size_t)r;
				}
			} else memcpy(p->slot[w][b], j->host_ptr + off, len);
			if (hipMemcpyAsync(j->dev_ptr + off, p->slot[w][b], len, hipMemcpyHostToDevice, st) != hipSuccess) { j->rc = -1; return 0; }
			if (hipEventRecord(p->ev[w][b], st) != hipSuccess) { j->rc = -1; return 0; }
		} else { // device -> host: DMA of slice k+1 overlaps the memcpy of slice k
			if (k == 0) {
				if (hipMemcpyAsync(p->slot[w][0], j->dev_ptr + off, len, hipMemcpyDeviceToHost, st) != hipSuccess) { j->rc = -1; return 0; }
				if (hipEventRecord(p->ev[w][0], st) != hipSuccess) { j->rc = -1; return 0; }
			}
			const size_t s2 = s + (size_t)j->n_workers;
			if (s2 < n_slices) {
				const size_t off2 = s2 * XF_SLOT, len2 = j->bytes - off2 < XF_SLOT ? j->bytes - off2 : XF_SLOT;
				if (hipMemcpyAsync(p->slot[w][b ^ 1], j->dev_ptr + off2, len2, hipMemcpyDeviceToHost, st) != hipSuccess) { j->rc = -1; return 0; }
				if (hipEventRecord(p->ev[w][b ^ 1], st) != hipSuccess) { j->rc = -1; return 0; }
			}
			if (hipEventSynchronize(p->ev[w][b]) != hipSuccess) { j->rc = -1; return 0; }
			memcpy(j->host_ptr + off, p->slot[w][b], len);
		}
	}
	if (hipStreamSynchronize(st) != hipSuccess) j->rc = -1;
	return 0;
}

extern "C" int ma_cpu_budget(void);
static int xfer_workers(int to_device)
{
	const char *s = getenv("MA_XFER_THREADS");
	long n = s ? atol(s) : ma_cpu_budget(); // (the control group's CPU quota counts, not the machine's core count: host/ingest_mt.c)
	// device -> host ends in a memcpy into pageable memory that is usually fresh (a page fault and a cleared page per 4 KB): 8 workers reach 21 GB/s, 16 reach 31
	// (8 GB of tie-walk keys, BASELINE configs[4], profiles/r03_tiewalk.txt); host -> device is at 45 GB/s with 8 and no faster with 16
	const long most = to_device ? 8 : 16;
	if (n < 1) n = 1;
	if (n > most && !s) n = most;
	if (n > XF_MAX_WORKERS) n = XF_MAX_WORKERS;
	return (int)n;
}

static int xfer_pool_init(mahip_ctx *c, int n)
{
	if (!c->xfer) c->xfer = new XferPool();
	XferPool *p = (XferPool*)c->xfer;
	if (p->n < n) p->n = n; // slots, events and the stream of a worker are created by the worker on first use
	return 0;
}

void xfer_pool_free(mahip_ctx *c)
{
	XferPool *p = (XferPool*)c->xfer;
	if (!p) return;
	for (int w = 0; w < p->n; ++w) {
		for (int b = 0; b < 2; ++b) { if (p->slot[w][b]) (void)hipHostFree(p->slot[w][b]); if (p->ev[w][b]) (void)hipEventDestroy(p->ev[w][b]); }
	}
	delete p;
	c->xfer = nullptr;
}

// synchronous with respect to the host; ordered after everything already queued on the context's stream
static int xfer_run(mahip_ctx *c, void *dev_ptr, void *host_ptr, int fd, size_t bytes, int to_device, size_t fd_off = 0)
{
	if (bytes == 0) return 0;
	HIPCHK(hipSetDevice(c->dev));
	if (fd < 0 && bytes < (8u << 20)) { // small: the runtime's own path
		HIPCHK(hipMemcpyAsync(to_device ? dev_ptr : host_ptr, to_device ? host_ptr : dev_ptr, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, c->st));
		HIPCHK(hipStreamSynchronize(c->st));
		return 0;
	}
	int W = xfer_workers(to_device);
	const size_t n_slices = (bytes + XF_SLOT - 1) / XF_SLOT;
	if ((size_t)W > n_slices) W = (int)n_slices;
	CHK(xfer_pool_init(c, W));
	XferJob job[XF_MAX_WORKERS];
	pthread_t th[XF_MAX_WORKERS];
	bool started[XF_MAX_WORKERS];
	for (int w = 0; w < W; ++w) {
		job[w].p = (XferPool*)c->xfer; job[w].st = c->st; job[w].w = w; job[w].n_workers = W; job[w].dev = c->dev; job[w].to_device = to_device; job[w].rc = 0;
		job[w].dev_ptr = (char*)dev_ptr; job[w].host_ptr = (char*)host_ptr; job[w].bytes = bytes; job[w].fd = fd; job[w].fd_off = fd_off;
		started[w] = pthread_create(&th[w], 0, xfer_worker, &job[w]) == 0;
		if (!started[w]) xfer_worker(&job[w]); // no thread: do this worker's slices here
	}
	int rc = 0;
	const bool timing = getenv("MA_PIPE_TIMING") != nullptr;
	struct timespec ts0, ts1;
	if (timing) clock_gettime(CLOCK_MONOTONIC, &ts0);
	for (int w = 0; w < W; ++w) {
		if (started[w]) pthread_join(th[w], 0);
		if (job[w].rc != 0) rc = -1;
	}
	if (timing) {
		clock_gettime(CLOCK_MONOTONIC, &ts1);
		double dt = (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
		fprintf(stderr, "[T::xfer] %s %.0f MB, %d workers: %.3f s (%.1f GB/s)\n", fd >= 0 ? "file->HBM" : to_device ? "host->HBM" : "HBM->host", (double)bytes / 1e6, W, dt, (double)bytes / dt / 1e9);
	}
	if (rc) { mahip_set_error("xfer_copy: staged copy failed (%s)", hipGetErrorString(hipGetLastError())); return -1; }
	return 0;
}

int xfer_copy(mahip_ctx *c, void *dev_ptr, void *host_ptr, size_t bytes, int to_device) { return xfer_run(c, dev_ptr, host_ptr, -1, bytes, to_device); }
// bytes [0, bytes) of an open file -> device memory
int xfer_from_fd(mahip_ctx *c, void *dev_ptr, int fd, size_t bytes) { return xfer_run(c, dev_ptr, nullptr, fd, bytes, 1); }
// bytes [off, off + bytes) of an open file -> device memory (a rank's own range of the text: host/ingest_sharded.c)
int xfer_from_fd_at(mahip_ctx *c, void *dev_ptr, int fd, size_t off, size_t bytes) { return xfer_run(c, dev_ptr, nullptr, fd, bytes, 1, off); }

extern "C" int mahip_memcpy_h2d(mahip_ctx_t *c, void *d_dst, const void *h_src, size_t bytes) { return xfer_copy(c, d_dst, (void*)h_src, bytes, 1); }
extern "C" int mahip_memcpy_d2h(mahip_ctx_t *c, void *h_dst, const void *d_src, size_t bytes) { return xfer_copy(c, (void*)d_src, h_dst, bytes, 0); }
