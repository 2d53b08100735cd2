// rg_comm.cpp -- the exchanges between ranks, behind the C ABI (include/raygraph.h, "multi-GPU exchange").
//
// Replaces, for this path, RayPlatform's per-message MPI traffic: RAY_MPI_TAG_VERTICES_DATA (sender
// KmerAcademyBuilder.cpp:187-196, receiver MessageProcessor.cpp:692-809), OUT_/IN_EDGES_DATA (VerticesExtractor.cpp:175-251),
// REQUEST_VERTEX_COVERAGE (EdgePurgerWorker.cpp:38-47, IndexerWorker.cpp:76-96; served by MessageProcessor.cpp:1224-1256) and
// COVERAGE_DATA (CoverageGatherer.cpp:164-181 -> MessageProcessor.cpp:1048-1061) by a handful of all-to-all(v) per pass.
// Host code only: it drives the public entry points (rg_records_local_part, rg_export_records, rg_count_records,
// rg_owner_counts, ...) and moves what they export with one of two transports:
//   * RCCL (librccl.so.1, opened at run time): grouped ncclSend / ncclRecv over xGMI -- every one of the 7 peers of a GPU
//     has its own link, so an all-to-all uses them all at once.  Two communicators: the small one carries the per-rank
//     counts (a few words, host round trip of ~50 us), the large one the payload on its own stream, so that asking for the
//     counts of slice p never waits for the payload of slice p - 1 still in flight.
//   * the caller's own collectives (rg_transport: e.g. MPI_Alltoallv of GPU-aware MPI inside Ray, which is an MPI program;
//     the tests' in-process stand-in).
// One rank (size == 1) needs neither: everything stays on the device.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the functions are looked up in the librccl the process already holds

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rg_internal.h"

namespace {

struct RcclApi {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	std::string error;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

// A process must hold ONE librccl (as it must hold one HIP runtime): "librccl.so.1" resolves to the copy already loaded
// (torch ships one under the same SONAME) or, in a plain C++ process, to ROCm's.
bool rccl_load() {
	std::lock_guard<std::mutex> lk(g_rccl_mu);
	if (g_rccl.lib) return true;
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	void *lib = nullptr;
	for (const char *n : names) {
		lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
		if (lib) break;
	}
	if (!lib) {
		g_rccl.error = std::string("cannot open librccl.so.1: ") + (dlerror() ? dlerror() : "?");
		return false;
	}
#define RG_SYM(field, name)                                                          \
	*(void **)(&g_rccl.field) = dlsym(lib, name);                                    \
	if (!g_rccl.field) {                                                             \
		g_rccl.error = std::string("librccl lacks ") + name;                         \
		return false;                                                                \
	}
	RG_SYM(GetUniqueId, "ncclGetUniqueId")
	RG_SYM(CommInitRank, "ncclCommInitRank")
	RG_SYM(CommDestroy, "ncclCommDestroy")
	RG_SYM(Send, "ncclSend")
	RG_SYM(Recv, "ncclRecv")
	RG_SYM(GroupStart, "ncclGroupStart")
	RG_SYM(GroupEnd, "ncclGroupEnd")
	RG_SYM(GetErrorString, "ncclGetErrorString")
#undef RG_SYM
	g_rccl.lib = lib;
	return true;
}

}  // namespace

struct rg_comm {
	int rank = 0, size = 1, device = 0;
	int kind = 0;  // 1 RCCL, 2 caller's transport
	std::string err;
	// RCCL
	ncclComm_t data = nullptr, ctl = nullptr;
	hipStream_t s_data = nullptr, s_ctl = nullptr;
	uint64_t *d_ctl = nullptr;  // [2][size][CTL_WORDS] counts out / in
	uint64_t *h_ctl = nullptr;  // pinned, same shape
	// caller's transport
	rg_transport t{};
	// statistics (rg_comm_get_stats): payload bytes this rank sent to other ranks / received from them, seconds the host waited for payload
	uint64_t bytes_sent = 0, bytes_received = 0, collectives = 0;
	double wait_seconds = 0.0;
};

namespace {

constexpr uint32_t CTL_WORDS = 8;

int cfail(rg_comm *c, int code, const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (c) c->err = buf;
	return code;
}

thread_local std::string g_comm_create_error;

#define NCCL_TRY(c, call)                                                                                      \
	do {                                                                                                       \
		ncclResult_t r_ = (call);                                                                              \
		if (r_ != ncclSuccess) return cfail(c, RG_ERR_COMM, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); \
	} while (0)
#define CHIP_TRY(c, call)                                                                                 \
	do {                                                                                                  \
		hipError_t e_ = (call);                                                                           \
		if (e_ != hipSuccess) return cfail(c, RG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
	} while (0)

double now_s() {
	timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// ---- the two primitives every exchange is made of ----------------------------------------------------------------
// counts: n words for every rank out, n words from every rank in (host memory; returns with `in` filled)
int comm_alltoall_u64(rg_comm *c, int size, const uint64_t *out, uint64_t *in, uint32_t n) {
	if (!c) {  // one rank without a communicator (one with a communicator goes through it: RCCL sends to itself)
		memcpy(in, out, sizeof(uint64_t) * n);
		return RG_OK;
	}
	if (c->kind == 2) {
		const int rc = c->t.alltoall_u64(c->t.user, out, in, n);
		return rc == 0 ? RG_OK : cfail(c, RG_ERR_COMM, "the caller's alltoall_u64 returned %d", rc);
	}
	if (n > CTL_WORDS) return cfail(c, RG_ERR_ARG, "at most %u words per rank", CTL_WORDS);
	const size_t words = (size_t)size * n;
	memcpy(c->h_ctl, out, words * 8);
	uint64_t *d_out = c->d_ctl, *d_in = c->d_ctl + (size_t)size * CTL_WORDS;
	CHIP_TRY(c, hipMemcpyAsync(d_out, c->h_ctl, words * 8, hipMemcpyHostToDevice, c->s_ctl));
	NCCL_TRY(c, g_rccl.GroupStart());
	for (int r = 0; r < size; ++r) {
		NCCL_TRY(c, g_rccl.Send(d_out + (size_t)r * n, n, ncclUint64, r, c->ctl, c->s_ctl));
		NCCL_TRY(c, g_rccl.Recv(d_in + (size_t)r * n, n, ncclUint64, r, c->ctl, c->s_ctl));
	}
	NCCL_TRY(c, g_rccl.GroupEnd());
	uint64_t *h_in = c->h_ctl + (size_t)size * CTL_WORDS;
	CHIP_TRY(c, hipMemcpyAsync(h_in, d_in, words * 8, hipMemcpyDeviceToHost, c->s_ctl));
	CHIP_TRY(c, hipStreamSynchronize(c->s_ctl));
	memcpy(in, h_in, words * 8);
	return RG_OK;
}

// payload: scount[r] items of `unit` bytes to rank r from d_send (grouped in rank order), rcount[r] items from rank r into
// d_recv (rank order).  start queues it (RCCL: on the payload stream; the data must be complete, which it is: the entry
// points that made it return with the handle's stream drained); wait returns when d_recv is complete and d_send reusable.
struct Pending {
	hipEvent_t ev = nullptr;
	bool active = false;
};

int comm_alltoallv_start(rg_comm *c, int size, hipStream_t local_stream, const void *d_send, const uint64_t *scount, void *d_recv,
                         const uint64_t *rcount, uint32_t unit, Pending *p) {
	p->active = false;
	if (!c) {  // one rank without a communicator: what it sends to itself
		if (scount[0] != rcount[0]) return cfail(c, RG_ERR_ARG, "one rank sends itself %llu items and expects %llu", (unsigned long long)scount[0], (unsigned long long)rcount[0]);
		if (scount[0]) {
			hipError_t e = hipMemcpyAsync(d_recv, d_send, scount[0] * unit, hipMemcpyDeviceToDevice, local_stream);
			if (e == hipSuccess) e = hipStreamSynchronize(local_stream);
			if (e != hipSuccess) return cfail(c, RG_ERR_HIP, "device copy failed: %s", hipGetErrorString(e));
		}
		return RG_OK;
	}
	for (int r = 0; r < size; ++r)
		if (r != c->rank) {
			c->bytes_sent += scount[r] * unit;
			c->bytes_received += rcount[r] * unit;
		}
	c->collectives++;
	if (c->kind == 2) {
		const double t0 = now_s();
		const int rc = c->t.alltoallv(c->t.user, d_send, scount, d_recv, rcount, unit);
		c->wait_seconds += now_s() - t0;
		return rc == 0 ? RG_OK : cfail(c, RG_ERR_COMM, "the caller's alltoallv returned %d", rc);
	}
	const char *sp = (const char *)d_send;
	char *rp = (char *)d_recv;
	NCCL_TRY(c, g_rccl.GroupStart());
	for (int r = 0; r < size; ++r) {
		if (scount[r]) NCCL_TRY(c, g_rccl.Send(sp, scount[r] * unit, ncclInt8, r, c->data, c->s_data));
		if (rcount[r]) NCCL_TRY(c, g_rccl.Recv(rp, rcount[r] * unit, ncclInt8, r, c->data, c->s_data));
		sp += scount[r] * unit;
		rp += rcount[r] * unit;
	}
	NCCL_TRY(c, g_rccl.GroupEnd());
	CHIP_TRY(c, hipEventCreateWithFlags(&p->ev, hipEventDisableTiming));
	CHIP_TRY(c, hipEventRecord(p->ev, c->s_data));
	p->active = true;
	return RG_OK;
}

int comm_wait(rg_comm *c, Pending *p) {
	if (!p->active) return RG_OK;
	const double t0 = now_s();
	hipError_t e = hipEventSynchronize(p->ev);
	c->wait_seconds += now_s() - t0;
	(void)hipEventDestroy(p->ev);
	p->active = false;
	p->ev = nullptr;
	if (e != hipSuccess) return cfail(c, RG_ERR_HIP, "waiting for the all-to-all failed: %s", hipGetErrorString(e));
	return RG_OK;
}

int comm_alltoallv(rg_comm *c, int size, hipStream_t local_stream, const void *d_send, const uint64_t *scount, void *d_recv, const uint64_t *rcount,
                   uint32_t unit) {
	Pending p;
	int rc = comm_alltoallv_start(c, size, local_stream, d_send, scount, d_recv, rcount, unit, &p);
	if (rc == RG_OK) rc = comm_wait(c, &p);
	return rc;
}

// a pool block that goes back to the pool when the scope is left
struct Block {
	int dev;
	void *p = nullptr;
	explicit Block(int d) : dev(d) {}
	Block(const Block &) = delete;
	Block &operator=(const Block &) = delete;
	Block(Block &&o) noexcept : dev(o.dev), p(o.p) { o.p = nullptr; }
	~Block() { reset(); }
	void reset() {
		if (p) rgi::pool_put(dev, p);
		p = nullptr;
	}
	hipError_t alloc(uint64_t bytes) {
		reset();
		return rgi::pool_get(dev, &p, bytes ? bytes : 256);
	}
	void *release() {
		void *q = p;
		p = nullptr;
		return q;
	}
};

uint64_t sum_of(const std::vector<uint64_t> &v) {
	uint64_t s = 0;
	for (uint64_t x : v) s += x;
	return s;
}

// the communicator must be the handle's job: same rank, same size, same GPU
int check(rg_handle *h, rg_comm *c, const char *what) {
	if (!h) return RG_ERR_ARG;
	const int size = rgi::size_of(h);
	if (size > 1 && !c) return rgi::fail(h, RG_ERR_ARG, "%s: %d ranks need a communicator (rg_comm_create_rccl / rg_comm_create_transport)", what, size);
	if (c && (c->size != size || c->rank != rgi::rank_of(h) || c->device != rgi::device_of(h)))
		return rgi::fail(h, RG_ERR_ARG, "%s: the communicator is rank %d of %d on device %d, the handle rank %d of %d on device %d", what, c->rank, c->size, c->device,
		                 rgi::rank_of(h), size, rgi::device_of(h));
	return RG_OK;
}

// an error of the communicator becomes the handle's error text
int lift(rg_handle *h, rg_comm *c, int rc, const char *what) {
	if (rc == RG_OK) return rc;
	return rgi::fail(h, rc, "%s: %s", what, c ? c->err.c_str() : "exchange failed");
}

#define H_TRY(call)                  \
	do {                             \
		const int rc_ = (call);      \
		if (rc_ != RG_OK) return rc_; \
	} while (0)
#define C_TRY(h, c, what, call)                       \
	do {                                              \
		const int rc_ = (call);                       \
		if (rc_ != RG_OK) return lift(h, c, rc_, what); \
	} while (0)
#define A_TRY(h, what, blk, bytes)                                                                                             \
	do {                                                                                                                       \
		if ((blk).alloc(bytes) != hipSuccess) {                                                                                \
			(void)hipGetLastError();                                                                                           \
			return rgi::fail(h, RG_ERR_NOMEM, "%s: out of device memory (%.2f GB)", what, (double)(bytes) / 1e9);              \
		}                                                                                                                      \
	} while (0)

}  // namespace

extern "C" {

int rg_comm_unique_id(void *id_out) {
	if (!id_out) return RG_ERR_ARG;
	if (!rccl_load()) {
		g_comm_create_error = g_rccl.error;
		return RG_ERR_COMM;
	}
	static_assert(2 * sizeof(ncclUniqueId) <= RG_COMM_ID_BYTES, "two RCCL ids must fit the id buffer");
	memset(id_out, 0, RG_COMM_ID_BYTES);
	ncclUniqueId ids[2];
	for (int i = 0; i < 2; ++i) {
		ncclResult_t r = g_rccl.GetUniqueId(&ids[i]);
		if (r != ncclSuccess) {
			g_comm_create_error = std::string("ncclGetUniqueId failed: ") + g_rccl.GetErrorString(r);
			return RG_ERR_COMM;
		}
	}
	memcpy(id_out, ids, sizeof ids);
	return RG_OK;
}

const char *rg_comm_last_error(const rg_comm *c) { return c ? c->err.c_str() : g_comm_create_error.c_str(); }

int rg_comm_destroy(rg_comm *c) {
	if (!c) return RG_OK;
	(void)hipSetDevice(c->device);
	if (c->s_data) (void)hipStreamSynchronize(c->s_data);
	if (c->s_ctl) (void)hipStreamSynchronize(c->s_ctl);
	if (c->data) (void)g_rccl.CommDestroy(c->data);
	if (c->ctl) (void)g_rccl.CommDestroy(c->ctl);
	if (c->s_data) (void)hipStreamDestroy(c->s_data);
	if (c->s_ctl) (void)hipStreamDestroy(c->s_ctl);
	if (c->d_ctl) (void)hipFree(c->d_ctl);
	if (c->h_ctl) (void)hipHostFree(c->h_ctl);
	delete c;
	return RG_OK;
}

int rg_comm_create_rccl(const void *id, int rank, int size, int device, rg_comm **out) {
	if (!out) return RG_ERR_ARG;
	*out = nullptr;
	if (!id || size < 1 || rank < 0 || rank >= size) {
		g_comm_create_error = "rg_comm_create_rccl: bad rank / size / id";
		return RG_ERR_ARG;
	}
	if (!rccl_load()) {
		g_comm_create_error = g_rccl.error;
		return RG_ERR_COMM;
	}
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess) {
		g_comm_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
		return RG_ERR_HIP;
	}
	rg_comm *c = new rg_comm();
	c->rank = rank;
	c->size = size;
	c->device = device;
	c->kind = 1;
	auto bail = [&](int code, const std::string &msg) {
		g_comm_create_error = msg;
		rg_comm_destroy(c);
		return code;
	};
	ncclUniqueId ids[2];
	memcpy(ids, id, sizeof ids);
	if (hipStreamCreateWithFlags(&c->s_data, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->s_ctl, hipStreamNonBlocking) != hipSuccess ||
	    hipMalloc((void **)&c->d_ctl, sizeof(uint64_t) * 2 * (size_t)size * CTL_WORDS) != hipSuccess ||
	    hipHostMalloc((void **)&c->h_ctl, sizeof(uint64_t) * 2 * (size_t)size * CTL_WORDS) != hipSuccess)
		return bail(RG_ERR_HIP, "rg_comm_create_rccl: streams / control buffers could not be created");
	ncclResult_t r = g_rccl.CommInitRank(&c->data, size, ids[0], rank);
	if (r != ncclSuccess) return bail(RG_ERR_COMM, std::string("ncclCommInitRank (payload) failed: ") + g_rccl.GetErrorString(r));
	r = g_rccl.CommInitRank(&c->ctl, size, ids[1], rank);
	if (r != ncclSuccess) return bail(RG_ERR_COMM, std::string("ncclCommInitRank (counts) failed: ") + g_rccl.GetErrorString(r));
	*out = c;
	return RG_OK;
}

int rg_comm_create_transport(const rg_transport *t, int rank, int size, int device, rg_comm **out) {
	if (!out) return RG_ERR_ARG;
	*out = nullptr;
	if (!t || !t->alltoall_u64 || !t->alltoallv || size < 1 || rank < 0 || rank >= size) {
		g_comm_create_error = "rg_comm_create_transport: bad rank / size or a missing callback";
		return RG_ERR_ARG;
	}
	rg_comm *c = new rg_comm();
	c->rank = rank;
	c->size = size;
	c->device = device;
	c->kind = 2;
	c->t = *t;
	*out = c;
	return RG_OK;
}

int rg_comm_get_stats(rg_comm *c, rg_comm_stats *out) {
	if (!c || !out) return RG_ERR_ARG;
	out->bytes_sent = c->bytes_sent;
	out->bytes_received = c->bytes_received;
	out->collectives = c->collectives;
	out->wait_seconds = c->wait_seconds;
	return RG_OK;
}

// ---- pass 1, first exchange of the record route -----------------------------------------------------------------------
// The local reads, cut into super-k-mer records, go to the rank that aggregates their bucket; that rank then counts what it
// received.  parts > 1: the reads are cut in slices; the transfer of slice p runs on the payload stream while the record
// kernel and the export scatter of slice p + 1 run on the handle's; at most two slices' send buffers exist at a time.
int rg_exchange_records(rg_handle *h, rg_comm *c, uint32_t parts) {
	static const char *what = "rg_exchange_records";
	H_TRY(check(h, c, what));
	if (parts == 0) parts = 1;
	const int size = rgi::size_of(h), dev = rgi::device_of(h);
	hipStream_t ls = rgi::stream_of(h);
	std::vector<uint64_t> scount(size), rcount(size);
	uint32_t unit = 0;
	Block big(dev);
	uint64_t big_cap = 0, big_used = 0;  // records
	std::vector<std::pair<Block, uint64_t>> extra;  // slices that did not fit the common receive buffer: (block, records)
	std::vector<Block> sends;
	std::vector<Pending> pend(parts);
	for (uint32_t p = 0; p < parts; ++p) {
		if (p >= 2) {  // slice p - 2 has arrived: its send buffer can go
			C_TRY(h, c, what, comm_wait(c, &pend[p - 2]));
			sends[p - 2].reset();
		}
		H_TRY(rg_records_local_part(h, p, parts));
		H_TRY(rg_record_counts(h, scount.data(), &unit));
		const uint64_t total = sum_of(scount);
		sends.emplace_back(dev);
		A_TRY(h, what, sends.back(), total * unit);
		H_TRY(rg_export_records(h, sends.back().p, total));  // returns with the buffer complete
		C_TRY(h, c, what, comm_alltoall_u64(c, size, scount.data(), rcount.data(), 1));
		const uint64_t n_recv = sum_of(rcount);
		if (p == 0) {  // one receive buffer for all slices, sized from the first, so that nothing is copied together later
			big_cap = (uint64_t)((double)n_recv * parts * 1.1) + 4096ull * parts;
			A_TRY(h, what, big, big_cap * unit);
		}
		void *dst;
		if (big_used + n_recv <= big_cap) {
			dst = (char *)big.p + big_used * unit;
			big_used += n_recv;
		} else {
			extra.emplace_back(Block(dev), n_recv);
			A_TRY(h, what, extra.back().first, n_recv * unit);
			dst = extra.back().first.p;
		}
		C_TRY(h, c, what, comm_alltoallv_start(c, size, ls, sends.back().p, scount.data(), dst, rcount.data(), unit, &pend[p]));
	}
	for (uint32_t p = parts >= 2 ? parts - 2 : 0; p < parts; ++p) C_TRY(h, c, what, comm_wait(c, &pend[p]));
	sends.clear();
	uint64_t n_all = big_used;
	if (!extra.empty()) {  // rare: a later slice received more than the first one promised; gather everything in one buffer
		for (auto &e : extra) n_all += e.second;
		Block all(dev);
		A_TRY(h, what, all, n_all * unit);
		hipError_t e = hipMemcpyAsync(all.p, big.p, big_used * unit, hipMemcpyDeviceToDevice, ls);
		uint64_t at = big_used;
		for (auto &x : extra) {
			if (e == hipSuccess) e = hipMemcpyAsync((char *)all.p + at * unit, x.first.p, x.second * unit, hipMemcpyDeviceToDevice, ls);
			at += x.second;
		}
		if (e == hipSuccess) e = hipStreamSynchronize(ls);
		if (e != hipSuccess) return rgi::fail(h, RG_ERR_HIP, "%s: gathering the received slices failed: %s", what, hipGetErrorString(e));
		extra.clear();
		big.reset();
		return rg_count_records(h, all.p, n_all);
	}
	return rg_count_records(h, n_all ? big.p : nullptr, n_all);  // the buffer goes back to the pool when it returns
}

// ---- pass 1, the owner exchange (both routes) -------------------------------------------------------------------------
// The aggregated entries go to the rank vertexRank names (hash_function_1 % size, Kmer.cpp:165-170): the one exchange that
// RAY_MPI_TAG_VERTICES_DATA was.  What a rank receives arrives in ONE piece and is admitted in place (rg_adopt_entries).
// Follow with rg_resolve_admission.
int rg_exchange_owner_buckets(rg_handle *h, rg_comm *c) {
	static const char *what = "rg_exchange_owner_buckets";
	H_TRY(check(h, c, what));
	const int size = rgi::size_of(h), dev = rgi::device_of(h);
	if (size == 1 && !rgi::has_local(h)) return RG_OK;  // one rank after rg_count_local: its table already is the partition
	std::vector<uint64_t> scount(size), rcount(size);
	H_TRY(rg_owner_counts(h, scount.data()));
	const uint64_t total = sum_of(scount);
	Block send(dev);
	A_TRY(h, what, send, total * RG_ENTRY_BYTES);
	H_TRY(rg_export_owner_buckets(h, send.p, total));
	H_TRY(rg_release_local(h));  // the buckets hold everything the aggregation knew: its memory is free for the receive buffer
	C_TRY(h, c, what, comm_alltoall_u64(c, size, scount.data(), rcount.data(), 1));
	const uint64_t n_recv = sum_of(rcount);
	Block recv(dev);
	A_TRY(h, what, recv, n_recv * RG_ENTRY_BYTES);
	C_TRY(h, c, what, comm_alltoallv(c, size, rgi::stream_of(h), send.p, scount.data(), recv.p, rcount.data(), RG_ENTRY_BYTES));
	send.reset();
	bool adopted = false;
	H_TRY(rgi::adopt_pool_block(h, n_recv ? recv.p : nullptr, n_recv, &adopted));
	if (adopted) (void)recv.release();  // the handle's now: back to the pool when rg_resolve_admission is done with it
	return RG_OK;
}

// ---- RAY_SLAVE_MODE_ADD_VERTICES across ranks in one call ---------------------------------------------------------------
int rg_count_kmers_exchange(rg_handle *h, rg_comm *c, int route, uint32_t parts, uint64_t local_reads_for_bloom) {
	if (!h) return RG_ERR_ARG;
	if (route == RG_ROUTE_RECORDS) H_TRY(rg_exchange_records(h, c, parts ? parts : 4));
	else if (route == RG_ROUTE_ENTRIES) H_TRY(rg_count_local(h));
	else return rgi::fail(h, RG_ERR_ARG, "rg_count_kmers_exchange: route must be RG_ROUTE_RECORDS or RG_ROUTE_ENTRIES");
	H_TRY(rg_exchange_owner_buckets(h, c));
	return rg_resolve_admission(h, local_reads_for_bloom);
}

// ---- RAY_SLAVE_MODE_SEND_DISTRIBUTION: the ranks' histograms merged, on every rank --------------------------------------
// (the reference merges on rank 0, MessageProcessor.cpp:1048-1061: map[coverage] += count; here every rank gets the result,
// so rank 0 can write CoverageDistribution.txt and nobody waits for the three values to be sent round, MachineHelper.cpp:343-353)
int rg_histogram_exchange(rg_handle *h, rg_comm *c, uint32_t *coverage, uint64_t *count, uint64_t cap, uint64_t *n) {
	static const char *what = "rg_histogram_exchange";
	H_TRY(check(h, c, what));
	if (!n) return RG_ERR_ARG;
	const int size = rgi::size_of(h), dev = rgi::device_of(h);
	uint64_t mine = 0;
	H_TRY(rg_histogram(h, nullptr, nullptr, 0, &mine));
	std::vector<uint32_t> cov(mine ? mine : 1);
	std::vector<uint64_t> cnt(mine ? mine : 1);
	H_TRY(rg_histogram(h, cov.data(), cnt.data(), mine, &mine));
	std::map<uint32_t, uint64_t> merged;
	if (!c) {
		for (uint64_t i = 0; i < mine; ++i) merged[cov[i]] += cnt[i];
	} else {
		// every rank sends its (coverage, count) pairs to every rank: a few thousand pairs of 16 bytes
		std::vector<uint64_t> scount(size, mine), rcount(size);
		C_TRY(h, c, what, comm_alltoall_u64(c, size, scount.data(), rcount.data(), 1));
		const uint64_t n_recv = sum_of(rcount);
		std::vector<uint64_t> out(2 * std::max<uint64_t>(1, mine) * size), in(2 * std::max<uint64_t>(1, n_recv));
		for (int r = 0; r < size; ++r)
			for (uint64_t i = 0; i < mine; ++i) {
				out[2 * ((uint64_t)r * mine + i)] = cov[i];
				out[2 * ((uint64_t)r * mine + i) + 1] = cnt[i];
			}
		Block d_out(dev), d_in(dev);
		A_TRY(h, what, d_out, out.size() * 8);
		A_TRY(h, what, d_in, in.size() * 8);
		hipStream_t ls = rgi::stream_of(h);
		hipError_t e = hipMemcpyAsync(d_out.p, out.data(), out.size() * 8, hipMemcpyHostToDevice, ls);
		if (e == hipSuccess) e = hipStreamSynchronize(ls);
		if (e != hipSuccess) return rgi::fail(h, RG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
		C_TRY(h, c, what, comm_alltoallv(c, size, ls, d_out.p, scount.data(), d_in.p, rcount.data(), 16));
		e = hipMemcpyAsync(in.data(), d_in.p, n_recv * 16, hipMemcpyDeviceToHost, ls);
		if (e == hipSuccess) e = hipStreamSynchronize(ls);
		if (e != hipSuccess) return rgi::fail(h, RG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
		for (uint64_t i = 0; i < n_recv; ++i) merged[(uint32_t)in[2 * i]] += in[2 * i + 1];
	}
	*n = merged.size();
	if (coverage && count) {
		uint64_t i = 0;
		for (auto &kv : merged) {
			if (i >= cap) break;
			coverage[i] = kv.first;
			count[i] = kv.second;
			++i;
		}
	}
	return RG_OK;
}

// one word from every rank (the GridTable sizes for GraphPartition.txt, EdgePurger.cpp:172-217; read counts; timings)
int rg_allgather_u64(rg_handle *h, rg_comm *c, uint64_t value, uint64_t *out) {
	static const char *what = "rg_allgather_u64";
	H_TRY(check(h, c, what));
	if (!out) return RG_ERR_ARG;
	const int size = rgi::size_of(h);
	std::vector<uint64_t> send(size, value);
	C_TRY(h, c, what, comm_alltoall_u64(c, size, send.data(), out, 1));
	return RG_OK;
}

// ---- RAY_SLAVE_MODE_PURGE_NULL_EDGES across ranks (EdgePurgerWorker.cpp:28-102) ------------------------------------------
int rg_purge_edges_exchange(rg_handle *h, rg_comm *c) {
	static const char *what = "rg_purge_edges_exchange";
	H_TRY(check(h, c, what));
	const int size = rgi::size_of(h), dev = rgi::device_of(h);
	const uint32_t unit = 8u * (uint32_t)rgi::kmer_words_of(h);
	hipStream_t ls = rgi::stream_of(h);
	std::vector<uint64_t> ask(size), asked(size);
	H_TRY(rg_purge_query_counts(h, ask.data()));
	const uint64_t total = sum_of(ask);
	Block keys(dev), d_asked(dev), answers(dev), back(dev);
	A_TRY(h, what, keys, total * unit);
	H_TRY(rg_purge_export_queries(h, keys.p, total));
	C_TRY(h, c, what, comm_alltoall_u64(c, size, ask.data(), asked.data(), 1));
	const uint64_t n_asked = sum_of(asked);
	A_TRY(h, what, d_asked, n_asked * unit);
	C_TRY(h, c, what, comm_alltoallv(c, size, ls, keys.p, ask.data(), d_asked.p, asked.data(), unit));
	keys.reset();
	A_TRY(h, what, answers, n_asked);
	H_TRY(rg_purge_answer(h, d_asked.p, n_asked, (uint8_t *)answers.p));
	d_asked.reset();
	A_TRY(h, what, back, total);
	C_TRY(h, c, what, comm_alltoallv(c, size, ls, answers.p, asked.data(), back.p, ask.data(), 1));
	return rg_purge_apply(h, (const uint8_t *)back.p, total);
}

// ---- SequencesIndexer's coverage look-ups across ranks (IndexerWorker.cpp:67-248) ----------------------------------------
int rg_read_markers_exchange(rg_handle *h, rg_comm *c, uint32_t parts, int32_t *d_forward, int32_t *d_reverse) {
	static const char *what = "rg_read_markers_exchange";
	H_TRY(check(h, c, what));
	if (parts == 0) parts = 1;
	const int size = rgi::size_of(h), dev = rgi::device_of(h);
	const uint32_t unit = 8u * (uint32_t)rgi::kmer_words_of(h);
	hipStream_t ls = rgi::stream_of(h);
	std::vector<uint64_t> ask(size), asked(size);
	for (uint32_t p = 0; p < parts; ++p) {
		H_TRY(rg_markers_query_counts_part(h, p, parts, ask.data()));
		const uint64_t total = sum_of(ask);
		Block keys(dev), d_asked(dev), answers(dev), back(dev);
		A_TRY(h, what, keys, total * unit);
		H_TRY(rg_markers_export_queries(h, keys.p, total));
		C_TRY(h, c, what, comm_alltoall_u64(c, size, ask.data(), asked.data(), 1));
		const uint64_t n_asked = sum_of(asked);
		A_TRY(h, what, d_asked, n_asked * unit);
		C_TRY(h, c, what, comm_alltoallv(c, size, ls, keys.p, ask.data(), d_asked.p, asked.data(), unit));
		keys.reset();
		A_TRY(h, what, answers, n_asked * 4);
		H_TRY(rg_markers_answer(h, d_asked.p, n_asked, (uint32_t *)answers.p));
		d_asked.reset();
		A_TRY(h, what, back, total * 4);
		C_TRY(h, c, what, comm_alltoallv(c, size, ls, answers.p, asked.data(), back.p, ask.data(), 4));
		H_TRY(rg_markers_apply(h, (const uint32_t *)back.p, total, d_forward, d_reverse));
	}
	return RG_OK;
}

}  // extern "C"
