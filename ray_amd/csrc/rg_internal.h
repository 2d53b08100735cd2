// rg_internal.h -- what rg_comm.cpp (the exchanges between ranks, host code over the public C ABI) needs from inside the
// library: the handle's stream and device, the device memory pool, the error text.  Not installed, not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/raygraph.h"

namespace rgi {
int fail(rg_handle *h, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
hipStream_t stream_of(rg_handle *h);
int device_of(const rg_handle *h);
int rank_of(const rg_handle *h);
int size_of(const rg_handle *h);
int kmer_words_of(const rg_handle *h);
uint64_t reads_of(const rg_handle *h);
bool has_local(const rg_handle *h);  // an aggregation of this rank's own reads / received records exists (rg_count_local with size > 1, rg_count_records)
hipError_t pool_get(int device, void **p, uint64_t bytes);
void pool_put(int device, void *p);
// rg_adopt_entries for a block that came from pool_get: *adopted = true means the handle now owns the block (it goes back to
// the pool when rg_resolve_admission is done with it), false that the entries were merged and the block is still the caller's
int adopt_pool_block(rg_handle *h, void *d_entries, uint64_t n_entries, bool *adopted);
}  // namespace rgi
