// Internals shared by the translation units of libtrx.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace trx {

// Per-(device, stream) scratch owned by the library.  Work on one stream is ordered, so a buffer can
// serve call after call on that stream without any allocator traffic; it only grows (the stream is
// synchronised first, then hipFree + hipMalloc), and it is released by trx_release_scratch().
//   slot 0  row-constant blocks of a likelihood call (rowc_kernel -> cells_kernel)
//   slot 1  trx_scenario_evidence: buffers sized by the number of draws
//   slot 2  trx_scenario_evidence: buffers sized by the number of masked draws
//   slot 3  (host, pinned) small results on their way back
constexpr int kScratchSlots = 4;
hipError_t stream_scratch(hipStream_t st, int slot, size_t bytes, void** out);

}  // namespace trx
