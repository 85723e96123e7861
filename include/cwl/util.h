// cwl/util.h -- error macros of the reference's cwl/include/cwl/util.h:11-34 on top of the C ABI.
#pragma once
#include <stdexcept>
#include <string>

#include "../fredholm_hip.h"

namespace cwl
{
// process-wide default context used by the buffer and post-process shims (the reference relies on the CUDA primary context)
inline fh_ctx*& default_context()
{
  static fh_ctx* ctx = nullptr;
  return ctx;
}
inline fh_ctx* require_context()
{
  fh_ctx*& c = default_context();
  if (!c) {
    if (fh_ctx_create(0, &c) != FH_OK) throw std::runtime_error(std::string("fh_ctx_create: ") + fh_last_error(nullptr));
  }
  return c;
}
inline void check(fh_ctx* ctx, int rc, const char* what)
{
  if (rc != FH_OK) throw std::runtime_error(std::string(what) + ": " + fh_last_error(ctx));
}
}  // namespace cwl
#define CUDA_SYNC_CHECK() ::cwl::check(::cwl::require_context(), fh_sync(::cwl::require_context()), "sync")

// The reference's applications spell two more things from this header's includes (cwl/include/cwl/util.h:7-21 pulls in <cuda_runtime.h>): the macro
// `CUDA_CHECK(call)` and, as its only argument anywhere in app/, `cudaFree(0)` -- CUDA's idiom for "create the primary context"
// (app/rtcamp8.cpp:75, app/controller.cpp:13).  These are APPLICATION-BOUNDARY NAMES for those two call sites, over the C ABI; nothing under
// fredholm_amd/csrc uses them and no other CUDA runtime name is provided.
typedef int cudaError_t;
static const cudaError_t cudaSuccess = FH_OK;
inline const char* cudaGetErrorString(cudaError_t) { return fh_last_error(::cwl::default_context()); }
inline cudaError_t cudaFree(void* device_ptr)
{
  fh_ctx*& c = ::cwl::default_context();
  if (!c) {
    const int rc = fh_ctx_create(0, &c);  // cudaFree(0): make the process-wide context
    if (rc != FH_OK) return rc;
  }
  return device_ptr ? fh_free(c, device_ptr) : FH_OK;
}
#define CUDA_CHECK(call)                                                                                                                    \
  do {                                                                                                                                      \
    cudaError_t fh_error_ = call;                                                                                                           \
    if (fh_error_ != cudaSuccess) {                                                                                                         \
      throw std::runtime_error(std::string("CUDA call (" #call " ) failed with error: '") + cudaGetErrorString(fh_error_) + "' (" __FILE__ ":" + \
                               std::to_string(__LINE__) + ")\n");                                                                           \
    }                                                                                                                                       \
  } while (0)
