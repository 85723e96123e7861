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
