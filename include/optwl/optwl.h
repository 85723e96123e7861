// optwl/optwl.h -- stand-in for the reference's OptiX context wrapper (optwl/include/optwl/optwl.h:41-81).
// There is no OptiX: the "device context" is the HIP context behind libfredholm_hip.so.
#pragma once
#include "../cwl/util.h"

using OptixDeviceContext = fh_ctx*;

namespace optwl
{
struct Context {
  OptixDeviceContext m_context = nullptr;
  explicit Context(bool /*enable_validation_mode*/ = false) { m_context = cwl::require_context(); }
  OptixDeviceContext get_context() const { return m_context; }
};
}  // namespace optwl
