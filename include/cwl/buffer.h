// cwl/buffer.h -- RAII device buffer with the interface of the reference's cwl::CUDABuffer
// (cwl/include/cwl/buffer.h:18-85), backed by the device-memory entry points of the C ABI.
#pragma once
#include <cstdint>
#include <vector>

#include "util.h"

namespace cwl
{
template <typename T>
class CUDABuffer
{
 public:
  explicit CUDABuffer(uint32_t buffer_size) : m_buffer_size(buffer_size)
  {
    if (buffer_size == 0) return;
    check(require_context(), fh_malloc(require_context(), uint64_t(buffer_size) * sizeof(T), &m_d_ptr), "fh_malloc");
  }
  CUDABuffer(uint32_t buffer_size, uint32_t value) : CUDABuffer<T>(buffer_size)
  {
    if (buffer_size == 0) return;
    check(require_context(), fh_memset(require_context(), m_d_ptr, int(value), uint64_t(buffer_size) * sizeof(T)), "fh_memset");
  }
  explicit CUDABuffer(const std::vector<T>& values) : CUDABuffer<T>(uint32_t(values.size()))
  {
    if (!values.empty()) copy_from_host_to_device(values);
  }
  CUDABuffer(const CUDABuffer<T>&) = delete;
  CUDABuffer(CUDABuffer<T>&& o) noexcept : m_d_ptr(o.m_d_ptr), m_buffer_size(o.m_buffer_size) { o.m_d_ptr = nullptr; o.m_buffer_size = 0; }
  ~CUDABuffer() { if (m_d_ptr) fh_free(require_context(), m_d_ptr); }

  void clear() const { check(require_context(), fh_memset(require_context(), m_d_ptr, 0, uint64_t(m_buffer_size) * sizeof(T)), "fh_memset"); }
  void copy_from_host_to_device(const std::vector<T>& value) const
  {
    check(require_context(), fh_copy_to_device(require_context(), m_d_ptr, value.data(), uint64_t(m_buffer_size) * sizeof(T)), "fh_copy_to_device");
  }
  void copy_from_device_to_host(std::vector<T>& value) const
  {
    value.resize(m_buffer_size);
    check(require_context(), fh_copy_to_host(require_context(), value.data(), m_d_ptr, uint64_t(m_buffer_size) * sizeof(T)), "fh_copy_to_host");
  }
  T* get_device_ptr() { return reinterpret_cast<T*>(m_d_ptr); }
  const T* get_const_device_ptr() const { return reinterpret_cast<const T*>(m_d_ptr); }
  uint32_t get_size() const { return m_buffer_size; }
  uint32_t get_size_in_bytes() const { return m_buffer_size * sizeof(T); }

 private:
  void* m_d_ptr = nullptr;
  uint32_t m_buffer_size = 0;
};

// cwl::CUDAGLBuffer (cwl/include/cwl/buffer.h:88-143): an OpenGL buffer object the renderer writes through a mapped device pointer, so the GUI can
// draw the AOV layers without a copy (app/controller.cpp:80-107).  Compiled only where the application has OpenGL (define FH_WITH_OPENGL before
// including this header, as app/gui does by linking GL); backed by fh_gl_register_buffer / fh_gl_unregister_buffer (hipGraphicsGLRegisterBuffer).
#ifdef FH_WITH_OPENGL
}  // namespace cwl
#include <cstring>

#include "../oglw/buffer.h"
namespace cwl
{
template <typename T>
struct CUDAGLBuffer {
  explicit CUDAGLBuffer(uint32_t buffer_size) : m_buffer_size(buffer_size)
  {
    std::vector<T> data(m_buffer_size);
    std::memset(data.data(), 0, m_buffer_size * sizeof(T));
    m_buffer.setData(data, GL_STATIC_DRAW);
    uint64_t bytes = 0;
    check(require_context(), fh_gl_register_buffer(require_context(), m_buffer.getName(), &m_resource, &m_d_buffer, &bytes), "fh_gl_register_buffer");
  }
  CUDAGLBuffer(const CUDAGLBuffer&) = delete;
  CUDAGLBuffer(CUDAGLBuffer&& other) : m_buffer(std::move(other.m_buffer)), m_buffer_size(other.m_buffer_size), m_resource(other.m_resource), m_d_buffer(other.m_d_buffer)
  {
    other.m_resource = nullptr;
    other.m_d_buffer = nullptr;
  }
  ~CUDAGLBuffer() noexcept(false)
  {
    if (m_resource) check(require_context(), fh_gl_unregister_buffer(require_context(), m_resource), "fh_gl_unregister_buffer");
  }
  void clear() { check(require_context(), fh_memset(require_context(), m_d_buffer, 0, uint64_t(m_buffer_size) * sizeof(T)), "fh_memset"); }
  void copy_from_device_to_host(std::vector<T>& value)
  {
    value.resize(m_buffer_size);
    check(require_context(), fh_copy_to_host(require_context(), value.data(), m_d_buffer, uint64_t(m_buffer_size) * sizeof(T)), "fh_copy_to_host");
  }
  const oglw::Buffer<T>& get_gl_buffer() const { return m_buffer; }
  T* get_device_ptr() const { return reinterpret_cast<T*>(m_d_buffer); }

  oglw::Buffer<T> m_buffer;
  uint32_t m_buffer_size = 0;
  void* m_resource = nullptr;
  void* m_d_buffer = nullptr;
};
#endif  // FH_WITH_OPENGL
}  // namespace cwl
