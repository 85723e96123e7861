// oglw/buffer.h -- the one class of the reference's OpenGL wrapper that the render path touches: oglw::Buffer<T>
// (oglw/include/oglw/buffer.h:9-76), the buffer object behind cwl::CUDAGLBuffer that app/gui.cpp binds as a shader-storage buffer to draw the
// AOVs (gui.cpp:330-347).  Same members; plain OpenGL 4.5 entry points (<GL/gl.h> + <GL/glext.h>) instead of the glad loader, no spdlog.
// Everything else of oglw (textures, framebuffers, shaders, the quad) is display code and out of scope.
#pragma once
#include <cstdint>
#include <vector>

#ifndef GL_GLEXT_PROTOTYPES
#define GL_GLEXT_PROTOTYPES 1
#endif
#include <GL/gl.h>
#include <GL/glext.h>

namespace oglw
{
template <typename T>
class Buffer
{
 private:
  GLuint buffer;
  uint32_t size;

 public:
  Buffer() : buffer{0}, size{0} { glCreateBuffers(1, &buffer); }
  Buffer(const Buffer&) = delete;
  Buffer(Buffer&& other) : buffer(other.buffer), size(other.size) { other.buffer = 0; }
  ~Buffer() { release(); }
  Buffer& operator=(const Buffer&) = delete;
  Buffer& operator=(Buffer&& other)
  {
    if (this != &other) {
      release();
      buffer = other.buffer;
      size = other.size;
      other.buffer = 0;
    }
    return *this;
  }
  void release()
  {
    if (buffer) {
      glDeleteBuffers(1, &buffer);
      buffer = 0;
    }
  }
  GLuint getName() const { return buffer; }
  uint32_t getLength() const { return size; }
  void setData(const std::vector<T>& data, GLenum usage)
  {
    glNamedBufferData(buffer, sizeof(T) * data.size(), data.data(), usage);
    size = uint32_t(data.size());
  }
  void bindToShaderStorageBuffer(GLuint binding_point_index) const { glBindBufferBase(GL_SHADER_STORAGE_BUFFER, binding_point_index, buffer); }
};
}  // namespace oglw
