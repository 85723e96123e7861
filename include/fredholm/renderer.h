// fredholm/renderer.h -- drop-in facade with the method surface of the reference's header-only
// fredholm::Renderer (fredholm/include/fredholm/renderer.h:29-846).  Every method forwards to the C ABI of
// libfredholm_hip.so; a non-zero status becomes std::runtime_error like the reference's CUDA_CHECK /
// OPTIX_CHECK.  The OptiX pipeline-construction calls are kept as no-ops so application code
// (app/controller.cpp:60-68, app/rtcamp8.cpp:75-146) compiles unchanged.
#pragma once
#include <cstdint>
#include <filesystem>
#include <stdexcept>

// the reference's header includes the application's logging library (renderer.h:12), and app/controller.cpp:311 relies on getting it from here
#if defined(__has_include)
#if __has_include("spdlog/spdlog.h")
#include "spdlog/spdlog.h"
#endif
#endif

#include "../cwl/util.h"
#include "../optwl/optwl.h"
#include "camera.h"
#include "scene.h"
#include "shared.h"

namespace fredholm
{

class Renderer
{
 public:
  explicit Renderer(const OptixDeviceContext& context) : m_ctx(context) {}
  ~Renderer() noexcept(false) {}

  // OptiX plumbing of the reference (renderer.h:124-352): nothing to build, the kernels live in the library
  void create_module(const std::filesystem::path&) {}
  void create_program_group() {}
  void create_pipeline() {}
  void create_sbt() {}

  // renderer.h:354-432
  void load_scene(const std::filesystem::path& filepath, bool clear = true)
  {
    m_scene.load_model(filepath, clear);
    upload_scene();
  }
  // same upload for a scene assembled in memory
  void load_scene(const Scene& scene)
  {
    m_scene = scene;
    upload_scene();
  }

  void build_gas() {}                                                            // renderer.h:434-496: folded into build_ias
  void build_ias() { cwl::check(m_ctx, fh_bvh_build(m_ctx), "fh_bvh_build"); }  // renderer.h:498-552

  void set_directional_light(const float3& le, const float3& dir, float angle)  // renderer.h:554-567
  {
    const float l[3] = {le.x, le.y, le.z}, d[3] = {dir.x, dir.y, dir.z};
    cwl::check(m_ctx, fh_set_directional_light(m_ctx, l, d, angle), "fh_set_directional_light");
  }
  void set_sky_intensity(float v) { cwl::check(m_ctx, fh_set_sky_intensity(m_ctx, v), "fh_set_sky_intensity"); }
  void load_ibl(const std::filesystem::path& filepath)  // renderer.h:574-581
  {
    const FloatTexture ibl(filepath);
    cwl::check(m_ctx, fh_load_ibl(m_ctx, reinterpret_cast<const float*>(ibl.m_data.data()), ibl.m_width, ibl.m_height), "fh_load_ibl");
  }
  void clear_ibl() { cwl::check(m_ctx, fh_clear_ibl(m_ctx), "fh_clear_ibl"); }  // renderer.h:583-586
  void load_arhosek_sky(float turbidity, float albedo) { cwl::check(m_ctx, fh_load_arhosek_sky(m_ctx, turbidity, albedo), "fh_load_arhosek_sky"); }
  void clear_arhosek_sky() { cwl::check(m_ctx, fh_clear_arhosek_sky(m_ctx), "fh_clear_arhosek_sky"); }

  void set_time(float time)  // renderer.h:614-640: animation update, transform re-upload, acceleration-structure rebuild
  {
    m_scene.update_animation(time);
    std::vector<float> o2w, w2o;
    m_scene.transforms_3x4(o2w, w2o);
    if (!o2w.empty()) cwl::check(m_ctx, fh_set_transforms(m_ctx, uint32_t(o2w.size() / 12), o2w.data(), w2o.data()), "fh_set_transforms");
    build_ias();
  }
  void set_resolution(uint32_t width, uint32_t height)  // renderer.h:642-648
  {
    m_width = width;
    m_height = height;
    cwl::check(m_ctx, fh_set_resolution(m_ctx, width, height), "fh_set_resolution");
  }
  void init_render_states() { cwl::check(m_ctx, fh_init_render_states(m_ctx), "fh_init_render_states"); }  // renderer.h:650-655

  // renderer.h:657-734
  void render(const Camera& camera, const float3& bg_color, const RenderLayer& render_layer, uint32_t n_samples, uint32_t max_depth)
  {
    const fh_camera cam = m_scene.m_has_camera_transform ? camera_from(m_scene.m_camera_transform, camera) : camera.to_c();
    const float bg[3] = {bg_color.x, bg_color.y, bg_color.z};
    fh_render_layers layers{reinterpret_cast<float*>(render_layer.beauty), reinterpret_cast<float*>(render_layer.position), render_layer.depth,
                            reinterpret_cast<float*>(render_layer.normal), reinterpret_cast<float*>(render_layer.texcoord), reinterpret_cast<float*>(render_layer.albedo)};
    cwl::check(m_ctx, fh_render(m_ctx, &cam, bg, &layers, n_samples, max_depth, 1u /* params.seed = 1, renderer.h:664 */), "fh_render");
  }
  void wait_for_completion() { cwl::check(m_ctx, fh_sync(m_ctx), "fh_sync"); }  // renderer.h:736

  // not in the reference: render(n_samples = k) as ONE reference launch of k samples, payload.firsthit quirk included (pt.cu:432-433;
  // rtcamp8 renders 16 per launch, rtcamp8.cpp:183-189), instead of k one-sample launches
  void set_reference_launch_semantics(bool on)
  {
    uint32_t flags = 0;  // (only this bit changes: timing / counting / serial-pass flags set through the C ABI stay as they are)
    cwl::check(m_ctx, fh_get_flags(m_ctx, &flags), "fh_get_flags");
    cwl::check(m_ctx, fh_set_flags(m_ctx, on ? (flags | FH_FLAG_REFERENCE_FIRSTHIT) : (flags & ~FH_FLAG_REFERENCE_FIRSTHIT)), "fh_set_flags");
  }

 private:
  static fh_camera camera_from(const Mat4& m, const Camera& camera)
  {
    fh_camera c = camera.to_c();
    for (int r = 0; r < 3; ++r)
      for (int col = 0; col < 4; ++col) c.transform[4 * r + col] = m[col][r];
    return c;
  }
  void upload_scene()
  {
    if (!m_scene.is_valid()) throw std::runtime_error("invalid scene");
    std::vector<float> o2w, w2o;
    m_scene.transforms_3x4(o2w, w2o);
    fh_scene_desc d{};
    d.n_vertices = uint32_t(m_scene.m_vertices.size());
    d.vertices = reinterpret_cast<const float*>(m_scene.m_vertices.data());
    d.normals = reinterpret_cast<const float*>(m_scene.m_normals.data());
    d.texcoords = reinterpret_cast<const float*>(m_scene.m_texcoords.data());
    d.n_faces = uint32_t(m_scene.m_indices.size());
    d.indices = reinterpret_cast<const uint32_t*>(m_scene.m_indices.data());
    d.material_ids = m_scene.m_material_ids.data();
    d.instance_ids = m_scene.m_instance_ids.empty() ? nullptr : m_scene.m_instance_ids.data();
    d.n_materials = uint32_t(m_scene.m_materials.size());
    d.materials = reinterpret_cast<const fh_material*>(m_scene.m_materials.data());
    d.n_instances = uint32_t(m_scene.m_transforms.size());
    d.object_to_world = o2w.empty() ? nullptr : o2w.data();
    d.world_to_object = w2o.empty() ? nullptr : w2o.data();
    std::vector<fh_texture_desc> tex;  // renderer.h:414-424: COLOR textures are sRGB-decoded by the texture unit
    for (const Texture& t : m_scene.m_textures)
      tex.push_back(fh_texture_desc{t.m_width, t.m_height, reinterpret_cast<const uint8_t*>(t.m_data.data()), t.m_texture_type == TextureType::COLOR ? 1 : 0});
    d.n_textures = uint32_t(tex.size());
    d.textures = tex.empty() ? nullptr : tex.data();
    cwl::check(m_ctx, fh_scene_upload(m_ctx, &d), "fh_scene_upload");
  }
  fh_ctx* m_ctx = nullptr;
  uint32_t m_width = 0, m_height = 0;
  Scene m_scene;
};

}  // namespace fredholm
