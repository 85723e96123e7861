// fredholm/scene.h -- flat scene container with the public members of the reference's fredholm::Scene
// (fredholm/include/fredholm/scene.h:103-135) and a from-scratch Wavefront .obj/.mtl reader that fills them
// the way the reference's tinyobjloader path does (fredholm/src/scene.cpp:119-443): triangulated faces,
// face normals / barycentric texcoords when absent (:361-377), Kd -> base_color, Ks -> specular_color,
// Pr/Pm/Pc, d -> transmission = 1 - d (:245), Tf, Ke, and the custom keys diffuse, diffuse_roughness, sheen*,
// subsurface*, thin_walled (:183-284); coat_roughness takes clearcoat_thickness as in the reference (:240-242).
// Texture statements (map_Kd, map_Ks, map_Pr, map_Pm, map_bump/bump, norm, map_d) load PNG / binary PPM files through
// image_io.h with the reference's conventions (scene.cpp:7-37,144-153: vertical flip, one Texture per distinct file name, its
// first use fixes COLOR / NONCOLOR).  Not supported in this build: JPEG files and glTF.
#pragma once
#include <cmath>
#include <filesystem>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "camera.h"
#include "image_io.h"
#include "shared.h"

namespace fredholm
{

// fredholm/include/fredholm/scene.h:60-79
enum class TextureType { COLOR, NONCOLOR };

struct Texture {
  uint32_t m_width = 0;
  uint32_t m_height = 0;
  std::vector<uchar4> m_data;
  TextureType m_texture_type = TextureType::NONCOLOR;

  Texture() {}
  Texture(const std::filesystem::path& filepath, const TextureType& texture_type) : m_texture_type(texture_type)
  {
    const image_io::Image8 img = image_io::load_rgba8(filepath, /*flip_vertically=*/true);  // scene.cpp:15-16
    m_width = uint32_t(img.width);
    m_height = uint32_t(img.height);
    m_data.resize(size_t(m_width) * m_height);
    for (size_t i = 0; i < m_data.size(); ++i) m_data[i] = uchar4{img.rgba[4 * i], img.rgba[4 * i + 1], img.rgba[4 * i + 2], img.rgba[4 * i + 3]};
  }
};

struct FloatTexture {
  uint32_t m_width = 0;
  uint32_t m_height = 0;
  std::vector<float4> m_data;

  explicit FloatTexture(const std::filesystem::path& filepath)
  {
    const image_io::ImageF img = image_io::load_hdr(filepath);  // scene.cpp:44-45: no flip
    m_width = uint32_t(img.width);
    m_height = uint32_t(img.height);
    m_data.resize(size_t(m_width) * m_height);
    for (size_t i = 0; i < m_data.size(); ++i) m_data[i] = float4{img.rgba[4 * i], img.rgba[4 * i + 1], img.rgba[4 * i + 2], img.rgba[4 * i + 3]};
  }
};

struct Scene {
  bool m_has_camera_transform = false;
  Mat4 m_camera_transform = {};
  std::vector<float3> m_vertices = {};
  std::vector<uint3> m_indices = {};
  std::vector<float2> m_texcoords = {};
  std::vector<float3> m_normals = {};
  std::vector<uint> m_material_ids = {};
  std::vector<Material> m_materials;
  std::vector<Texture> m_textures;
  std::vector<uint> m_submesh_offsets = {};
  std::vector<uint> m_submesh_n_faces = {};
  std::vector<uint> m_instance_ids = {};
  std::vector<Mat4> m_transforms = {};

  bool is_valid() const { return !m_vertices.empty() && !m_indices.empty() && !m_normals.empty() && m_vertices.size() == m_normals.size(); }

  void clear()
  {
    m_vertices.clear(); m_indices.clear(); m_texcoords.clear(); m_normals.clear(); m_material_ids.clear(); m_materials.clear(); m_textures.clear();
    m_submesh_offsets.clear(); m_submesh_n_faces.clear(); m_instance_ids.clear(); m_transforms.clear();
  }

  void load_model(const std::filesystem::path& filepath, bool do_clear)
  {
    if (do_clear) clear();
    const std::string ext = filepath.extension().string();
    if (ext == ".obj") load_obj(filepath);
    else throw std::runtime_error("unsupported model format in this build: " + filepath.generic_string());
  }

  void update_animation(float /*time*/) {}  // .obj scenes have no animation (scene.cpp:862-898 handles glTF only)

  void load_obj(const std::filesystem::path& filepath)
  {
    std::ifstream in(filepath);
    if (!in) throw std::runtime_error("failed to load " + filepath.generic_string());
    std::vector<float3> pos, nrm;
    std::vector<float2> tex;
    std::map<std::string, int> mat_index, unique_textures;
    const size_t material_base = m_materials.size();
    int current_material = -1;
    bool need_default = false;
    const size_t first_face = m_indices.size();
    std::string line;
    auto resolve = [](int idx, size_t n) { return idx > 0 ? idx - 1 : int(n) + idx; };
    while (std::getline(in, line)) {
      std::istringstream ss(line);
      std::string tag;
      if (!(ss >> tag) || tag[0] == '#') continue;
      if (tag == "v") { float3 p; ss >> p.x >> p.y >> p.z; pos.push_back(p); }
      else if (tag == "vn") { float3 n; ss >> n.x >> n.y >> n.z; nrm.push_back(n); }
      else if (tag == "vt") { float2 t{0, 0}; ss >> t.x >> t.y; tex.push_back(t); }
      else if (tag == "mtllib") { std::string name; ss >> name; load_mtl(filepath.parent_path() / name, mat_index, unique_textures); }
      else if (tag == "usemtl") { std::string name; ss >> name; current_material = mat_index.count(name) ? mat_index[name] : -1; }
      else if (tag == "f") {
        struct Corner { int v, t, n; };
        std::vector<Corner> cs;
        std::string tok;
        while (ss >> tok) {
          Corner c{0, 0, 0};
          std::string part[3];
          int k = 0;
          for (char ch : tok) { if (ch == '/') { if (++k > 2) break; } else part[k] += ch; }
          c.v = part[0].empty() ? 0 : std::stoi(part[0]);
          c.t = part[1].empty() ? 0 : std::stoi(part[1]);
          c.n = part[2].empty() ? 0 : std::stoi(part[2]);
          cs.push_back(c);
        }
        for (size_t k = 1; k + 1 < cs.size(); ++k) {  // fan triangulation
          const Corner tri[3] = {cs[0], cs[k], cs[k + 1]};
          float3 p[3];
          for (int c = 0; c < 3; ++c) p[c] = pos.at(resolve(tri[c].v, pos.size()));
          const bool has_n = tri[0].n && tri[1].n && tri[2].n, has_t = tri[0].t && tri[1].t && tri[2].t;
          float3 fn{0, 0, 0};
          if (!has_n) {  // scene.cpp:361-371: normalize(cross(normalize(e1), normalize(e2)))
            const float3 e1 = norm3(sub3(p[1], p[0])), e2 = norm3(sub3(p[2], p[0]));
            fn = norm3(make_float3(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x));
          }
          const float2 bary[3] = {{0, 0}, {1, 0}, {0, 1}};
          const uint base = uint(m_vertices.size());
          for (int c = 0; c < 3; ++c) {
            m_vertices.push_back(p[c]);
            m_normals.push_back(has_n ? nrm.at(resolve(tri[c].n, nrm.size())) : fn);
            m_texcoords.push_back(has_t ? tex.at(resolve(tri[c].t, tex.size())) : bary[c]);
          }
          m_indices.push_back(make_uint3(base, base + 1, base + 2));
          if (current_material < 0) need_default = true;
          m_material_ids.push_back(current_material < 0 ? 0xffffffffu : uint(current_material));
          m_instance_ids.push_back(0);
        }
      }
    }
    if (need_default) {  // faces without usemtl: the reference indexes materials[-1]; give them a default material instead
      const uint id = uint(m_materials.size());
      m_materials.push_back(Material{});
      for (size_t f = first_face; f < m_material_ids.size(); ++f)
        if (m_material_ids[f] == 0xffffffffu) m_material_ids[f] = id;
    }
    (void)material_base;
    m_submesh_offsets.push_back(uint(first_face));
    m_submesh_n_faces.push_back(uint(m_indices.size() - first_face));
    if (m_transforms.empty()) m_transforms.push_back(Mat4{});
  }

 private:
  static float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
  static float3 norm3(float3 a)
  {
    const float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return make_float3(a.x / l, a.y / l, a.z / l);
  }

  void load_mtl(const std::filesystem::path& path, std::map<std::string, int>& mat_index, std::map<std::string, int>& unique_textures)
  {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("failed to load " + path.generic_string());
    Material* cur = nullptr;
    float clearcoat_thickness = 0.0f;
    std::string line;
    while (std::getline(in, line)) {
      std::istringstream ss(line);
      std::string tag;
      if (!(ss >> tag) || tag[0] == '#') continue;
      if (tag == "newmtl") {
        std::string name;
        ss >> name;
        mat_index[name] = int(m_materials.size());
        Material m;
        m.base_color = make_float3(0, 0, 0);      // tinyobjloader initialises Kd / Ks to 0 (scene.cpp:196, :210)
        m.specular_color = make_float3(0, 0, 0);
        m_materials.push_back(m);
        cur = &m_materials.back();
        clearcoat_thickness = 0.0f;
        continue;
      }
      if (!cur) continue;
      auto f1 = [&]() { float v = 0; ss >> v; return v; };
      auto f3 = [&]() { float3 v{0, 0, 0}; ss >> v.x >> v.y >> v.z; return v; };
      if (tag == "Kd") cur->base_color = f3();
      else if (tag == "Ks") cur->specular_color = f3();
      else if (tag == "Pr") { const float v = f1(); if (v > 0) cur->specular_roughness = v; }
      else if (tag == "Pm") cur->metalness = f1();
      else if (tag == "Pc") { clearcoat_thickness = f1(); if (clearcoat_thickness > 0) cur->coat = clearcoat_thickness; }
      else if (tag == "Pcr") { const float v = f1(); if (v > 0) cur->coat_roughness = clearcoat_thickness; }
      else if (tag == "d") cur->transmission = std::fmax(1.0f - f1(), 0.0f);
      else if (tag == "Tr") cur->transmission = std::fmax(f1(), 0.0f);
      else if (tag == "Tf") { const float3 v = f3(); if (v.x > 0 || v.y > 0 || v.z > 0) cur->transmission_color = v; }
      else if (tag == "Ke") { const float3 v = f3(); if (v.x > 0 || v.y > 0 || v.z > 0) { cur->emission = 1.0f; cur->emission_color = v; } }
      else if (tag == "diffuse") cur->diffuse = f1();
      else if (tag == "diffuse_roughness") cur->diffuse_roughness = f1();
      else if (tag == "sheen") cur->sheen = f1();
      else if (tag == "sheen_color") cur->sheen_color = f3();
      else if (tag == "sheen_roughness") cur->sheen_roughness = f1();
      else if (tag == "subsurface") cur->subsurface = f1();
      else if (tag == "subsurface_color") cur->subsurface_color = f3();
      else if (tag == "thin_walled") cur->thin_walled = f1();
      else if (tag == "map_Kd" || tag == "map_Ks" || tag == "map_Pr" || tag == "map_Pm" || tag == "map_bump" || tag == "map_Bump" || tag == "bump" || tag == "norm" || tag == "map_d") {
        std::string name, tok;
        while (ss >> tok) name = tok;  // options ("-bm 1") precede the file name
        const bool color = tag == "map_Kd" || tag == "map_Ks";
        if (!unique_textures.count(name)) {  // scene.cpp:144-153
          unique_textures[name] = int(m_textures.size());
          m_textures.push_back(Texture(path.parent_path() / name, color ? TextureType::COLOR : TextureType::NONCOLOR));
        }
        const int id = unique_textures[name];
        if (tag == "map_Kd") cur->base_color_texture_id = id;
        else if (tag == "map_Ks") cur->specular_color_texture_id = id;
        else if (tag == "map_Pr") cur->specular_roughness_texture_id = id;
        else if (tag == "map_Pm") cur->metalness_texture_id = id;
        else if (tag == "norm") cur->normalmap_texture_id = id;
        else if (tag == "map_d") cur->alpha_texture_id = id;
        else cur->heightmap_texture_id = id;
      } else if (tag.rfind("map_", 0) == 0)
        throw std::runtime_error(tag + " is not a texture slot of the reference's .mtl mapping (" + path.generic_string() + ")");
    }
  }
};

}  // namespace fredholm
