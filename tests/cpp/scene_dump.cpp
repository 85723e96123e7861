// test helper: load model files with the drop-in fredholm::Scene and dump every array the renderer uploads
//   scene_dump out.bin time file1 [file2 ...]       (files after the first are appended: load_model(path, false))
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fredholm/scene.h"

template <typename T>
static void put(std::FILE* f, const std::vector<T>& v)
{
  const unsigned long long n = v.size() * sizeof(T);
  std::fwrite(&n, sizeof n, 1, f);
  if (n) std::fwrite(v.data(), 1, n, f);
}

int main(int argc, char** argv)
{
  if (argc < 4) return 2;
  try {
    fredholm::Scene scene;
    for (int i = 3; i < argc; ++i) scene.load_model(argv[i], i == 3);
    const float time = float(std::atof(argv[2]));
    if (time >= 0.0f) scene.update_animation(time);
    std::FILE* f = std::fopen(argv[1], "wb");
    if (!f) return 3;
    put(f, scene.m_vertices); put(f, scene.m_normals); put(f, scene.m_texcoords); put(f, scene.m_indices); put(f, scene.m_material_ids); put(f, scene.m_instance_ids);
    put(f, scene.m_materials);
    std::vector<float> o2w, w2o;
    scene.transforms_3x4(o2w, w2o);
    put(f, o2w); put(f, w2o);
    std::vector<float> cam;
    cam.push_back(scene.m_has_camera_transform ? 1.0f : 0.0f);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) cam.push_back(scene.m_camera_transform[c][r]);
    put(f, cam);
    put(f, scene.m_submesh_offsets); put(f, scene.m_submesh_n_faces);
    std::vector<unsigned> tex_hdr;
    for (const auto& t : scene.m_textures) { tex_hdr.push_back(t.m_width); tex_hdr.push_back(t.m_height); tex_hdr.push_back(t.m_texture_type == fredholm::TextureType::COLOR ? 1u : 0u); }
    put(f, tex_hdr);
    for (const auto& t : scene.m_textures) put(f, t.m_data);
    std::fclose(f);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
