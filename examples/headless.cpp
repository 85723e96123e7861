// examples/headless.cpp -- a headless batch driver shaped like the reference's app/rtcamp8.cpp:47-303
// (load scene, set sky, per-frame: clear layers, init_render_states, render N spp, post-process, copy back, write a
// PPM), written against the drop-in facade in include/.  Build:
//   g++ -std=c++17 -Iinclude examples/headless.cpp -Lfredholm_amd -lfredholm_hip -Wl,-rpath,$PWD/fredholm_amd -o headless
#include <cstdio>
#include <fstream>
#include <vector>

#include "cwl/buffer.h"
#include "fredholm/renderer.h"
#include "kernels/post-process.h"
#include "optwl/optwl.h"

int main(int argc, char** argv)
{
  if (argc < 2) { std::fprintf(stderr, "usage: %s scene.obj [out.ppm] [width height spp depth] [ibl.hdr]\n", argv[0]); return 2; }
  const char* out_path = argc > 2 ? argv[2] : "out.ppm";
  const uint32_t width = argc > 3 ? std::atoi(argv[3]) : 512, height = argc > 4 ? std::atoi(argv[4]) : 512;
  const uint32_t n_spp = argc > 5 ? std::atoi(argv[5]) : 16, max_depth = argc > 6 ? std::atoi(argv[6]) : 5;
  try {
    optwl::Context context;
    fredholm::Camera camera(make_float3(0.0f, 1.0f, 1.0f), 0.5f * float(M_PI), 100.0f, 10000.0f);
    fredholm::Renderer renderer(context.get_context());
    renderer.create_module("pt.ptx");
    renderer.create_program_group();
    renderer.create_pipeline();
    renderer.set_resolution(width, height);
    renderer.load_scene(argv[1]);
    renderer.build_gas();
    renderer.build_ias();
    renderer.create_sbt();
    if (argc > 7) renderer.load_ibl(argv[7]);  // rtcamp8.cpp loads its environment the same way

    cwl::CUDABuffer<float4> beauty(width * height), position(width * height), normal(width * height), texcoord(width * height), albedo(width * height);
    cwl::CUDABuffer<float> depth(width * height);
    cwl::CUDABuffer<float4> hi(width * height), tmp(width * height), pp(width * height);
    beauty.clear(); position.clear(); normal.clear(); texcoord.clear(); albedo.clear(); depth.clear();
    renderer.init_render_states();

    fredholm::RenderLayer layer{beauty.get_device_ptr(), position.get_device_ptr(), depth.get_device_ptr(), normal.get_device_ptr(), texcoord.get_device_ptr(), albedo.get_device_ptr()};
    renderer.render(camera, make_float3(0, 0, 0), layer, n_spp, max_depth);
    renderer.wait_for_completion();

    PostProcessParams params{false, 2.0f, 5.0f, 80.0f, 1.0f};
    post_process_kernel_launch(beauty.get_device_ptr(), hi.get_device_ptr(), tmp.get_device_ptr(), int(width), int(height), params, pp.get_device_ptr());
    CUDA_SYNC_CHECK();

    std::vector<float4> img;
    pp.copy_from_device_to_host(img);
    std::ofstream out(out_path, std::ios::binary);
    out << "P6\n" << width << " " << height << "\n255\n";
    for (const float4& p : img) {
      const unsigned char rgb[3] = {(unsigned char)(255.0f * std::fmin(std::fmax(p.x, 0.0f), 1.0f)), (unsigned char)(255.0f * std::fmin(std::fmax(p.y, 0.0f), 1.0f)),
                                    (unsigned char)(255.0f * std::fmin(std::fmax(p.z, 0.0f), 1.0f))};
      out.write(reinterpret_cast<const char*>(rgb), 3);
    }
    std::printf("wrote %s (%ux%u, %u spp, depth %u)\n", out_path, width, height, n_spp, max_depth);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
