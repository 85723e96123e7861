// rtcamp.cpp -- headless animation batch driver shaped like the reference's app/rtcamp8.cpp:47-303, against the drop-in headers
// of include/: for every frame { clear layers, init_render_states, set_time, render, (denoise: pass-through), post-process,
// copy to host } on a render thread, while a second thread converts finished frames to 8-bit and writes them as PNG files.
//
//   rtcamp --scene a.obj [--scene b.gltf ...] [--out DIR] [--width W --height H --spp N --depth D]
//          [--fps F --max-time T] [--bloom] [--sun] [--sky] [--ibl env.hdr] [--fov deg --F f --focus d]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "cwl/buffer.h"
#include "cwl/util.h"
#include "fredholm/camera.h"
#include "fredholm/denoiser.h"
#include "fredholm/image_io.h"
#include "fredholm/renderer.h"
#include "kernels/post-process.h"
#include "optwl/optwl.h"

int main(int argc, char** argv)
{
  std::vector<std::string> scene_files;
  std::string out_dir = "output", ibl;
  int width = 1920, height = 1080, n_spp = 16, max_depth = 5;
  float fps = 24.0f, max_time = 9.5f, fov_deg = 60.0f, F = 100.0f, focus = 8.0f;
  bool bloom = false, sun = false, sky = false, reference_launches = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value after %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
    if (a == "--scene") scene_files.push_back(next());
    else if (a == "--out") out_dir = next();
    else if (a == "--width") width = std::atoi(next());
    else if (a == "--height") height = std::atoi(next());
    else if (a == "--spp") n_spp = std::atoi(next());
    else if (a == "--depth") max_depth = std::atoi(next());
    else if (a == "--fps") fps = float(std::atof(next()));
    else if (a == "--max-time") max_time = float(std::atof(next()));
    else if (a == "--fov") fov_deg = float(std::atof(next()));
    else if (a == "--F") F = float(std::atof(next()));
    else if (a == "--focus") focus = float(std::atof(next()));
    else if (a == "--ibl") ibl = next();
    else if (a == "--bloom") bloom = true;
    else if (a == "--sun") sun = true;
    else if (a == "--sky") sky = true;
    else if (a == "--reference-launches") reference_launches = true;  // one launch of --spp samples per frame exactly as the reference computes it (firsthit quirk)
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  if (scene_files.empty()) { std::fprintf(stderr, "usage: %s --scene file.obj|file.gltf [--scene ...] [--out DIR] [--width W --height H --spp N --depth D] [--fps F --max-time T] [--bloom] [--sun] [--sky] [--ibl env.hdr]\n", argv[0]); return 2; }
  const float time_step = 1.0f / fps;
  try {
    std::filesystem::create_directories(out_dir);
    optwl::Context context;
    fredholm::Renderer renderer(context.get_context());
    renderer.create_module("pt.ptx");
    renderer.create_program_group();
    renderer.create_pipeline();
    renderer.set_resolution(uint32_t(width), uint32_t(height));
    if (reference_launches) renderer.set_reference_launch_semantics(true);

    const size_t n_px = size_t(width) * size_t(height);
    cwl::CUDABuffer<float4> layer_beauty(n_px), layer_position(n_px), layer_normal(n_px), layer_texcoord(n_px), layer_albedo(n_px), layer_denoised(n_px);
    cwl::CUDABuffer<float> layer_depth(n_px);
    cwl::CUDABuffer<float4> layer_denoised_pp(n_px), denoised_high_luminance(n_px), denoised_temp(n_px);
    // the post-process grid is floor(w/16) x floor(h/16) tiles (post-process.cu:9-11): border pixels of these buffers are never written
    // but the bloom taps read them, so give them a defined value once
    layer_denoised.clear(); layer_denoised_pp.clear(); denoised_high_luminance.clear(); denoised_temp.clear();
    fredholm::Denoiser denoiser(context.get_context(), uint32_t(width), uint32_t(height), layer_beauty.get_device_ptr(), layer_normal.get_device_ptr(), layer_albedo.get_device_ptr(),
                                layer_denoised.get_device_ptr(), false);

    for (size_t k = 0; k < scene_files.size(); ++k) renderer.load_scene(scene_files[k], k == 0);  // rtcamp8.cpp:114-115
    renderer.build_gas();
    renderer.build_ias();
    renderer.create_sbt();

    fredholm::Camera camera;
    camera.m_fov = fov_deg / 180.0f * float(M_PI);
    camera.m_F = F;
    camera.m_focus = focus;
    fredholm::RenderLayer render_layer{layer_beauty.get_device_ptr(), layer_position.get_device_ptr(), layer_depth.get_device_ptr(), layer_normal.get_device_ptr(),
                                       layer_texcoord.get_device_ptr(), layer_albedo.get_device_ptr()};
    if (sun) renderer.set_directional_light(make_float3(20, 20, 20), make_float3(-0.1f, 1, 0.1f), 1.0f);  // rtcamp8.cpp:133-134
    if (sky) renderer.load_arhosek_sky(3.0f, 0.3f);                                                       // rtcamp8.cpp:137
    if (!ibl.empty()) renderer.load_ibl(ibl);

    std::queue<std::pair<int, std::vector<float4>>> queue;
    std::mutex queue_mutex;
    bool render_finished = false;
    std::string failure;

    std::thread render_thread([&] {
      try {
        int frame_idx = 0;
        float time = 0.0f;
        for (;;) {
          if (time > max_time) break;
          const auto t0 = std::chrono::steady_clock::now();
          layer_beauty.clear(); layer_position.clear(); layer_normal.clear(); layer_depth.clear(); layer_texcoord.clear(); layer_albedo.clear();
          renderer.init_render_states();
          renderer.set_time(time);
          renderer.render(camera, make_float3(0, 0, 0), render_layer, uint32_t(n_spp), uint32_t(max_depth));
          CUDA_SYNC_CHECK();
          denoiser.denoise();
          PostProcessParams params{bloom, 2.0f, 5.0f, 80.0f, 1.0f};  // rtcamp8.cpp:57-60,207-212
          post_process_kernel_launch(layer_denoised.get_device_ptr(), denoised_high_luminance.get_device_ptr(), denoised_temp.get_device_ptr(), width, height, params,
                                     layer_denoised_pp.get_device_ptr());
          CUDA_SYNC_CHECK();
          std::vector<float4> image;
          layer_denoised_pp.copy_from_device_to_host(image);
          const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          std::printf("[Render] frame %d, time %.4f: %.2f ms\n", frame_idx, double(time), ms);
          { std::lock_guard<std::mutex> lock(queue_mutex); queue.push({frame_idx, std::move(image)}); }
          frame_idx++;
          time += time_step;
        }
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lock(queue_mutex);
        failure = e.what();
      }
      std::lock_guard<std::mutex> lock(queue_mutex);
      render_finished = true;
    });

    std::thread save_thread([&] {
      for (;;) {
        int frame_idx = -1;
        std::vector<float4> image;
        {
          std::lock_guard<std::mutex> lock(queue_mutex);
          if (queue.empty()) { if (render_finished) break; }
          else { frame_idx = queue.front().first; image = std::move(queue.front().second); queue.pop(); }
        }
        if (frame_idx < 0) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }
        std::vector<uint8_t> rgba(image.size() * 4);
        for (size_t i = 0; i < image.size(); ++i) {  // rtcamp8.cpp:266-279
          const float4& v = image[i];
          rgba[4 * i] = static_cast<unsigned char>(std::fmin(std::fmax(255.0f * v.x, 0.0f), 255.0f));
          rgba[4 * i + 1] = static_cast<unsigned char>(std::fmin(std::fmax(255.0f * v.y, 0.0f), 255.0f));
          rgba[4 * i + 2] = static_cast<unsigned char>(std::fmin(std::fmax(255.0f * v.z, 0.0f), 255.0f));
          rgba[4 * i + 3] = 255;
        }
        const std::filesystem::path file = std::filesystem::path(out_dir) / (std::to_string(frame_idx) + ".png");
        fredholm::image_io::write_png_rgba8(file, width, height, rgba.data());
        std::printf("[Image Write] %s saved\n", file.generic_string().c_str());
      }
    });

    render_thread.join();
    save_thread.join();
    if (!failure.empty()) throw std::runtime_error(failure);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
