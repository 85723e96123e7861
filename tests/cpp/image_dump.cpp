// test helper: decode an image with include/fredholm/image_io.h and dump width, height and the raw texels to stdout
#include <cstdio>
#include <cstring>
#include <string>

#include "fredholm/image_io.h"

int main(int argc, char** argv)
{
  if (argc < 3) return 2;
  try {
    const std::string mode = argv[1];
    if (mode == "rewrite") {  // decode argv[2], encode it again with the PNG writer into argv[3]
      if (argc < 4) return 2;
      const fredholm::image_io::Image8 img = fredholm::image_io::load_rgba8(argv[2], false);
      fredholm::image_io::write_png_rgba8(argv[3], img.width, img.height, img.rgba.data());
    } else if (mode == "rgba8" || mode == "rgba8_flip") {
      const fredholm::image_io::Image8 img = fredholm::image_io::load_rgba8(argv[2], mode == "rgba8_flip");
      const int hdr[2] = {img.width, img.height};
      std::fwrite(hdr, sizeof hdr, 1, stdout);
      std::fwrite(img.rgba.data(), 1, img.rgba.size(), stdout);
    } else {
      const fredholm::image_io::ImageF img = fredholm::image_io::load_hdr(argv[2]);
      const int hdr[2] = {img.width, img.height};
      std::fwrite(hdr, sizeof hdr, 1, stdout);
      std::fwrite(img.rgba.data(), sizeof(float), img.rgba.size(), stdout);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
