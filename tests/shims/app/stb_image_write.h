// TEST SHIM: the one stb_image_write entry the reference's applications call (rtcamp8.cpp), declared only -- the syntax check never links.
#pragma once
extern "C" int stbi_write_png(char const* filename, int w, int h, int comp, const void* data, int stride_in_bytes);
extern "C" int stbi_write_jpg(char const* filename, int w, int h, int comp, const void* data, int quality);
extern "C" void stbi_flip_vertically_on_write(int flag);
