// TEST SHIM: definitions for the declarations of tests/shims/app/stb_image_write.h so that the link check of the reference's applications resolves; never run.
extern "C" int stbi_write_png(char const*, int, int, int, const void*, int) { return 0; }
extern "C" int stbi_write_jpg(char const*, int, int, int, const void*, int) { return 0; }
extern "C" void stbi_flip_vertically_on_write(int) {}
