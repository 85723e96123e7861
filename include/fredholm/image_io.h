// image_io.h -- image-file readers for the drop-in Scene / Renderer (host side, header only, no dependencies).
//
// The reference decodes images with stb_image (externals/stb, an empty submodule in the checkout):
//   Texture      (fredholm/src/scene.cpp:7-37):  stbi_load(..., STBI_rgb_alpha) with stbi_set_flip_vertically_on_load(true)
//   FloatTexture (fredholm/src/scene.cpp:39-66): stbi_loadf(..., STBI_rgb_alpha) without the flip (IBL)
// This header restates the published formats it needs from their specifications: PNG (W3C PNG 2nd ed.: zlib/deflate RFC 1950/1951,
// the five scanline filters, colour types 0/2/3/4/6 at 8 or 16 bits, non-interlaced), binary PPM/PGM, and Radiance RGBE .hdr
// (flat and new-style run-length scanlines).  JPEG and interlaced PNG are rejected with an exception, never decoded wrongly.
// Conversion conventions follow stb_image's documented behaviour: grey -> r=g=b, missing alpha -> 255 (1.0f for .hdr),
// 16-bit samples -> high byte, palette -> RGBA through PLTE/tRNS, .hdr texel = mantissa * 2^(exponent - 136).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace fredholm::image_io {

inline std::vector<uint8_t> read_file(const std::filesystem::path& path)
{
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("failed to load " + path.generic_string());
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// ---------------------------------------------------------------------------------------------- inflate (RFC 1951)
class Inflater {
 public:
  Inflater(const uint8_t* data, size_t size) : m_in(data), m_size(size) {}

  std::vector<uint8_t> run()
  {
    std::vector<uint8_t> out;
    bool last = false;
    while (!last) {
      last = bits(1) != 0;
      const uint32_t type = bits(2);
      if (type == 0) stored(out);
      else if (type == 1) { fixed_tables(); codes(out); }
      else if (type == 2) { dynamic_tables(); codes(out); }
      else throw std::runtime_error("inflate: invalid block type");
    }
    return out;
  }

 private:
  struct Huffman { uint16_t count[16]; uint16_t symbol[288]; };
  const uint8_t* m_in;
  size_t m_size, m_pos = 0;
  uint32_t m_bitbuf = 0;
  int m_bitcnt = 0;
  Huffman m_len{}, m_dist{};

  uint32_t bits(int need)
  {
    uint32_t val = m_bitbuf;
    while (m_bitcnt < need) {
      if (m_pos >= m_size) throw std::runtime_error("inflate: out of input");
      val |= (uint32_t)m_in[m_pos++] << m_bitcnt;
      m_bitcnt += 8;
    }
    m_bitbuf = need < 32 ? val >> need : 0;
    m_bitcnt -= need;
    return need < 32 ? val & ((1u << need) - 1u) : val;
  }

  void stored(std::vector<uint8_t>& out)
  {
    m_bitbuf = 0; m_bitcnt = 0;
    if (m_pos + 4 > m_size) throw std::runtime_error("inflate: out of input");
    const uint32_t len = m_in[m_pos] | (m_in[m_pos + 1] << 8), nlen = m_in[m_pos + 2] | (m_in[m_pos + 3] << 8);
    m_pos += 4;
    if ((len ^ 0xffffu) != nlen) throw std::runtime_error("inflate: stored block length mismatch");
    if (m_pos + len > m_size) throw std::runtime_error("inflate: out of input");
    out.insert(out.end(), m_in + m_pos, m_in + m_pos + len);
    m_pos += len;
  }

  static void construct(Huffman& h, const uint16_t* length, int n)
  {
    for (int i = 0; i < 16; ++i) h.count[i] = 0;
    for (int i = 0; i < n; ++i) h.count[length[i]]++;
    uint16_t offs[16];
    offs[1] = 0;
    for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + h.count[i];
    for (int i = 0; i < n; ++i)
      if (length[i]) h.symbol[offs[length[i]]++] = (uint16_t)i;
  }

  int decode(const Huffman& h)
  {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
      code |= (int)bits(1);
      const int count = h.count[len];
      if (code - count < first) return h.symbol[index + (code - first)];
      index += count;
      first += count;
      first <<= 1;
      code <<= 1;
    }
    throw std::runtime_error("inflate: invalid code");
  }

  void fixed_tables()
  {
    uint16_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    construct(m_len, l, 288);
    for (int i = 0; i < 30; ++i) l[i] = 5;
    construct(m_dist, l, 30);
  }

  void dynamic_tables()
  {
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
    if (nlen > 286 || ndist > 30) throw std::runtime_error("inflate: bad counts");
    uint16_t l[320];
    for (int i = 0; i < 19; ++i) l[i] = 0;
    for (int i = 0; i < ncode; ++i) l[order[i]] = (uint16_t)bits(3);
    Huffman lencode{};
    construct(lencode, l, 19);
    int idx = 0;
    while (idx < nlen + ndist) {
      int sym = decode(lencode);
      if (sym < 16) l[idx++] = (uint16_t)sym;
      else {
        uint16_t prev = 0;
        int rep;
        if (sym == 16) {
          if (idx == 0) throw std::runtime_error("inflate: repeat without a previous length");
          prev = l[idx - 1];
          rep = 3 + (int)bits(2);
        } else if (sym == 17) rep = 3 + (int)bits(3);
        else rep = 11 + (int)bits(7);
        if (idx + rep > nlen + ndist) throw std::runtime_error("inflate: too many lengths");
        while (rep--) l[idx++] = prev;
      }
    }
    if (l[256] == 0) throw std::runtime_error("inflate: no end-of-block code");
    construct(m_len, l, nlen);
    construct(m_dist, l + nlen, ndist);
  }

  void codes(std::vector<uint8_t>& out)
  {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
      int sym = decode(m_len);
      if (sym < 256) out.push_back((uint8_t)sym);
      else if (sym == 256) return;
      else {
        sym -= 257;
        if (sym >= 29) throw std::runtime_error("inflate: invalid length symbol");
        const int len = lbase[sym] + (int)bits(lext[sym]);
        const int ds = decode(m_dist);
        if (ds >= 30) throw std::runtime_error("inflate: invalid distance symbol");
        const size_t dist = dbase[ds] + bits(dext[ds]);
        if (dist > out.size()) throw std::runtime_error("inflate: distance too far back");
        const size_t from = out.size() - dist;
        for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
      }
    }
  }
};

inline std::vector<uint8_t> zlib_decompress(const uint8_t* data, size_t size)
{
  if (size < 6 || (data[0] & 0x0f) != 8 || ((data[0] << 8) | data[1]) % 31 != 0 || (data[1] & 0x20)) throw std::runtime_error("zlib: bad header");
  return Inflater(data + 2, size - 2).run();  // (the Adler-32 trailer is not verified; PNG chunks carry CRCs of their own)
}

// ---------------------------------------------------------------------------------------------- 8-bit images
struct Image8 { int width = 0, height = 0; std::vector<uint8_t> rgba; };  // row 0 = top row of the file

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline Image8 decode_png(const std::vector<uint8_t>& file)
{
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("png: bad signature");
  size_t pos = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1;
  std::vector<uint8_t> idat, plte, trns;
  bool end = false;
  while (!end && pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const std::string type((const char*)&file[pos + 4], 4);
    if (pos + 12 + (size_t)len > file.size()) throw std::runtime_error("png: truncated chunk");
    const uint8_t* d = &file[pos + 8];
    if (type == "IHDR") {
      if (len != 13) throw std::runtime_error("png: bad IHDR");
      w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9];
      if (d[10] != 0 || d[11] != 0) throw std::runtime_error("png: unknown compression/filter method");
      if (d[12] != 0) throw std::runtime_error("png: interlaced images are not supported");
    } else if (type == "PLTE") plte.assign(d, d + len);
    else if (type == "tRNS") trns.assign(d, d + len);
    else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
    else if (type == "IEND") end = true;
    pos += 12 + (size_t)len;
  }
  if (ctype < 0 || w == 0 || h == 0) throw std::runtime_error("png: missing IHDR");
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw std::runtime_error("png: bad colour type");
  }
  if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) throw std::runtime_error("png: bad bit depth");
  if (ctype == 3 && depth == 16) throw std::runtime_error("png: bad bit depth");
  const size_t bpp = (size_t)(channels * depth + 7) / 8;          // filter unit
  const size_t stride = ((size_t)w * channels * depth + 7) / 8;  // bytes per scanline
  std::vector<uint8_t> raw = zlib_decompress(idat.data(), idat.size());
  if (raw.size() < (stride + 1) * h) throw std::runtime_error("png: not enough image data");
  // undo the scanline filters in place
  std::vector<uint8_t> prev(stride, 0);
  for (uint32_t y = 0; y < h; ++y) {
    uint8_t* row = &raw[(stride + 1) * y];
    const uint8_t ft = row[0];
    uint8_t* cur = row + 1;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int pred;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: throw std::runtime_error("png: bad filter type");
      }
      cur[i] = (uint8_t)(cur[i] + pred);
    }
    std::memcpy(prev.data(), cur, stride);
  }
  Image8 img;
  img.width = (int)w; img.height = (int)h;
  img.rgba.resize((size_t)w * h * 4);
  auto sample = [&](const uint8_t* line, size_t idx) -> uint32_t {  // idx-th sample of a scanline, scaled to 8 bits except palette indices
    if (depth == 8) return line[idx];
    if (depth == 16) return line[2 * idx];
    const uint32_t per = 8u / depth, v = (line[idx / per] >> ((per - 1 - idx % per) * depth)) & ((1u << depth) - 1u);
    return ctype == 3 ? v : v * 255u / ((1u << depth) - 1u);
  };
  // tRNS colour key for grey / RGB images (compared on the unscaled samples as the specification says; 16-bit keys use both bytes)
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* line = &raw[(stride + 1) * y + 1];
    for (uint32_t x = 0; x < w; ++x) {
      uint8_t* o = &img.rgba[((size_t)y * w + x) * 4];
      if (ctype == 3) {
        const uint32_t pi = sample(line, x);
        if (3 * pi + 2 >= plte.size()) throw std::runtime_error("png: palette index out of range");
        o[0] = plte[3 * pi]; o[1] = plte[3 * pi + 1]; o[2] = plte[3 * pi + 2];
        o[3] = pi < trns.size() ? trns[pi] : 255;
      } else if (ctype == 0 || ctype == 4) {
        const uint32_t g = sample(line, (size_t)x * channels);
        o[0] = o[1] = o[2] = (uint8_t)g;
        o[3] = ctype == 4 ? (uint8_t)sample(line, (size_t)x * 2 + 1) : 255;
        if (ctype == 0 && trns.size() >= 2) {
          uint32_t rawv;
          if (depth == 16) rawv = (line[2 * x] << 8) | line[2 * x + 1];
          else if (depth == 8) rawv = line[x];
          else { const uint32_t per = 8u / depth; rawv = (line[x / per] >> ((per - 1 - x % per) * depth)) & ((1u << depth) - 1u); }
          if (rawv == (uint32_t)((trns[0] << 8) | trns[1])) o[3] = 0;
        }
      } else {
        o[0] = (uint8_t)sample(line, (size_t)x * channels);
        o[1] = (uint8_t)sample(line, (size_t)x * channels + 1);
        o[2] = (uint8_t)sample(line, (size_t)x * channels + 2);
        o[3] = ctype == 6 ? (uint8_t)sample(line, (size_t)x * 4 + 3) : 255;
        if (ctype == 2 && trns.size() >= 6) {
          bool key = true;
          for (int c = 0; c < 3; ++c) {
            const uint32_t rawv = depth == 16 ? (uint32_t)((line[2 * (3 * x + c)] << 8) | line[2 * (3 * x + c) + 1]) : line[3 * x + c];
            key = key && rawv == (uint32_t)((trns[2 * c] << 8) | trns[2 * c + 1]);
          }
          if (key) o[3] = 0;
        }
      }
    }
  }
  return img;
}

inline Image8 decode_pnm(const std::vector<uint8_t>& file)  // binary P5 / P6, maxval <= 255
{
  size_t pos = 0;
  auto token = [&]() {
    std::string t;
    for (;;) {
      while (pos < file.size() && std::isspace(file[pos])) ++pos;
      if (pos < file.size() && file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') ++pos; continue; }
      break;
    }
    while (pos < file.size() && !std::isspace(file[pos])) t.push_back((char)file[pos++]);
    return t;
  };
  const std::string magic = token();
  if (magic != "P5" && magic != "P6") throw std::runtime_error("pnm: only binary P5/P6 are supported");
  const int w = std::stoi(token()), h = std::stoi(token()), maxv = std::stoi(token());
  if (w <= 0 || h <= 0 || maxv <= 0 || maxv > 255) throw std::runtime_error("pnm: bad header");
  ++pos;  // the single whitespace byte after maxval
  const int ch = magic == "P6" ? 3 : 1;
  if (pos + (size_t)w * h * ch > file.size()) throw std::runtime_error("pnm: truncated");
  Image8 img;
  img.width = w; img.height = h;
  img.rgba.resize((size_t)w * h * 4);
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    const uint8_t* s = &file[pos + i * ch];
    uint8_t* o = &img.rgba[i * 4];
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((ch == 3 ? s[c] : s[0]) * 255 / maxv);
    o[3] = 255;
  }
  return img;
}

// stbi_load(path, ..., STBI_rgb_alpha); flip = stbi_set_flip_vertically_on_load
inline Image8 load_rgba8(const std::filesystem::path& path, bool flip_vertically)
{
  const std::vector<uint8_t> file = read_file(path);
  Image8 img;
  if (file.size() >= 8 && file[0] == 0x89 && file[1] == 'P') img = decode_png(file);
  else if (file.size() >= 2 && file[0] == 'P' && (file[1] == '5' || file[1] == '6')) img = decode_pnm(file);
  else throw std::runtime_error("failed to load " + path.generic_string() + ": only PNG and binary PPM/PGM images are supported in this build");
  if (flip_vertically) {
    const size_t rb = (size_t)img.width * 4;
    std::vector<uint8_t> tmp(rb);
    for (int y = 0; y < img.height / 2; ++y) {
      uint8_t* a = &img.rgba[(size_t)y * rb];
      uint8_t* b = &img.rgba[(size_t)(img.height - 1 - y) * rb];
      std::memcpy(tmp.data(), a, rb); std::memcpy(a, b, rb); std::memcpy(b, tmp.data(), rb);
    }
  }
  return img;
}

// ---------------------------------------------------------------------------------------------- Radiance .hdr
struct ImageF { int width = 0, height = 0; std::vector<float> rgba; };

inline ImageF load_hdr(const std::filesystem::path& path)  // stbi_loadf(path, ..., STBI_rgb_alpha), no flip
{
  const std::vector<uint8_t> f = read_file(path);
  size_t pos = 0;
  auto line = [&]() {
    std::string s;
    while (pos < f.size() && f[pos] != '\n') s.push_back((char)f[pos++]);
    ++pos;
    return s;
  };
  const std::string magic = line();
  if (magic != "#?RADIANCE" && magic != "#?RGBE") throw std::runtime_error("hdr: bad signature in " + path.generic_string());
  bool fmt = false;
  for (;;) {
    if (pos >= f.size()) throw std::runtime_error("hdr: truncated header");
    const std::string s = line();
    if (s.empty()) break;
    if (s == "FORMAT=32-bit_rle_rgbe") fmt = true;
  }
  if (!fmt) throw std::runtime_error("hdr: unsupported format");
  int w = 0, h = 0;
  {
    const std::string s = line();
    if (std::sscanf(s.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) throw std::runtime_error("hdr: unsupported data layout");
  }
  ImageF img;
  img.width = w; img.height = h;
  img.rgba.resize((size_t)w * h * 4);
  auto put = [&](size_t px, const uint8_t* rgbe) {
    float* o = &img.rgba[px * 4];
    if (rgbe[3] != 0) {
      const float s = std::ldexp(1.0f, (int)rgbe[3] - (128 + 8));
      o[0] = rgbe[0] * s; o[1] = rgbe[1] * s; o[2] = rgbe[2] * s;
    } else o[0] = o[1] = o[2] = 0.0f;
    o[3] = 1.0f;
  };
  std::vector<uint8_t> scan((size_t)w * 4);
  for (int y = 0; y < h; ++y) {
    bool rle = false;
    if (w >= 8 && w < 32768 && pos + 4 <= f.size() && f[pos] == 2 && f[pos + 1] == 2 && (f[pos + 2] & 0x80) == 0) {
      if (((f[pos + 2] << 8) | f[pos + 3]) != w) throw std::runtime_error("hdr: bad scanline width");
      rle = true;
      pos += 4;
    }
    if (rle) {
      for (int c = 0; c < 4; ++c) {
        int x = 0;
        while (x < w) {
          if (pos >= f.size()) throw std::runtime_error("hdr: truncated");
          int count = f[pos++];
          if (count > 128) {
            count -= 128;
            if (count == 0 || x + count > w || pos >= f.size()) throw std::runtime_error("hdr: bad run");
            const uint8_t v = f[pos++];
            for (int i = 0; i < count; ++i) scan[(size_t)(x++) * 4 + c] = v;
          } else {
            if (count == 0 || x + count > w || pos + count > f.size()) throw std::runtime_error("hdr: bad run");
            for (int i = 0; i < count; ++i) scan[(size_t)(x++) * 4 + c] = f[pos++];
          }
        }
      }
      for (int x = 0; x < w; ++x) put((size_t)y * w + x, &scan[(size_t)x * 4]);
    } else {
      if (pos + (size_t)w * 4 > f.size()) throw std::runtime_error("hdr: truncated");
      for (int x = 0; x < w; ++x) put((size_t)y * w + x, &f[pos + (size_t)x * 4]);
      pos += (size_t)w * 4;
    }
  }
  return img;
}

// ---------------------------------------------------------------------------------------------- PNG writer
// stbi_write_png's role in app/rtcamp8.cpp:287-289.  8-bit RGBA, filter 0, zlib stream of stored (uncompressed) deflate blocks:
// valid for every PNG reader, no compressor needed.
inline uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0)
{
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
  return ~crc;
}

inline void write_png_rgba8(const std::filesystem::path& path, int width, int height, const uint8_t* rgba)
{
  std::vector<uint8_t> raw;
  raw.reserve((size_t(width) * 4 + 1) * size_t(height));
  for (int y = 0; y < height; ++y) {
    raw.push_back(0);
    raw.insert(raw.end(), rgba + size_t(y) * width * 4, rgba + size_t(y + 1) * width * 4);
  }
  std::vector<uint8_t> z = {0x78, 0x01};
  uint32_t a = 1, b = 0;  // Adler-32
  for (size_t pos = 0; pos < raw.size() || pos == 0;) {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n >= raw.size() ? 1 : 0);
    z.push_back(uint8_t(n & 0xff)); z.push_back(uint8_t(n >> 8));
    z.push_back(uint8_t(~n & 0xff)); z.push_back(uint8_t((~n >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + long(pos), raw.begin() + long(pos + n));
    for (size_t i = pos; i < pos + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    pos += n;
    if (n == 0) break;
  }
  const uint32_t adler = (b << 16) | a;
  for (int k = 3; k >= 0; --k) z.push_back(uint8_t(adler >> (8 * k)));
  std::ofstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("failed to write " + path.generic_string());
  auto be = [](uint32_t v, uint8_t* o) { o[0] = uint8_t(v >> 24); o[1] = uint8_t(v >> 16); o[2] = uint8_t(v >> 8); o[3] = uint8_t(v); };
  auto chunk = [&](const char* type, const std::vector<uint8_t>& body) {
    std::vector<uint8_t> c(8 + body.size() + 4);
    be(uint32_t(body.size()), c.data());
    std::memcpy(c.data() + 4, type, 4);
    if (!body.empty()) std::memcpy(c.data() + 8, body.data(), body.size());
    be(crc32(c.data() + 4, 4 + body.size()), c.data() + 8 + body.size());
    f.write(reinterpret_cast<const char*>(c.data()), long(c.size()));
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  f.write(reinterpret_cast<const char*>(sig), 8);
  std::vector<uint8_t> ihdr(13);
  be(uint32_t(width), ihdr.data()); be(uint32_t(height), ihdr.data() + 4);
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk("IHDR", ihdr);
  chunk("IDAT", z);
  chunk("IEND", {});
}

}  // namespace fredholm::image_io
