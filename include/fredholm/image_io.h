// image_io.h -- image-file readers for the drop-in Scene / Renderer (host side, header only, no dependencies).
//
// The reference decodes images with stb_image (externals/stb, an empty submodule in the checkout):
//   Texture      (fredholm/src/scene.cpp:7-37):  stbi_load(..., STBI_rgb_alpha) with stbi_set_flip_vertically_on_load(true)
//   FloatTexture (fredholm/src/scene.cpp:39-66): stbi_loadf(..., STBI_rgb_alpha) without the flip (IBL)
// This header restates the published formats it needs from their specifications: PNG (W3C PNG 2nd ed.: zlib/deflate RFC 1950/1951,
// the five scanline filters, colour types 0/2/3/4/6 at 8 or 16 bits, non-interlaced), JPEG (ITU-T T.81 Huffman DCT: sequential and progressive),
// binary PPM/PGM, and Radiance RGBE .hdr (flat and new-style run-length scanlines).  Arithmetic-coded JPEG and interlaced PNG are
// rejected with an exception, never decoded wrongly.
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

// ---------------------------------------------------------------------------------------------- JPEG (ITU-T T.81)
// Sequential (SOF0 / SOF1) and progressive (SOF2: spectral selection + successive approximation, end-of-band runs) Huffman DCT frames, 8-bit,
// 1 or 3 components, sampling factors 1 and 2, restart intervals, JFIF YCbCr.  Arithmetic-coded, lossless, 12-bit and 4-component files are rejected.  Everything after entropy decoding is integer arithmetic,
// so that this decoder and fredholm_amd/image_io.py produce identical bytes:
//   IDCT      Loeffler-Ligtenberg-Moschytz 1-D flow graph, 13-bit constants, two passes (the scaling of the IJG "slow integer" IDCT)
//   upsample  triangle filter: 3/4 nearer + 1/4 farther sample per axis, rounding constants 1,2 (one axis) and 8,7 (two axes)
//   colour    R = Y + 1.402 Cr', G = Y - 0.344136 Cb' - 0.714136 Cr', B = Y + 1.772 Cb' in 16-bit fixed point, clamped
// stb_image, which the reference uses, has its own IDCT and resampling code: decoded texels can differ by a unit in the last place.
class JpegDecoder {
 public:
  explicit JpegDecoder(const std::vector<uint8_t>& file) : m_f(file) {}

  Image8 run()
  {
    if (m_f.size() < 4 || m_f[0] != 0xff || m_f[1] != 0xd8) throw std::runtime_error("jpeg: bad signature");
    m_pos = 2;
    bool seen_scan = false;
    for (;;) {
      const int marker = next_marker();
      if (marker == 0xd9) {
        if (m_progressive && seen_scan) return finish_progressive();
        throw std::runtime_error("jpeg: no image data");
      }
      if (marker >= 0xd0 && marker <= 0xd7) continue;  // a restart marker left over at the end of a scan
      if (marker == 0xda) {
        if (!m_progressive) break;
        scan_progressive();
        seen_scan = true;
        continue;
      }
      const size_t len = seg_length();
      const uint8_t* d = &m_f[m_pos + 2];
      const size_t n = len - 2;
      if (marker == 0xc0 || marker == 0xc1 || marker == 0xc2) {
        if (!m_comp.empty()) throw std::runtime_error("jpeg: more than one frame");
        frame(d, n);
        if (marker == 0xc2) begin_progressive();
      }
      else if (marker >= 0xc3 && marker <= 0xcf && marker != 0xc4 && marker != 0xc8 && marker != 0xcc) throw std::runtime_error("jpeg: unsupported coding process");
      else if (marker == 0xc4) huffman_tables(d, n);
      else if (marker == 0xdb) quant_tables(d, n);
      else if (marker == 0xdd) { if (n < 2) throw std::runtime_error("jpeg: bad DRI"); m_restart = (d[0] << 8) | d[1]; }
      else if (marker == 0xee && n >= 12 && std::memcmp(d, "Adobe", 5) == 0) { m_adobe = true; m_adobe_transform = d[11]; }
      m_pos += len;
    }
    scan();
    return finish();
  }

 private:
  struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0, bw = 0, bh = 0;
    std::vector<uint8_t> plane;
    int nbx = 0, nby = 0;        // progressive: blocks a scan of this component alone covers (its own size, not the MCU-padded one)
    std::vector<int> coef;       // progressive: quantised coefficients of every block of the padded grid, natural order
  };
  bool m_progressive = false;
  struct Table { uint16_t count[17] = {}; uint8_t symbol[256] = {}; bool set = false; };
  const std::vector<uint8_t>& m_f;
  size_t m_pos = 0;
  int m_width = 0, m_height = 0, m_restart = 0, m_hmax = 1, m_vmax = 1;
  bool m_adobe = false;
  int m_adobe_transform = 0;
  std::vector<Component> m_comp;
  uint16_t m_q[4][64] = {};
  bool m_q_set[4] = {false, false, false, false};
  Table m_dc[4], m_ac[4];
  uint32_t m_bits = 0;
  int m_nbits = 0;
  bool m_hit_marker = false;

  int next_marker()
  {
    while (m_pos + 1 < m_f.size()) {
      if (m_f[m_pos] != 0xff) { ++m_pos; continue; }
      const int m = m_f[m_pos + 1];
      if (m == 0x00 || m == 0xff) { ++m_pos; continue; }
      m_pos += 2;
      return m;
    }
    throw std::runtime_error("jpeg: truncated");
  }
  size_t seg_length() const
  {
    if (m_pos + 2 > m_f.size()) throw std::runtime_error("jpeg: truncated");
    const size_t len = (size_t(m_f[m_pos]) << 8) | m_f[m_pos + 1];
    if (len < 2 || m_pos + len > m_f.size()) throw std::runtime_error("jpeg: bad segment length");
    return len;
  }
  void frame(const uint8_t* d, size_t n)
  {
    if (n < 6 || d[0] != 8) throw std::runtime_error("jpeg: only 8-bit samples are supported");
    m_height = (d[1] << 8) | d[2];
    m_width = (d[3] << 8) | d[4];
    const int nc = d[5];
    if (m_width <= 0 || m_height <= 0) throw std::runtime_error("jpeg: bad dimensions");
    if (nc != 1 && nc != 3) throw std::runtime_error("jpeg: only 1- and 3-component images are supported");
    if (n < size_t(6 + 3 * nc)) throw std::runtime_error("jpeg: bad SOF");
    m_comp.assign(size_t(nc), Component());
    for (int i = 0; i < nc; ++i) {
      Component& c = m_comp[size_t(i)];
      c.id = d[6 + 3 * i]; c.h = d[7 + 3 * i] >> 4; c.v = d[7 + 3 * i] & 15; c.tq = d[8 + 3 * i];
      if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) throw std::runtime_error("jpeg: unsupported sampling factors");
      m_hmax = std::max(m_hmax, c.h); m_vmax = std::max(m_vmax, c.v);
    }
    if (nc == 1) { m_comp[0].h = m_comp[0].v = 1; m_hmax = m_vmax = 1; }
    for (Component& c : m_comp)
      if (m_hmax % c.h || m_vmax % c.v) throw std::runtime_error("jpeg: unsupported sampling factors");
  }
  void quant_tables(const uint8_t* d, size_t n)
  {
    static const uint8_t zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    size_t p = 0;
    while (p < n) {
      const int pq = d[p] >> 4, tq = d[p] & 15;
      ++p;
      if (tq > 3 || pq > 1 || p + size_t(64 * (pq + 1)) > n) throw std::runtime_error("jpeg: bad DQT");
      for (int i = 0; i < 64; ++i) { m_q[tq][zz[i]] = pq ? uint16_t((d[p] << 8) | d[p + 1]) : d[p]; p += size_t(pq + 1); }
      m_q_set[tq] = true;
    }
  }
  void huffman_tables(const uint8_t* d, size_t n)
  {
    size_t p = 0;
    while (p < n) {
      if (p + 17 > n) throw std::runtime_error("jpeg: bad DHT");
      const int tc = d[p] >> 4, th = d[p] & 15;
      if (tc > 1 || th > 3) throw std::runtime_error("jpeg: bad DHT");
      Table& t = tc ? m_ac[th] : m_dc[th];
      int total = 0;
      for (int i = 1; i <= 16; ++i) { t.count[i] = d[p + size_t(i)]; total += t.count[i]; }
      p += 17;
      if (total > 256 || p + size_t(total) > n) throw std::runtime_error("jpeg: bad DHT");
      std::memcpy(t.symbol, d + p, size_t(total));
      t.set = true;
      p += size_t(total);
    }
  }
  // entropy-coded segment: 0xff 0x00 is a stuffed 0xff; any other marker ends the data (zero bits follow)
  int bit()
  {
    if (m_nbits == 0) {
      uint8_t b = 0;
      if (!m_hit_marker && m_pos < m_f.size()) {
        b = m_f[m_pos];
        if (b == 0xff) {
          const uint8_t b2 = m_pos + 1 < m_f.size() ? m_f[m_pos + 1] : 0xd9;
          if (b2 == 0x00) m_pos += 2;
          else { m_hit_marker = true; b = 0; }
        } else ++m_pos;
      }
      m_bits = b;
      m_nbits = 8;
    }
    --m_nbits;
    return int((m_bits >> m_nbits) & 1u);
  }
  int receive(int n) { int v = 0; for (int i = 0; i < n; ++i) v = (v << 1) | bit(); return v; }
  static int extend(int v, int n) { return n == 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); }
  int decode(const Table& t)
  {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 16; ++len) {
      code |= bit();
      const int count = t.count[len];
      if (code - count < first) return t.symbol[index + (code - first)];
      index += count;
      first = (first + count) << 1;
      code <<= 1;
    }
    throw std::runtime_error("jpeg: bad Huffman code");
  }
  void restart()
  {
    m_nbits = 0;
    m_hit_marker = false;
    // skip to the RSTn marker
    while (m_pos + 1 < m_f.size() && !(m_f[m_pos] == 0xff && m_f[m_pos + 1] >= 0xd0 && m_f[m_pos + 1] <= 0xd7)) ++m_pos;
    if (m_pos + 1 < m_f.size()) m_pos += 2;
    for (Component& c : m_comp) c.pred = 0;
  }

  static void idct(const int* in, uint8_t* out, int stride)  // in: dequantised coefficients, natural order
  {
    constexpr int CB = 13, P1 = 2;
    constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819,
                  F2562 = 20995, F3072 = 25172;
    int ws[64];
    for (int pass = 0; pass < 2; ++pass) {
      for (int i = 0; i < 8; ++i) {
        const int* s = pass == 0 ? in + i : ws + 8 * i;
        const int st = pass == 0 ? 8 : 1;
        const long d0 = s[0], d1 = s[st], d2 = s[2 * st], d3 = s[3 * st], d4 = s[4 * st], d5 = s[5 * st], d6 = s[6 * st], d7 = s[7 * st];
        long z1 = (d2 + d6) * F0541;
        const long t2 = z1 + d6 * (-F1847), t3 = z1 + d2 * F0765;
        const long t0 = (d0 + d4) << CB, t1 = (d0 - d4) << CB;
        const long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        long a0 = d7, a1 = d5, a2 = d3, a3 = d1;
        z1 = a0 + a3;
        long z2 = a1 + a2, z3 = a0 + a2, z4 = a1 + a3;
        const long z5 = (z3 + z4) * F1175;
        a0 *= F0298; a1 *= F2053; a2 *= F3072; a3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        a0 += z1 + z3; a1 += z2 + z4; a2 += z2 + z3; a3 += z1 + z4;
        const int sh = pass == 0 ? CB - P1 : CB + P1 + 3;
        const long rnd = 1L << (sh - 1);
        const long o[8] = {t10 + a3, t11 + a2, t12 + a1, t13 + a0, t13 - a0, t12 - a1, t11 - a2, t10 - a3};
        for (int k = 0; k < 8; ++k) {
          const long v = (o[k] + rnd) >> sh;
          if (pass == 0) ws[8 * k + i] = int(v);
          else { const long px = v + 128; out[i * stride + k] = uint8_t(px < 0 ? 0 : (px > 255 ? 255 : px)); }
        }
      }
    }
  }

  void scan()
  {
    const size_t len = seg_length();
    const uint8_t* d = &m_f[m_pos + 2];
    if (m_comp.empty()) throw std::runtime_error("jpeg: scan before frame");
    const int ns = d[0];
    if (ns != int(m_comp.size()) || len < size_t(6 + 2 * ns)) throw std::runtime_error("jpeg: non-interleaved scans are not supported");
    for (int i = 0; i < ns; ++i) {
      Component* c = nullptr;
      for (Component& cc : m_comp) if (cc.id == d[1 + 2 * i]) c = &cc;
      if (!c) throw std::runtime_error("jpeg: bad scan component");
      c->td = d[2 + 2 * i] >> 4; c->ta = d[2 + 2 * i] & 15;
      if (c->td > 3 || c->ta > 3 || !m_dc[c->td].set || !m_ac[c->ta].set || !m_q_set[c->tq]) throw std::runtime_error("jpeg: missing table");
    }
    m_pos += len;
    static const uint8_t zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    const int mcu_w = 8 * m_hmax, mcu_h = 8 * m_vmax;
    const int mx = (m_width + mcu_w - 1) / mcu_w, my = (m_height + mcu_h - 1) / mcu_h;
    for (Component& c : m_comp) {
      c.bw = mx * c.h * 8; c.bh = my * c.v * 8;
      c.plane.assign(size_t(c.bw) * size_t(c.bh), 0);
      c.pred = 0;
    }
    int count = 0;
    for (int y = 0; y < my; ++y)
      for (int x = 0; x < mx; ++x) {
        if (m_restart && count && count % m_restart == 0) restart();
        ++count;
        for (Component& c : m_comp)
          for (int by = 0; by < c.v; ++by)
            for (int bx = 0; bx < c.h; ++bx) {
              int coef[64] = {0};
              const int t = decode(m_dc[c.td]);
              if (t > 11) throw std::runtime_error("jpeg: bad DC size");
              c.pred += extend(receive(t), t);
              coef[0] = c.pred * m_q[c.tq][0];
              for (int k = 1; k < 64;) {
                const int rs = decode(m_ac[c.ta]);
                const int r = rs >> 4, sz = rs & 15;
                if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                k += r;
                if (k > 63) throw std::runtime_error("jpeg: bad AC run");
                coef[zz[k]] = extend(receive(sz), sz) * m_q[c.tq][zz[k]];
                ++k;
              }
              idct(coef, &c.plane[size_t((y * c.v + by) * 8) * size_t(c.bw) + size_t((x * c.h + bx) * 8)], c.bw);
            }
      }
  }

  // ---- progressive DCT (SOF2, annex G): every scan adds a band of coefficients (spectral selection) or one more bit of them (successive
  // approximation) to the coefficient store; dequantisation and the IDCT run once, after the last scan
  void begin_progressive()
  {
    m_progressive = true;
    const int mcu_w = 8 * m_hmax, mcu_h = 8 * m_vmax;
    const int mx = (m_width + mcu_w - 1) / mcu_w, my = (m_height + mcu_h - 1) / mcu_h;
    for (Component& c : m_comp) {
      c.bw = mx * c.h * 8; c.bh = my * c.v * 8;
      const int cw = (m_width * c.h + m_hmax - 1) / m_hmax, ch = (m_height * c.v + m_vmax - 1) / m_vmax;
      c.nbx = (cw + 7) / 8; c.nby = (ch + 7) / 8;
      c.coef.assign(size_t(c.bw / 8) * size_t(c.bh / 8) * 64, 0);
    }
  }
  static const uint8_t* zigzag()
  {
    static const uint8_t zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return zz;
  }
  void refine_nonzero(int& v, int p1, int m1)
  {
    if (bit() && (v & p1) == 0) v += v >= 0 ? p1 : m1;
  }
  void scan_progressive()
  {
    const size_t len = seg_length();
    const uint8_t* d = &m_f[m_pos + 2];
    if (m_comp.empty()) throw std::runtime_error("jpeg: scan before frame");
    const int ns = d[0];
    if (ns < 1 || ns > int(m_comp.size()) || len < size_t(6 + 2 * ns)) throw std::runtime_error("jpeg: bad SOS");
    Component* sc[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < ns; ++i) {
      for (Component& cc : m_comp) if (cc.id == d[1 + 2 * i]) sc[i] = &cc;
      if (!sc[i]) throw std::runtime_error("jpeg: bad scan component");
      sc[i]->td = d[2 + 2 * i] >> 4; sc[i]->ta = d[2 + 2 * i] & 15;
    }
    const int Ss = d[1 + 2 * ns], Se = d[2 + 2 * ns], Ah = d[3 + 2 * ns] >> 4, Al = d[3 + 2 * ns] & 15;
    if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13 || (Ah && Ah != Al + 1)) throw std::runtime_error("jpeg: bad progressive scan parameters");
    for (int i = 0; i < ns; ++i)
      if (sc[i]->td > 3 || sc[i]->ta > 3 || (Ss == 0 && Ah == 0 && !m_dc[sc[i]->td].set) || (Ss > 0 && !m_ac[sc[i]->ta].set)) throw std::runtime_error("jpeg: missing table");
    m_pos += len;
    m_nbits = 0;
    m_hit_marker = false;
    for (Component& c : m_comp) c.pred = 0;
    const uint8_t* zz = zigzag();
    const int mcu_w = 8 * m_hmax, mcu_h = 8 * m_vmax;
    const int mx = (m_width + mcu_w - 1) / mcu_w, my = (m_height + mcu_h - 1) / mcu_h;
    const int ux = ns > 1 ? mx : sc[0]->nbx, uy = ns > 1 ? my : sc[0]->nby;
    const int p1 = 1 << Al, m1 = -(1 << Al);
    int eobrun = 0, count = 0;
    auto block = [&](Component& c, int by, int bx) {
      int* blk = &c.coef[(size_t(by) * size_t(c.bw / 8) + size_t(bx)) * 64];
      if (Ss == 0) {
        if (Ah == 0) {
          const int t = decode(m_dc[c.td]);
          if (t > 11) throw std::runtime_error("jpeg: bad DC size");
          c.pred += extend(receive(t), t);
          blk[0] = c.pred * (1 << Al);
        } else if (bit()) blk[0] |= p1;
        return;
      }
      const Table& tab = m_ac[c.ta];
      if (Ah == 0) {
        if (eobrun) { --eobrun; return; }
        for (int k = Ss; k <= Se;) {
          const int rs = decode(tab);
          const int r = rs >> 4, sz = rs & 15;
          if (sz == 0) {
            if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += receive(r); break; }
            k += 16;
            continue;
          }
          k += r;
          if (k > Se) throw std::runtime_error("jpeg: bad AC run");
          blk[zz[k]] = extend(receive(sz), sz) * (1 << Al);
          ++k;
        }
        return;
      }
      int k = Ss;  // refinement of an AC band (annex G.2.3)
      if (eobrun == 0) {
        for (; k <= Se; ++k) {
          const int rs = decode(tab);
          int r = rs >> 4;
          const int sz = rs & 15;
          int value = 0;
          if (sz) {
            if (sz != 1) throw std::runtime_error("jpeg: bad refinement code");
            value = bit() ? p1 : m1;
          } else if (r != 15) {
            eobrun = 1 << r;
            if (r) eobrun += receive(r);
            break;
          }
          for (; k <= Se; ++k) {
            int& v = blk[zz[k]];
            if (v != 0) refine_nonzero(v, p1, m1);
            else if (--r < 0) break;
          }
          if (value) {
            if (k > Se) throw std::runtime_error("jpeg: bad AC run");
            blk[zz[k]] = value;
          }
        }
      }
      if (eobrun > 0) {
        for (; k <= Se; ++k) {
          int& v = blk[zz[k]];
          if (v != 0) refine_nonzero(v, p1, m1);
        }
        --eobrun;
      }
    };
    for (int y = 0; y < uy; ++y)
      for (int x = 0; x < ux; ++x) {
        if (m_restart && count && count % m_restart == 0) { restart(); eobrun = 0; }
        ++count;
        if (ns > 1) {
          for (int i = 0; i < ns; ++i)
            for (int by = 0; by < sc[i]->v; ++by)
              for (int bx = 0; bx < sc[i]->h; ++bx) block(*sc[i], y * sc[i]->v + by, x * sc[i]->h + bx);
        } else block(*sc[0], y, x);
      }
  }
  Image8 finish_progressive()
  {
    for (Component& c : m_comp) {
      if (!m_q_set[c.tq]) throw std::runtime_error("jpeg: missing table");
      c.plane.assign(size_t(c.bw) * size_t(c.bh), 0);
      for (int by = 0; by < c.bh / 8; ++by)
        for (int bx = 0; bx < c.bw / 8; ++bx) {
          const int* q = &c.coef[(size_t(by) * size_t(c.bw / 8) + size_t(bx)) * 64];
          int coef[64];
          for (int i = 0; i < 64; ++i) coef[i] = q[i] * m_q[c.tq][i];
          idct(coef, &c.plane[size_t(by * 8) * size_t(c.bw) + size_t(bx * 8)], c.bw);
        }
    }
    return finish();
  }

  // full-resolution plane of a component (triangle-filter upsampling)
  std::vector<uint8_t> upsampled(const Component& c) const
  {
    const int fx = m_hmax / c.h, fy = m_vmax / c.v;
    const int W = c.bw * fx, H = c.bh * fy;
    if (fx == 1 && fy == 1) return c.plane;
    std::vector<uint8_t> out(size_t(W) * size_t(H));
    auto at = [&](int x, int y) { x = x < 0 ? 0 : (x >= c.bw ? c.bw - 1 : x); y = y < 0 ? 0 : (y >= c.bh ? c.bh - 1 : y); return int(c.plane[size_t(y) * size_t(c.bw) + size_t(x)]); };
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        int v;
        if (fx == 2 && fy == 1) {
          const int sx = x >> 1;
          v = (x & 1) ? (3 * at(sx, y) + at(sx + 1, y) + 2) >> 2 : (3 * at(sx, y) + at(sx - 1, y) + 1) >> 2;
        } else if (fx == 1 && fy == 2) {
          const int sy = y >> 1;
          v = (y & 1) ? (3 * at(x, sy) + at(x, sy + 1) + 2) >> 2 : (3 * at(x, sy) + at(x, sy - 1) + 1) >> 2;
        } else {
          const int sx = x >> 1, sy = y >> 1, oy = (y & 1) ? sy + 1 : sy - 1, ox = (x & 1) ? sx + 1 : sx - 1;
          const int near_col = 3 * at(sx, sy) + at(sx, oy), far_col = 3 * at(ox, sy) + at(ox, oy);
          v = (3 * near_col + far_col + ((x & 1) ? 7 : 8)) >> 4;
        }
        out[size_t(y) * size_t(W) + size_t(x)] = uint8_t(v);
      }
    return out;
  }

  Image8 finish() const
  {
    Image8 img;
    img.width = m_width; img.height = m_height;
    img.rgba.resize(size_t(m_width) * size_t(m_height) * 4);
    if (m_comp.size() == 1) {
      const Component& c = m_comp[0];
      for (int y = 0; y < m_height; ++y)
        for (int x = 0; x < m_width; ++x) {
          uint8_t* o = &img.rgba[(size_t(y) * size_t(m_width) + size_t(x)) * 4];
          o[0] = o[1] = o[2] = c.plane[size_t(y) * size_t(c.bw) + size_t(x)];
          o[3] = 255;
        }
      return img;
    }
    const std::vector<uint8_t> p0 = upsampled(m_comp[0]), p1 = upsampled(m_comp[1]), p2 = upsampled(m_comp[2]);
    const int W = m_comp[0].bw * (m_hmax / m_comp[0].h);
    const bool ycc = m_adobe ? m_adobe_transform != 0 : true;
    auto clamp8 = [](int v) { return uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (int y = 0; y < m_height; ++y)
      for (int x = 0; x < m_width; ++x) {
        const size_t i = size_t(y) * size_t(W) + size_t(x);
        uint8_t* o = &img.rgba[(size_t(y) * size_t(m_width) + size_t(x)) * 4];
        if (ycc) {
          const int Y = p0[i], cb = int(p1[i]) - 128, cr = int(p2[i]) - 128;
          o[0] = clamp8(Y + ((91881 * cr + 32768) >> 16));
          o[1] = clamp8(Y + ((-22554 * cb - 46802 * cr + 32768) >> 16));
          o[2] = clamp8(Y + ((116130 * cb + 32768) >> 16));
        } else { o[0] = p0[i]; o[1] = p1[i]; o[2] = p2[i]; }
        o[3] = 255;
      }
    return img;
  }
};

inline Image8 decode_jpeg(const std::vector<uint8_t>& file) { return JpegDecoder(file).run(); }

// stbi_load(path, ..., STBI_rgb_alpha); flip = stbi_set_flip_vertically_on_load
inline Image8 load_rgba8(const std::filesystem::path& path, bool flip_vertically)
{
  const std::vector<uint8_t> file = read_file(path);
  Image8 img;
  if (file.size() >= 8 && file[0] == 0x89 && file[1] == 'P') img = decode_png(file);
  else if (file.size() >= 2 && file[0] == 'P' && (file[1] == '5' || file[1] == '6')) img = decode_pnm(file);
  else if (file.size() >= 2 && file[0] == 0xff && file[1] == 0xd8) img = decode_jpeg(file);
  else throw std::runtime_error("failed to load " + path.generic_string() + ": only PNG, JPEG (sequential or progressive Huffman) and binary PPM/PGM images are supported in this build");
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
