// capi.hip -- implementation of the C ABI declared in include/fredholm_hip.h (context, scene
// upload, environment, frame state, stats, device-memory helpers, tile pack/unpack).
// Mirrors the host duties of fredholm::Renderer (fredholm/include/fredholm/renderer.h).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "../../include/fredholm/image_io.h"
#include <hip/hip_gl_interop.h>

#include "context.h"
#include "fh_bsdf.h"
#include "fh_trace.h"

// generated at build time from fredholm_amd/data/*.{u32,f32} by tools/gen_tables_inc.py
#include "gen/tables.inc"

namespace fh {

// errors of calls that have no context (fh_ctx_create, fh_image_load_rgba8) are kept per thread: a failing call on one thread
// (e.g. a texture loader thread) never reallocates a string another thread is reading through fh_last_error(NULL)
static thread_local std::string g_create_error;

int fail(fh_ctx* ctx, int code, const std::string& msg)
{
  g_create_error = msg;
  if (ctx) ctx->err = msg;
  return code;
}

namespace {

m34 load_m34(const float* m)
{
  m34 r;
  for (int k = 0; k < 3; ++k) r.r[k] = make_float4(m[4 * k], m[4 * k + 1], m[4 * k + 2], m[4 * k + 3]);
  return r;
}
const float kIdentity12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};

bool material_emissive(const fh_material& m) { return m.emission_color[0] > 0 || m.emission_color[1] > 0 || m.emission_color[2] > 0 || m.emission_texture_id != -1; }
bool material_textured(const fh_material& m)
{
  return m.base_color_texture_id != -1 || m.specular_color_texture_id != -1 || m.specular_roughness_texture_id != -1 || m.metalness_texture_id != -1 ||
         m.metallic_roughness_texture_id != -1 || m.coat_texture_id != -1 || m.coat_roughness_texture_id != -1 || m.emission_texture_id != -1 || m.heightmap_texture_id != -1 ||
         m.normalmap_texture_id != -1 || m.alpha_texture_id != -1;
}

// lobes a material can ever enable (see fh_bsdf.h).  Seen from the FRONT, metalness == 1 zeroes the weight and the layering
// multiplier of every lobe after the metal one.  Seen from BEHIND the reference resets metalness, specular, sheen and diffuse to 0
// (bsdf.cu:56-62), so transmission and diffuse transmission get their full weight back (weights[3], weights[5], bsdf.cu:74-84): those
// two stay whenever their own parameters enable them.  specular / sheen / diffuse are zero on both sides of a full metal and are dropped.
uint32_t material_lobes(const fh_material& m)
{
  // A texture on metalness (or the glTF metallic-roughness texture) makes metalness unknowable on the host: the metal lobe AND the lobes
  // a non-metal needs are kept.  Lobes whose OWN switch is a constant zero stay dropped whatever the textures say: transmission, sheen and
  // subsurface have no texture slot (shared.h:100-142), so a glTF material (scene.cpp:487-549 sets none of them) shades with
  // coat? + metal + specular + diffuse instead of the generic seven-lobe kernel.
  const bool metal_unknown = m.metalness_texture_id >= 0 || m.metallic_roughness_texture_id >= 0;
  uint32_t l = 0;
  const float coat = clampf(m.coat, 0.0f, 1.0f);
  if (coat > 0.0f || m.coat_texture_id >= 0) l |= L_COAT;
  if (m.metalness > 0.0f || metal_unknown) l |= L_METAL;
  const bool metal_full = m.metalness == 1.0f && !metal_unknown;
  if (m.transmission > 0.0f) l |= L_TRANS;
  if (m.subsurface * m.thin_walled > 0.0f) l |= L_DT;
  if (!metal_full) {
    if (m.specular_color_texture_id >= 0 || m.specular * lum(mk3(m.specular_color[0], m.specular_color[1], m.specular_color[2])) > 0.0f) l |= L_SPEC;
    if (m.sheen * lum(mk3(m.sheen_color[0], m.sheen_color[1], m.sheen_color[2])) != 0.0f) l |= L_SHEEN;
    if (m.diffuse > 0.0f) l |= L_DIFF;
  }
  return l;
}

// world-space face record of every face (layout: fh_device.h) from the object-space arrays and the instance transforms:
// positions by object_to_world, normals by the transpose of world_to_object (shared.h:42-50), as pt.cu:141-179 does per hit
__global__ void __launch_bounds__(256) k_face_records(uint32_t nf, const float* vertices, const float* normals, const float* texcoords, const uint32_t* indices, const uint2* meta,
                                                      const float4* o2w, const float4* w2o, float4* rec)
{
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const uint2 mi = meta[f];
  m34 a, b;
#pragma unroll
  for (int k = 0; k < 3; ++k) { a.r[k] = o2w[3 * (size_t)mi.y + k]; b.r[k] = w2o[3 * (size_t)mi.y + k]; }
  f3 p[3], n[3];
  float uv[3][2];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const size_t v = indices[3 * (size_t)f + k];
    p[k] = xform_point(a, mk3(vertices[3 * v], vertices[3 * v + 1], vertices[3 * v + 2]));
    n[k] = xform_normal(b, mk3(normals[3 * v], normals[3 * v + 1], normals[3 * v + 2]));
    uv[k][0] = texcoords[2 * v];
    uv[k][1] = texcoords[2 * v + 1];
  }
  float4* r = rec + kFaceRec * (size_t)f;
  r[0] = mk4(p[0], uv[0][0]); r[1] = mk4(p[1], uv[0][1]); r[2] = mk4(p[2], uv[1][0]);
  r[3] = mk4(n[0], uv[1][1]); r[4] = mk4(n[1], uv[2][0]); r[5] = mk4(n[2], uv[2][1]);
  r[6] = make_float4(__uint_as_float(mi.x), __uint_as_float(mi.y), 0.0f, 0.0f);
}

int transform_faces(fh_ctx* ctx)
{
  const uint32_t ni = (uint32_t)ctx->h_o2w.size() / 12;
  if (ctx->n_xf_alloc < ni) {
    if (ctx->d_o2w) (void)hipFree(ctx->d_o2w);
    if (ctx->d_w2o) (void)hipFree(ctx->d_w2o);
    ctx->d_o2w = ctx->d_w2o = nullptr; ctx->n_xf_alloc = 0;
    FH_HIP(hipMalloc((void**)&ctx->d_o2w, 48ull * ni));
    FH_HIP(hipMalloc((void**)&ctx->d_w2o, 48ull * ni));
    ctx->n_xf_alloc = ni;
  }
  FH_HIP(hipMemcpyAsync(ctx->d_o2w, ctx->h_o2w.data(), 48ull * ni, hipMemcpyHostToDevice, ctx->stream));
  FH_HIP(hipMemcpyAsync(ctx->d_w2o, ctx->h_w2o.data(), 48ull * ni, hipMemcpyHostToDevice, ctx->stream));
  if (ctx->n_faces)
    hipLaunchKernelGGL(k_face_records, dim3((ctx->n_faces + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_faces, ctx->d_obj_vertices, ctx->d_obj_normals, ctx->d_obj_texcoords, ctx->d_obj_indices,
                       ctx->d_face_meta, ctx->d_o2w, ctx->d_w2o, ctx->d_face_rec);
  FH_HIP(hipGetLastError());
  FH_HIP(hipStreamSynchronize(ctx->stream));  // the host copies of the transforms may change right after this call
  ctx->bvh_valid = false;
  return FH_OK;
}

// ---- Opacity classes of cut-out faces (round 5).  The any-hit programs (pt.cu:545-678) discard a candidate hit when the filtered alpha of the base-colour texture or the
// filtered red of the alpha texture is below 0.5 at the hit's texture coordinate.  A filtered value is a convex combination of the four texels around the coordinate (weights
// are products of 1.8 fixed-point fractions and sum to exactly 1, include/fh_texture_unit.h), so over the part of a texture a FACE can ever address the outcome is known in
// advance when all of those texels agree: every byte >= 128 (0.50196) -- the test always passes, the face needs no test at all -- or every byte <= 127 (0.49804) -- it never
// passes, the face can never be hit.  The footprint is the rectangle of texels the face's three texture coordinates span, one texel wider on every side for the bilinear taps
// and a sixteenth more for the rounding of the interpolated coordinate, wrapped like the texture unit wraps; faces with a footprint of more than 2^18 texels are left to the
// test.  0 = test as before, 1 = always passes, 2 = never passes.  FH_OPACITY_CLASSES=0 (read at upload): everything 0.
static int texel_span(float lo, float hi, uint32_t n, int& first, int& count)
{
  // texel columns (rows) a fetch at normalised coordinates in [lo, hi] can read: floor(u * n - 0.5) and its successor, before wrapping
  const double a = (double)lo * n - 0.5, b = (double)hi * n - 0.5;
  const double slack = 0.0625 + 1e-5 * (std::fabs(a) > std::fabs(b) ? std::fabs(a) : std::fabs(b));
  const double f0 = std::floor(a - slack), f1 = std::floor(b + slack) + 1.0;
  if (!(f1 - f0 < (double)n)) { first = 0; count = (int)n; return 1; }
  first = (int)(((long long)f0 % (long long)n + (long long)n) % (long long)n);
  count = (int)(f1 - f0) + 1;
  return 1;
}
// 1: every texel of the footprint passes the 0.5 threshold, 2: none does, 0: mixed (or not decided).  uv: n_pts texture coordinates whose bounding rectangle is the footprint
static uint32_t footprint_class(const uint8_t* rgba8, uint32_t w, uint32_t h, uint32_t channel, const float* decode, const float* uv, int n_pts = 3)
{
  if (!rgba8 || w == 0 || h == 0) return 0u;
  float ulo = uv[0], uhi = uv[0], vlo = uv[1], vhi = uv[1];
  for (int k = 1; k < n_pts; ++k) { ulo = std::fmin(ulo, uv[2 * k]); uhi = std::fmax(uhi, uv[2 * k]); vlo = std::fmin(vlo, uv[2 * k + 1]); vhi = std::fmax(vhi, uv[2 * k + 1]); }
  if (!(std::fabs(ulo) < 1e6f) || !(std::fabs(uhi) < 1e6f) || !(std::fabs(vlo) < 1e6f) || !(std::fabs(vhi) < 1e6f)) return 0u;  // (huge or non-finite coordinates: left to the test)
  int x0, nx, y0, ny;
  texel_span(ulo, uhi, w, x0, nx);
  texel_span(vlo, vhi, h, y0, ny);
  if ((long long)nx * ny > (1ll << 18)) return 0u;
  bool all_pass = true, none_pass = true;
  for (int j = 0; j < ny && (all_pass || none_pass); ++j) {
    const size_t row = (size_t)((y0 + j) % (int)h) * w;
    for (int i = 0; i < nx; ++i) {
      const uint8_t b = rgba8[(row + (size_t)((x0 + i) % (int)w)) * 4u + channel];
      const float v = decode ? decode[b] : (float)b * (1.0f / 255.0f);
      if (v >= 0.50196f) none_pass = false;       // (byte 128 of a linear channel)
      else if (v <= 0.49804f) all_pass = false;   // (byte 127)
      else { all_pass = none_pass = false; }      // an sRGB-decoded value between the two: not decided
    }
  }
  return all_pass ? 1u : (none_pass ? 2u : 0u);
}

// Rebuild everything derived from the flat scene + transforms: face records, classes, lights.
int rebuild_device_scene(fh_ctx* ctx)
{
  const uint32_t nf = (uint32_t)ctx->h_indices.size() / 3;
  const uint32_t nv = (uint32_t)ctx->h_vertices.size() / 3;
  const uint32_t nm = (uint32_t)ctx->h_materials.size();
  const uint32_t ni = (uint32_t)ctx->h_o2w.size() / 12;
  // classes
  std::vector<MaterialDev> mats(nm);
  ctx->n_classes = 0;
  for (uint32_t i = 0; i < nm; ++i) {
    static_assert(sizeof(fh_material) == 180, "fh_material must match the reference Material (shared.h:100-142)");
    std::memcpy(mats[i].w, &ctx->h_materials[i], 180);
    const uint32_t lobes = material_lobes(ctx->h_materials[i]);
    mats[i].lobes = lobes;
    mats[i].emissive = material_emissive(ctx->h_materials[i]) ? 1u : 0u;
    {  // alpha: 1 = some fetch of the material's textures can fail the any-hit test; 2 = only a non-finite texture coordinate can (the texture unit returns 0 for it)
      const int32_t bt = ctx->h_materials[i].base_color_texture_id, at = ctx->h_materials[i].alpha_texture_id;
      const bool cuts = (bt >= 0 && ctx->h_tex_alpha_cuts[(size_t)bt]) || (at >= 0 && ctx->h_tex_red_cuts[(size_t)at]);
      mats[i].alpha = cuts ? 1u : ((bt >= 0 || at >= 0) ? 2u : 0u);
    }
    uint32_t c = 0;
    for (; c < ctx->n_classes; ++c)
      if (ctx->class_lobes[c] == lobes) break;
    if (c == ctx->n_classes) {
      if (ctx->n_classes < kMaxClasses) ctx->class_lobes[ctx->n_classes++] = lobes;
      else { c = kMaxClasses - 1; ctx->class_lobes[c] = L_ALL; }  // overflow: fold into one generic class
    }
    mats[i].cls = c;
  }
  if (ctx->n_classes == kMaxClasses)  // materials folded into the generic class must see the generic mask
    for (uint32_t i = 0; i < nm; ++i)
      if (mats[i].cls == kMaxClasses - 1) ctx->class_lobes[kMaxClasses - 1] |= mats[i].lobes;

  // per-face class bytes, the light list and index validation on the host (cheap integer work); the world-space face records -- the
  // transform of 3 positions and 3 normals per face -- on the device from the object-space arrays resident there, so that a frame of an
  // animation (fh_set_transforms) uploads 96 bytes per instance instead of recomputing and re-uploading 112 bytes per face
  std::vector<uint8_t> cls(nf);
  std::vector<uint2> meta(nf);
  std::vector<AreaLightDev> lights;
  std::vector<uint8_t> alpha_bits(nf, 0);  // bit 0: test the base-colour texture's alpha, bit 1: test the alpha texture's red
  bool any_alpha = false;
  bool opacity_classes = true;
  if (const char* e = getenv("FH_OPACITY_CLASSES")) opacity_classes = e[0] != '0';
  for (int k = 0; k < 4; ++k) ctx->alpha_face_counts[k] = 0;
  for (int k = 0; k < 3; ++k) ctx->alpha_cell_counts[k] = 0;
  float srgb_table[256];
  for (int i = 0; i < 256; ++i) srgb_table[i] = fht_srgb_to_linear((float)i * (1.0f / 255.0f));
  for (uint32_t f = 0; f < nf; ++f) {
    const uint32_t inst = ctx->h_instance_ids.empty() ? 0u : ctx->h_instance_ids[f];
    if (inst >= ni) return fail(ctx, FH_E_INVALID, "instance id out of range");
    const uint32_t mid = ctx->h_material_ids[f];
    if (mid >= nm) return fail(ctx, FH_E_INVALID, "material id out of range");
    for (int k = 0; k < 3; ++k)
      if (ctx->h_indices[3ull * f + k] >= nv) return fail(ctx, FH_E_INVALID, "vertex index out of range");
    meta[f] = make_uint2(mid, inst);
    bool alpha = mats[mid].alpha == 1u, never = false;
    if (mats[mid].alpha) {
      bool wild = false;  // opaque textures: the test can only fail where the interpolated coordinate is NaN or overflows
      for (int k = 0; k < 3; ++k) {
        const size_t v = ctx->h_indices[3ull * f + k];
        if (!(std::fabs(ctx->h_texcoords[2 * v]) < 1e30f) || !(std::fabs(ctx->h_texcoords[2 * v + 1]) < 1e30f)) wild = true;
      }
      alpha = alpha || wild;
      const int32_t bt = ctx->h_materials[mid].base_color_texture_id, at = ctx->h_materials[mid].alpha_texture_id;
      alpha_bits[f] = (uint8_t)(((bt >= 0 && (wild || ctx->h_tex_alpha_cuts[(size_t)bt])) ? 1u : 0u) | ((at >= 0 && (wild || ctx->h_tex_red_cuts[(size_t)at])) ? 2u : 0u));
      if (alpha) ctx->alpha_face_counts[0]++;
      if (alpha && !wild && opacity_classes && !ctx->h_tex_host.empty()) {  // what the face's own footprint says (above)
        float uv[6];
        for (int k = 0; k < 3; ++k) { const size_t v = ctx->h_indices[3ull * f + k]; uv[2 * k] = ctx->h_texcoords[2 * v]; uv[2 * k + 1] = ctx->h_texcoords[2 * v + 1]; }
        uint32_t cb = 1u, ca = 1u;  // a texture the face does not test passes by itself
        if (alpha_bits[f] & 1u) { const fh_ctx::HostTexture& t = ctx->h_tex_host[(size_t)bt]; cb = footprint_class(t.rgba8.data(), t.width, t.height, 3u, nullptr, uv); }
        if (alpha_bits[f] & 2u) { const fh_ctx::HostTexture& t = ctx->h_tex_host[(size_t)at]; ca = footprint_class(t.rgba8.data(), t.width, t.height, 0u, t.srgb ? srgb_table : nullptr, uv); }
        if (cb == 1u && ca == 1u) { alpha = false; alpha_bits[f] = 0; ctx->alpha_face_counts[1]++; }          // always passes: an ordinary opaque face
        else if (cb == 2u || ca == 2u) { never = true; alpha = false; alpha_bits[f] = 0; ctx->alpha_face_counts[2]++; }  // never passes: no ray can hit it
      }
    }
    cls[f] = (uint8_t)(mats[mid].cls | (mats[mid].emissive ? 0x80u : 0u) | (alpha ? 0x40u : 0u) | (never ? 0x20u : 0u));
    if (alpha) any_alpha = true;
    if (mats[mid].emissive) lights.push_back({f, mid});  // renderer.h:388-402, face order
  }
  auto re_alloc = [&](auto*& ptr, size_t bytes) -> hipError_t {
    if (ptr) { (void)hipFree(ptr); ptr = nullptr; }
    return hipMalloc((void**)&ptr, bytes ? bytes : 16);
  };
  FH_HIP(re_alloc(ctx->d_face_rec, (unsigned long long)kFaceRec * nf * sizeof(float4)));
  FH_HIP(re_alloc(ctx->d_face_cls, cls.size()));
  FH_HIP(re_alloc(ctx->d_materials, mats.size() * sizeof(MaterialDev)));
  FH_HIP(re_alloc(ctx->d_lights, lights.size() * sizeof(AreaLightDev)));
  FH_HIP(re_alloc(ctx->d_obj_vertices, ctx->h_vertices.size() * sizeof(float)));
  FH_HIP(re_alloc(ctx->d_obj_normals, ctx->h_normals.size() * sizeof(float)));
  FH_HIP(re_alloc(ctx->d_obj_texcoords, ctx->h_texcoords.size() * sizeof(float)));
  FH_HIP(re_alloc(ctx->d_obj_indices, ctx->h_indices.size() * sizeof(uint32_t)));
  FH_HIP(re_alloc(ctx->d_face_meta, meta.size() * sizeof(uint2)));
  FH_HIP(hipMemcpy(ctx->d_face_cls, cls.data(), cls.size(), hipMemcpyHostToDevice));
  FH_HIP(hipMemcpy(ctx->d_materials, mats.data(), mats.size() * sizeof(MaterialDev), hipMemcpyHostToDevice));
  if (!lights.empty()) FH_HIP(hipMemcpy(ctx->d_lights, lights.data(), lights.size() * sizeof(AreaLightDev), hipMemcpyHostToDevice));
  FH_HIP(hipMemcpy(ctx->d_obj_vertices, ctx->h_vertices.data(), ctx->h_vertices.size() * sizeof(float), hipMemcpyHostToDevice));
  FH_HIP(hipMemcpy(ctx->d_obj_normals, ctx->h_normals.data(), ctx->h_normals.size() * sizeof(float), hipMemcpyHostToDevice));
  FH_HIP(hipMemcpy(ctx->d_obj_texcoords, ctx->h_texcoords.data(), ctx->h_texcoords.size() * sizeof(float), hipMemcpyHostToDevice));
  FH_HIP(hipMemcpy(ctx->d_obj_indices, ctx->h_indices.data(), ctx->h_indices.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  FH_HIP(hipMemcpy(ctx->d_face_meta, meta.data(), meta.size() * sizeof(uint2), hipMemcpyHostToDevice));
  ctx->n_faces = nf;
  ctx->n_lights = (uint32_t)lights.size();
  ctx->n_materials = nm;
  ctx->has_alpha = any_alpha;
  ctx->alpha_face_counts[3] = ctx->alpha_face_counts[0] - ctx->alpha_face_counts[1] - ctx->alpha_face_counts[2];
  if (getenv("FH_DEBUG_BVH") && ctx->alpha_face_counts[0])
    fprintf(stderr, "[alpha] %u faces whose textures can cut: %u always pass, %u never pass, %u keep their any-hit test\n", ctx->alpha_face_counts[0], ctx->alpha_face_counts[1], ctx->alpha_face_counts[2],
            ctx->alpha_face_counts[3]);
  if (ctx->d_alpha_rec) { (void)hipFree(ctx->d_alpha_rec); ctx->d_alpha_rec = nullptr; }
  if (any_alpha) {
    // everything the any-hit test of a face reads, in one 64-byte line: the three texture coordinates and the two textures it may have to look at
    // ... and, behind them (second 64-byte line), the face's OPACITY MICROMAP: the face cut into 16 x 16 cells of its barycentrics (cell = (floor(16 u), floor(16 v)), the
    // v1 / v2 weights of a hit), two bits per cell -- 0: test, 1: the test passes everywhere in the cell, 2: nowhere -- decided like the per-face classes above from the texels
    // the cell can address (the cell's four corners, moved out by 10^-4 for the rounding of the barycentrics and of the cell index).  alpha_pass looks the cell of a candidate up
    // before anything else (one dword).  Exact: same hits, same images.  FH_OPACITY_MICROMAP=0 (read at upload): all cells 0.
    bool micromap = opacity_classes;
    if (const char* e = getenv("FH_OPACITY_MICROMAP")) micromap = micromap && e[0] != '0';
    unsigned long long cells_total = 0, cells_pass = 0, cells_never = 0;
    std::vector<uint4> rec(8ull * nf, make_uint4(0u, 0u, 0u, 0u));
    auto bits = [](float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; };
    for (uint32_t f = 0; f < nf; ++f) {
      if (!alpha_bits[f]) continue;
      const fh_material& m = ctx->h_materials[ctx->h_material_ids[f]];
      const float* tc = ctx->h_texcoords.data();
      const size_t v0 = ctx->h_indices[3ull * f], v1 = ctx->h_indices[3ull * f + 1], v2 = ctx->h_indices[3ull * f + 2];
      rec[8ull * f] = make_uint4(bits(tc[2 * v0]), bits(tc[2 * v0 + 1]), bits(tc[2 * v1]), bits(tc[2 * v1 + 1]));
      uint32_t flags = alpha_bits[f];
      if ((flags & 2u) && ctx->h_tex_desc[(size_t)m.alpha_texture_id].srgb) flags |= 4u;
      rec[8ull * f + 1] = make_uint4(bits(tc[2 * v2]), bits(tc[2 * v2 + 1]), flags, 0u);
      if (flags & 1u) {
        const fht_texture& t = ctx->h_tex_desc[(size_t)m.base_color_texture_id];
        const unsigned long long p = (unsigned long long)(uintptr_t)t.rgba8;
        rec[8ull * f + 2] = make_uint4((uint32_t)p, (uint32_t)(p >> 32), t.width, t.height);
      }
      if (flags & 2u) {
        const fht_texture& t = ctx->h_tex_desc[(size_t)m.alpha_texture_id];
        const unsigned long long p = (unsigned long long)(uintptr_t)t.rgba8;
        rec[8ull * f + 3] = make_uint4((uint32_t)p, (uint32_t)(p >> 32), t.width, t.height);
      }
      const bool finite_uv = std::fabs(tc[2 * v0]) < 1e6f && std::fabs(tc[2 * v0 + 1]) < 1e6f && std::fabs(tc[2 * v1]) < 1e6f && std::fabs(tc[2 * v1 + 1]) < 1e6f && std::fabs(tc[2 * v2]) < 1e6f &&
                             std::fabs(tc[2 * v2 + 1]) < 1e6f;
      if (micromap && finite_uv && !ctx->h_tex_host.empty()) {
        uint32_t words[16] = {};
        const double u0 = tc[2 * v0], w0 = tc[2 * v0 + 1], du1 = (double)tc[2 * v1] - u0, dw1 = (double)tc[2 * v1 + 1] - w0, du2 = (double)tc[2 * v2] - u0, dw2 = (double)tc[2 * v2 + 1] - w0;
        for (int cj = 0; cj < 16; ++cj)
          for (int ci = 0; ci + cj <= 16 && ci < 16; ++ci) {  // (cells beyond the hypotenuse are never addressed: they stay 0)
            // barycentric range of the cell, widened: the last cell of a row / column also takes weights of 1 (and a hair more), the first a hair below 0
            const double a0 = ci / 16.0 - 1e-4, a1 = (ci == 15 ? 1.0 : (ci + 1) / 16.0) + 1e-4, b0 = cj / 16.0 - 1e-4, b1 = (cj == 15 ? 1.0 : (cj + 1) / 16.0) + 1e-4;
            const double cb[4][2] = {{a0, b0}, {a1, b0}, {a0, b1}, {a1, b1}};
            float uv[8];
            for (int k = 0; k < 4; ++k) { uv[2 * k] = (float)(u0 + cb[k][0] * du1 + cb[k][1] * du2); uv[2 * k + 1] = (float)(w0 + cb[k][0] * dw1 + cb[k][1] * dw2); }
            uint32_t cb_ = 1u, ca_ = 1u;
            if (flags & 1u) { const fh_ctx::HostTexture& t = ctx->h_tex_host[(size_t)m.base_color_texture_id]; cb_ = footprint_class(t.rgba8.data(), t.width, t.height, 3u, nullptr, uv, 4); }
            if (flags & 2u) { const fh_ctx::HostTexture& t = ctx->h_tex_host[(size_t)m.alpha_texture_id]; ca_ = footprint_class(t.rgba8.data(), t.width, t.height, 0u, t.srgb ? srgb_table : nullptr, uv, 4); }
            const uint32_t st = (cb_ == 1u && ca_ == 1u) ? 1u : ((cb_ == 2u || ca_ == 2u) ? 2u : 0u);
            const int cell = cj * 16 + ci;
            words[cell >> 4] |= st << (2 * (cell & 15));
            ++cells_total; cells_pass += st == 1u; cells_never += st == 2u;
          }
        for (int k = 0; k < 4; ++k) rec[8ull * f + 4 + k] = make_uint4(words[4 * k], words[4 * k + 1], words[4 * k + 2], words[4 * k + 3]);
      }
    }
    ctx->alpha_cell_counts[0] = cells_total; ctx->alpha_cell_counts[1] = cells_pass; ctx->alpha_cell_counts[2] = cells_never;
    if (getenv("FH_DEBUG_BVH") && cells_total)
      fprintf(stderr, "[alpha] micromap: %llu cells of the faces that keep their test: %.1f %% always pass, %.1f %% never pass\n", cells_total, 100.0 * cells_pass / cells_total, 100.0 * cells_never / cells_total);
    FH_HIP(hipMalloc((void**)&ctx->d_alpha_rec, rec.size() * sizeof(uint4)));
    FH_HIP(hipMemcpy(ctx->d_alpha_rec, rec.data(), rec.size() * sizeof(uint4), hipMemcpyHostToDevice));
  }
  // (the host copies of the textures that can cut were needed for the classes and the micromaps above only: every upload brings its textures again)
  for (fh_ctx::HostTexture& t : ctx->h_tex_host) std::vector<uint8_t>().swap(t.rgba8);
  ctx->refit_ok = false;  // new topology: the next build is a full one
  return transform_faces(ctx);
}

// texels of all textures in one device blob + descriptors + the 256-entry sRGB table (cwl/texture.h:13-75 per texture)
int upload_textures(fh_ctx* ctx, uint32_t n, const fh_texture_desc* descs)
{
  if (ctx->d_texels) { (void)hipFree(ctx->d_texels); ctx->d_texels = nullptr; }
  if (ctx->d_textures) { (void)hipFree(ctx->d_textures); ctx->d_textures = nullptr; }
  ctx->n_textures = n;
  if (!ctx->d_srgb_lut) {
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = fht_srgb_to_linear((float)i * (1.0f / 255.0f));
    FH_HIP(hipMalloc((void**)&ctx->d_srgb_lut, sizeof lut));
    FH_HIP(hipMemcpy(ctx->d_srgb_lut, lut, sizeof lut, hipMemcpyHostToDevice));
  }
  ctx->h_tex_alpha_cuts.assign(n, 0);
  ctx->h_tex_red_cuts.assign(n, 0);
  ctx->h_tex_desc.clear();
  ctx->h_tex_host.clear();
  if (n == 0) return FH_OK;
  size_t total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (!descs[i].rgba8 || descs[i].width == 0 || descs[i].height == 0) return fail(ctx, FH_E_INVALID, "fh_scene_upload: empty texture");
    total += (size_t)descs[i].width * descs[i].height * 4;
    // The any-hit programs discard a hit when the filtered alpha (base-colour texture) or red (alpha texture) is below 0.5 (pt.cu:545-678).  A filtered
    // value is a convex combination of texel values (weights are multiples of 2^-16 that sum to exactly 1), so a texture whose every texel decodes to
    // >= 0.501 can never discard a hit: faces that only reference such textures need no any-hit test at all (most base-colour maps are opaque).
    const uint8_t* px = descs[i].rgba8;
    const size_t n_px = (size_t)descs[i].width * descs[i].height;
    uint8_t min_a = 255, min_r = 255;
    for (size_t k = 0; k < n_px; ++k) { min_r = px[4 * k] < min_r ? px[4 * k] : min_r; min_a = px[4 * k + 3] < min_a ? px[4 * k + 3] : min_a; }
    const float red = descs[i].srgb ? fht_srgb_to_linear((float)min_r * (1.0f / 255.0f)) : (float)min_r * (1.0f / 255.0f);
    ctx->h_tex_alpha_cuts[i] = (float)min_a * (1.0f / 255.0f) < 0.501f ? 1 : 0;
    ctx->h_tex_red_cuts[i] = red < 0.501f ? 1 : 0;
  }
  // host copies of the textures that can cut: what the per-face opacity classes are decided from (rebuild_device_scene)
  ctx->h_tex_host.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    fh_ctx::HostTexture& t = ctx->h_tex_host[i];
    t.width = descs[i].width; t.height = descs[i].height; t.srgb = descs[i].srgb ? 1u : 0u;
    if (ctx->h_tex_alpha_cuts[i] || ctx->h_tex_red_cuts[i]) t.rgba8.assign(descs[i].rgba8, descs[i].rgba8 + (size_t)descs[i].width * descs[i].height * 4);
  }
  FH_HIP(hipMalloc((void**)&ctx->d_texels, total));
  std::vector<fht_texture> t(n);
  size_t off = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const size_t bytes = (size_t)descs[i].width * descs[i].height * 4;
    FH_HIP(hipMemcpy(ctx->d_texels + off, descs[i].rgba8, bytes, hipMemcpyHostToDevice));
    t[i] = fht_texture{ctx->d_texels + off, nullptr, descs[i].width, descs[i].height, descs[i].srgb ? 1u : 0u};
    off += bytes;
  }
  FH_HIP(hipMalloc((void**)&ctx->d_textures, n * sizeof(fht_texture)));
  FH_HIP(hipMemcpy(ctx->d_textures, t.data(), n * sizeof(fht_texture), hipMemcpyHostToDevice));
  ctx->h_tex_desc = t;
  return FH_OK;
}

// the pixels of rank `rank` of `world` in the order the library packs them: tiles t with t % world == rank, row-major tile order; inside a tile 8 x 8 blocks, row-major,
// and row-major inside a block -- so the 64 lanes of a wave of k_generate (64 consecutive list entries) hold a compact 8 x 8 patch of the image instead of two rows of 32:
// their camera rays share more nodes and their first hits more faces (r6-4: closest-hit -2 %, configs[2] / [3] +0.9 %; 4 x 4 blocks measure the same).  FH_PIXEL_BLOCK=0:
// rows of the whole tile, as before round 6.  fredholm_amd/distributed.py: tile_ownership is the same list; tests/test_gpu_parity.py compares the two.
constexpr uint32_t kPixelBlock = 8;
static void owned_list(uint32_t width, uint32_t height, uint32_t tw, uint32_t th, uint32_t rank, uint32_t world, std::vector<uint32_t>& owned, std::vector<uint32_t>* owned_xy)
{
  static const uint32_t blk = [] { const char* e = getenv("FH_PIXEL_BLOCK"); const int v = e ? atoi(e) : (int)kPixelBlock; return (uint32_t)(v > 0 && v <= 64 ? v : 65536); }();
  const uint32_t tx = (width + tw - 1) / tw, ty = (height + th - 1) / th;
  for (uint32_t t = rank; t < tx * ty; t += world) {
    const uint32_t x0 = (t % tx) * tw, y0 = (t / tx) * th;
    for (uint32_t by = y0; by < y0 + th && by < height; by += (blk < th ? blk : th))
      for (uint32_t bx = x0; bx < x0 + tw && bx < width; bx += (blk < tw ? blk : tw))
        for (uint32_t y = by; y < by + blk && y < y0 + th && y < height; ++y)
          for (uint32_t x = bx; x < bx + blk && x < x0 + tw && x < width; ++x) { owned.push_back(x + width * y); if (owned_xy) owned_xy->push_back(x | (y << 16)); }
  }
}

int rebuild_ownership(fh_ctx* ctx)
{
  if (ctx->d_owned) { (void)hipFree(ctx->d_owned); ctx->d_owned = nullptr; }
  if (ctx->d_owned_xy) { (void)hipFree(ctx->d_owned_xy); ctx->d_owned_xy = nullptr; }
  ctx->n_owned = 0;
  if (ctx->width == 0 || ctx->height == 0) return FH_OK;
  if (ctx->width > 65535u || ctx->height > 65535u) return fail(ctx, FH_E_UNSUPPORTED, "frames wider or higher than 65535 pixels are not supported");
  std::vector<uint32_t> owned, owned_xy;
  owned_list(ctx->width, ctx->height, ctx->tile_w, ctx->tile_h, ctx->shard_rank, ctx->shard_world, owned, &owned_xy);
  ctx->n_owned = (uint32_t)owned.size();
  FH_HIP(hipMalloc((void**)&ctx->d_owned, owned.empty() ? 16 : owned.size() * 4));
  FH_HIP(hipMalloc((void**)&ctx->d_owned_xy, owned.empty() ? 16 : owned.size() * 4));
  if (!owned.empty()) FH_HIP(hipMemcpy(ctx->d_owned, owned.data(), owned.size() * 4, hipMemcpyHostToDevice));
  if (!owned.empty()) FH_HIP(hipMemcpy(ctx->d_owned_xy, owned_xy.data(), owned_xy.size() * 4, hipMemcpyHostToDevice));
  return FH_OK;
}

__global__ void k_pack(const float* layer, const uint32_t* owned, uint32_t n, uint32_t fpp, float* packed)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n * fpp; i += gridDim.x * blockDim.x) packed[i] = layer[(size_t)owned[i / fpp] * fpp + i % fpp];
}
__global__ void k_unpack(const float* packed, const uint32_t* owned, uint32_t n, uint32_t fpp, float* layer)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n * fpp; i += gridDim.x * blockDim.x) layer[(size_t)owned[i / fpp] * fpp + i % fpp] = packed[i];
}
// every rank's shard in ONE launch (fh_unpack_shards): `all_owned` is the ownership lists of ranks 0 .. world - 1 one after the other (a permutation of the frame's pixels),
// start[r] where rank r's list begins in it, packed[r] that rank's packed shard
struct ShardSources { const float* packed[kMaxShardsPerLaunch]; uint32_t start[kMaxShardsPerLaunch + 1]; uint32_t world; };
__global__ void k_unpack_all(ShardSources src, const uint32_t* all_owned, uint32_t n, uint32_t fpp, float* layer)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n * fpp; i += gridDim.x * blockDim.x) {
    const uint32_t j = i / fpp, c = i % fpp;
    uint32_t r = 0;
    while (r + 1u < src.world && j >= src.start[r + 1u]) ++r;
    layer[(size_t)all_owned[j] * fpp + c] = src.packed[r][(size_t)(j - src.start[r]) * fpp + c];
  }
}

}  // namespace
}  // namespace fh

using namespace fh;

#define CTX_CHECK(ctx)                      \
  if (!(ctx)) return FH_E_INVALID;          \
  if (hipSetDevice((ctx)->device) != hipSuccess) return fh::fail(ctx, FH_E_HIP, "hipSetDevice failed")

namespace fh {
// render.hip: pool_ensure -- path record 64, radiance 16, identity 8, flags 4, first-hit AOVs 64, a 48-byte place per kind of secondary ray, the pending
// light ray 32 (emitters only); queues: two radiance + one spare + secondary + its sorted copy 5 x 4, two 16-bit keys, one entry per shading class
uint64_t pool_bytes_per_path(const fh_ctx* ctx)
{
  const uint32_t sec = 2u + (ctx->has_dir ? 1u : 0u) + (ctx->n_lights > 0 ? 1u : 0u);
  const uint32_t classes = ctx->n_classes < 1u ? 1u : ctx->n_classes;
  return 64u + 16u + 8u + 4u + 64u + 48u * sec + (ctx->n_lights > 0 ? 32u : 0u) + 20u + 4u + 4u + 4u * classes;  // (... + queue keys and sorted queues 20, q_sec 4, q_prim 4, class queues)
}
}  // namespace fh

extern "C" {

int fh_ctx_create(int device, fh_ctx** out)
{
  if (!out) return FH_E_INVALID;
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { g_create_error = "no HIP device available: the fredholm HIP path has no CPU fallback"; return FH_E_HIP; }
  if (device < 0 || device >= n_dev) { g_create_error = "device index out of range"; return FH_E_INVALID; }
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return FH_E_HIP; }
  fh_ctx* ctx = new fh_ctx;
  ctx->device = device;
  auto bail = [&](const char* what) { g_create_error = what; delete ctx; return FH_E_HIP; };
  auto make_stream = [&](hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); };
  if (make_stream(&ctx->stream) != hipSuccess) return bail("hipStreamCreate failed");
  if (hipMalloc((void**)&ctx->d_sobol, kSobolMatricesBytes) != hipSuccess) return bail("hipMalloc failed");
  if (hipMalloc((void**)&ctx->d_lut_refl, kLutReflectionBytes) != hipSuccess) return bail("hipMalloc failed");
  if (hipMalloc((void**)&ctx->d_lut_sheen, kLutSheenBytes) != hipSuccess) return bail("hipMalloc failed");
  if (hipMalloc((void**)&ctx->d_trace_counters, 32 * sizeof(unsigned long long)) != hipSuccess) return bail("hipMalloc failed");
  if (hipMalloc((void**)&ctx->d_hosek, sizeof(fh::HosekSky)) != hipSuccess || hipMemsetAsync(ctx->d_hosek, 0, sizeof(fh::HosekSky), ctx->stream) != hipSuccess) return bail("hipMalloc failed");
  if (hipMemcpy(ctx->d_sobol, kSobolMatrices, kSobolMatricesBytes, hipMemcpyHostToDevice) != hipSuccess) return bail("table upload failed");
  {
    // byte-indexed form of the generator matrices: entry [dim][k][b] = XOR of the columns 8k + j selected by the bits j of b, so that the XOR over the 32 index
    // bits (sobol.cu:10716-10727) becomes four table reads (fh_sampler.h: sobol_row_bytes); the integrator kernels stage the tables of their dimensions in LDS
    std::vector<uint32_t> bytes((size_t)1024 * 4 * 256);
    const uint32_t* mat = kSobolMatrices;
    for (uint32_t dim = 0; dim < 1024u; ++dim)
      for (uint32_t k = 0; k < 4u; ++k) {
        uint32_t* t = &bytes[((size_t)dim * 4 + k) * 256];
        t[0] = 0u;
        for (uint32_t b = 1; b < 256u; ++b) t[b] = t[b & (b - 1u)] ^ mat[dim * 52u + 8u * k + (uint32_t)__builtin_ctz(b)];
      }
    if (hipMalloc((void**)&ctx->d_sobol_bytes, bytes.size() * sizeof(uint32_t)) != hipSuccess) return bail("hipMalloc failed");
    if (hipMemcpy(ctx->d_sobol_bytes, bytes.data(), bytes.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return bail("table upload failed");
  }
  if (hipMemcpy(ctx->d_lut_refl, kLutReflection, kLutReflectionBytes, hipMemcpyHostToDevice) != hipSuccess) return bail("table upload failed");
  if (hipMemcpy(ctx->d_lut_sheen, kLutSheen, kLutSheenBytes, hipMemcpyHostToDevice) != hipSuccess) return bail("table upload failed");
  (void)hipMemsetAsync(ctx->d_trace_counters, 0, 32 * sizeof(unsigned long long), ctx->stream);  // (never hipMemset: it is asynchronous and ordered with nothing on a non-blocking stream)
  for (int k = 0; k < 2; ++k)
    if (make_stream(&ctx->aux_stream[k]) != hipSuccess) return bail("hipStreamCreate failed");
  for (int k = 0; k < 3; ++k) {
    (void)hipHostMalloc((void**)&ctx->h_counters[k], sizeof(uint32_t) * fh::kCounterStride * 66);
    (void)hipEventCreate(&ctx->ev_counters[k]);
    (void)hipEventCreateWithFlags(&ctx->ev_gen[k], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&ctx->ev_acc[k], hipEventDisableTiming);
  }
  (void)hipEventCreateWithFlags(&ctx->ev_enter, hipEventDisableTiming);
  if (const char* e = getenv("FH_PIPELINE")) ctx->n_slots = e[0] == '0' ? 1 : (e[0] == '2' ? 2 : 3);
  {
    fh_ctx::Tunables& t = ctx->tun;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) t.n_cus = (uint32_t)prop.multiProcessorCount;
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) == hipSuccess && v >= 64 * 1024) t.lds_per_cu = (uint32_t)v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, device) == hipSuccess && v >= 64 * 1024) t.lds_per_block = (uint32_t)v;
    if (t.lds_per_block > t.lds_per_cu) t.lds_per_block = t.lds_per_cu;
    auto env_uint = [](const char* name, int lo, int hi, uint32_t& dst) { if (const char* e = getenv(name)) { const int v = atoi(e); if (v >= lo && v <= hi) dst = (uint32_t)v; } };
    auto env_off = [](const char* name, bool& dst) { if (const char* e = getenv(name)) dst = e[0] != '0'; };
    env_uint("FH_COOP_T", 1, 64, t.coop_flush);
    t.coop_flush_fixed = getenv("FH_COOP_T") != nullptr;
    env_off("FH_COOP", t.coop);
    env_off("FH_STREAM", t.stream);
    t.stream_forced = t.stream && getenv("FH_STREAM") != nullptr;
    env_uint("FH_STREAM_REFILL", 1, 64, t.stream_refill);
    env_uint("FH_STREAM_MIN_RAYS", 0, 65535, t.stream_min_rays);
    env_off("FH_SORT_SMALL", t.sort_small);
    env_off("FH_OVERLAP", t.overlap_secondary);
    env_off("FH_MERGE", t.merge_trace);
    env_uint("FH_SKY_BLOCKS", 0, 64, t.sky_blocks_per_cu);
    if (const char* e = getenv("FH_POISON")) t.poison_pools = e[0] == '1';
    env_off("FH_SKY_SPLIT", t.sky_split);
    env_uint("FH_SKY_SPLIT_MIN_LOG2", 0, 40, t.sky_split_min_log2);
    env_uint("FH_STACK_LDS", 1, 99, t.stack_lds_entries);
    env_uint("FH_SHADE_WGS", 2, 3, t.shade_wgs);
    env_uint("FH_STREAM_CHUNK", 16, 65535, t.stream_chunk);
    t.stream_chunk_fixed = getenv("FH_STREAM_CHUNK") != nullptr;
    if (t.stream_chunk_fixed) t.stream_chunk_closest = t.stream_chunk;  // (FH_STREAM_CHUNK alone sets both launches)
    env_uint("FH_STREAM_CHUNK_CLOSEST", 16, 65535, t.stream_chunk_closest);
    env_uint("FH_TAIL_DEPTH", 0, 64, t.tail_depth);
    env_uint("FH_TAIL_PATHS", 64, 1 << 30, t.tail_paths);
    env_off("FH_SORT", t.sort_queues);
    env_uint("FH_SORT_ONEPASS", 0, 2, t.sort_onepass);
    env_uint("FH_BOTTOM_UP", 0, 2, t.bottom_up);
    t.debug_tail = getenv("FH_DEBUG_TAIL") != nullptr;
    if (const char* e = getenv("FH_FORCE_ALPHA")) t.force_alpha = e[0] == '1';
  }
  {  // the sky-pixel kernel's stream has the lowest priority: its workgroups -- pure arithmetic, 110 registers -- take what the passes leave instead of the wave slots the
     // traversal launches want (FH_SKY_PRIO=0: default priority; profiles/README.md r4-13)
    int least = 0, greatest = 0;
    const char* e = getenv("FH_SKY_PRIO");
    const bool low = !(e && e[0] == '0') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
    if ((low ? hipStreamCreateWithPriority(&ctx->sky_stream, hipStreamNonBlocking, least) : hipStreamCreateWithFlags(&ctx->sky_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate failed");
  }
  (void)hipEventCreateWithFlags(&ctx->ev_sky, hipEventDisableTiming);
  if (hipMalloc((void**)&ctx->d_split_counters, 16) != hipSuccess) return bail("hipMalloc failed");
  (void)hipMemsetAsync(ctx->d_split_counters, 0, 16, ctx->stream);
  (void)hipEventCreate(&ctx->ev_render_begin);
  (void)hipEventCreate(&ctx->ev_render_end);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail("hipStreamSynchronize failed");  // the fills above are done before any upload (synchronous copies, ordered with no stream) touches what they cleared
  *out = ctx;
  return FH_OK;
}

int fh_ctx_destroy(fh_ctx* ctx)
{
  if (!ctx) return FH_E_INVALID;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  pool_release(ctx);
  void* ptrs[] = {ctx->d_sample_issued, ctx->d_sobol, ctx->d_sobol_bytes, ctx->d_alpha_rec, ctx->d_lut_refl, ctx->d_lut_sheen, ctx->d_face_rec, ctx->d_face_cls, ctx->d_materials, ctx->d_lights, ctx->d_bvh2_nodes, ctx->d_bvh2_tris,
                  ctx->d_bvh8_nodes, ctx->d_bvh8_tris, ctx->d_sample_count, ctx->d_owned, ctx->d_trace_counters, ctx->d_texels, ctx->d_textures, ctx->d_srgb_lut, ctx->d_ibl,
                  ctx->d_bloom_weights, ctx->d_quirk_seen, ctx->d_quirk_aov, ctx->d_obj_vertices, ctx->d_obj_normals, ctx->d_obj_texcoords, ctx->d_obj_indices, ctx->d_face_meta, ctx->d_o2w, ctx->d_w2o,
                  ctx->d_bvh8_box, ctx->d_denoise_tmp[0], ctx->d_denoise_tmp[1], ctx->d_hosek, ctx->d_owned_xy, ctx->d_stack_spill, ctx->d_bvh8_parent, ctx->d_face_node};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (fh_ctx::ShardList& c : ctx->shard_lists)
    if (c.d_owned) (void)hipFree(c.d_owned);
  if (ctx->frame_map.d_all) (void)hipFree(ctx->frame_map.d_all);
  for (int k = 0; k < 4; ++k) if (ctx->d_split[k]) (void)hipFree(ctx->d_split[k]);
  if (ctx->d_split_counters) (void)hipFree(ctx->d_split_counters);
  if (ctx->sky_stream) { (void)hipStreamSynchronize(ctx->sky_stream); (void)hipStreamDestroy(ctx->sky_stream); }
  if (ctx->ev_sky) (void)hipEventDestroy(ctx->ev_sky);
  for (auto& s : ctx->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  for (auto e : ctx->ev_bounce) (void)hipEventDestroy(e);
  for (int k = 0; k < 3; ++k) {
  }
  for (int k = 0; k < 3; ++k) {
    if (ctx->h_counters[k]) (void)hipHostFree(ctx->h_counters[k]);
    (void)hipEventDestroy(ctx->ev_counters[k]);
    (void)hipEventDestroy(ctx->ev_gen[k]);
    (void)hipEventDestroy(ctx->ev_acc[k]);
  }
  (void)hipEventDestroy(ctx->ev_enter);
  for (int k = 0; k < 2; ++k) (void)hipStreamDestroy(ctx->aux_stream[k]);
  (void)hipEventDestroy(ctx->ev_render_begin);
  (void)hipEventDestroy(ctx->ev_render_end);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return FH_OK;
}

const char* fh_last_error(fh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int fh_set_flags(fh_ctx* ctx, uint32_t flags)
{
  CTX_CHECK(ctx);
  ctx->flags = flags;
  return FH_OK;
}
int fh_get_flags(fh_ctx* ctx, uint32_t* flags)
{
  CTX_CHECK(ctx);
  if (!flags) return fail(ctx, FH_E_INVALID, "fh_get_flags: null argument");
  *flags = ctx->flags;
  return FH_OK;
}

int fh_set_path_pool(fh_ctx* ctx, uint32_t target)
{
  CTX_CHECK(ctx);
  if (target == 0) return fail(ctx, FH_E_INVALID, "fh_set_path_pool: zero");
  (void)hipStreamSynchronize(ctx->stream);
  pool_release(ctx);
  ctx->pool_target = target;
  ctx->pool_target_by_caller = true;
  return FH_OK;
}

int fh_path_pool_bytes(fh_ctx* ctx, uint64_t* bytes_per_path, uint32_t* pools)
{
  CTX_CHECK(ctx);
  if (!bytes_per_path || !pools) return fail(ctx, FH_E_INVALID, "fh_path_pool_bytes: null argument");
  *bytes_per_path = pool_bytes_per_path(ctx);
  *pools = (uint32_t)ctx->n_slots;
  return FH_OK;
}

int fh_kat_face_classes(fh_ctx* ctx, uint8_t* out, uint32_t n)
{
  CTX_CHECK(ctx);
  if (!out || n != ctx->n_faces) return fail(ctx, FH_E_INVALID, "fh_kat_face_classes: one byte per face of the uploaded scene");
  if (n) FH_HIP(hipMemcpy(out, ctx->d_face_cls, n, hipMemcpyDeviceToHost));
  return FH_OK;
}

int fh_alpha_face_counts(fh_ctx* ctx, uint32_t counts[4])
{
  CTX_CHECK(ctx);
  if (!counts) return fail(ctx, FH_E_INVALID, "fh_alpha_face_counts: null argument");
  for (int k = 0; k < 4; ++k) counts[k] = ctx->alpha_face_counts[k];
  return FH_OK;
}

int fh_alpha_cell_counts(fh_ctx* ctx, uint64_t counts[3])
{
  CTX_CHECK(ctx);
  if (!counts) return fail(ctx, FH_E_INVALID, "fh_alpha_cell_counts: null argument");
  for (int k = 0; k < 3; ++k) counts[k] = ctx->alpha_cell_counts[k];
  return FH_OK;
}

int fh_kat_alpha_records(fh_ctx* ctx, uint32_t* out, uint32_t n_faces)
{
  CTX_CHECK(ctx);
  if (!out || n_faces != ctx->n_faces) return fail(ctx, FH_E_INVALID, "fh_kat_alpha_records: 32 words per face of the uploaded scene");
  if (!ctx->d_alpha_rec) { std::memset(out, 0, 128ull * n_faces); return FH_OK; }
  FH_HIP(hipMemcpy(out, ctx->d_alpha_rec, 128ull * n_faces, hipMemcpyDeviceToHost));
  return FH_OK;
}

int fh_path_pool_allocated(fh_ctx* ctx, uint64_t* bytes, uint64_t* paths)
{
  CTX_CHECK(ctx);
  if (!bytes || !paths) return fail(ctx, FH_E_INVALID, "fh_path_pool_allocated: null argument");
  *bytes = 0; *paths = 0;
  for (int k = 0; k < 3; ++k) { *bytes += ctx->pool_alloc_bytes[k]; *paths += ctx->pool[k].capacity; }
  return FH_OK;
}

int fh_set_tail_depth(fh_ctx* ctx, uint32_t depth)
{
  CTX_CHECK(ctx);
  ctx->tail_depth = depth;  // 0 = adaptive; bounce 0 always runs as wavefront kernels (it writes the first-hit AOVs)
  return FH_OK;
}

int fh_scene_upload(fh_ctx* ctx, const fh_scene_desc* s)
{
  CTX_CHECK(ctx);
  if (!s || !s->vertices || !s->normals || !s->texcoords || !s->indices || !s->material_ids || !s->materials || s->n_materials == 0)
    return fail(ctx, FH_E_INVALID, "fh_scene_upload: missing arrays");
  for (uint32_t i = 0; i < s->n_materials; ++i) {
    const fh_material& m = s->materials[i];
    const int32_t ids[11] = {m.base_color_texture_id, m.specular_color_texture_id, m.specular_roughness_texture_id, m.metalness_texture_id, m.metallic_roughness_texture_id, m.coat_texture_id,
                             m.coat_roughness_texture_id, m.emission_texture_id, m.heightmap_texture_id, m.normalmap_texture_id, m.alpha_texture_id};
    for (int32_t id : ids)
      if (id < -1 || id >= (int32_t)s->n_textures) return fail(ctx, FH_E_INVALID, "fh_scene_upload: material references a texture id outside [0, n_textures)");
  }
  if (s->n_textures && !s->textures) return fail(ctx, FH_E_INVALID, "fh_scene_upload: n_textures > 0 but textures == NULL");
  {
    const int rc = upload_textures(ctx, s->n_textures, s->textures);
    if (rc) return rc;
  }
  ctx->h_vertices.assign(s->vertices, s->vertices + 3ull * s->n_vertices);
  ctx->h_normals.assign(s->normals, s->normals + 3ull * s->n_vertices);
  ctx->h_texcoords.assign(s->texcoords, s->texcoords + 2ull * s->n_vertices);
  ctx->h_indices.assign(s->indices, s->indices + 3ull * s->n_faces);
  ctx->h_material_ids.assign(s->material_ids, s->material_ids + s->n_faces);
  if (s->instance_ids) ctx->h_instance_ids.assign(s->instance_ids, s->instance_ids + s->n_faces);
  else ctx->h_instance_ids.clear();
  ctx->h_materials.assign(s->materials, s->materials + s->n_materials);
  if (s->n_instances && s->object_to_world && s->world_to_object) {
    ctx->h_o2w.assign(s->object_to_world, s->object_to_world + 12ull * s->n_instances);
    ctx->h_w2o.assign(s->world_to_object, s->world_to_object + 12ull * s->n_instances);
  } else {
    ctx->h_o2w.assign(kIdentity12, kIdentity12 + 12);
    ctx->h_w2o.assign(kIdentity12, kIdentity12 + 12);
  }
  const int rc = rebuild_device_scene(ctx);
  if (rc) return rc;
  ctx->scene_loaded = true;
  ctx->builder_choice = 0;  // new geometry: let the next build choose its builder again
  return FH_OK;
}

int fh_set_transforms(fh_ctx* ctx, uint32_t n, const float* o2w, const float* w2o)
{
  CTX_CHECK(ctx);
  if (!ctx->scene_loaded || !o2w || !w2o || n == 0) return fail(ctx, FH_E_INVALID, "fh_set_transforms: no scene / null arrays");
  // a frame of an animation that only moves the camera (rtcamp8's scene) leaves every instance where it was: keep the world-space
  // geometry and the acceleration structure (fh_bvh_build returns at once while bvh_valid holds)
  if (ctx->h_o2w.size() == 12ull * n && ctx->h_w2o.size() == 12ull * n && std::memcmp(ctx->h_o2w.data(), o2w, 48ull * n) == 0 &&
      std::memcmp(ctx->h_w2o.data(), w2o, 48ull * n) == 0)
    return FH_OK;
  (void)hipStreamSynchronize(ctx->stream);
  for (int k = 0; k < 2; ++k) (void)hipStreamSynchronize(ctx->aux_stream[k]);  // no pass may still be reading the face records
  uint32_t max_inst = 0;
  for (uint32_t i : ctx->h_instance_ids) max_inst = i > max_inst ? i : max_inst;
  if (max_inst >= n) return fail(ctx, FH_E_INVALID, "fh_set_transforms: fewer transforms than the scene has instances");
  ctx->h_o2w.assign(o2w, o2w + 12ull * n);
  ctx->h_w2o.assign(w2o, w2o + 12ull * n);
  return transform_faces(ctx);  // topology, classes and lights are unchanged: only the world-space records move (and the BVH is refitted)
}

int fh_bvh_build(fh_ctx* ctx)
{
  CTX_CHECK(ctx);
  if (!ctx->scene_loaded) return fail(ctx, FH_E_INVALID, "fh_bvh_build: no scene");
  if (ctx->bvh_valid) return FH_OK;  // geometry unchanged since the last build
  return bvh_build_device(ctx);
}

int fh_scene_n_lights(fh_ctx* ctx, uint32_t* out)
{
  CTX_CHECK(ctx);
  if (!out) return FH_E_INVALID;
  *out = ctx->n_lights;
  return FH_OK;
}

int fh_set_directional_light(fh_ctx* ctx, const float* le, const float* dir, float angle)
{
  CTX_CHECK(ctx);
  if (!le || !dir) return FH_E_INVALID;
  const f3 d = normalize(mk3(dir[0], dir[1], dir[2]));  // renderer.h:560
  ctx->has_dir = true;
  for (int k = 0; k < 3; ++k) ctx->dir_le[k] = le[k];
  ctx->dir_dir[0] = d.x; ctx->dir_dir[1] = d.y; ctx->dir_dir[2] = d.z;
  ctx->sun_dir[0] = d.x; ctx->sun_dir[1] = d.y; ctx->sun_dir[2] = d.z;  // renderer.h:563
  ctx->dir_angle = angle;
  return FH_OK;
}
int fh_clear_directional_light(fh_ctx* ctx)
{
  CTX_CHECK(ctx);
  ctx->has_dir = false;
  return FH_OK;
}
int fh_set_sky_intensity(fh_ctx* ctx, float v)
{
  CTX_CHECK(ctx);
  ctx->sky_intensity = v;
  return FH_OK;
}
int fh_load_arhosek_sky(fh_ctx* ctx, float turbidity, float albedo)
{
  CTX_CHECK(ctx);
  if (!(turbidity >= 1.0f && turbidity <= 10.0f)) return fail(ctx, FH_E_INVALID, "turbidity must be in [1,10]");
  const float elevation = (float)(0.5f * 3.14159265358979323846 - fhe_acos(clampf(ctx->sun_dir[1], -1.0f, 1.0f)));  // renderer.h:592-601
  ctx->hosek = hosek_cook(kHosekRgb, turbidity, albedo, elevation);
  ctx->has_hosek = true;
  FH_HIP(hipMemcpyAsync(ctx->d_hosek, &ctx->hosek, sizeof ctx->hosek, hipMemcpyHostToDevice, ctx->stream));  // (ordered after the frames already submitted)
  FH_HIP(hipStreamSynchronize(ctx->stream));
  return FH_OK;
}
int fh_clear_arhosek_sky(fh_ctx* ctx)
{
  CTX_CHECK(ctx);
  ctx->has_hosek = false;
  return FH_OK;
}
int fh_load_ibl(fh_ctx* ctx, const float* rgba, uint32_t w, uint32_t h)
{
  CTX_CHECK(ctx);
  if (!rgba || w == 0 || h == 0) return fail(ctx, FH_E_INVALID, "fh_load_ibl: empty image");
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_ibl) { (void)hipFree(ctx->d_ibl); ctx->d_ibl = nullptr; }
  FH_HIP(hipMalloc((void**)&ctx->d_ibl, sizeof(float) * 4ull * w * h));
  FH_HIP(hipMemcpy(ctx->d_ibl, rgba, sizeof(float) * 4ull * w * h, hipMemcpyHostToDevice));
  ctx->ibl_w = w;
  ctx->ibl_h = h;
  return FH_OK;
}
int fh_clear_ibl(fh_ctx* ctx)
{
  CTX_CHECK(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_ibl) { (void)hipFree(ctx->d_ibl); ctx->d_ibl = nullptr; }
  return FH_OK;
}

int fh_init_render_states(fh_ctx* ctx)
{
  CTX_CHECK(ctx);
  if (!ctx->d_sample_count) return fail(ctx, FH_E_INVALID, "resolution not set");
  FH_HIP(hipMemsetAsync(ctx->d_sample_count, 0, 4ull * ctx->width * ctx->height, ctx->stream));
  FH_HIP(hipMemsetAsync(ctx->d_sample_issued, 0, 4ull * ctx->width * ctx->height, ctx->stream));
  return FH_OK;
}

int fh_set_resolution(fh_ctx* ctx, uint32_t w, uint32_t h)
{
  CTX_CHECK(ctx);
  if (w == 0 || h == 0) return fail(ctx, FH_E_INVALID, "zero resolution");
  if (w > 65535u || h > 65535u) return fail(ctx, FH_E_UNSUPPORTED, "frames wider or higher than 65535 pixels are not supported");  // (before anything is changed: the context keeps its resolution)
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_sample_count) { (void)hipFree(ctx->d_sample_count); ctx->d_sample_count = nullptr; }
  if (ctx->d_sample_issued) { (void)hipFree(ctx->d_sample_issued); ctx->d_sample_issued = nullptr; }
  ctx->width = w; ctx->height = h;
  FH_HIP(hipMalloc((void**)&ctx->d_sample_count, 4ull * w * h));
  FH_HIP(hipMalloc((void**)&ctx->d_sample_issued, 4ull * w * h));
  const int rc = rebuild_ownership(ctx);
  if (rc) return rc;
  return fh_init_render_states(ctx);
}

int fh_set_tile_shard(fh_ctx* ctx, uint32_t rank, uint32_t world, uint32_t tw, uint32_t th)
{
  CTX_CHECK(ctx);
  if (world == 0 || rank >= world || tw == 0 || th == 0) return fail(ctx, FH_E_INVALID, "bad shard");
  (void)hipStreamSynchronize(ctx->stream);
  ctx->shard_rank = rank; ctx->shard_world = world; ctx->tile_w = tw; ctx->tile_h = th;
  return rebuild_ownership(ctx);
}

int fh_owned_pixel_count(fh_ctx* ctx, uint32_t* out)
{
  CTX_CHECK(ctx);
  if (!out) return FH_E_INVALID;
  *out = ctx->n_owned;
  return FH_OK;
}

int fh_pack_owned(fh_ctx* ctx, const float* layer, uint32_t fpp, float* packed)
{
  CTX_CHECK(ctx);
  if (!layer || !packed || fpp == 0) return FH_E_INVALID;
  if (ctx->n_owned) hipLaunchKernelGGL(k_pack, dim3((ctx->n_owned * fpp + 255) / 256), dim3(256), 0, ctx->stream, layer, ctx->d_owned, ctx->n_owned, fpp, packed);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int fh_unpack_shard(fh_ctx* ctx, uint32_t rank, uint32_t world, const float* packed, uint32_t fpp, float* layer)
{
  CTX_CHECK(ctx);
  if (!layer || !packed || fpp == 0 || world == 0 || rank >= world) return FH_E_INVALID;
  if (ctx->width > 65535u || ctx->height > 65535u) return fail(ctx, FH_E_UNSUPPORTED, "frames wider or higher than 65535 pixels are not supported");
  // ownership list of (rank, world) with this context's tile size and resolution: built once and kept (a presented frame unpacks every rank's shard), so that the call
  // is one asynchronous launch on the context stream -- no allocation, no host copy, no synchronisation per frame
  fh_ctx::ShardList* sl = nullptr;
  for (fh_ctx::ShardList& c : ctx->shard_lists)
    if (c.rank == rank && c.world == world && c.width == ctx->width && c.height == ctx->height && c.tile_w == ctx->tile_w && c.tile_h == ctx->tile_h) sl = &c;
  if (!sl) {
    fh_ctx tmp;
    tmp.device = ctx->device; tmp.width = ctx->width; tmp.height = ctx->height; tmp.tile_w = ctx->tile_w; tmp.tile_h = ctx->tile_h; tmp.shard_rank = rank; tmp.shard_world = world;
    const int rc = rebuild_ownership(&tmp);
    if (rc) { ctx->err = tmp.err; return rc; }
    if (tmp.d_owned_xy) (void)hipFree(tmp.d_owned_xy);
    if (ctx->shard_lists.size() >= 64u) {  // (a caller cycling through many splits: drop the oldest)
      (void)hipStreamSynchronize(ctx->stream);
      if (ctx->shard_lists.front().d_owned) (void)hipFree(ctx->shard_lists.front().d_owned);
      ctx->shard_lists.erase(ctx->shard_lists.begin());
    }
    ctx->shard_lists.push_back(fh_ctx::ShardList{rank, world, ctx->width, ctx->height, ctx->tile_w, ctx->tile_h, tmp.d_owned, tmp.n_owned});
    sl = &ctx->shard_lists.back();
  }
  if (sl->n_owned) hipLaunchKernelGGL(k_unpack, dim3((sl->n_owned * fpp + 255) / 256), dim3(256), 0, ctx->stream, packed, sl->d_owned, sl->n_owned, fpp, layer);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int fh_unpack_shards(fh_ctx* ctx, uint32_t world, const float* const* packed, uint32_t fpp, float* layer)
{
  CTX_CHECK(ctx);
  if (!layer || !packed || fpp == 0 || world == 0) return FH_E_INVALID;
  for (uint32_t r = 0; r < world; ++r)
    if (!packed[r]) return fail(ctx, FH_E_INVALID, "fh_unpack_shards: null shard pointer");
  if (world > kMaxShardsPerLaunch) {  // (more ranks than one launch's argument block holds: one launch per rank)
    for (uint32_t r = 0; r < world; ++r) { const int rc = fh_unpack_shard(ctx, r, world, packed[r], fpp, layer); if (rc) return rc; }
    return FH_OK;
  }
  if (ctx->width > 65535u || ctx->height > 65535u) return fail(ctx, FH_E_UNSUPPORTED, "frames wider or higher than 65535 pixels are not supported");
  // the ownership lists of all `world` ranks, one after the other: built once per (world, resolution, tile size) and kept, so a presented frame is ONE asynchronous launch
  fh_ctx::FrameMap& fm = ctx->frame_map;
  if (!(fm.d_all && fm.world == world && fm.width == ctx->width && fm.height == ctx->height && fm.tile_w == ctx->tile_w && fm.tile_h == ctx->tile_h)) {
    (void)hipStreamSynchronize(ctx->stream);
    if (fm.d_all) (void)hipFree(fm.d_all);
    fm = fh_ctx::FrameMap{};
    std::vector<uint32_t> all;
    fm.start.assign(world + 1u, 0u);
    for (uint32_t r = 0; r < world; ++r) { owned_list(ctx->width, ctx->height, ctx->tile_w, ctx->tile_h, r, world, all, nullptr); fm.start[r + 1u] = (uint32_t)all.size(); }
    FH_HIP(hipMalloc((void**)&fm.d_all, all.empty() ? 16 : all.size() * 4));
    if (!all.empty()) FH_HIP(hipMemcpy(fm.d_all, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    fm.world = world; fm.width = ctx->width; fm.height = ctx->height; fm.tile_w = ctx->tile_w; fm.tile_h = ctx->tile_h;
  }
  ShardSources src{};
  src.world = world;
  for (uint32_t r = 0; r < world; ++r) { src.packed[r] = packed[r]; src.start[r] = fm.start[r]; }
  src.start[world] = fm.start[world];
  const uint32_t n = fm.start[world];
  if (n) hipLaunchKernelGGL(k_unpack_all, dim3((uint32_t)(((size_t)n * fpp + 255) / 256)), dim3(256), 0, ctx->stream, src, fm.d_all, n, fpp, layer);
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int fh_render(fh_ctx* ctx, const fh_camera* cam, const float* bg, const fh_render_layers* layers, uint32_t n_samples, uint32_t max_depth, uint32_t seed)
{
  CTX_CHECK(ctx);
  if (!cam || !bg || !layers || !layers->beauty || !layers->position || !layers->depth || !layers->normal || !layers->texcoord || !layers->albedo)
    return fail(ctx, FH_E_INVALID, "fh_render: null argument");
  return render_submit(ctx, cam, bg, layers, n_samples, max_depth, seed);
}

int fh_sync(fh_ctx* ctx)
{
  CTX_CHECK(ctx);
  if (ctx->render_pending) (void)hipEventRecord(ctx->ev_render_end, ctx->stream);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->render_pending) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->ev_render_begin, ctx->ev_render_end) == hipSuccess) ctx->stats.render_ms += ms;
    ctx->render_pending = false;
  }
  for (auto& s : ctx->spans) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      if (s.kind == 0) ctx->stats.trace_closest_ms += ms;
      else if (s.kind == 1) ctx->stats.trace_shadow_ms += ms;
      else if (s.kind == 3) ctx->stats.tail_ms += ms;
      else if (s.kind == 4) ctx->stats.generate_ms += ms;
      else if (s.kind == 5) ctx->stats.accumulate_ms += ms;
      else if (s.kind == 6) ctx->stats.queue_ms += ms;
      else if (s.kind == 7) ctx->stats.post_ms += ms;
      else ctx->stats.shade_ms += ms;
    }
    ctx->event_pool.push_back(s.a);
    ctx->event_pool.push_back(s.b);
  }
  ctx->spans.clear();
  if (ctx->stats.sky_pixel_samples && ctx->d_split_counters) {  // k_sky_pixels re-runs the scene-bounds test of every sample it renders: a hit there means the conservative split was wrong
    uint32_t violations = 0;
    FH_HIP(hipMemcpy(&violations, ctx->d_split_counters + 2, 4, hipMemcpyDeviceToHost));
    if (violations) {
      (void)hipMemsetAsync(ctx->d_split_counters + 2, 0, 4, ctx->stream);
      return fail(ctx, FH_E_HIP, "sky-pixel split: " + std::to_string(violations) + " pixels classified as sky have rays that reach the scene bounds (set FH_SKY_SPLIT=0 and report)");
    }
  }
  if (ctx->flags & FH_FLAG_TIME_KERNELS) {  // shader cycles and 100 MHz ticks the waves of the streaming traversal kernels have summed up (fh_device.h: ClockStamp)
    unsigned long long c[4];
    FH_HIP(hipMemcpy(c, ctx->d_trace_counters + 27, sizeof c, hipMemcpyDeviceToHost));
    ctx->stats.clk_cycles_closest = c[0]; ctx->stats.clk_ticks_closest = c[1]; ctx->stats.clk_cycles_shadow = c[2]; ctx->stats.clk_ticks_shadow = c[3];
  }
  if (ctx->flags & FH_FLAG_COUNT_TRAVERSAL) {
    unsigned long long c[32];
    FH_HIP(hipMemcpy(c, ctx->d_trace_counters, sizeof c, hipMemcpyDeviceToHost));
    ctx->stats.wave_node_steps_closest = c[6]; ctx->stats.wave_tri_steps_closest = c[7]; ctx->stats.wave_node_steps_shadow = c[8]; ctx->stats.wave_tri_steps_shadow = c[9];
    for (int k = 0; k < 8; ++k) { ctx->stats.hist_nodes_closest[k] = c[10 + k]; ctx->stats.hist_nodes_shadow[k] = c[18 + k]; }
    ctx->stats.nodes_closest = c[0]; ctx->stats.tris_closest = c[1]; ctx->stats.rays_closest = c[2];
    ctx->stats.nodes_shadow = c[3]; ctx->stats.tris_shadow = c[4]; ctx->stats.rays_shadow = c[5];
    ctx->stats.shaded_hits = c[26];
  }
  return FH_OK;
}

int fh_get_stats(fh_ctx* ctx, fh_stats* out)
{
  CTX_CHECK(ctx);
  if (!out) return FH_E_INVALID;
  *out = ctx->stats;
  return FH_OK;
}
int fh_kernel_info(fh_ctx* ctx, int which, uint32_t out[6])
{
  CTX_CHECK(ctx);
  if (!out || which < 0 || which > 1 + (int)kMaxClasses) return fail(ctx, FH_E_INVALID, "fh_kernel_info: which must be 0 (closest hit), 1 (secondary rays) or 2 + a shading class of the scene");
  return kernel_info(ctx, which, out);
}
// test hook (fredholm_hip_test.h): where the scene's first-hit rays start, and what the probing passes have counted so far
int fh_kat_ray_start(fh_ctx* ctx, double out[5])
{
  CTX_CHECK(ctx);
  if (!out) return fail(ctx, FH_E_INVALID, "fh_kat_ray_start: null argument");
  out[0] = (double)ctx->bu_choice; out[1] = ctx->bu_items[0]; out[2] = ctx->bu_items[1]; out[3] = ctx->bu_cost[0]; out[4] = ctx->bu_cost[1];
  return FH_OK;
}
int fh_reset_stats(fh_ctx* ctx)
{
  CTX_CHECK(ctx);
  const double build_ms = ctx->stats.bvh_build_ms;
  const uint64_t nodes = ctx->stats.bvh_nodes, nb = ctx->stats.bvh_node_bytes, tb = ctx->stats.bvh_tri_bytes, depth = ctx->stats.bvh_depth;
  ctx->stats = fh_stats{};
  ctx->stats.bvh_build_ms = build_ms; ctx->stats.bvh_nodes = nodes; ctx->stats.bvh_node_bytes = nb; ctx->stats.bvh_tri_bytes = tb; ctx->stats.bvh_depth = depth;
  FH_HIP(hipMemsetAsync(ctx->d_trace_counters, 0, 32 * sizeof(unsigned long long), ctx->stream));
  return FH_OK;
}

int fh_post_process(fh_ctx* ctx, const float* in, float* hi, float* tmp, int w, int h, const fh_post_params* pp, float* out)
{
  CTX_CHECK(ctx);
  if (!in || !hi || !tmp || !out || !pp || w <= 0 || h <= 0) return fail(ctx, FH_E_INVALID, "fh_post_process: bad argument");
  return post_process_submit(ctx, in, hi, tmp, w, h, pp, out);
}

int fh_denoise(fh_ctx* ctx, uint32_t width, uint32_t height, const float* beauty, const float* normal, const float* albedo, float* denoised, int upscale2x)
{
  CTX_CHECK(ctx);
  if (!beauty || !normal || !albedo || !denoised || width == 0 || height == 0 || width > 32768 || height > 32768) return fail(ctx, FH_E_INVALID, "fh_denoise: bad argument");
  return denoise_submit(ctx, (int)width, (int)height, beauty, normal, albedo, denoised, upscale2x ? 1 : 0);
}

// OpenGL interop (cwl::CUDAGLBuffer, cwl/include/cwl/buffer.h:88-143: cuGraphicsGLRegisterBuffer + map + mapped pointer, unmapped and
// unregistered in the destructor).  Needs a current OpenGL context on the calling thread, like the reference.
int fh_gl_register_buffer(fh_ctx* ctx, unsigned int gl_buffer, void** resource, void** device_ptr, uint64_t* bytes)
{
  CTX_CHECK(ctx);
  if (!resource || !device_ptr) return fail(ctx, FH_E_INVALID, "fh_gl_register_buffer: null argument");
  *resource = nullptr; *device_ptr = nullptr;
  hipGraphicsResource_t res = nullptr;
  FH_HIP(hipGraphicsGLRegisterBuffer(&res, (GLuint)gl_buffer, hipGraphicsRegisterFlagsNone));
  hipError_t e = hipGraphicsMapResources(1, &res, ctx->stream);
  size_t size = 0;
  void* ptr = nullptr;
  if (e == hipSuccess) e = hipGraphicsResourceGetMappedPointer(&ptr, &size, res);
  if (e != hipSuccess) {
    (void)hipGraphicsUnregisterResource(res);
    return fail(ctx, FH_E_HIP, std::string("fh_gl_register_buffer: ") + hipGetErrorString(e));
  }
  *resource = (void*)res; *device_ptr = ptr;
  if (bytes) *bytes = (uint64_t)size;
  return FH_OK;
}
int fh_gl_unregister_buffer(fh_ctx* ctx, void* resource)
{
  CTX_CHECK(ctx);
  if (!resource) return FH_OK;
  hipGraphicsResource_t res = (hipGraphicsResource_t)resource;
  (void)hipStreamSynchronize(ctx->stream);
  FH_HIP(hipGraphicsUnmapResources(1, &res, ctx->stream));
  FH_HIP(hipGraphicsUnregisterResource(res));
  return FH_OK;
}

int fh_malloc(fh_ctx* ctx, uint64_t bytes, void** out)
{
  CTX_CHECK(ctx);
  if (!out) return FH_E_INVALID;
  FH_HIP(hipMalloc(out, bytes ? bytes : 16));
  return FH_OK;
}
int fh_free(fh_ctx* ctx, void* p)
{
  CTX_CHECK(ctx);
  FH_HIP(hipFree(p));
  return FH_OK;
}
int fh_memset(fh_ctx* ctx, void* p, int v, uint64_t bytes)
{
  CTX_CHECK(ctx);
  FH_HIP(hipMemsetAsync(p, v, bytes, ctx->stream));
  return FH_OK;
}
int fh_copy_to_device(fh_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
  CTX_CHECK(ctx);
  FH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  FH_HIP(hipStreamSynchronize(ctx->stream));
  return FH_OK;
}
int fh_image_load_rgba8(const char* path, int flip_vertically, uint32_t* width, uint32_t* height, uint8_t** rgba8)
{
  if (!path || !width || !height || !rgba8) { g_create_error = "fh_image_load_rgba8: null argument"; return FH_E_INVALID; }
  try {
    const fredholm::image_io::Image8 img = fredholm::image_io::load_rgba8(path, flip_vertically != 0);
    uint8_t* out = (uint8_t*)std::malloc(img.rgba.size() ? img.rgba.size() : 1);
    if (!out) { g_create_error = "fh_image_load_rgba8: out of memory"; return FH_E_INVALID; }
    std::memcpy(out, img.rgba.data(), img.rgba.size());
    *width = (uint32_t)img.width; *height = (uint32_t)img.height; *rgba8 = out;
    return FH_OK;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return FH_E_INVALID;
  }
}
void fh_image_free(uint8_t* rgba8) { std::free(rgba8); }
int fh_copy_on_device(fh_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
  CTX_CHECK(ctx);
  FH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return FH_OK;
}
int fh_copy_to_host(fh_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
  CTX_CHECK(ctx);
  FH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  FH_HIP(hipStreamSynchronize(ctx->stream));
  return FH_OK;
}
void* fh_stream(fh_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

}  // extern "C"
