// fredholm/shared.h -- host-visible POD types of the reference's fredholm/include/fredholm/shared.h that the
// applications touch: Material (:100-142), RenderLayer (:201-208), DirectionalLight (:155-159).  No OptiX types.
#pragma once
#include "../fredholm_hip.h"
#include "types.h"

namespace fredholm
{

struct Material {
  float diffuse = 1.0f;
  float3 base_color = make_float3(1, 1, 1);
  int base_color_texture_id = -1;
  float diffuse_roughness = 0.0f;
  float specular = 1.0f;
  float3 specular_color = make_float3(1, 1, 1);
  int specular_color_texture_id = -1;
  float specular_roughness = 0.2f;
  int specular_roughness_texture_id = -1;
  float metalness = 0;
  int metalness_texture_id = -1;
  int metallic_roughness_texture_id = -1;
  float coat = 0;
  int coat_texture_id = -1;
  float3 coat_color = make_float3(1, 1, 1);
  float coat_roughness = 0.1f;
  int coat_roughness_texture_id = -1;
  float transmission = 0;
  float3 transmission_color = make_float3(1, 1, 1);
  float sheen = 0.0f;
  float3 sheen_color = make_float3(1.0f, 1.0f, 1.0f);
  float sheen_roughness = 0.3f;
  float subsurface = 0;
  float3 subsurface_color = make_float3(1.0f, 1.0f, 1.0f);
  float thin_walled = 0.0f;
  float emission = 0;
  float3 emission_color = make_float3(0, 0, 0);
  int emission_texture_id = -1;
  int heightmap_texture_id = -1;
  int normalmap_texture_id = -1;
  int alpha_texture_id = -1;
};
static_assert(sizeof(Material) == sizeof(fh_material), "Material must stay layout-compatible with fh_material");

struct RenderLayer {
  float4* beauty;
  float4* position;
  float* depth;
  float4* normal;
  float4* texcoord;
  float4* albedo;
};

}  // namespace fredholm
