// fredholm/camera.h -- host camera with the interface of the reference's fredholm::Camera
// (fredholm/include/fredholm/camera.h:22-135) without the glm dependency.  m_transform is the
// camera-to-world matrix = inverse(lookAt(origin, origin + 0.01 forward, up)), column-major like glm::mat4
// (m_transform[column][row]).
#pragma once
#include <cmath>

#include "../fredholm_hip.h"
#include "types.h"

namespace fredholm
{

enum class CameraMovement { FORWARD, BACKWARD, RIGHT, LEFT, UP, DOWN };

struct Vec3 {
  float x = 0, y = 0, z = 0;
};
struct Mat4 {
  float m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  float* operator[](int c) { return m[c]; }
  const float* operator[](int c) const { return m[c]; }
};

struct Camera {
  Mat4 m_transform;
  float m_fov;
  float m_F;
  float m_focus;
  float m_movement_speed;
  float m_look_around_speed;
  Vec3 m_origin, m_forward, m_right, m_up;
  float m_phi, m_theta;

  Camera() : m_fov(0.5f * float(M_PI)), m_F(8.0f), m_focus(10000.0f), m_movement_speed(10.0f), m_look_around_speed(0.1f), m_phi(270.0f), m_theta(90.0f) {}

  Camera(const float3& origin, float fov = 0.5f * float(M_PI), float F = 8.0f, float focus = 10000.0f, float movement_speed = 1.0f, float look_around_speed = 0.1f)
      : m_fov(fov), m_F(F), m_focus(focus), m_movement_speed(movement_speed), m_look_around_speed(look_around_speed), m_phi(270.0f), m_theta(90.0f)
  {
    m_origin = {origin.x, origin.y, origin.z};
    m_forward = {0, 0, -1};
    m_right = normalize(cross(m_forward, Vec3{0, 1, 0}));
    m_up = normalize(cross(m_right, m_forward));
    update_transform();
  }

  float3 get_origin() const { return make_float3(m_origin.x, m_origin.y, m_origin.z); }
  void set_origin(const float3& origin)
  {
    m_origin = {origin.x, origin.y, origin.z};
    update_transform();
  }

  void move(const CameraMovement& direction, float dt)
  {
    const float v = m_movement_speed * dt;
    switch (direction) {
      case CameraMovement::FORWARD: m_origin = add(m_origin, scale(m_forward, v)); break;
      case CameraMovement::BACKWARD: m_origin = add(m_origin, scale(m_forward, -v)); break;
      case CameraMovement::RIGHT: m_origin = add(m_origin, scale(m_right, v)); break;
      case CameraMovement::LEFT: m_origin = add(m_origin, scale(m_right, -v)); break;
      case CameraMovement::UP: m_origin = add(m_origin, scale(m_up, v)); break;
      case CameraMovement::DOWN: m_origin = add(m_origin, scale(m_up, -v)); break;
    }
    update_transform();
  }

  void lookAround(float d_phi, float d_theta)
  {
    m_phi += m_look_around_speed * d_phi;
    if (m_phi < 0.0f) m_phi = 360.0f;
    if (m_phi > 360.0f) m_phi = 0.0f;
    m_theta += m_look_around_speed * d_theta;
    if (m_theta < 0.0f) m_theta = 180.0f;
    if (m_theta > 180.0f) m_theta = 0.0f;
    const float phi = m_phi / 180.0f * float(M_PI), theta = m_theta / 180.0f * float(M_PI);
    m_forward = {std::cos(phi) * std::sin(theta), std::cos(theta), std::sin(phi) * std::sin(theta)};
    m_right = normalize(cross(m_forward, Vec3{0.0f, 1.0f, 0.0f}));
    m_up = normalize(cross(m_right, m_forward));
    update_transform();
  }

  // CameraParams (shared.h:59-64) as the C ABI wants them: 3x4 rows of m_transform (renderer.h:679-689)
  fh_camera to_c() const
  {
    fh_camera c{};
    for (int r = 0; r < 3; ++r)
      for (int col = 0; col < 4; ++col) c.transform[4 * r + col] = m_transform[col][r];
    c.fov = m_fov;
    c.F = m_F;
    c.focus = m_focus;
    return c;
  }

 private:
  static Vec3 add(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
  static Vec3 scale(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
  static Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
  static Vec3 normalize(Vec3 a)
  {
    const float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x / l, a.y / l, a.z / l};
  }
  void update_transform()
  {
    // inverse of a right-handed lookAt: columns right, up, -forward, origin
    const Vec3 f = normalize(m_forward);
    const Vec3 s = normalize(cross(f, m_up));
    const Vec3 u = cross(s, f);
    m_transform = Mat4{};
    m_transform[0][0] = s.x; m_transform[0][1] = s.y; m_transform[0][2] = s.z;
    m_transform[1][0] = u.x; m_transform[1][1] = u.y; m_transform[1][2] = u.z;
    m_transform[2][0] = -f.x; m_transform[2][1] = -f.y; m_transform[2][2] = -f.z;
    m_transform[3][0] = m_origin.x; m_transform[3][1] = m_origin.y; m_transform[3][2] = m_origin.z;
  }
};

}  // namespace fredholm
