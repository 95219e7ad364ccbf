// camera.cc — reference: VerStarting/camera.cc.  Sensor::Reset stays on the
// host on purpose: sin/cos must be glibc's for the start/delta vectors to be
// bit-identical to the reference's; the kernel only does GetRay's arithmetic.
#include "camera.h"

#include <cstring>

namespace raytracer {

V3D Camera::GetDirection() const {
  // roll does not move the forward vector
  const V3D forward{0.0, 0.0, 1.0};
  return M4D::RotationYDeg(yaw) * M4D::RotationXDeg(pitch) * forward;
}

Camera::Sensor Camera::GetSensor(int width, int height) const {
  Sensor s;
  s.width = width;
  s.height = height;
  s.cam = this;
  s.Reset();
  return s;
}

void Camera::Sensor::Reset() {
  // vertical angle of view from the aspect ratio (camera.cc:29)
  const double aov_v = (V3D::basetype(height) / V3D::basetype(width)) * cam->aov;

  // frustum corner directions = Z-rotation (up/down) after Y-rotation
  // (left/right) of the forward vector (camera.cc:32-45)
  const M4D to_left = M4D::RotationYDeg(cam->aov / 2.0);
  const M4D to_right = M4D::RotationYDeg(-cam->aov / 2.0);
  const M4D to_top = M4D::RotationZDeg(aov_v / 2.0);
  const M4D to_bottom = M4D::RotationZDeg(-aov_v / 2.0);
  const V3D forward{0.0, 0.0, 1.0};
  V3D top_left = (to_top * to_left) * forward;
  // the reference builds its "right top" corner from the *bottom* rotation
  // (camera.cc:38); parity requires the same
  V3D top_right = (to_bottom * to_right) * forward;
  V3D bottom_left = (to_bottom * to_left) * forward;

  // aim the frustum (camera.cc:48-55)
  const M4D aim =
      M4D::RotationYDeg(cam->yaw) * M4D::RotationXDeg(cam->pitch) * M4D::RotationZDeg(cam->roll);
  top_left = aim * top_left;
  top_right = aim * top_right;
  bottom_left = aim * bottom_left;

  delta_scanline = (bottom_left - top_left) / V3D::basetype(height);
  delta_pixel = (top_right - top_left) / V3D::basetype(width);
  start_point = top_left;
}

Ray Camera::Sensor::GetRay(int x, int y) const {
  V3D direction = start_point + (delta_scanline * y) + (delta_pixel * x);
  direction.Norm();
  return {cam->origin, direction};
}

// 56-byte blob: origin, pitch, yaw, roll, aov as raw doubles (camera.cc:71-96)
void Camera::Serialize(std::vector<uint8_t>* bytes) {
  const double fields[7] = {origin.v[0], origin.v[1], origin.v[2], pitch, yaw, roll, aov};
  bytes->resize(kSerializedSize);
  memcpy(bytes->data(), fields, sizeof fields);
}

bool Camera::Deserialize(const std::vector<uint8_t>& bytes) {
  if (bytes.size() != kSerializedSize) return false;
  double fields[7];
  memcpy(fields, bytes.data(), sizeof fields);
  origin = {fields[0], fields[1], fields[2]};
  pitch = fields[3];
  yaw = fields[4];
  roll = fields[5];
  aov = fields[6];
  return true;
}

}  // namespace raytracer
