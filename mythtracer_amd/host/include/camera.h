// camera.h — reference: VerStarting/camera.h:12-47, camera.cc.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include "math3d.h"
#include "ray.h"

namespace raytracer {
using math3d::M4D;
using math3d::V3D;

class Camera {
 public:
  // The image plane: per-pixel direction = start + ds*y + dp*x, normalised.
  class Sensor {
   public:
    Ray GetRay(int x, int y) const;
    // what the GPU kernel needs (mt_sensor of the C ABI)
    const V3D& StartPoint() const { return start_point; }
    const V3D& DeltaScanline() const { return delta_scanline; }
    const V3D& DeltaPixel() const { return delta_pixel; }

   private:
    void Reset();
    V3D delta_scanline, delta_pixel, start_point;
    int width, height;
    const Camera* cam;
    friend Camera;
  };

  V3D origin;
  V3D::basetype pitch, yaw, roll;  // degrees, about X, Y, Z
  V3D::basetype aov;               // horizontal angle of view, degrees

  V3D GetDirection() const;
  Sensor GetSensor(int width, int height) const;

  static const size_t kSerializedSize = sizeof(V3D) + 4 * sizeof(V3D::basetype);
  void Serialize(std::vector<uint8_t>* bytes);
  bool Deserialize(const std::vector<uint8_t>& bytes);
};

}  // namespace raytracer
