/* TEST STUB: rv::Point3f as the hot path sees it -- 16 bytes x, y, z, 1 (src/rv/geometry.h:331-345). */
#ifndef REF_SHELLS_STUB_RV_GEOMETRY_H_
#define REF_SHELLS_STUB_RV_GEOMETRY_H_
namespace rv {
struct Point3f {
  Point3f() : x(0.f), y(0.f), z(0.f), w(1.f) {}
  Point3f(float xx, float yy, float zz) : x(xx), y(yy), z(zz), w(1.f) {}
  float x, y, z, w;
};
}  // namespace rv
#endif
