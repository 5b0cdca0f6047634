/* TEST STUB (tests/cpp/ref_shells): the slice of jbehley/glow's GlBuffer<T> that the reference's callers of the hot
 * path touch (SurfelMapping.cpp:323-331: assign; Frame.h:70-72: the members) -- a host std::vector behind the same
 * member names, so that the reference-signature shells of ref_shells.cpp compile without glow / GL.  Not shipped. */
#ifndef REF_SHELLS_STUB_GLOW_GLBUFFER_H_
#define REF_SHELLS_STUB_GLOW_GLBUFFER_H_
#include <cstddef>
#include <cstdint>
#include <vector>
namespace glow {
enum class BufferTarget { ARRAY_BUFFER, TEXTURE_BUFFER };
enum class BufferUsage { STATIC_DRAW, DYNAMIC_DRAW, STREAM_DRAW };
template <class T>
class GlBuffer {
 public:
  GlBuffer(BufferTarget = BufferTarget::ARRAY_BUFFER, BufferUsage = BufferUsage::DYNAMIC_DRAW) {}
  void assign(const std::vector<T>& data) { host_ = data; }
  void assign(const GlBuffer<T>& other) { host_ = other.host_; }
  void get(std::vector<T>& data) const { data = host_; }
  void reserve(uint32_t n) { host_.reserve(n); }
  size_t size() const { return host_.size(); }
  const std::vector<T>& host() const { return host_; } /* stub only: what glBufferData received */
 private:
  std::vector<T> host_;
};
}  // namespace glow
#endif
