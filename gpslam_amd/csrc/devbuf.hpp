// devbuf.hpp -- owning device buffer used by the host side of libgpslam_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

namespace gps {

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, n ? n : 8);
    if (e == hipSuccess) bytes = n ? n : 8;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <typename U> U *as() const { return reinterpret_cast<U *>(p); }
};

// synchronous host -> device copy of a vector (the stream is drained so temporaries may die)
template <typename V> inline hipError_t upload_vec(hipStream_t stream, DevBuf &buf, const std::vector<V> &v) {
  hipError_t e = buf.reserve(v.size() * sizeof(V));
  if (e != hipSuccess || v.empty()) return e;
  if ((e = hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
  return hipStreamSynchronize(stream);
}

}  // namespace gps
