// hbm_rates.hip -- what the HBM of this MI355X delivers to plain streaming kernels (the roof the row-table kernels live under)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/hbm_rates.hip -o /tmp/hbm_rates && /tmp/hbm_rates
// Patterns: read-only, write-only (16-byte stores, whole 64-byte lines per 4 lanes), copy, and the write pattern of the
// GP-prior row store (96-byte segments at a 192-byte pitch, the other half written later).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double V2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_write(V2 *p, size_t n, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = V2{v, v};
}
__global__ void __launch_bounds__(256) k_read(const V2 *p, size_t n, double *out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { V2 x = p[i]; s += x.x + x.y; }
  if (s == 123.456) out[0] = s;
}
// U independent 16-byte loads in flight per lane
template <int U> __global__ void __launch_bounds__(256) k_read_u(const V2 *p, size_t n, double *out) {
  double s = 0.0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    V2 x[U];
#pragma unroll
    for (int u = 0; u < U; u++) x[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) s += x[u].x + x[u].y;
  }
  if (s == 123.456) out[0] = s;
}
template <int U> __global__ void __launch_bounds__(256) k_write_u(V2 *p, size_t n, double v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; u++) p[i + u * stride] = V2{v, v + u};
  }
}
__global__ void __launch_bounds__(256) k_copy(const V2 *a, V2 *b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// half rows: piece i of 6 per 96-byte segment, segments at a 192-byte pitch; pass 0 writes the right halves, pass 1 the left
__global__ void __launch_bounds__(256) k_write_half(V2 *p, size_t nseg, int half, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nseg * 6; i += (size_t)gridDim.x * blockDim.x) {
    const size_t seg = i / 6, piece = i - seg * 6;
    p[seg * 12 + half * 6 + piece] = V2{v, v};
  }
}

int main() {
  const size_t bytes = 480ull << 20, n = bytes / 16;
  V2 *a, *b;
  double *out;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&out, 8));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grids[] = {512, 1024, 2048, 4096, 16384};
  for (int g : grids) {
    float ms[5] = {0, 0, 0, 0, 0}, mu[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 6; rep++) {
      float t;
      CHECK(hipEventRecord(e0)); k_write<<<g, 256>>>(a, n, 1.0 + rep); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[0] += t;
      CHECK(hipEventRecord(e0)); k_read<<<g, 256>>>(a, n, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[1] += t;
      CHECK(hipEventRecord(e0)); k_read_u<4><<<g, 256>>>(a, n, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep) mu[0] += t;
      CHECK(hipEventRecord(e0)); k_read_u<8><<<g, 256>>>(a, n, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep) mu[1] += t;
      CHECK(hipEventRecord(e0)); k_write_u<4><<<g, 256>>>(a, n, 1.5); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep) mu[2] += t;
      CHECK(hipEventRecord(e0)); k_write_u<8><<<g, 256>>>(a, n, 2.5); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep) mu[3] += t;
      CHECK(hipEventRecord(e0)); k_copy<<<g, 256>>>(a, b, n); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[2] += t;
      CHECK(hipEventRecord(e0)); k_write_half<<<g, 256>>>(b, n / 12, 1, 2.0); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[3] += t;
      CHECK(hipEventRecord(e0)); k_write_half<<<g, 256>>>(b, n / 12, 0, 3.0); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[4] += t;
    }
    const double gb = bytes / 1e9;
    std::printf("grid %6d x 256: write %.2f TB/s | read %.2f TB/s | copy %.2f TB/s (read + written bytes) | half-row writes %.2f / %.2f TB/s\n", g,
                gb / (ms[0] / 5) , gb / (ms[1] / 5), 2 * gb / (ms[2] / 5), 0.5 * gb / (ms[3] / 5), 0.5 * gb / (ms[4] / 5));
    std::printf("                   4 / 8 loads in flight per lane: read %.2f / %.2f TB/s;  4 / 8 stores: write %.2f / %.2f TB/s\n", gb / (mu[0] / 5), gb / (mu[1] / 5),
                gb / (mu[2] / 5), gb / (mu[3] / 5));
  }
  return 0;
}
