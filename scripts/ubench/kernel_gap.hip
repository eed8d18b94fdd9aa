// kernel_gap.hip -- what sits between two dependent launches on one stream: the idle time before a tiny kernel as a function of
// what the kernel in front of it did (bytes written / read, store policy, grid size).  The Gauss-Newton iteration at 1e5 states
// shows 6-10 us before k_fused_level0, k_multi_forward and k_retract and none between the small launches (profiles/round4_v5).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/kernel_gap.hip -o scripts/ubench/kernel_gap
//   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -o g -- scripts/ubench/kernel_gap ; python scripts/ubench/kernel_gap.py gpurun_out/gap
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double V2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: plain stores, 1: nontemporal, 2: agent scope (sc1), 3: read-only
template <int MODE> __global__ void __launch_bounds__(256) k_big(V2 *p, size_t n, double v, double *out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (MODE == 0) p[i] = V2{v, v};
    else if constexpr (MODE == 1) __builtin_nontemporal_store(V2{v, v}, &p[i]);
    else if constexpr (MODE == 2) { V2 x = {v, v}; asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(&p[i]), "v"(x) : "memory"); }
    else { V2 x = p[i]; s += x.x + x.y; }
  }
  if (MODE == 3 && s == 123.456) out[0] = s;
}
// the kernel behind: one workgroup; DEP: reads what the big kernel wrote last
__global__ void __launch_bounds__(64) k_tiny(const V2 *p, size_t n, double *out) {
  if (threadIdx.x == 0) out[1] = p[n - 1].x + 1.0;
}
// the same with a big grid of idle workgroups (is it the size of the NEXT launch?)
__global__ void __launch_bounds__(128) k_wide(const V2 *p, size_t n, double *out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = p[n - 1].x + 1.0;
}

int main() {
  const size_t maxb = (size_t)1 << 30;
  V2 *p; double *out;
  CHECK(hipMalloc(&p, maxb)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(p, 0, maxb));
  const size_t sizes[] = {(size_t)1 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20, (size_t)512 << 20};
  for (int rep = 0; rep < 6; rep++) {
    for (size_t b : sizes) {
      const size_t n = b / sizeof(V2);
      const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
      k_big<0><<<grid, 256>>>(p, n, 1.0 + rep, out); k_tiny<<<1, 64>>>(p, n, out);
      k_big<1><<<grid, 256>>>(p, n, 2.0 + rep, out); k_tiny<<<1, 64>>>(p, n, out);
      k_big<2><<<grid, 256>>>(p, n, 3.0 + rep, out); k_tiny<<<1, 64>>>(p, n, out);
      k_big<3><<<grid, 256>>>(p, n, 4.0 + rep, out); k_tiny<<<1, 64>>>(p, n, out);
      k_big<0><<<grid, 256>>>(p, n, 5.0 + rep, out); k_wide<<<1000, 128>>>(p, n, out);
      k_tiny<<<1, 64>>>(p, n, out); k_tiny<<<1, 64>>>(p, n, out);      // tiny behind tiny: the floor
    }
  }
  CHECK(hipDeviceSynchronize());
  std::printf("done\n");
  return 0;
}
