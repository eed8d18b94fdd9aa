// Issue cost (cycles per wave instruction) of the VALU instructions the solver kernels are made of, one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rates.hip -o scripts/ubench/valu_rates && scripts/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE> __global__ void k(double *p, long long *cyc, int iters) {
  double a0 = p[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double b = p[64 + threadIdx.x];
  int i0 = (int)a0, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7, ib = (int)b;
  long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {        // 8 independent v_fma_f64 chains
      REP16(asm volatile("v_fma_f64 %0, %0, %8, %8\n\tv_fma_f64 %1, %1, %8, %8\n\tv_fma_f64 %2, %2, %8, %8\n\tv_fma_f64 %3, %3, %8, %8\n\t"
                         "v_fma_f64 %4, %4, %8, %8\n\tv_fma_f64 %5, %5, %8, %8\n\tv_fma_f64 %6, %6, %8, %8\n\tv_fma_f64 %7, %7, %8, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 1) { // v_mov_b64_dpp row_newbcast
      REP16(asm volatile("v_mov_b64_dpp %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 2) { // v_mov_b32_dpp row_newbcast (8 per line -> 4 doubles)
      REP16(asm volatile("v_mov_b32_dpp %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ib));)
    } else if (MODE == 3) { // v_readlane_b32 into 8 SGPRs
      REP16(asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %0, 4\n\tv_readlane_b32 s22, %0, 5\n\tv_readlane_b32 s23, %0, 6\n\t"
                         "v_readlane_b32 s24, %0, 7\n\tv_readlane_b32 s25, %0, 8\n\tv_readlane_b32 s26, %0, 9\n\tv_readlane_b32 s27, %0, 10"
                         :: "v"(ib) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (MODE == 4) { // one dependent v_fma_f64 chain (latency)
      REP16(asm volatile("v_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\t"
                         "v_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1"
                         : "+v"(a0) : "v"(b));)
    } else if (MODE == 5) { // dpp64 -> dependent fma pairs (what the elimination does)
      REP16(asm volatile("v_mov_b64_dpp %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b64_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fma_f64 %4, %0, %8, %4\n\tv_fma_f64 %5, %1, %8, %5\n\tv_fma_f64 %6, %2, %8, %6\n\tv_fma_f64 %7, %3, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 6) { // v_mul_f64
      REP16(asm volatile("v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %3, %3, %8\n\t"
                         "v_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 8) { // v_fmac_f64_dpp row_newbcast: D += bcast(S0) * S1 -- broadcast fused into the multiply-add
      REP16(asm volatile("v_fmac_f64_dpp %0, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %4, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %6, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 7) { // v_rcp_f64
      REP16(asm volatile("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\t"
                         "v_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  p[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
}

template <int MODE> void run(const char *name, int waves_per_simd) {
  double *p; long long *c;
  hipMalloc(&p, 4096); hipMemset(p, 0, 4096); hipMalloc(&c, 8 * 4096);
  const int iters = 200, blocks = 256 * 4 * waves_per_simd;   // 64-thread blocks: waves_per_simd on every SIMD
  k<MODE><<<blocks, 64>>>(p, c, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<MODE><<<blocks, 64>>>(p, c, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const double n = 200.0 * 16 * 8;
  printf("%-44s waves/SIMD %d: %.2f clock64 ticks per instr per wave, %.3f ns per instr per SIMD (wall)\n", name, waves_per_simd,
         (double)h / n, ms * 1e6 / (n * waves_per_simd));
  hipFree(p); hipFree(c);
}

// semantics check of v_fmac_f64_dpp row_newbcast:k : d[lane] += s0[16 * (lane / 16) + k] * s1[lane]
__global__ void k_sem(double *out) {
  const int lane = threadIdx.x;
  double d = 1000.0 + lane, s0 = 1.0 + lane, s1 = 0.5 * lane;
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s0), "v"(s1));
  out[lane] = d;
}

int main() {
  {
    double *o; hipMalloc(&o, 512); k_sem<<<1, 64>>>(o); double h[64]; hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) { const double want = 1000.0 + l + (1.0 + 16 * (l / 16) + 5) * 0.5 * l; if (h[l] != want) bad++; }
    printf("v_fmac_f64_dpp row_newbcast semantics (d += s0[row lane k] * s1): %s\n", bad ? "MISMATCH" : "ok");
    hipFree(o);
  }
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f64 (8 independent chains)", w);
    run<6>("v_mul_f64", w);
    run<1>("v_mov_b64_dpp row_newbcast", w);
    run<2>("v_mov_b32_dpp row_newbcast", w);
    run<3>("v_readlane_b32", w);
    run<4>("v_fma_f64 dependent chain", w);
    run<5>("4 x dpp64 + 4 dependent fma", w);
    run<7>("v_rcp_f64", w);
    run<8>("v_fmac_f64_dpp row_newbcast", w);
  }
  return 0;
}
