// dpp.hpp -- wave64 / DPP building blocks shared by the solver kernels (kernels.hpp, upper.hip): LDS-only synchronisation,
// lane broadcasts, the fast reciprocal, and the 64-bit DPP row operations of the row-layout eliminations
// (v_mov_b64_dpp / v_fmac_f64_dpp row_newbcast) written out as inline assembly.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>

namespace gps {

// Kernels that run one wave per workgroup (the solver, the landmark solve) exchange data between lanes through LDS.  DS operations of one
// wave execute in order, so a compiler-level ordering point is all that is needed; __syncthreads() would add an
// s_barrier and, worse, drain vmcnt to zero, i.e. wait for the prefetched next-block operands and for the factor
// stores of the current block at every step.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double lane_bcast(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// 1 / p to full precision: hardware reciprocal + two Newton steps (the 15-instruction IEEE division sequence is
// the single most expensive piece of a pivot step; p is a positive pivot, no special values to honour)
__device__ __forceinline__ double fast_rcp(double p) {
  double r = __builtin_amdgcn_rcp(p);
  r = fma(fma(-p, r, 1.0), r, r);
  r = fma(fma(-p, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float fast_rcp(float p) {
  float r = __builtin_amdgcn_rcpf(p);
  r = fmaf(fmaf(-p, r, 1.0f), r, r);
  return r;
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// v_mov_b64 is one of the few 64-bit DPP instructions (row_newbcast only); the compiler splits a broadcast double into
// two v_mov_b32_dpp, so the 64-bit form is written out.  A DPP read of a VGPR needs two wait states after the VALU
// write of that VGPR, which the compiler cannot insert for inline asm: every block starts with s_nop 1 (the block
// itself only reads its sources).
template <int K> __device__ __forceinline__ double row_bcast(double v) {
  double o;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=&v"(o) : "v"(v), "n"(K));
  return o;
}
template <int K> __device__ __forceinline__ void row_bcast12(const double *s, double *d) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b64_dpp %0, %12 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %1, %13 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %2, %14 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %3, %15 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %4, %16 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %5, %17 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %6, %18 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %7, %19 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %8, %20 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %9, %21 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %10, %22 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %11, %23 row_newbcast:%24 row_mask:0xf bank_mask:0xf"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]),
        "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11])
      : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]),
        "v"(s[10]), "v"(s[11]), "n"(K));
}


// v_fmac_f64_dpp D, S0, S1 row_newbcast:K  computes  D += S0[lane K of the 16-lane row] * S1  -- the broadcast of a pivot /
// source row fused INTO the multiply-add.  It is the one double-precision ALU instruction gfx950's assembler accepts with
// a DPP operand besides v_mov_b64 (v_fma_f64 / v_add_f64 / v_mul_f64 are rejected), it issues at the plain v_fma_f64
// rate (scripts/ubench/valu_rates.hip: 4.96 cycles per wave instruction, semantics checked there) and it performs the
// same single rounding as fma(-m, bcast, d) with the negated multiplier passed in -- so every "broadcast the row, then
// multiply-add" pair of the row-layout kernels becomes ONE instruction with bit-identical results.  As with the
// v_mov_b64_dpp blocks, each block opens with s_nop 1 (DPP read of a freshly written VGPR) and never reads through DPP a
// register that an earlier instruction of the same block wrote.
#define GPS_FMAC_ROW "row_mask:0xf bank_mask:0xf\n\t"
// d[q] += bcast_K(s[q]) * m, q = 0..11
template <int K, bool NEG = false> __device__ __forceinline__ void fmac_bcast12(double *d, const double *s, double m) {
  if constexpr (NEG) {      // d[q] -= bcast_K(s[q]) * m: the source modifier of the DPP encoding, the same single rounding
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %12, -%24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %13, -%24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %2, %14, -%24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %15, -%24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %4, %16, -%24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %5, %17, -%24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %6, %18, -%24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %7, %19, -%24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %8, %20, -%24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %9, %21, -%24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %10, %22, -%24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %11, %23, -%24 row_newbcast:%25 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]),
          "+v"(d[10]), "+v"(d[11])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]), "v"(s[10]),
          "v"(s[11]), "v"(m), "n"(K));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %12, %24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %13, %24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %2, %14, %24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %15, %24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %4, %16, %24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %5, %17, %24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %6, %18, %24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %7, %19, %24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %8, %20, %24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %9, %21, %24 row_newbcast:%25 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %10, %22, %24 row_newbcast:%25 " GPS_FMAC_ROW "v_fmac_f64_dpp %11, %23, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]),
          "+v"(d[10]), "+v"(d[11])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]), "v"(s[10]),
          "v"(s[11]), "v"(m), "n"(K));
  }
}
// d[q] += bcast_K(d[q]) * m, q = 0..11 (a Gauss-Jordan row operation: the pivot row is lane K of the same registers)
template <int K> __device__ __forceinline__ void fmac_self12(double *d, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %0, %12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %1, %12 row_newbcast:%13 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %2, %12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %3, %12 row_newbcast:%13 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %4, %4, %12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %5, %5, %12 row_newbcast:%13 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %6, %6, %12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %7, %7, %12 row_newbcast:%13 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %8, %8, %12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %9, %9, %12 row_newbcast:%13 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %10, %10, %12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %11, %11, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]),
        "+v"(d[10]), "+v"(d[11])
      : "v"(m), "n"(K));
}
// the same on four consecutive entries (the columns right of the pivot shrink as the elimination proceeds)
template <int K> __device__ __forceinline__ void fmac_self4(double *d, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%5 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %1, %4 row_newbcast:%5 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %2, %4 row_newbcast:%5 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
      : "v"(m), "n"(K));
}
// two scalars: d0 += bcast_K(s) * m0, d1 += bcast_K(s) * m1 (d0 may be s itself only through fmac_self1)
template <int K, bool NEG = false> __device__ __forceinline__ void fmac_bcast2(double &d0, double &d1, double s, double m0, double m1) {
  if constexpr (NEG) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %2, -%3 row_newbcast:%5 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %2, -%4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(d0), "+v"(d1)
        : "v"(s), "v"(m0), "v"(m1), "n"(K));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %2, %3 row_newbcast:%5 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(d0), "+v"(d1)
        : "v"(s), "v"(m0), "v"(m1), "n"(K));
  }
}
template <int K> __device__ __forceinline__ void fmac_bcast1(double &d0, double s, double m0) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d0) : "v"(s), "v"(m0), "n"(K));
}
template <int K> __device__ __forceinline__ void fmac_self1(double &d0, double m0) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d0) : "v"(m0), "n"(K));
}
// ---- the same operations WITHOUT the leading s_nop (round 6).  The hazard the s_nop covers is "a VALU instruction writes VGPR x, one
// of the next two instructions reads x THROUGH DPP".  In the inner loops of the eliminations the register read through DPP is a panel
// column that was last written a whole pivot step (dozens of instructions) earlier, and the freshly computed operand -- the multiplier
// -- is the plain source: the 2 x 157 s_nop 1 per block step of k_fused_level0 guarded nothing.  These forms are single instructions
// (asm volatile keeps their order), to be used ONLY where the caller can name the last writer of the DPP source; every phase that uses
// them opens with one guarded instruction or an explicit dpp_guard().
__device__ __forceinline__ void dpp_guard() { asm volatile("s_nop 1"); }
template <int K> __device__ __forceinline__ void fmac_self1_nn(double &d0, double m0) {
  asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d0) : "v"(m0), "n"(K));
}
template <int K, bool NEG = false> __device__ __forceinline__ void fmac_bcast1_nn(double &d0, double s, double m0) {
  if constexpr (NEG) asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d0) : "v"(s), "v"(m0), "n"(K));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d0) : "v"(s), "v"(m0), "n"(K));
}
// d[q] += bcast_K(d[q]) * m for q in [LO, HI)
template <int K, int LO, int HI> __device__ __forceinline__ void fmac_self_range_nn(double *d, double m) {
  static_for<LO, HI>([&](auto qq) { fmac_self1_nn<K>(d[decltype(qq)::value], m); });
}
// d[q] (+/-)= bcast_K(s[q]) * m for q in [0, N)
template <int K, int N, bool NEG = false> __device__ __forceinline__ void fmac_bcast_n_nn(double *d, const double *s, double m) {
  static_for<0, N>([&](auto qq) { fmac_bcast1_nn<K, NEG>(d[decltype(qq)::value], s[decltype(qq)::value], m); });
}

// d[k] += s[lane K0 + k] * m, k = 0..N-1 (the gather form below, unguarded)
template <int N, int K0 = 0> __device__ __forceinline__ void fmac_gather_nn(double *d, double s, double m) {
  static_for<0, N>([&](auto kk) { fmac_bcast1_nn<K0 + decltype(kk)::value>(d[decltype(kk)::value], s, m); });
}

// d[k] += s[lane k] * m, k = 0..N-1: the GATHER form (one source register, twelve broadcast lanes) of the assembly wave
template <int N> __device__ __forceinline__ void fmac_gather(double *d, double s, double m);
template <> __device__ __forceinline__ void fmac_gather<12>(double *d, double s, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %12, %13 row_newbcast:0 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %12, %13 row_newbcast:1 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %12, %13 row_newbcast:2 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %12, %13 row_newbcast:3 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %4, %12, %13 row_newbcast:4 " GPS_FMAC_ROW "v_fmac_f64_dpp %5, %12, %13 row_newbcast:5 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %6, %12, %13 row_newbcast:6 " GPS_FMAC_ROW "v_fmac_f64_dpp %7, %12, %13 row_newbcast:7 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %8, %12, %13 row_newbcast:8 " GPS_FMAC_ROW "v_fmac_f64_dpp %9, %12, %13 row_newbcast:9 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %10, %12, %13 row_newbcast:10 " GPS_FMAC_ROW "v_fmac_f64_dpp %11, %12, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]),
        "+v"(d[10]), "+v"(d[11])
      : "v"(s), "v"(m));
}
template <> __device__ __forceinline__ void fmac_gather<6>(double *d, double s, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %6, %7 row_newbcast:0 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %6, %7 row_newbcast:1 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %6, %7 row_newbcast:2 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %6, %7 row_newbcast:3 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %4, %6, %7 row_newbcast:4 " GPS_FMAC_ROW "v_fmac_f64_dpp %5, %6, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])
      : "v"(s), "v"(m));
}

template <> __device__ __forceinline__ void fmac_gather<3>(double *d, double s, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %3, %4 row_newbcast:0 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %3, %4 row_newbcast:1 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %3, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])
      : "v"(s), "v"(m));
}

// three lanes from lane K0 on: d[k] += s[lane K0 + k] * m, k = 0..2 (the block-triangular halves of the SE(3) records)
template <int K0> __device__ __forceinline__ void fmac_gather3_at(double *d, double s, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %3, %4 row_newbcast:%5 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%6 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %3, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])
      : "v"(s), "v"(m), "n"(K0), "n"(K0 + 1), "n"(K0 + 2));
}

// two dot products against six lanes of ONE register: a0 += sum_k s[lane K0 + k] * m0[k], a1 += sum_k s[lane K0 + k] * m1[k]
// (the interpolated measurement rows of k_fused_level0<4>: mu lives in lanes 6..11 of the row's register).  The two sums alternate
// so that consecutive instructions never wait for one another's result.
template <int K0> __device__ __forceinline__ void fmac_dot6x2_at(double &a0, double &a1, double s, const double *m0, const double *m1) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %2, %3 row_newbcast:%15 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %2, %9 row_newbcast:%15 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%16 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %2, %10 row_newbcast:%16 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %0, %2, %5 row_newbcast:%17 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %2, %11 row_newbcast:%17 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %0, %2, %6 row_newbcast:%18 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %2, %12 row_newbcast:%18 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %0, %2, %7 row_newbcast:%19 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %2, %13 row_newbcast:%19 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %0, %2, %8 row_newbcast:%20 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %2, %14 row_newbcast:%20 row_mask:0xf bank_mask:0xf"
      : "+v"(a0), "+v"(a1)
      : "v"(s), "v"(m0[0]), "v"(m0[1]), "v"(m0[2]), "v"(m0[3]), "v"(m0[4]), "v"(m0[5]), "v"(m1[0]), "v"(m1[1]), "v"(m1[2]), "v"(m1[3]), "v"(m1[4]),
        "v"(m1[5]), "n"(K0), "n"(K0 + 1), "n"(K0 + 2), "n"(K0 + 3), "n"(K0 + 4), "n"(K0 + 5));
}
// four lanes of one register broadcast to all lanes of the row: d[k] = s[lane K0 + k]
template <int K0> __device__ __forceinline__ void row_bcast4_at(double s, double *d) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b64_dpp %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %1, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %2, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %3, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
      : "v"(s), "n"(K0), "n"(K0 + 1), "n"(K0 + 2), "n"(K0 + 3));
}

template <int N> __device__ __forceinline__ void lane_gather(double v, double *d);
template <> __device__ __forceinline__ void lane_gather<12>(double v, double *d) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b64_dpp %0, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %1, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %2, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %3, %12 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %4, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %5, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %6, %12 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %7, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %8, %12 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %9, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %10, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %11, %12 row_newbcast:11 row_mask:0xf bank_mask:0xf"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]),
        "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11])
      : "v"(v));
}
template <> __device__ __forceinline__ void lane_gather<6>(double v, double *d) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b64_dpp %0, %6 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %1, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %2, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %3, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %4, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b64_dpp %5, %6 row_newbcast:5 row_mask:0xf bank_mask:0xf"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5])
      : "v"(v));
}

// ---- d[i] += M(m0 + i * S) * m for i < N, where matrix element e lives in lane (e & 15) of every 16-lane row of register
// Mr[e >> 4] (wave-uniform matrices held ONCE per DPP row instead of being re-read from LDS as broadcast operands: k_fs_sweep_syrk).
// The registers Mr come from memory loads, not from VALU writes, but the block still opens with s_nop 1 (a register copy the
// allocator might place in front of the block would be a VALU write of a DPP source).
template <int M0, int S> __device__ __forceinline__ void fmac_mat1(double *d, const double *Mr, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0])
      : "v"(Mr[(M0 + 0 * S) >> 4]), "v"(m), "n"((M0 + 0 * S) & 15));
}
template <int M0, int S> __device__ __forceinline__ void fmac_mat2(double *d, const double *Mr, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%5 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1])
      : "v"(Mr[(M0 + 0 * S) >> 4]), "v"(Mr[(M0 + 1 * S) >> 4]), "v"(m), "n"((M0 + 0 * S) & 15), "n"((M0 + 1 * S) & 15));
}
template <int M0, int S> __device__ __forceinline__ void fmac_mat3(double *d, const double *Mr, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %3, %6 row_newbcast:%7 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%8 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %5, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2])
      : "v"(Mr[(M0 + 0 * S) >> 4]), "v"(Mr[(M0 + 1 * S) >> 4]), "v"(Mr[(M0 + 2 * S) >> 4]), "v"(m), "n"((M0 + 0 * S) & 15), "n"((M0 + 1 * S) & 15), "n"((M0 + 2 * S) & 15));
}
template <int M0, int S> __device__ __forceinline__ void fmac_mat4(double *d, const double *Mr, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%9 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%10 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%11 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %3, %7, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
      : "v"(Mr[(M0 + 0 * S) >> 4]), "v"(Mr[(M0 + 1 * S) >> 4]), "v"(Mr[(M0 + 2 * S) >> 4]), "v"(Mr[(M0 + 3 * S) >> 4]), "v"(m), "n"((M0 + 0 * S) & 15), "n"((M0 + 1 * S) & 15), "n"((M0 + 2 * S) & 15), "n"((M0 + 3 * S) & 15));
}
template <int M0, int S> __device__ __forceinline__ void fmac_mat5(double *d, const double *Mr, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %5, %10 row_newbcast:%11 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %1, %6, %10 row_newbcast:%12 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %7, %10 row_newbcast:%13 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %3, %8, %10 row_newbcast:%14 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %4, %9, %10 row_newbcast:%15 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4])
      : "v"(Mr[(M0 + 0 * S) >> 4]), "v"(Mr[(M0 + 1 * S) >> 4]), "v"(Mr[(M0 + 2 * S) >> 4]), "v"(Mr[(M0 + 3 * S) >> 4]), "v"(Mr[(M0 + 4 * S) >> 4]), "v"(m), "n"((M0 + 0 * S) & 15), "n"((M0 + 1 * S) & 15), "n"((M0 + 2 * S) & 15), "n"((M0 + 3 * S) & 15), "n"((M0 + 4 * S) & 15));
}
template <int M0, int S> __device__ __forceinline__ void fmac_mat6(double *d, const double *Mr, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %6, %12 row_newbcast:%13 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %1, %7, %12 row_newbcast:%14 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %8, %12 row_newbcast:%15 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %3, %9, %12 row_newbcast:%16 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %4, %10, %12 row_newbcast:%17 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %5, %11, %12 row_newbcast:%18 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])
      : "v"(Mr[(M0 + 0 * S) >> 4]), "v"(Mr[(M0 + 1 * S) >> 4]), "v"(Mr[(M0 + 2 * S) >> 4]), "v"(Mr[(M0 + 3 * S) >> 4]), "v"(Mr[(M0 + 4 * S) >> 4]), "v"(Mr[(M0 + 5 * S) >> 4]), "v"(m), "n"((M0 + 0 * S) & 15), "n"((M0 + 1 * S) & 15), "n"((M0 + 2 * S) & 15), "n"((M0 + 3 * S) & 15), "n"((M0 + 4 * S) & 15), "n"((M0 + 5 * S) & 15));
}
template <int N, int M0, int S> __device__ __forceinline__ void fmac_mat(double *d, const double *Mr, double m) {
  if constexpr (N == 1) fmac_mat1<M0, S>(d, Mr, m);
  else if constexpr (N == 2) fmac_mat2<M0, S>(d, Mr, m);
  else if constexpr (N == 3) fmac_mat3<M0, S>(d, Mr, m);
  else if constexpr (N == 4) fmac_mat4<M0, S>(d, Mr, m);
  else if constexpr (N == 5) fmac_mat5<M0, S>(d, Mr, m);
  else fmac_mat6<M0, S>(d, Mr, m);
}


// LDS-only workgroup barrier: the two waves exchange nothing but LDS, so neither the factor stores nor the row loads
// in flight are drained (which __syncthreads() would do)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the same row operations for block sizes 6 and 4 (the upper solver levels of the planar / rotation chains), and
// width-generic front ends:  fmac_self_n<K, N>(d, m): d[q] += bcast_K(d[q]) * m;  fmac_bcast_n<K, N>(d, s, m): d[q] += bcast_K(s[q]) * m
template <int K> __device__ __forceinline__ void fmac_self6(double *d, double m) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %0, %6 row_newbcast:%7 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %1, %6 row_newbcast:%7 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %2, %2, %6 row_newbcast:%7 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %3, %6 row_newbcast:%7 " GPS_FMAC_ROW
      "v_fmac_f64_dpp %4, %4, %6 row_newbcast:%7 " GPS_FMAC_ROW "v_fmac_f64_dpp %5, %5, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])
      : "v"(m), "n"(K));
}
template <int K, bool NEG = false> __device__ __forceinline__ void fmac_bcast6(double *d, const double *s, double m) {
  if constexpr (NEG) {      // d[q] -= bcast_K(s[q]) * m: the source modifier of the DPP encoding, the same single rounding
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %6, -%12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %7, -%12 row_newbcast:%13 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %2, %8, -%12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %9, -%12 row_newbcast:%13 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %4, %10, -%12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %5, %11, -%12 row_newbcast:%13 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(m), "n"(K));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %6, %12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %7, %12 row_newbcast:%13 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %2, %8, %12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %9, %12 row_newbcast:%13 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %4, %10, %12 row_newbcast:%13 " GPS_FMAC_ROW "v_fmac_f64_dpp %5, %11, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(m), "n"(K));
  }
}
template <int K, bool NEG = false> __device__ __forceinline__ void fmac_bcast4(double *d, const double *s, double m) {
  if constexpr (NEG) {      // d[q] -= bcast_K(s[q]) * m: the source modifier of the DPP encoding, the same single rounding
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %4, -%8 row_newbcast:%9 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %5, -%8 row_newbcast:%9 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %2, %6, -%8 row_newbcast:%9 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %7, -%8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(m), "n"(K));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%9 " GPS_FMAC_ROW "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%9 " GPS_FMAC_ROW
        "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%9 " GPS_FMAC_ROW "v_fmac_f64_dpp %3, %7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(m), "n"(K));
  }
}
template <int K, int N> __device__ __forceinline__ void fmac_self_n(double *d, double m) {
  static_assert(N == 4 || N == 6 || N == 12, "block sizes of the chain solver");
  if constexpr (N == 12) fmac_self12<K>(d, m);
  else if constexpr (N == 6) fmac_self6<K>(d, m);
  else fmac_self4<K>(d, m);
}
template <int K, int N, bool NEG = false> __device__ __forceinline__ void fmac_bcast_n(double *d, const double *s, double m) {
  static_assert(N == 4 || N == 6 || N == 12, "block sizes of the chain solver");
  if constexpr (N == 12) fmac_bcast12<K, NEG>(d, s, m);
  else if constexpr (N == 6) fmac_bcast6<K, NEG>(d, s, m);
  else fmac_bcast4<K, NEG>(d, s, m);
}

}  // namespace gps
