// api_impl64.hip -- the precision-dependent half of the library with RowT = double (Jacobian row tables); Real (normal equations,
// solver) is double in both halves.  See api_impl.inc.
#include "api_common.hpp"

namespace impl64 {
#include "api_decl.inc"
}
namespace impl32 {
#include "api_decl.inc"
}

namespace impl64 {
typedef double Real;
typedef double RowT;
#define IMPL_NS impl64
#include "api_impl.inc"
#undef IMPL_NS
}  // namespace impl64

#ifdef GPS_TRACE_KLIN
extern "C" int gpslam_hip_debug_klin_trace(unsigned long long *out) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gps::g_klin_trace), sizeof(gps::g_klin_trace)) == hipSuccess ? 0 : -2;
}
#endif
