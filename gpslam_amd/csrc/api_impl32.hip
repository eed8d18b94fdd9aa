// api_impl32.hip -- the precision-dependent half of the library with RowT = float (Jacobian row tables); Real (normal equations,
// solver) is double in both halves.  See api_impl.inc.
#include "api_common.hpp"

namespace impl64 {
#include "api_decl.inc"
}
namespace impl32 {
#include "api_decl.inc"
}

namespace impl32 {
typedef double Real;
typedef float RowT;
#define IMPL_NS impl32
#include "api_impl.inc"
#undef IMPL_NS
}  // namespace impl32
