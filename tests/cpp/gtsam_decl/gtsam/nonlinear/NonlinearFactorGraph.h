// declaration-only stand-in (see ../gtsam_decl.hpp): type-checking gtsam_adapter.hpp / bench/gtsam_reference.cpp without GTSAM
#include "../../gtsam_decl.hpp"
