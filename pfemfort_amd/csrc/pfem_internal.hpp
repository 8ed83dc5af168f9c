// pfem_internal.hpp -- declarations shared by the translation units of libpfem_amd.so
#pragma once

#include "../../include/pfem_amd.h"
#include "pfem_elem.hpp"

#include <cstdint>
#include <string>
#include <vector>

namespace pfem {

// per-kind geometry of the element families on the hot path
inline int kind_npelem(int kind) { return (kind == PFEM_POISSON_TET || kind == PFEM_ELAST_TET) ? 4 : 3; }
inline int kind_ndof(int kind) { return kind == PFEM_ELAST_TET ? 3 : (kind == PFEM_ELAST_TRIA ? 2 : 1); }
inline int kind_ndim(int kind) { return (kind == PFEM_POISSON_TET || kind == PFEM_ELAST_TET) ? 3 : 2; }
inline bool kind_valid(int kind) { return kind >= PFEM_POISSON_TRIA && kind <= PFEM_ELAST_TRIA; }

void set_last_error(const std::string &msg);

}  // namespace pfem
