// explicit instantiations of the scalar kernels (wf_scalar_impl.h): compiled as a separate translation unit
#include "wf_scalar_impl.h"

namespace wf {
namespace scalar {
WF_SCALAR_SHAPE(, 3, 32)
}  // namespace scalar
}  // namespace wf
