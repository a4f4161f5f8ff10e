// explicit instantiations of the scalar kernels (wf_scalar_impl.h): compiled as a separate translation unit
#include "wf_scalar_impl.h"

namespace wf {
namespace scalar {
WF_SCALAR_SHAPE(, 7, 32)
WF_SCALAR_SHAPE(, 8, 32)
}  // namespace scalar
}  // namespace wf
