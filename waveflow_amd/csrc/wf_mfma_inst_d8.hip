// explicit instantiations of the MFMA kernel (wf_mfma_impl.h): compiled as a separate translation unit
#include "wf_mfma_impl.h"

namespace wf {
namespace mfma {
template int launch_dw<8, 1, 8, 1>(const MfmaDev*, int, int, const float*, int64_t, float*, float*, int32_t*, hipStream_t);

}  // namespace mfma
}  // namespace wf
