// explicit instantiations of the MFMA kernel (wf_mfma_impl.h): compiled as a separate translation unit
#include "wf_mfma_impl.h"

namespace wf {
namespace mfma {
template int launch_dw<8, 1, 8, 1>(const MfmaDev*, int, int, const float*, int64_t, float*, float*, int32_t*, hipStream_t);
#ifdef WF_D8_WAVES_ALL   // experiment build: WF_MFMA_WAVES = 12 / 16 select these (wf_kernels_mfma.hip)
template int launch_dw<8, 1, 12, 1>(const MfmaDev*, int, int, const float*, int64_t, float*, float*, int32_t*, hipStream_t);
template int launch_dw<8, 1, 16, 1>(const MfmaDev*, int, int, const float*, int64_t, float*, float*, int32_t*, hipStream_t);
#endif

}  // namespace mfma
}  // namespace wf
