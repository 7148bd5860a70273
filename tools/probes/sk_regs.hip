// register / spill check of single kernels of gps_gemm.hip (seconds instead of a minute per edit):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -c -Rpass-analysis=kernel-resource-usage tools/probes/sk_regs.hip -o /tmp/sk_regs.o
#define GPS_GEMM_NO_ENTRY 1
#include "../../sceneverse_amd/csrc/gps_gemm.hip"
namespace gps_gemm {
#ifndef SK_EPI
#define SK_EPI 0
#endif
#ifndef SK_BTR
#define SK_BTR false
#endif
#ifndef SK_RAGGED
#define SK_RAGGED false
#endif
template __global__ void gemm8p_sk_kernel<SK_BTR, SK_EPI, SK_RAGGED>(const Params);
}
