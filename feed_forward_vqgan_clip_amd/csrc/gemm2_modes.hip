// gemm2_modes.hip — instantiations of the LDS-DMA GEMM for the operand-mode pairs with a reduction-major ("TRANS")
// operand: NN (token mixing, attention PV), TN and TT (wgrad when forced onto this path).  Split from gemm2.hip only so
// that the translation units compile in parallel.
#include "gemm2_kernels.h"

int ffvc_gemm2_launch_nn(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg) {
  return launch2_cfg<FFVC_OP_KMAJOR, FFVC_OP_TRANS>(d, st, vec_ok, zero, cfg);
}
int ffvc_gemm2_launch_tn(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg) {
  return launch2_cfg<FFVC_OP_TRANS, FFVC_OP_KMAJOR>(d, st, vec_ok, zero, cfg);
}
int ffvc_gemm2_launch_tt(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg) {
  return launch2_cfg<FFVC_OP_TRANS, FFVC_OP_TRANS>(d, st, vec_ok, zero, cfg);
}
