// gemm2_conv.hip — instantiations of the LDS-DMA GEMM for the generic implicit-GEMM 3x3 convolution (im2col addressing
// in the DMA source; the haloed row-tile kernel lives in gemm2.hip).  Separate translation unit for parallel compilation.
#include "gemm2_kernels.h"

int ffvc_gemm2_launch_conv(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg) {
  return launch2_cfg<FFVC_OP_CONV3X3, FFVC_OP_KMAJOR>(d, st, vec_ok, zero, cfg);
}
