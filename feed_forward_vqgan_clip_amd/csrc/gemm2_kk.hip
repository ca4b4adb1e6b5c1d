// gemm2_kk.hip — instantiations of the LDS-DMA GEMM for K-major x K-major operands (Linear forward / dgrad, 1x1 convs,
// attention products): lean, complete and per-activation epilogue classes x three tile configurations x two 16-bit formats.
// The largest family, in its own translation unit so that it compiles next to gemm2.hip (conv row tile, dispatch).
#include "gemm2_kernels.h"

int ffvc_gemm2_launch_kk(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg) {
  return launch2_cfg<FFVC_OP_KMAJOR, FFVC_OP_KMAJOR>(d, st, vec_ok, zero, cfg);
}
