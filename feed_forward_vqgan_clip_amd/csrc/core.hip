// core.hip — error plumbing, device info and hardware probes of libffvc_hip.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ffvc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ffvc_last_error(void) { return g_err; }

extern "C" int ffvc_version(void) { return 100; }

extern "C" int ffvc_device_info(int32_t* n_cu, int32_t* clock_khz, int64_t* hbm_bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    ffvc_set_error("hipGetDevice: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    ffvc_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
    return (int)e;
  }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return 0;
}

// out[2 x + 0] = shader-clock counter (s_memtime, ticks at the CURRENT engine clock), out[2 x + 1] = the constant 100 MHz counter
// (s_memrealtime), sampled on XCD x (0..7; the counters are per XCD, so a pair of samples is only comparable on the same XCD):
// 64 one-thread workgroups are dealt round-robin to the XCDs, each writes its XCD's slot.  Two samples bracketing a region give
// its average effective clock per XCD, (d out[2x]) / (d out[2x+1] * 10 ns) — the chip clocks to its power budget under MFMA load
// (profiles/r03_power_ceiling.txt).  out: 16 x uint64, zero-filled first (an XCD that ran no block keeps zeros).
__global__ void clock_sample_kernel(unsigned long long* out) {
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  out[2 * xcc] = __builtin_amdgcn_s_memtime();
  out[2 * xcc + 1] = __builtin_amdgcn_s_memrealtime();
}

extern "C" int ffvc_clock_sample(uint64_t* out, void* stream) {
  FFVC_CHECK_ARG(out, "ffvc_clock_sample: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, 16 * sizeof(uint64_t), st) != hipSuccess) return FFVC_E_BADARG;
  hipLaunchKernelGGL(clock_sample_kernel, dim3(64), dim3(1), 0, st, (unsigned long long*)out);
  FFVC_LAUNCH_CHECK();
  return 0;
}

// Each lane supplies the address of 4 consecutive shorts (lane l -> lds[4l..4l+3], lds[i] = i);
// the dump shows which source element lands in (lane, j).
__global__ void probe_tr16_kernel(int16_t* out) {
  __shared__ __attribute__((aligned(16))) int16_t lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (int16_t)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  s16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}

extern "C" int ffvc_probe_tr16(int16_t* out, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  FFVC_LAUNCH_CHECK();
  return 0;
}
