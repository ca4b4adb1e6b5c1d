// gemm2_kernels.h — kernel templates and launchers of the LDS-DMA GEMM path; included by gemm2.hip (K-major x
// K-major, conv row tile, dispatch) and gemm2_modes.hip (the other operand-mode pairs) so the two translation
// units compile in parallel.  See gemm2.hip for the design notes.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <type_traits>

#include "gemm_common.h"

extern int g_force_gemm8;   // gemm2.hip; ffvc_set_option("gemm8", 1): every eligible launch takes the 8-phase kernel (tests)

namespace {

constexpr int BK = 64;
constexpr int EPC = 8;                         // bf16 per 16-byte chunk
// An operand tile of ROWS rows is ROWS x 128 B (K-major) == 64 x (2*ROWS) B (reduction-major): ROWS/32 DMA pieces
// per thread per K step.  The W tile is always 128 rows; the X tile 128 or 256 (block tile BM x 128).

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

__device__ __forceinline__ void dma16(const void* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)lds_wave_base, 16, 0, 0);
}

// ---- buffer-resource variants (buffer_load_dwordx4 ... lds): 32-bit byte offsets against a wave-uniform descriptor instead of
// 64-bit per-lane addresses — half the address VALU per piece — and out-of-range rows / K tails / conv halo are simply an
// offset beyond num_records (the hardware writes zeros), so no zero page and no select on pointers.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
// The scalar offset of a buffer instruction is NOT part of the hardware's range check (raw buffers check voffset + inst_offset
// against num_records), so an out-of-range piece keeps the wave-uniform scalar offset: a per-lane select on it (`ok ? so : 0`,
// rounds 1-2) made hipcc wrap every LDS-DMA piece in a readfirstlane "waterfall" loop — the 100-190 cycles per piece that
// tools/cr_timing.py measured in conv_row_kernel's DMA issue (profiles/r03_conv_row_timing.txt).
constexpr uint32_t DMA_OOB = 0xFFFFFFF0u;
constexpr int DMA_NUMREC = 0x7FFFFF00;            // every valid offset must stay below 2 GiB (launcher-checked)
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, DMA_NUMREC, 0x00020000);
}
__device__ __forceinline__ void dma16b(rsrc_t rs, uint32_t voff, unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vp)lds_wave_base, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ void dma16bs(rsrc_t rs, uint32_t voff, uint32_t soff, unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vp)lds_wave_base, 16, voff, soff, 0, 0);
}

// ---- K-major operand: 4 DMA pieces per thread per K step ------------------------------------------------
// piece j of wave w covers LDS chunk positions p = (4j + w) * 64 + lane; line = p >> 4, slot = p & 15,
// source chunk c' = slot ^ (line & 15): row = 2 * line + (c' >> 3), k-chunk = c' & 7.
template <int ROWS, int NW>
struct KMajorDma {
  static constexpr int NP = ROWS / (8 * NW);   // 1-KiB DMA pieces per thread per K step
  // position p = (4j + w) * 64 + lane  ->  line = p >> 4 = 16 j + 4 w + (lane >> 4), so (line & 15) and therefore the
  // source k-chunk are the SAME for every piece j, and the source row advances by exactly 32 per piece: the whole
  // per-thread state is one row pointer, the k-chunk offset and the first row index.
  const uint16_t* row0p;   // pointer to (first row, k = 0)
  int64_t step;            // elements between piece j and j + 1 (32 rows), when the row map is linear
  int r0, rows, kc;
  int kseg, mi;
  int64_t kso, ld, so;
  const uint16_t* base;
  __device__ __forceinline__ void init(const uint16_t* base_, int64_t ld_, int row0, int rows_, int kseg_, int64_t kso_,
                                       int tid, int mi_, int64_t so_) {
    const int lane = tid & 63, w = tid >> 6;
    const int line = 4 * w + (lane >> 4);        // + 4*NW per piece: (line & 15) is piece-independent (4*NW % 16 == 0)
    const int cp = (lane & 15) ^ (line & 15);
    r0 = row0 + 2 * line + (cp >> 3);
    kc = (cp & 7) * EPC;
    rows = rows_;
    kseg = kseg_;
    kso = kso_;
    ld = ld_;
    mi = mi_;
    so = so_;
    base = base_;
    step = (8 * NW) * ld_;
    const int rr = r0 < rows ? r0 : 0;
    row0p = base + (int64_t)rr * ld_;
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int k0, int kend, const uint16_t* zero, int tid) {
    const int w = tid >> 6;
    const int64_t koff = (kseg ? (int64_t)(k0 / kseg) * kso + (k0 % kseg) : (int64_t)k0) + kc;
    const bool kok = k0 + kc + EPC <= kend;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = r0 + (8 * NW) * j;
      const bool ok = kok && r < rows;
      const uint16_t* src;
      if (mi)
        src = base + (int64_t)(r / mi) * so + (int64_t)(r % mi) * ld + koff;
      else
        src = row0p + j * step + koff;
      dma16(ok ? (const void*)src : (const void*)zero, tile + (NW * j + w) * 1024);
    }
  }
  // pieces J0 and J1 only (rows 8*NW*J .. of the tile): the half-tile granularity of the 8-phase kernel.  Plain operands
  // only (no K segments, no split row map: the launcher checks), so the source is one add away from the row pointer.
  template <int J0, int J1>
  __device__ __forceinline__ void issue2(unsigned char* tile, int k0, int kend, const uint16_t* zero, int tid) {
    const int w = tid >> 6;
    const bool kok = k0 + kc + EPC <= kend;
    const uint16_t* p0 = row0p + k0 + kc;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = jj ? J1 : J0;
      const bool ok = kok && (r0 + (8 * NW) * j) < rows;
      dma16(ok ? (const void*)(p0 + j * step) : (const void*)zero, tile + (NW * j + w) * 1024);
    }
  }
};

// ---- implicit im2col (3x3, pad 1, optional fused nearest 2x upsample) ------------------------------------
template <int ROWS, int NW>
struct ConvDma {
  static constexpr int NP = ROWS / (8 * NW);
  const uint16_t* base;
  int pix[NP], oy[NP], ox[NP], kc[NP];
  bool rvalid[NP];
  int H, W, Win, Cin, ups;
  __device__ __forceinline__ void init(const uint16_t* base_, int row0, int rows, int H_, int W_, int Cin_, int ups_,
                                       int tid) {
    const int lane = tid & 63, w = tid >> 6;
    base = base_;
    H = H_;
    W = W_;
    Cin = Cin_;
    ups = ups_;
    Win = W_ >> ups_;
    const int Hin = H_ >> ups_;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = (NW * j + w) * 64 + lane;
      const int line = p >> 4, cp = (p & 15) ^ (line & 15);
      const int r = row0 + 2 * line + (cp >> 3);
      kc[j] = (cp & 7) * EPC;
      rvalid[j] = r < rows;
      const int rr = rvalid[j] ? r : 0;
      const int b = rr / (H * W);
      const int rem = rr - b * (H * W);
      oy[j] = rem / W;
      ox[j] = rem - oy[j] * W;
      pix[j] = b * Hin * Win;
    }
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int k0, int /*kend*/, const uint16_t* zero, int tid) {
    const int w = tid >> 6;
    const int tap = k0 / Cin;
    const int ci0 = k0 - tap * Cin;
    const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int iy = oy[j] + kh - 1, ix = ox[j] + kw - 1;
      const bool ok = rvalid[j] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const int64_t off = ((int64_t)(pix[j] + (iy >> ups) * Win + (ix >> ups))) * Cin + ci0 + kc[j];
      dma16(ok ? (const void*)(base + off) : (const void*)zero, tile + (NW * j + w) * 1024);
    }
  }
  template <int J0, int J1>
  __device__ __forceinline__ void issue2(unsigned char* tile, int k0, int /*kend*/, const uint16_t* zero, int tid) {
    const int w = tid >> 6;
    const int tap = k0 / Cin;
    const int ci0 = k0 - tap * Cin;
    const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = jj ? J1 : J0;
      const int iy = oy[j] + kh - 1, ix = ox[j] + kw - 1;
      const bool ok = rvalid[j] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const int64_t off = ((int64_t)(pix[j] + (iy >> ups) * Win + (ix >> ups))) * Cin + ci0 + kc[j];
      dma16(ok ? (const void*)(base + off) : (const void*)zero, tile + (NW * j + w) * 1024);
    }
  }
};

// ---- 3x3 conv, haloed row tile ----------------------------------------------------------------------------
// When 256 % W == 0 a 256-pixel output tile is R = 256/W whole image rows.  For one kernel row kh and one
// 64-channel block the LDS tile holds those R input rows WITH a one-pixel halo on both sides:
//     tile_row = seg * (W + 2) + (ix + 1),   ix = -1 .. W,   seg = 0 .. R-1        (<= 264 rows of 128 B)
// and serves the three kw taps by fragment reads shifted by kw rows: the activation is fetched from L2 3x per
// (kh, channel block) less often than with the generic im2col loader (9 taps -> 3 row loads).
struct ConvRowDma {
  static constexpr int NP = 9;      // ceil(264 * 8 / 256) 1-KiB pieces per thread
  const uint16_t* base;
  int rowinfo[NP];                  // (seg << 16) | (ix + 1), or -1 when the position is beyond the tile
  int kc;
  int H, W, Win, Hin, Cin, ups, img, oy0, x0;
  // A 256-pixel tile is 256 / Wt whole rows of width Wt = W (W <= 256) or one 256-pixel segment of a wider row
  // (W a multiple of 256: x0 = first column of the segment); each tile row carries one halo pixel on either side.
  __device__ __forceinline__ void init(const uint16_t* base_, int m0, int H_, int W_, int Cin_, int ups_, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    base = base_;
    H = H_;
    W = W_;
    Cin = Cin_;
    ups = ups_;
    Win = W_ >> ups_;
    Hin = H_ >> ups_;
    img = m0 / (H * W);
    const int rem = m0 - img * (H * W);
    oy0 = rem / W;
    const int Wt = W < 256 ? W : 256;
    x0 = W > 256 ? rem - oy0 * W : 0;
    const int line = 4 * w + (lane >> 4);
    const int cp = (lane & 15) ^ (line & 15);
    kc = (cp & 7) * EPC;
    const int r0 = 2 * line + (cp >> 3);
    const int tr = (256 / Wt) * (Wt + 2);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = r0 + 32 * j;
      rowinfo[j] = r < tr ? (((r / (Wt + 2)) << 16) | (r % (Wt + 2))) : -1;
    }
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int kh, int ci0, const uint16_t* zero, int tid) {
    const int w = tid >> 6;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if (rowinfo[j] < 0) continue;                      // exec-masked lanes simply do not write
      const int seg = rowinfo[j] >> 16, ix = x0 + (rowinfo[j] & 0xffff) - 1;
      const int iy = oy0 + seg + kh - 1;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const int64_t off = ((int64_t)((img * Hin + (iy >> ups)) * Win + (ix >> ups))) * Cin + ci0 + kc;
      dma16(ok ? (const void*)(base + off) : (const void*)zero, tile + (4 * j + w) * 1024);
    }
  }
};

// buffer-descriptor variant of ConvRowDma (same tile image): per piece the column part of the address is a per-thread
// constant, the kernel row contributes one shift-multiply-add, the channel block rides in the scalar offset
struct ConvRowDmaB {
  static constexpr int NP = 9;
  rsrc_t rs;
  int colpart[NP];                  // ((img*Hin)*Win + (ix >> ups)) * Cin + kc, or -1: position beyond the tile, -2: column halo
  int seg[NP];
  int H, ups, oy0, wincin, w;
  __device__ __forceinline__ void init(const uint16_t* base, int m0, int H_, int W_, int Cin_, int ups_, int tid) {
    const int lane = tid & 63;
    w = __builtin_amdgcn_readfirstlane(tid >> 6);
    rs = make_rsrc(base);
    H = H_;
    ups = ups_;
    const int Win = W_ >> ups_, Hin = H_ >> ups_;
    wincin = Win * Cin_;
    const int img = m0 / (H_ * W_);
    const int rem = m0 - img * (H_ * W_);
    oy0 = rem / W_;
    const int Wt = W_ < 256 ? W_ : 256;
    const int x0 = W_ > 256 ? rem - oy0 * W_ : 0;
    const int line = 4 * w + (lane >> 4);
    const int cp = (lane & 15) ^ (line & 15);
    const int kc = (cp & 7) * EPC;
    const int r0 = 2 * line + (cp >> 3);
    const int tr = (256 / Wt) * (Wt + 2);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = r0 + 32 * j;
      if (r < tr) {
        seg[j] = r / (Wt + 2);
        const int ix = x0 + (r % (Wt + 2)) - 1;
        colpart[j] = ((unsigned)ix < (unsigned)W_) ? (img * Hin * Win + (ix >> ups_)) * Cin_ + kc : -2;
      } else {
        seg[j] = 0;
        colpart[j] = -1;
      }
    }
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int kh, int ci0, const uint16_t*, int) {
    const uint32_t so = (uint32_t)ci0 * 2u;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if (colpart[j] == -1) continue;                     // exec-masked lanes simply do not write
      const int iy = oy0 + seg[j] + kh - 1;
      const bool ok = colpart[j] >= 0 && (unsigned)iy < (unsigned)H;
      const uint32_t off = (uint32_t)(colpart[j] + (iy >> ups) * wincin) * 2u;
      dma16bs(rs, ok ? off : DMA_OOB, so, tile + (4 * j + w) * 1024);
    }
  }
  // one piece (the pipelined kernel spreads a tile's nine pieces over two K steps)
  __device__ __forceinline__ void issue1(unsigned char* tile, int j, int kh, int ci0) {
    if (4 * j + w >= 33) return;        // wave-uniform: this piece lies beyond the 33-KiB tile (rows past the halo'd rows of a
                                        // partly used last piece get zeros instead of an exec-masked skip: no divergence here)
    const int iy = oy0 + seg[j] + kh - 1;
    const bool ok = colpart[j] >= 0 && (unsigned)iy < (unsigned)H;
    const uint32_t off = (uint32_t)(colpart[j] + (iy >> ups) * wincin) * 2u;
    dma16bs(rs, ok ? off : DMA_OOB, (uint32_t)ci0 * 2u, tile + (4 * j + w) * 1024);
  }
};

// ---- reduction-major operand: tile [64 k][128 cols]; position p: krow = p >> 4, slot = p & 15,
// source column chunk = slot ^ ((krow & 3) << 2) ------------------------------------------------------------
// Load bound of a reduction-major operand ([k][cols], row stride ld): the DMA moves whole 8-element chunks, so an extent that is not
// a multiple of 8 is read up to the next multiple when the row stride has room for it (the padded qkv / w_out operands of the VitGAN
// blocks: 3060 of 3064, 1020 of 1024).  The extra columns only reach accumulator rows / columns beyond M / N, which no epilogue stores.
__device__ __forceinline__ int tr_cols(int cols, int64_t ld) {
#ifdef FFVC_NO_TR_COLS      // debugging build: the loaders' bound is the extent itself
  return cols;
#endif
  const int up = (cols + 7) & ~7;
  return (int64_t)up <= ld ? up : cols;
}

template <int ROWS, int NW>
struct TransDma {
  static constexpr int NP = ROWS / (8 * NW);
  static constexpr int CPR = ROWS / 8;      // 16-byte chunks per k-row
  const uint16_t* colp[NP];
  int krow[NP];
  bool cvalid[NP];
  int64_t ld;
  __device__ __forceinline__ void init(const uint16_t* base, int64_t ld_, int col0, int cols, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    ld = ld_;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = (NW * j + w) * 64 + lane;
      krow[j] = p / CPR;
      const int c = col0 + (((p % CPR) ^ ((krow[j] & 3) << 2)) * EPC);
      cvalid[j] = c + EPC <= cols;
      colp[j] = base + (cvalid[j] ? c : 0);
    }
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int k0, int kend, const uint16_t* zero, int tid) {
    const int w = tid >> 6;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int k = k0 + krow[j];
      const bool ok = cvalid[j] && k < kend;
      dma16(ok ? (const void*)(colp[j] + (int64_t)k * ld) : (const void*)zero, tile + (NW * j + w) * 1024);
    }
  }
};

template <int ROWS, int NW>
struct KMajorDmaB {
  static constexpr int NP = ROWS / (8 * NW);
  rsrc_t rs;
  uint32_t voff[NP];        // byte offset of this thread's chunk in piece j at k = 0; DMA_OOB for rows beyond the operand
  int kc, w, kseg;
  int64_t kso;
  // Same arguments as KMajorDma::init.  The descriptor starts at the tile's first row (plain row map) or at the operand
  // (split row map), every offset is relative to that.
  __device__ __forceinline__ void init(const uint16_t* base, int64_t ld, int row0, int rows, int kseg_, int64_t kso_, int tid,
                                       int mi, int64_t so) {
    const int lane = tid & 63;
    w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int line = 4 * w + (lane >> 4);
    const int cp = (lane & 15) ^ (line & 15);
    const int r0 = row0 + 2 * line + (cp >> 3);
    kc = (cp & 7) * EPC;
    kseg = kseg_;
    kso = kso_;
    const int64_t origin = mi ? 0 : (int64_t)row0 * ld;
    rs = make_rsrc(base + origin);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = r0 + (8 * NW) * j;
      const int64_t ro = mi ? (int64_t)(r / mi) * so + (int64_t)(r % mi) * ld : (int64_t)r * ld;
      voff[j] = r < rows ? (uint32_t)((ro - origin + kc) * 2) : DMA_OOB;
    }
  }
  // gemm8's shorthand: plain operand, base already at the tile's first row
  __device__ __forceinline__ void init(const uint16_t* base, int64_t ld, int rows_, int tid) {
    init(base, ld, 0, rows_, 0, 0, tid, 0, 0);
  }
  // k0 is wave-uniform: it travels in the instruction's scalar offset, so an interior K tile costs no VALU at all
  __device__ __forceinline__ void issue1(unsigned char* tile, int j, int k0, int kend) {
    const uint32_t so = (uint32_t)((kseg ? (int64_t)(k0 / kseg) * kso + (k0 % kseg) : (int64_t)k0) * 2);
    if (k0 + BK <= kend) {
      dma16bs(rs, voff[j], so, tile + (NW * j + w) * 1024);
    } else {                 // K tail: chunks beyond kend read as zero
      const bool ok = k0 + kc + EPC <= kend;
      dma16bs(rs, ok ? voff[j] : DMA_OOB, so, tile + (NW * j + w) * 1024);
    }
  }
  template <int J0, int J1>
  __device__ __forceinline__ void issue2(unsigned char* tile, int k0, int kend) {
    issue1(tile, J0, k0, kend);
    issue1(tile, J1, k0, kend);
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int k0, int kend) {
    const uint32_t so = (uint32_t)((kseg ? (int64_t)(k0 / kseg) * kso + (k0 % kseg) : (int64_t)k0) * 2);
    if (k0 + BK <= kend) {
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16bs(rs, voff[j], so, tile + (NW * j + w) * 1024);
    } else {
      const bool ok = k0 + kc + EPC <= kend;
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16bs(rs, ok ? voff[j] : DMA_OOB, so, tile + (NW * j + w) * 1024);
    }
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int k0, int kend, const uint16_t*, int) { issue(tile, k0, kend); }
};

// reduction-major operand [k][cols] through a buffer descriptor: the k index of a K tile is the scalar offset
template <int ROWS, int NW>
struct TransDmaB {
  static constexpr int NP = ROWS / (8 * NW);
  static constexpr int CPR = ROWS / 8;
  rsrc_t rs;
  uint32_t voff[NP];
  int krow[NP];
  int w;
  int64_t ld;
  __device__ __forceinline__ void init(const uint16_t* base, int64_t ld_, int col0, int cols, int tid) {
    const int lane = tid & 63;
    w = __builtin_amdgcn_readfirstlane(tid >> 6);
    ld = ld_;
    rs = make_rsrc(base + col0);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = (NW * j + w) * 64 + lane;
      krow[j] = p / CPR;
      const int c = ((p % CPR) ^ ((krow[j] & 3) << 2)) * EPC;
      voff[j] = (col0 + c + EPC <= cols) ? (uint32_t)(((int64_t)krow[j] * ld_ + c) * 2) : DMA_OOB;
    }
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int k0, int kend, const uint16_t*, int) {
    const uint32_t so = (uint32_t)((int64_t)k0 * ld * 2);
    if (k0 + BK <= kend) {
#pragma unroll
      for (int j = 0; j < NP; ++j) dma16bs(rs, voff[j], so, tile + (NW * j + w) * 1024);
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const bool ok = k0 + krow[j] < kend;
        dma16bs(rs, ok ? voff[j] : DMA_OOB, so, tile + (NW * j + w) * 1024);
      }
    }
  }
};

template <int ROWS, int NW>
struct ConvDmaB {
  static constexpr int NP = ROWS / (8 * NW);
  rsrc_t rs;
  int pixo[NP], oy[NP], ox[NP];   // (image base + channel chunk) in elements, output coordinates (-4 marks a row beyond M)
  int H, W, Win, Cin, ups, w;
  int tap_k0, kh, kw, ci0;        // tap decomposition of the K tile last seen (wave-uniform, recomputed when k0 changes)
  __device__ __forceinline__ void init(const uint16_t* base, int row0, int rows, int H_, int W_, int Cin_, int ups_, int tid) {
    const int lane = tid & 63;
    w = __builtin_amdgcn_readfirstlane(tid >> 6);
    rs = make_rsrc(base);
    H = H_;
    W = W_;
    Cin = Cin_;
    ups = ups_;
    Win = W_ >> ups_;
    const int Hin = H_ >> ups_;
    tap_k0 = -1;
    kh = kw = ci0 = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = (NW * j + w) * 64 + lane;
      const int line = p >> 4, cp = (p & 15) ^ (line & 15);
      const int r = row0 + 2 * line + (cp >> 3);
      const bool valid = r < rows;
      const int rr = valid ? r : 0;
      const int b = rr / (H * W);
      const int rem = rr - b * (H * W);
      oy[j] = valid ? rem / W : -4;
      ox[j] = rem - (rem / W) * W;
      pixo[j] = b * Hin * Win * Cin + (cp & 7) * EPC;
    }
  }
  __device__ __forceinline__ void issue1(unsigned char* tile, int j, int k0, int /*kend*/) {
    if (k0 != tap_k0) {           // scalar: once per K tile, not once per piece
      tap_k0 = k0;
      const int tap = k0 / Cin;
      ci0 = k0 - tap * Cin;
      kh = tap / 3;
      kw = tap - 3 * kh;
    }
    const int iy = oy[j] + kh - 1, ix = ox[j] + kw - 1;
    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const uint32_t off = (uint32_t)(pixo[j] + ((iy >> ups) * Win + (ix >> ups)) * Cin) * 2u;
    dma16bs(rs, ok ? off : DMA_OOB, (uint32_t)ci0 * 2u, tile + (NW * j + w) * 1024);
  }
  template <int J0, int J1>
  __device__ __forceinline__ void issue2(unsigned char* tile, int k0, int kend) {
    issue1(tile, J0, k0, kend);
    issue1(tile, J1, k0, kend);
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int k0, int kend) {
#pragma unroll
    for (int j = 0; j < NP; ++j) issue1(tile, j, k0, kend);
  }
  __device__ __forceinline__ void issue(unsigned char* tile, int k0, int kend, const uint16_t*, int) { issue(tile, k0, kend); }
};

// ---- fragment reads (one 16-byte chunk = 8 bf16, k = 16*sub + 8*h + e) -----------------------------------
__device__ __forceinline__ u32x4_t frag_kmajor(const unsigned char* tile, int row, int sub, int lane) {
  const int line = row >> 1;
  const int cp = (((row & 1) << 3) | (2 * sub + (lane >> 5))) ^ (line & 15);
  return *(const u32x4_t*)(tile + line * 256 + cp * 16);
}
// 16x16x32 fragment of the same K-major image: row = block base + (lane & 15), 16-byte chunk 4 * sub32 + (lane >> 4) of the
// 64-deep K step.  The four 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) hit 16 distinct slots: conflict-free.
__device__ __forceinline__ u32x4_t frag16_kmajor(const unsigned char* tile, int row, int sub32, int lane) {
  const int line = row >> 1;
  const int cp = (((row & 1) << 3) | (4 * sub32 + (lane >> 4))) ^ (line & 15);
  return *(const u32x4_t*)(tile + line * 256 + cp * 16);
}
template <int ROWS>
__device__ __forceinline__ u32x4_t frag_trans(const unsigned char* tile, int row, int sub, int lane) {
  constexpr int RS = ROWS * 2;   // bytes per k-row
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  const int c = lane & 15;
  const int i = (row - (lane & 31)) + 16 * ((lane >> 4) & 1) + (c & 3) * 4;      // column of this lane's 4 elements
  const int k = 16 * sub + 8 * (lane >> 5) + (c >> 2);                          // k & 3 == c >> 2 for both reads
  const int slot = (i >> 3) ^ ((k & 3) << 2);
  const unsigned char* a0 = tile + k * RS + slot * 16 + (i & 7) * 2;
  union {
    s16x4_t hh[2];
    u32x4_t v;
  } u;
  u.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
  u.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * RS));
  return u.v;
}

// L = 16-bit storage format of the operands (uint16_t = bf16, f16_t = IEEE half): tiles move as raw 16-bit words, only
// the MFMA opcode and the epilogue conversions depend on it.

// Tile configurations (BM x BN, waves as 2(M) x BN/64(N), wave tile (BM/2) x 64):
//   128x128: 4 waves x 64x64,  2-stage ring (64 KiB),  2 workgroups / CU            -- small / ragged grids
//   256x128: 4 waves x 128x64, ONE stage (48 KiB), <=256 regs, 2 workgroups / CU alternate load / MFMA
//   256x256: 8 waves x 128x64, 2-stage ring (128 KiB), <=256 regs, 1 workgroup / CU = 2 waves / SIMD; 128 FLOP per
//            L2 byte, the only shape that is not capped by the ~53 B/clk a CU can pull from L2.
#ifdef FFVC_NO_FRAG_PREFETCH_256
constexpr bool FRAG_PREFETCH_256 = false;
#else
constexpr bool FRAG_PREFETCH_256 = true;   // 256-row tiles also fetch fragments one sub-step ahead (two register sets)
#endif
#ifndef FFVC_MFMA16
#define FFVC_MFMA16 1       // K-major x K-major / implicit-im2col kernels on v_mfma_f32_16x16x32 (0: 32x32x16 as in round 2)
#endif
#ifndef FFVC_MFMA16_CONVROW
#define FFVC_MFMA16_CONVROW FFVC_MFMA16   // the haloed-row convolution kernel (gemm2.hip) separately, for A/B builds
#endif
#ifndef FFVC_EXP_MODE
#define FFVC_EXP_MODE 0     // timing experiments only (wrong results): 1 = no DMA after the first stage, 2 = no vmcnt wait / barrier
#endif
#ifdef FFVC_NO_DMA_SPREAD
constexpr bool DMA_SPREAD = false;
#else
constexpr bool DMA_SPREAD = true;
#endif
#ifdef FFVC_EXP_SKIPFRAG
constexpr bool EXP_SKIP = true;    // timing experiment only (wrong results): emulate the LDS traffic of 128x128 wave tiles
#else
constexpr bool EXP_SKIP = false;
#endif
// BUF: K-major / implicit-im2col operands go through buffer-descriptor DMA (buffer_load ... lds, scalar K offset) instead of
// global_load_lds with 64-bit per-lane addresses; plain operands only (no K segments, no split row map, offsets < 2 GiB).
// EPI: epilogue class of the instantiation (gemm_common.h): the launcher picks the lean kernel for launches that use neither
// an activation nor the column / GroupNorm sums.
template <typename L, int XMODE, int WMODE, int BM, int BN, bool BUF = false, int EPI = ffvc_gemm_detail::EPI_ALL>
__global__ __launch_bounds__(64 * 2 * (BN / 64), (BM == 256 ? 2 : 1)) void gemm2_kernel(
    const ffvc_gemm_desc p, int tiles_n, int n_tiles, int ksplit_len, int vec_ok, const uint16_t* zero, int gm) {
  constexpr int MT = BM / 64;                        // 32-row MFMA tiles per wave along M (wave tile (32*MT) x 64)
  constexpr int NW = 2 * (BN / 64);                  // waves per workgroup
  {
    // Epilogue desynchronisation (FFVC_STAGGER_TICKS, 100 MHz ticks in gm's upper bits): equal tiles make every CU reach its
    // store burst at the same moment, an HBM-write-bound phase with idle matrix pipes.  Half of the FIRST round's workgroups
    // start late by about half a tile time, so afterwards the two halves of the chip alternate between K loop and epilogue.
    const int ticks = gm >> 8;
    gm &= 255;
    // workgroups are dealt round-robin to the 8 XCDs and then to an XCD's 32 CUs: with two workgroups per CU the pair on
    // a CU is (k, k + 32) of that XCD -> delay the second slot; with one per CU every other CU
    const int k = blockIdx.x >> 3;
    const bool late = (BM == 256 && BN == 256) ? (k < 32 && (k & 1)) : (k >= 32 && k < 64);
    if (ticks > 0 && blockIdx.y == 0 && blockIdx.z == 0 && late) {
      const uint64_t t0 = wall_clock64();
      while ((int64_t)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    }
  }
  constexpr int XTILE = BM * 128, WTILE = BN * 128;  // bytes
  constexpr int STAGE = XTILE + WTILE;
  constexpr bool RING = !(BM == 256 && BN == 128);   // 2-stage ring except for the single-stage 256x128 variant
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  const int l31 = lane & 31;

  // in-kernel split-K (see the combine in front of the epilogue): uniform mode = blockIdx.z is the K slice of every tile;
  // tail mode (p.sk_slices > 1, 1-D grid) = work items below p.sk_full are whole tiles, the rest are K slices of the tiles of
  // the last, partly filled round
  constexpr bool SKFIX = (FFVC_MFMA16 && XMODE == FFVC_OP_KMAJOR && WMODE == FFVC_OP_KMAJOR && ((BM <= 128 && BN == 128) || (BM == 256 && BN == 256))) ||
                         (XMODE == FFVC_OP_TRANS && WMODE == FFVC_OP_TRANS && BM == 256 && BN == 256);        // + weight gradients
  int wl = blockIdx.x, sk_slice = blockIdx.z, sk_n = gridDim.z;
  if constexpr (SKFIX) {
    if (p.sk_slices > 1) {
      sk_slice = 0;
      sk_n = 1;
      if (wl >= p.sk_full) {
        const int r = wl - p.sk_full, t = r / p.sk_slices;
        sk_slice = r - t * p.sk_slices;
        sk_n = p.sk_slices;
        wl = p.sk_full + t;
      }
    }
  }
  int tile;
  {
    const int bid = wl;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // Tiles are walked in groups of `gm` tile-rows, column-major inside a group: the ~32 workgroups that run together on
  // one XCD then cover a compact gm x (32/gm) block of C and share gm + 32/gm operand panels through that XCD's L2
  // instead of 1 + 32 (row-major order), which is what the fabric behind the L2s has to deliver.
  int tm, tn;
  if (gm > 1) {
    const int width = gm * tiles_n;
    const int grp = tile / width, rem = tile - grp * width;
    const int first = grp * gm;
    const int gsz = min(n_tiles / tiles_n - first, gm);
    tn = rem / gsz;
    tm = first + (rem - tn * gsz);
  } else {
    tm = tile / tiles_n;
    tn = tile - tm * tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const int zo = z / p.batch_inner, zi = z - zo * p.batch_inner;
  int k_begin = blockIdx.z * ksplit_len;
  int k_end = min(p.K, k_begin + ksplit_len);
  if constexpr (SKFIX) {
    if (p.sk_slices > 1 && sk_n > 1) {
      const int ks = (p.K + BK - 1) / BK, len = (ks + sk_n - 1) / sk_n;
      k_begin = sk_slice * len * BK;
      k_end = min(p.K, k_begin + len * BK);          // may be empty: the slice then parks a zero tile
    }
  }
  const uint16_t* xb = (const uint16_t*)p.x + zo * p.xbo + zi * p.xbi;
  const uint16_t* wb = (const uint16_t*)p.w + zo * p.wbo + zi * p.wbi;
  if constexpr (XMODE == FFVC_OP_TRANS && WMODE == FFVC_OP_TRANS && BM == 256 && BN == 256) {
    // grouped weight gradients (ffvc.h grp_*): every batch entry is another layer's operand pair, taken from an offset table in
    // the kernel arguments (wave-uniform index -> scalar loads); y keeps the constant stride ybo
    if (p.grp_n > 0) {
      xb = (const uint16_t*)p.x + p.grp_xoff[z & 7];
      wb = (const uint16_t*)p.w + p.grp_woff[z & 7];
    }
  }

  using XDma = typename std::conditional<
      XMODE == FFVC_OP_CONV3X3, typename std::conditional<BUF, ConvDmaB<BM, NW>, ConvDma<BM, NW>>::type,
      typename std::conditional<XMODE == FFVC_OP_TRANS, typename std::conditional<BUF, TransDmaB<BM, NW>, TransDma<BM, NW>>::type,
                                typename std::conditional<BUF, KMajorDmaB<BM, NW>, KMajorDma<BM, NW>>::type>::type>::type;
  using WDma = typename std::conditional<WMODE == FFVC_OP_TRANS, typename std::conditional<BUF, TransDmaB<BN, NW>, TransDma<BN, NW>>::type,
                                         typename std::conditional<BUF, KMajorDmaB<BN, NW>, KMajorDma<BN, NW>>::type>::type;
  XDma sx;
  WDma sw;
  if constexpr (XMODE == FFVC_OP_CONV3X3)
    sx.init(xb, m0, p.M, p.conv_H, p.conv_W, p.conv_Cin, (p.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0, tid);
  else if constexpr (XMODE == FFVC_OP_TRANS)
    sx.init(xb, p.ldx, m0, tr_cols(p.M, p.ldx), tid);
  else
    sx.init(xb, p.ldx, m0, p.M, p.kseg, p.xkso, tid, p.x_mi, p.x_so);
  if constexpr (WMODE == FFVC_OP_TRANS)
    sw.init(wb, p.ldw, n0, tr_cols(p.N, p.ldw), tid);
  else
    sw.init(wb, p.ldw, n0, p.N, p.kseg, p.wkso, tid, 0, 0);

  // MFMA shape: 16x16x32 for K-major / implicit-im2col operands (fragment = one 16-byte read per 16 rows x 32 k), 32x32x16
  // where an operand is reduction-major (its fragments come through ds_read_b64_tr_b16 in the 32x32 arrangement)
  constexpr bool M16 = FFVC_MFMA16 && XMODE != FFVC_OP_TRANS && WMODE != FFVC_OP_TRANS;
  f32x16_t acc[M16 ? 1 : 2][M16 ? 1 : MT];
  f32x4_t acc16[M16 ? 4 : 1][M16 ? 2 * MT : 1];
  if constexpr (M16) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2 * MT; ++b) acc16[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
  }

  const int nk = (k_end - k_begin + BK - 1) / BK;
  // `between(sub)` runs after the MFMAs of sub-step `sub` have been issued: the ring loop uses it to spread the next
  // stage's DMA issue over the first sub-steps, so the matrix pipe already has work queued while a wave is busy issuing
  // loads (issued in one burst right after the barrier, both waves of a SIMD leave the pipe idle for that long).
  auto compute = [&](const unsigned char* sX, const unsigned char* sW, auto&& between) {
    if constexpr (M16) {
      // two 32-deep sub-steps of 4 x 2MT MFMAs; the second one's fragments are fetched under the first one's MFMAs;
      // between(0) / between(1) (the next stage's DMA issue) fall after the first and second quarter of the MFMAs
      const int l15 = lane & 15;
      // two fragment sets (the second sub-step's under the first one's MFMAs) where the registers allow it: the implicit-im2col
      // loader keeps per-piece pixel coordinates, its 256-row tile gets ONE set (the partner wave covers the LDS latency)
      constexpr bool PF = !(XMODE == FFVC_OP_CONV3X3 && BM == 256);
      u32x4_t fa[PF ? 2 : 1][4], fb[PF ? 2 : 1][2 * MT];
      auto fetch = [&](int sub, u32x4_t (&a)[4], u32x4_t (&b)[2 * MT]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = frag16_kmajor(sW, wn * 64 + t * 16 + l15, sub, lane);
#pragma unroll
        for (int t = 0; t < 2 * MT; ++t) b[t] = frag16_kmajor(sX, wm * (32 * MT) + t * 16 + l15, sub, lane);
      };
      fetch(0, fa[0], fb[0]);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if (PF && sub == 0) fetch(1, fa[PF ? 1 : 0], fb[PF ? 1 : 0]);
        if (!PF && sub == 1) fetch(1, fa[0], fb[0]);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
          for (int b = 0; b < 2 * MT; ++b) mma16_lo<L>(acc16[a][b], fa[PF ? sub : 0][a], fb[PF ? sub : 0][b]);
          if (a == 1) between(2 * sub);
          if (a == 3) between(2 * sub + 1);
        }
      }
    } else if constexpr (BM == 256 && !FRAG_PREFETCH_256) {
      // one fragment set (register budget): the partner wave on the SIMD covers the LDS latency
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        u32x4_t fa[2], fb[MT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int rw = wn * 64 + t * 32 + l31;
          fa[t] = (WMODE == FFVC_OP_TRANS) ? frag_trans<BN>(sW, rw, sub, lane) : frag_kmajor(sW, rw, sub, lane);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int rx = wm * (32 * MT) + t * 32 + l31;
          if (EXP_SKIP && t >= 2) { fb[t] = fb[t - 2]; continue; }
          fb[t] = (XMODE == FFVC_OP_TRANS) ? frag_trans<BM>(sX, rx, sub, lane) : frag_kmajor(sX, rx, sub, lane);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < MT; ++b) mma_lo<L>(acc[a][b], fa[a], fb[b]);
        between(sub);
      }
    } else {
      // fragments are fetched one sub-step ahead of the MFMAs that consume them (two register sets)
      u32x4_t fa[2][2], fb[2][MT];
      auto fetch = [&](int sub, u32x4_t (&a)[2], u32x4_t (&b)[MT]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int rw = wn * 64 + t * 32 + l31;
          a[t] = (WMODE == FFVC_OP_TRANS) ? frag_trans<BN>(sW, rw, sub, lane) : frag_kmajor(sW, rw, sub, lane);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int rx = wm * (32 * MT) + t * 32 + l31;
          b[t] = (XMODE == FFVC_OP_TRANS) ? frag_trans<BM>(sX, rx, sub, lane) : frag_kmajor(sX, rx, sub, lane);
        }
      };
      fetch(0, fa[0], fb[0]);
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        if (sub < 3) fetch(sub + 1, fa[(sub + 1) & 1], fb[(sub + 1) & 1]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < MT; ++b) mma_lo<L>(acc[a][b], fa[sub & 1][a], fb[sub & 1][b]);
        between(sub);
      }
    }
  };
  if constexpr (!RING) {
    for (int kt = 0; kt < nk; ++kt) {
      sx.issue(smem, k_begin + kt * BK, k_end, zero, tid);
      sw.issue(smem + XTILE, k_begin + kt * BK, k_end, zero, tid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(smem, smem + XTILE, [](int) {});
      __syncthreads();
    }
  } else {
    if (nk > 0) {
      sx.issue(smem, k_begin, k_end, zero, tid);
      sw.issue(smem + XTILE, k_begin, k_end, zero, tid);
    }
    for (int kt = 0; kt < nk; ++kt) {
      if (FFVC_EXP_MODE != 2 || kt == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      unsigned char* cur = smem + (kt & 1) * STAGE;
      unsigned char* nxt = smem + ((kt + 1) & 1) * STAGE;
      const bool more = (kt + 1 < nk) && FFVC_EXP_MODE != 1;
      const int kn = k_begin + (kt + 1) * BK;
      if (DMA_SPREAD) {
#ifdef FFVC_DMA_STAGGER
        // experiment: the two waves that share a SIMD (w and w + NW/2) issue their pieces in different sub-steps, so one
        // of them always has MFMAs queued while the other is stuck in LDS-DMA issue
        const int s0 = (NW == 8 && wid >= 4) ? 2 : 0;
        compute(cur, cur + XTILE, [&](int sub) {
          if (more && sub == s0) sx.issue(nxt, kn, k_end, zero, tid);
          if (more && sub == s0 + 1) sw.issue(nxt + XTILE, kn, k_end, zero, tid);
        });
#else
        compute(cur, cur + XTILE, [&](int sub) {
          if (more && sub == 0) sx.issue(nxt, kn, k_end, zero, tid);
          if (more && sub == 1) sw.issue(nxt + XTILE, kn, k_end, zero, tid);
        });
#endif
      } else {
        if (more) {
          sx.issue(nxt, kn, k_end, zero, tid);
          sw.issue(nxt + XTILE, kn, k_end, zero, tid);
        }
        compute(cur, cur + XTILE, [](int) {});
      }
    }
  }
  // In-kernel split-K (under-filled grids, launch2 decides): every K slice parks its fp32 accumulators in accumulator order
  // (16 bytes per lane, fully coalesced), the LAST workgroup to arrive on the tile sums all slices in slice order (the result
  // does not depend on which one that was) and carries on into the ordinary epilogue; the others are done.
  // The slices of a tile may run on different XCDs, whose L2s are not coherent with each other: the partial tiles travel as
  // relaxed device-scope 64-bit atomic stores / loads (sc1: coherent at the memory side, no ordering of their own) and the
  // ticket is a device-scope atomic taken after the stores have retired.  Measured alternatives: a release fence
  // (buffer_wbl2 = write back the XCD's whole L2) made the split launch slower than the under-filled one (cfg3 78.3 vs
  // 72.4 ms); `volatile` accesses are followed by a full vmcnt(0) each (16 serial round trips per slice, +25 us per launch).
  if constexpr (SKFIX) {
    if (p.sk_ws != nullptr && sk_n > 1) {
      constexpr int NT = 64 * NW, TILE = BM * BN, NP = 16 * MT;              // 8-byte pieces per thread (32 MT accumulator floats)
      // piece pc of the thread's accumulators, whichever MFMA shape filled them
      // (element access by value: a reference cannot bind to a vector element)
      auto piece_get = [&](int pc, int e) -> float {
        if constexpr (M16) return acc16[pc / (2 * MT * 2)][(pc / 2) % (2 * MT)][2 * (pc % 2) + e];
        else return acc[pc / (MT * 8)][(pc / 8) % MT][2 * (pc % 8) + e];
      };
      auto piece_set = [&](int pc, int e, float v) {
        if constexpr (M16) acc16[pc / (2 * MT * 2)][(pc / 2) % (2 * MT)][2 * (pc % 2) + e] = v;
        else acc[pc / (MT * 8)][(pc / 8) % MT][2 * (pc % 8) + e] = v;
      };
      const int nz = sk_n;
      const int64_t tile_id = p.sk_slices > 1 ? (int64_t)(wl - p.sk_full) : (int64_t)blockIdx.y * n_tiles + tile;
      uint64_t* base = (uint64_t*)(p.sk_ws + tile_id * nz * TILE);
      uint64_t* mine = base + (int64_t)sk_slice * (TILE / 2);
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) {
        const f32x2_t v = {piece_get(pc, 0), piece_get(pc, 1)};
        __hip_atomic_store(mine + (pc * NT + tid), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the partial tile has left this CU before the ticket is taken
      __syncthreads();
      if (tid == 0) *(volatile unsigned*)smem = atomicAdd(p.sk_cnt + tile_id, 1u);
      __syncthreads();
      const unsigned ticket = *(volatile unsigned*)smem;
      if (ticket != (unsigned)(nz - 1)) return;
      if (tid == 0) p.sk_cnt[tile_id] = 0;   // ready for the next launch on this stream
      // the combine walks the accumulator in chunks of CH pieces with the loads of ALL slices of a chunk in flight together
      // (memory-side round trips; one slice per trip made the combine as long as the K loop); additions keep slice order
      constexpr int CH = (BM == 256) ? 4 : 8, SMAX = (BM == 256) ? 8 : 16;
#pragma unroll
      for (int c0 = 0; c0 < NP; c0 += CH) {
        uint64_t part[SMAX][CH];
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
          if (s < nz) {
            const uint64_t* src = base + (int64_t)s * (TILE / 2);
#pragma unroll
            for (int i = 0; i < CH; ++i) part[s][i] = __hip_atomic_load(src + ((c0 + i) * NT + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        f32x2_t sum[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) sum[i] = f32x2_t{0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SMAX; ++s)
          if (s < nz) {
#pragma unroll
            for (int i = 0; i < CH; ++i) sum[i] += __builtin_bit_cast(f32x2_t, part[s][i]);
          }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          piece_set(c0 + i, 0, sum[i][0]);
          piece_set(c0 + i, 1, sum[i][1]);
        }
      }
      __syncthreads();
    }
  }
  if constexpr (M16 && (EPI & ffvc_gemm_detail::EPI_K_VQ) != 0) {
    ffvc_gemm_detail::gemm_epilogue_vq16<MT>(p, acc16, m0, n0, wm, wn, lane);      // FFVC_F_VQ_ARGMIN: nothing is stored
  } else if constexpr (M16) {
    if (vec_ok == 2)
      // register exchange on the convolutions and on the activation-forward kinds that also store act' (two 16-bit tensors out: 178 vs
      // 189 us at 16384x4096x1024, profiles/r06_gemm3_ab.txt), LDS pads on the other K-major x K-major launches (gemm_common.h)
      ffvc_gemm_detail::gemm_epilogue_out16<L, MT, EPI, (XMODE == FFVC_OP_CONV3X3 || (EPI & ffvc_gemm_detail::EPI_K_FWDG) != 0
                                                             ? (FFVC_EPI_PERM != 0) : (FFVC_EPI_PERM != 0 && FFVC_EPI_PERM_NT != 0))>(
          p, acc16, m0, n0, wm, wn, lane, zo, zi, smem + (RING ? 2 : 1) * STAGE + wid * 4096);
    else
      ffvc_gemm_detail::gemm_epilogue16<L, MT, true>(p, acc16, m0, n0, wm, wn, lane, zo, zi, 1);
  } else {
    if (vec_ok == 2)
      ffvc_gemm_detail::gemm_epilogue_rows<L, MT, false, EPI>(p, acc, m0, n0, wm, wn, lane, zo, zi,
                                                               smem + (RING ? 2 : 1) * STAGE + wid * 4096);
    else
      ffvc_gemm_detail::gemm_epilogue<L, MT, true>(p, acc, m0, n0, wm, wn, lane, zo, zi, 1);
  }
}

// ---- 8-phase 256x256 kernel (K-major W; K-major or implicit-im2col X) --------------------------------------------------
// The ring kernel above spends ~25 % of its time with both waves of a SIMD stuck in LDS-DMA issue / fragment reads at the
// same moment (they run the same schedule in lock step after every barrier), so the matrix pipe idles.  Here a K tile is
// four phases, each a LOAD segment (fragment reads of ONE 64x32 quadrant of the wave tile + two DMA pieces) and an MFMA
// segment (8 MFMAs), separated by workgroup barriers, and the two wave groups (waves 0-3 / 4-7 = the two waves of every
// SIMD) run ONE SEGMENT APART: while one wave of a SIMD loads, the other multiplies.
//
//   quadrants of the 128(M) x 64(N) wave tile per K tile:  Q1 = XA x WA, Q2 = XA x WB, Q3 = XB x WB, Q4 = XB x WA
//     XA / XB = first / second 64 rows of the wave's X rows (DMA pieces {0,2} / {1,3} of the 256-row X tile),
//     WA / WB = first / second 32 of its W rows; the W rows of a wave are 32*wn.. and 128+32*wn.., so WA of all waves is
//     W pieces {0,1} and WB is {2,3} (epilogue column map NSPLIT).
//   L1 reads XA, WA (12 x b128) | L2 reads WB (4) | L3 reads XB (8) | L4 reads nothing (WA fragments are kept).
//   Half-tile slots die early: XA, WA after L1, WB after L2, XB after L3 (of BOTH groups, i.e. one segment later), which
//   is what lets the loader run a full K tile ahead inside a 2-tile ring.  The DMA pieces are issued INSIDE the MFMA
//   segments (two per segment, between the MFMAs: measured with s_memtime, a load segment that also issued them was
//   twice as long as an MFMA segment and set the pace):
//     M1(t) stages (t+1).XB -> other buffer     M2(t) stages (t+2).XA -> this buffer
//     M3(t) stages (t+2).WA -> this buffer      M4(t) stages (t+2).WB -> this buffer
//   One counted wait per K tile, right before the barrier that ends group 0's M4 / group 1's L4 (the same barrier):
//   group 0 leaves its three youngest issues in flight (vmcnt(6)), group 1 its two (vmcnt(4)): (t+1).XB and everything
//   older is then complete, which covers every read up to the next wait.  Fragment reads are retired (lgkmcnt(0)) BEFORE
//   the barrier that ends their load segment, so a slot may be restaged one segment after its last read.
#ifndef FFVC_NO_GEMM8
#ifdef FFVC_G8_TIMING
__device__ unsigned long long g8_stamps[2][64];     // [group][event]: s_memtime at segment boundaries of K tiles 4..5, block 0
#define G8_STAMP(idx)                                                                              \
  do {                                                                                             \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (tid & 255) == 0 && t >= 4 && t < 6) \
      g8_stamps[grp][(t - 4) * 16 + (idx)] = __builtin_amdgcn_s_memtime();                        \
  } while (0)
#else
#define G8_STAMP(idx)
#endif
template <typename L, int XMODE>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const ffvc_gemm_desc p, int tiles_n, int n_tiles, int ksplit_len,
                                                       int vec_ok, const uint16_t* zero, int gm) {
  constexpr int BM = 256, BN = 256, MT = 4, NW = 8;
  constexpr int XTILE = BM * 128, STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // the ONLY LDS object: 2 stages | 8 x 4 KiB pads
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1, grp = wid >> 2;
  const int l31 = lane & 31;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  if (gm > 1) {
    const int width = gm * tiles_n;
    const int g = tile / width, rem = tile - g * width;
    const int first = g * gm;
    const int gsz = min(n_tiles / tiles_n - first, gm);
    tn = rem / gsz;
    tm = first + (rem - tn * gsz);
  } else {
    tm = tile / tiles_n;
    tn = tile - tm * tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const int zo = z / p.batch_inner, zi = z - zo * p.batch_inner;
  const int k_begin = blockIdx.z * ksplit_len;
  const int k_end = min(p.K, k_begin + ksplit_len);
  const uint16_t* xb = (const uint16_t*)p.x + zo * p.xbo + zi * p.xbi;
  const uint16_t* wb = (const uint16_t*)p.w + zo * p.wbo + zi * p.wbi;
  using XDma = typename std::conditional<XMODE == FFVC_OP_CONV3X3, ConvDmaB<BM, NW>, KMajorDmaB<BM, NW>>::type;
  XDma sx;
  KMajorDmaB<BN, NW> sw;
  if constexpr (XMODE == FFVC_OP_CONV3X3)
    sx.init(xb, m0, p.M, p.conv_H, p.conv_W, p.conv_Cin, (p.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0, tid);
  else
    sx.init(xb + (int64_t)m0 * p.ldx, p.ldx, p.M - m0, tid);
  sw.init(wb + (int64_t)n0 * p.ldw, p.ldw, p.N - n0, tid);
  (void)zero;

  f32x16_t acc[2][MT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  const int nk = (k_end - k_begin + BK - 1) / BK;
  auto kof = [&](int t) { return k_begin + t * BK; };
  // prologue: tile 0 completely, tile 1 without XB (L1(0) stages it)
  if (nk > 0) {
    sx.issue(smem, kof(0), k_end);
    sw.issue(smem + XTILE, kof(0), k_end);
  }
  if (nk > 1) {
    sx.template issue2<0, 2>(smem + STAGE, kof(1), k_end);
    sw.issue(smem + STAGE + XTILE, kof(1), k_end);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();          // group 1 runs one segment behind group 0

  const int xrow = wm * 128 + l31;                 // + 32 b
  const int wrowA = wn * 32 + l31, wrowB = 128 + wn * 32 + l31;
  u32x4_t fx[2][4], fwA[4], fwB[4];
  auto end_load = [&]() {                          // fragment reads retired, then the segment's barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto end_mma = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one MFMA segment: quadrant (a; b0, b0+1) += fw x fx, with the segment's two DMA pieces issued between the MFMAs (the
  // matrix pipe keeps draining its queue while the wave is busy in LDS-DMA issue)
  auto mma_seg = [&](int a, int b0, const u32x4_t (&fw)[4], auto&& stage) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int b = 0; b < 2; ++b) mma_lo<L>(acc[a][b0 + b], fw[s], fx[b][s]);
      if (s == 0) stage(0);
      if (s == 2) stage(1);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  for (int t = 0; t < nk; ++t) {
    unsigned char* cur = smem + (t & 1) * STAGE;
    unsigned char* oth = smem + ((t + 1) & 1) * STAGE;
    const unsigned char* sX = cur;
    const unsigned char* sW = cur + XTILE;
    const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
    const int k1 = kof(t + 1), k2 = kof(t + 2);
    // ---- L1: XA, WA fragments
    G8_STAMP(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) fwA[s] = frag_kmajor(sW, wrowA, s, lane);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s) fx[b][s] = frag_kmajor(sX, xrow + 32 * b, s, lane);
    G8_STAMP(1);
    end_load();
    G8_STAMP(3);
    // ---- M1: Q1 = XA x WA; stage (t+1).XB -> other buffer
    mma_seg(0, 0, fwA, [&](int i) { if (n1) sx.issue1(oth, i ? 3 : 1, k1, k_end); });
    G8_STAMP(4);
    end_mma();
    G8_STAMP(5);
    // ---- L2: WB fragments
#pragma unroll
    for (int s = 0; s < 4; ++s) fwB[s] = frag_kmajor(sW, wrowB, s, lane);
    G8_STAMP(6);
    end_load();
    G8_STAMP(8);
    // ---- M2: Q2 = XA x WB; stage (t+2).XA -> this buffer (XA was last read in L1, two segments ago for either group)
    mma_seg(1, 0, fwB, [&](int i) { if (n2) sx.issue1(cur, i ? 2 : 0, k2, k_end); });
    G8_STAMP(9);
    end_mma();
    G8_STAMP(10);
    // ---- L3: XB fragments
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s) fx[b][s] = frag_kmajor(sX, xrow + 64 + 32 * b, s, lane);
    G8_STAMP(11);
    end_load();
    G8_STAMP(13);
    // ---- M3: Q3 = XB x WB; stage (t+2).WA
    mma_seg(1, 2, fwB, [&](int i) { if (n2) sw.issue1(cur + XTILE, i, k2, k_end); });
    G8_STAMP(14);
    end_mma();
    G8_STAMP(15);
    // ---- L4: nothing to read (WA fragments are kept); group 1's counted wait: its M1..M3 issues of this tile are behind
    // it, (t+1).XB (M1) has to be complete, the two younger ones may stay in flight
    if (grp) {
      if (n2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    end_load();
    // ---- M4: Q4 = XB x WA; stage (t+2).WB; group 0's counted wait (M2..M4 issues may stay in flight)
    mma_seg(0, 2, fwA, [&](int i) { if (n2) sw.issue1(cur + XTILE, 2 + i, k2, k_end); });
    if (!grp) {
      if (n2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    end_mma();
  }
  if (!grp) __builtin_amdgcn_s_barrier();         // balance group 1's extra barrier
  if (vec_ok == 2)
    ffvc_gemm_detail::gemm_epilogue_rows<L, MT, true>(p, acc, m0, n0, wm, wn, lane, zo, zi, smem + 2 * STAGE + wid * 4096);
  else
    ffvc_gemm_detail::gemm_epilogue<L, MT, true, true>(p, acc, m0, n0, wm, wn, lane, zo, zi, 1);
}
#endif  // FFVC_NO_GEMM8

#ifdef FFVC_BUILD_PERSIST   // opt-in build (adds ~2 min of compile time): make CXXEXTRA=-DFFVC_BUILD_PERSIST
// ---- persistent variant of the ring kernel ------------------------------------------------------------------------------
// One workgroup per CU slot walks work items w = blockIdx.x, + gridDim.x, ... (tile x batch x K-split).  The K loop is
// ONE stream of stages across work items: while the last stage of an item is being multiplied, the DMA state is
// re-initialised for the next item and its first stage is already on its way into the free half of the ring, so the
// epilogue (stores, activations) of item i overlaps the first loads of item i+1 and no workgroup launch / drain sits
// between tiles.  MEASURED (round 1): 5-8 % slower than the one-workgroup-per-tile launch on every shape of the step
// (static striding loses the hardware's dynamic tile dispatch and the longer-lived DMA state costs registers; a counted
// vmcnt wait that lets the next item start without draining the previous item's stores did not change that) -> kept
// behind FFVC_PERSIST=1|2 for further work, off by default.
struct WorkItem {
  int m0, n0, zo, zi, zs, k_begin, k_end;
};

template <int BM, int BN>
__device__ __forceinline__ WorkItem decode_work(const ffvc_gemm_desc& p, int w, int tiles_n, int n_tiles, int nbatch,
                                                int ksplit_len, int gm) {
  const int rest = w / n_tiles, t_lin = w - rest * n_tiles;
  const int zs = rest / nbatch, zb = rest - zs * nbatch;
  const int q = n_tiles >> 3, r = n_tiles & 7, xcd = t_lin & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t_lin >> 3);
  int tm, tn;
  if (gm > 1) {
    const int width = gm * tiles_n;
    const int grp = tile / width, rem = tile - grp * width;
    const int first = grp * gm;
    const int gsz = min(n_tiles / tiles_n - first, gm);
    tn = rem / gsz;
    tm = first + (rem - tn * gsz);
  } else {
    tm = tile / tiles_n;
    tn = tile - tm * tiles_n;
  }
  WorkItem it;
  it.m0 = tm * BM;
  it.n0 = tn * BN;
  it.zo = zb / p.batch_inner;
  it.zi = zb - it.zo * p.batch_inner;
  it.zs = zs;
  it.k_begin = zs * ksplit_len;
  it.k_end = min(p.K, it.k_begin + ksplit_len);
  return it;
}

template <typename L, int XMODE, int WMODE, int BM, int BN>
__global__ __launch_bounds__(64 * 2 * (BN / 64), (BM == 256 ? 2 : 1)) void gemm2p_kernel(
    const ffvc_gemm_desc p, int tiles_n, int n_tiles, int nbatch, int total, int ksplit_len, int vec_ok,
    const uint16_t* zero, int gm, int exact) {
  constexpr int MT = BM / 64;
  constexpr int NW = 2 * (BN / 64);
  constexpr int XTILE = BM * 128, WTILE = BN * 128;
  constexpr int STAGE = XTILE + WTILE;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  const int l31 = lane & 31;
  int w = blockIdx.x;
  if (w >= total) return;

  using XDma = typename std::conditional<XMODE == FFVC_OP_CONV3X3, ConvDma<BM, NW>,
                                         typename std::conditional<XMODE == FFVC_OP_TRANS, TransDma<BM, NW>, KMajorDma<BM, NW>>::type>::type;
  using WDma = typename std::conditional<WMODE == FFVC_OP_TRANS, TransDma<BN, NW>, KMajorDma<BN, NW>>::type;
  XDma sx;
  WDma sw;
  auto init_dma = [&](const WorkItem& it) {
    const uint16_t* xb = (const uint16_t*)p.x + it.zo * p.xbo + it.zi * p.xbi;
    const uint16_t* wb = (const uint16_t*)p.w + it.zo * p.wbo + it.zi * p.wbi;
    if constexpr (XMODE == FFVC_OP_CONV3X3)
      sx.init(xb, it.m0, p.M, p.conv_H, p.conv_W, p.conv_Cin, (p.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0, tid);
    else if constexpr (XMODE == FFVC_OP_TRANS)
      sx.init(xb, p.ldx, it.m0, tr_cols(p.M, p.ldx), tid);
    else
      sx.init(xb, p.ldx, it.m0, p.M, p.kseg, p.xkso, tid, p.x_mi, p.x_so);
    if constexpr (WMODE == FFVC_OP_TRANS)
      sw.init(wb, p.ldw, it.n0, tr_cols(p.N, p.ldw), tid);
    else
      sw.init(wb, p.ldw, it.n0, p.N, p.kseg, p.wkso, tid, 0, 0);
  };

  f32x16_t acc[2][MT];
  auto compute = [&](const unsigned char* sX, const unsigned char* sW, auto&& between) {
    u32x4_t fa[2][2], fb[2][MT];
    auto fetch = [&](int sub, u32x4_t (&a)[2], u32x4_t (&b)[MT]) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int rw = wn * 64 + t * 32 + l31;
        a[t] = (WMODE == FFVC_OP_TRANS) ? frag_trans<BN>(sW, rw, sub, lane) : frag_kmajor(sW, rw, sub, lane);
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int rx = wm * (32 * MT) + t * 32 + l31;
        b[t] = (XMODE == FFVC_OP_TRANS) ? frag_trans<BM>(sX, rx, sub, lane) : frag_kmajor(sX, rx, sub, lane);
      }
    };
    fetch(0, fa[0], fb[0]);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      if (sub < 3) fetch(sub + 1, fa[(sub + 1) & 1], fb[(sub + 1) & 1]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) mma_lo<L>(acc[a][b], fa[sub & 1][a], fb[sub & 1][b]);
      between(sub);
    }
  };

  WorkItem cur = decode_work<BM, BN>(p, w, tiles_n, n_tiles, nbatch, ksplit_len, gm);
  init_dma(cur);
  sx.issue(smem, cur.k_begin, cur.k_end, zero, tid);
  sw.issue(smem + XTILE, cur.k_begin, cur.k_end, zero, tid);
  int g = 0;                                        // running stage count: ring half = g & 1
  while (true) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    const int nk = (cur.k_end - cur.k_begin + BK - 1) / BK;
    const int wnext = w + gridDim.x;
    const bool has_next = wnext < total;
    WorkItem nxt = cur;
    for (int kt = 0; kt < nk; ++kt) {
      // First stage of a follow-up item: its DMA loads were issued BEFORE the previous item's epilogue stores.  vmcnt
      // retires in issue order and counts stores too, so waiting for "at most 4*MT operations outstanding" (every
      // wave issues at least that many row stores per epilogue when all tiles are interior: `exact`) releases the
      // wave as soon as the loads have landed, without draining the stores.
      if (kt == 0 && g > 0 && exact) {
        if constexpr (MT == 4)
          asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      unsigned char* cb = smem + (g & 1) * STAGE;
      unsigned char* nb = smem + ((g + 1) & 1) * STAGE;
      ++g;
      const bool last = kt + 1 == nk;
      int kn = cur.k_begin + (kt + 1) * BK, ke = cur.k_end;
      if (last && has_next) {
        // every load of the current item has been issued: the DMA state can move on to the next item now (fragment
        // registers are dead here), its first stage then streams in under this stage's MFMAs and the epilogue
        nxt = decode_work<BM, BN>(p, wnext, tiles_n, n_tiles, nbatch, ksplit_len, gm);
        init_dma(nxt);
        kn = nxt.k_begin;
        ke = nxt.k_end;
      }
      const bool more = !last || has_next;
      compute(cb, cb + XTILE, [&](int sub) {
        if (more && sub == 0) sx.issue(nb, kn, ke, zero, tid);
        if (more && sub == 1) sw.issue(nb + XTILE, kn, ke, zero, tid);
      });
    }
    if (vec_ok == 2)
      ffvc_gemm_detail::gemm_epilogue_rows<L, MT>(p, acc, cur.m0, cur.n0, wm, wn, lane, cur.zo, cur.zi,
                                                         smem + 2 * STAGE + wid * 4096, cur.zs);
    else
      ffvc_gemm_detail::gemm_epilogue<L, MT, true>(p, acc, cur.m0, cur.n0, wm, wn, lane, cur.zo, cur.zi, 1, cur.zs);
    if (!has_next) break;
    cur = nxt;
    w = wnext;
  }
}

#endif  // FFVC_BUILD_PERSIST

// the buffer-descriptor DMA addresses an operand with 32-bit byte offsets below 2 GiB from its descriptor's origin
template <int MODE>
inline bool dma_operand_ok(const ffvc_gemm_desc& d, bool is_x) {
  const int64_t lim = 0x7FFFFF00ll;
  const int64_t ld = is_x ? d.ldx : d.ldw;
  const int64_t kspan = d.kseg ? (int64_t)(d.K / d.kseg) * (is_x ? d.xkso : d.wkso) + d.kseg : (int64_t)d.K;
  if (MODE == FFVC_OP_CONV3X3) {
    const int ups = (d.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0;
    const int64_t images = d.M / ((int64_t)d.conv_H * d.conv_W);
    return images * (d.conv_H >> ups) * (d.conv_W >> ups) * d.conv_Cin * 2 < lim;
  }
  if (MODE == FFVC_OP_TRANS) return ((int64_t)d.K * ld + 256) * 2 < lim;           // k * ld + column
  if (is_x && d.x_mi) return ((int64_t)(d.M / d.x_mi + 1) * d.x_so + (int64_t)d.x_mi * ld + kspan) * 2 < lim;
  return (256 * ld + kspan) * 2 < lim;                                                // tile rows + k
}
template <int XMODE>
inline bool g8_offsets_ok(const ffvc_gemm_desc& d) {
  return dma_operand_ok<XMODE>(d, true) && dma_operand_ok<FFVC_OP_KMAJOR>(d, false);
}

// Per-stream scratch of the in-kernel split-K: partial tiles + one (self-resetting) ticket counter per tile.
inline bool skfix_scratch(hipStream_t st, size_t ws_bytes, size_t n_cnt, float** ws, uint32_t** cnt) {
  struct Slot {
    float* ws = nullptr;
    size_t ws_bytes = 0;
    uint32_t* cnt = nullptr;
    size_t n_cnt = 0;
  };
  static std::mutex mu;
  static std::map<hipStream_t, Slot> pool;
  std::lock_guard<std::mutex> lk(mu);
  Slot& s = pool[st];
  if (s.ws_bytes < ws_bytes) {
    if (s.ws) {
      (void)hipStreamSynchronize(st);
      (void)hipFree(s.ws);
      s.ws = nullptr;
      s.ws_bytes = 0;
    }
    const size_t want = ws_bytes < (32u << 20) ? (32u << 20) : ws_bytes;
    void* q = nullptr;
    if (hipMalloc(&q, want) != hipSuccess) return false;
    s.ws = (float*)q;
    s.ws_bytes = want;
  }
  if (s.n_cnt < n_cnt) {
    if (s.cnt) {
      (void)hipStreamSynchronize(st);
      (void)hipFree(s.cnt);
      s.cnt = nullptr;
      s.n_cnt = 0;
    }
    const size_t want = n_cnt < 4096 ? 4096 : n_cnt;
    void* q = nullptr;
    if (hipMalloc(&q, want * sizeof(uint32_t)) != hipSuccess) return false;
    if (hipMemsetAsync(q, 0, want * sizeof(uint32_t), st) != hipSuccess) return false;
    s.cnt = (uint32_t*)q;
    s.n_cnt = want;
  }
  *ws = s.ws;
  *cnt = s.cnt;
  return true;
}

template <typename L, int XMODE, int WMODE, int BM, int BN>
int launch2(const ffvc_gemm_desc& d_in, hipStream_t st, int vec_ok, const uint16_t* zero) {
  ffvc_gemm_desc d = d_in;
  d.sk_ws = nullptr;
  d.sk_cnt = nullptr;
  d.sk_full = 0;
  d.sk_slices = 0;
  const int tiles_m = ceil_div(d.M, BM), tiles_n = ceil_div(d.N, BN);
  const int n_tiles = tiles_m * tiles_n;
  int split = d.split_k < 1 ? 1 : d.split_k;
  const int ksteps = ceil_div(d.K, BK);
  if constexpr (FFVC_MFMA16 && XMODE == FFVC_OP_KMAJOR && WMODE == FFVC_OP_KMAJOR && BM <= 128 && BN == 128) {
    // Under-filled grid with a long reduction (VitGAN / x-transformer linears at a per-GPU batch of 16-32 samples: 512 rows):
    // 32 workgroups walking K = 4096 leave 7/8 of the chip idle (52 TFLOP/s).  Split K inside the launch; FFVC_SK_FIXUP=0 off.
    static int fix_opt = -1;
    if (fix_opt < 0) {
      const char* e = getenv("FFVC_SK_FIXUP");
      fix_opt = e ? atoi(e) : 1;
    }
    const int64_t wgs = (int64_t)n_tiles * d.batch;
    // (64x128 tiles, round 5: two workgroups fit a CU and a tile is half the work -> split up to 128 tiles)
    if (fix_opt && split == 1 && d.slab_stride == 0 && wgs <= (BM == 64 ? 128 : 64) && ksteps >= 16 && dma_operand_ok<XMODE>(d, true) &&
        dma_operand_ok<WMODE>(d, false)) {
      int want = (int)((fix_opt > 1 ? fix_opt : 256) / wgs);     // ~one workgroup per CU
      if (want > ksteps / 4) want = ksteps / 4;                   // at least 4 K steps (256 deep) per slice
      if (want > 16) want = 16;                                   // the last workgroup reads every slice of its tile
      if (want >= 2) {
        const int len = ceil_div(ksteps, want);
        const int nz = ceil_div(ksteps, len);
        float* ws = nullptr;
        uint32_t* cnt = nullptr;
        if (nz >= 2 && skfix_scratch(st, (size_t)wgs * nz * BM * BN * sizeof(float), (size_t)wgs, &ws, &cnt)) {
          split = nz;
          d.sk_ws = ws;
          d.sk_cnt = cnt;
        }
      }
    }
  }
  if constexpr (FFVC_MFMA16 && XMODE == FFVC_OP_KMAJOR && WMODE == FFVC_OP_KMAJOR && BM == 256 && BN == 256) {
    // A 256x256-tile grid that fills a fraction of the chip with a long reduction (ViT linears at 16-32 samples per GPU, the text
    // tower): 2-4 K slices per tile, combined inside the launch (6400x768x3072, 75 tiles: 68 -> 55 us; 4928x512x6144, 40 tiles:
    // 123 -> 60 us; 8192x1024x4096, 128 tiles: 85 -> 76 us).  FFVC_SK_FIXUP=0 off.
    static int fix256 = -1;
    if (fix256 < 0) {
      const char* e = getenv("FFVC_SK_FIXUP");
      fix256 = e ? atoi(e) : 1;
    }
    if (fix256 && split == 1 && !(d.flags & (FFVC_F_SPLITK_INKERNEL | FFVC_F_VQ_ARGMIN)) && d.slab_stride == 0 && d.batch == 1 && n_tiles >= 24 && n_tiles <= 100 &&
        ksteps >= 32 && dma_operand_ok<XMODE>(d, true) && dma_operand_ok<WMODE>(d, false)) {
      int want = 200 / n_tiles;
      if (want > 4) want = 4;
      if (want > ksteps / 12) want = ksteps / 12;
      if (want >= 2) {
        const int len = ceil_div(ksteps, want);
        const int nz = ceil_div(ksteps, len);
        float* ws = nullptr;
        uint32_t* cnt = nullptr;
        if (nz >= 2 && skfix_scratch(st, (size_t)n_tiles * nz * BM * BN * sizeof(float), (size_t)n_tiles, &ws, &cnt)) {
          split = nz;
          d.sk_ws = ws;
          d.sk_cnt = cnt;
        }
      }
    }
  }
  if (d.flags & FFVC_F_SPLITK_INKERNEL) {
    // the caller's explicit form: d.split_k K slices per tile, combined inside the launch (weight gradients: 64 tiles x 4)
    constexpr bool CAP = (FFVC_MFMA16 && XMODE == FFVC_OP_KMAJOR && WMODE == FFVC_OP_KMAJOR && ((BM <= 128 && BN == 128) || (BM == 256 && BN == 256))) ||
                         (XMODE == FFVC_OP_TRANS && WMODE == FFVC_OP_TRANS && BM == 256 && BN == 256);
    constexpr int SMAXH = (BM == 256) ? 8 : 16;
    if (split > 1) {
      if (!CAP || split > SMAXH || split > ksteps) {
        ffvc_set_error("ffvc_gemm: FFVC_F_SPLITK_INKERNEL: split_k=%d not available for this kernel (%dx%d tile, modes %d/%d)", split, BM, BN,
                       XMODE, WMODE);
        return FFVC_E_BADARG;
      }
      const int64_t wgs = (int64_t)n_tiles * d.batch;
      float* ws = nullptr;
      uint32_t* cnt = nullptr;
      if (!skfix_scratch(st, (size_t)wgs * split * BM * BN * sizeof(float), (size_t)wgs, &ws, &cnt)) {
        ffvc_set_error("ffvc_gemm: FFVC_F_SPLITK_INKERNEL: scratch allocation failed");
        return FFVC_E_BADARG;
      }
      d.sk_ws = ws;
      d.sk_cnt = cnt;
    }
  }
  if (split > ksteps) split = ksteps < 1 ? 1 : ksteps;
  const int ksplit_len = ceil_div(ksteps, split) * BK;
  split = ceil_div(d.K, ksplit_len);
  if (split < 1) split = 1;
  // Partial slabs: the caller sized and will reduce exactly d.split_k slabs, so launch that many K slices; a slice
  // whose K range is empty runs zero K steps and stores a zero tile (never leave a slab unwritten).
  if (d.slab_stride != 0 && d.split_k > split) split = d.split_k;
  dim3 grid(n_tiles, d.batch, split);
  constexpr int nthreads = 64 * 2 * (BN / 64);
  // stage ring + one 4 KiB row-store pad per wave (256x256: 128 + 32 = all 160 KiB of the CU)
  constexpr int lds = ((BM == 256 && BN == 128) ? 1 : 2) * (BM * 128 + BN * 128) + 2 * (BN / 64) * 4096;
  static bool attr_set = false;
  if (!attr_set) {   // > 64 KiB of dynamic LDS needs the opt-in attribute (once per instantiation)
    (void)hipFuncSetAttribute((const void*)gemm2_kernel<L, XMODE, WMODE, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  static int persist = -1, n_cu = 0;
  if (persist < 0) {
    const char* e = getenv("FFVC_PERSIST");
    persist = e ? atoi(e) : 0;   // 0 off | 1 every ring launch | 2 short reductions (K <= 512) with interior tiles only
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  static int gm_opt = -1;
  if (gm_opt < 0) {
    const char* e = getenv("FFVC_TILE_GM");
    gm_opt = e ? atoi(e) : 4;
  }
  // workgroups resident per XCD: 32 CUs x (1 | 2) -> aim at a square-ish block
  // measured (profiles/r01_gemm_micro.txt): grouping only pays for very wide outputs (8192^3: +7 %); the step's own
  // shapes (tiles_n <= 16) are neutral to slightly worse, so they keep the row-major order
  int gm = gm_opt >= 0 && getenv("FFVC_TILE_GM") ? gm_opt : (tiles_n > 16 ? 4 : 1);
  // weight gradients whose output is wider than tall (dW2 of the channel MLP: 1024 x 4096 from a 16384-long reduction):
  // pairs of tile rows share the wide operand's panel, 774 -> 847 TFLOP/s isolated (tools/tt_gm_ab.py)
  if (XMODE == FFVC_OP_TRANS && WMODE == FFVC_OP_TRANS && !getenv("FFVC_TILE_GM") && tiles_n > tiles_m) gm = 2;
  if (gm > tiles_m) gm = tiles_m;
  {
    static int stagger = -1;
    if (stagger < 0) {
      const char* e = getenv("FFVC_STAGGER_TICKS");
      stagger = e ? atoi(e) : 0;
    }
    // only grids with several full rounds of equal tiles
    if (stagger > 0 && (int64_t)n_tiles * d.batch * split >= 3 * n_cu) gm |= stagger << 8;
  }
#ifdef FFVC_BUILD_PERSIST
  constexpr bool ring = !(BM == 256 && BN == 128);
  if constexpr (ring) {
    const bool interior = (d.M % BM) == 0 && (d.N % BN) == 0 && vec_ok == 2 && !(d.flags & (FFVC_F_OUT_F32 | FFVC_F_ACCUM_OUT | FFVC_F_ATOMIC_OUT));
    // gemm2p_kernel has no in-kernel split-K combine: a launch whose K slices meet through sk_ws (or that would store unsynchronised
    // partial tiles) stays on gemm2_kernel
    const bool p_split_ok = d.sk_ws == nullptr && !(split > 1 && d.slab_stride == 0 && !(d.flags & FFVC_F_ATOMIC_OUT));
    if (p_split_ok && !(d.flags & FFVC_F_VQ_ARGMIN) && (persist == 1 || (persist == 2 && d.K <= 512 && interior && (int64_t)n_tiles * d.batch * split > n_cu))) {
      const int64_t total = (int64_t)n_tiles * d.batch * split;
      const int slots = n_cu * ((BM == 256) ? 1 : 2);
      int pgrid = total < slots ? (int)total : slots;
      static bool pattr_set = false;
      if (!pattr_set) {
        (void)hipFuncSetAttribute((const void*)gemm2p_kernel<L, XMODE, WMODE, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        pattr_set = true;
      }
      if (total < (1ll << 31)) {
        hipLaunchKernelGGL((gemm2p_kernel<L, XMODE, WMODE, BM, BN>), dim3(pgrid), dim3(nthreads), lds, st, d, tiles_n, n_tiles,
                           d.batch, (int)total, ksplit_len, vec_ok, zero, gm, interior ? 1 : 0);
        hipError_t pe = hipGetLastError();
        if (pe != hipSuccess) {
          ffvc_set_error("gemm2p launch failed: %s", hipGetErrorString(pe));
          return -(int)pe - 1000;
        }
        return 1;
      }
    }
  }
#else
  (void)persist;
#endif
#ifndef FFVC_NO_GEMM8
  if constexpr (BM == 256 && BN == 256 && WMODE == FFVC_OP_KMAJOR && (XMODE == FFVC_OP_KMAJOR || XMODE == FFVC_OP_CONV3X3)) {
    // FFVC_GEMM8: 0 (default) = ring kernel only: once the ring kernel got the buffer-descriptor DMA it passed the 8-phase
    // kernel on every shape (4096^3: 1175 vs 1004 TFLOP/s, profiles/r02_gemm8_ab.txt) | 1 = 8-phase for plain K-major x
    // K-major launches with K >= 2048 | 2 = every eligible launch; ffvc_set_option("gemm8", 1) forces it for the tests
    static int use8 = -1;
    if (use8 < 0) {
      const char* e = getenv("FFVC_GEMM8");
      use8 = e ? atoi(e) : 0;
    }
    const bool pays = XMODE == FFVC_OP_KMAJOR && d.K >= 2048;
    // gemm8_kernel has no ticket / combine either: the auto split-K above (24..100 tiles, K >= 2048) must not reach it
    const bool g8_split_ok = d.sk_ws == nullptr && !(split > 1 && d.slab_stride == 0 && !(d.flags & FFVC_F_ATOMIC_OUT));
    if (g8_split_ok && !(d.flags & FFVC_F_VQ_ARGMIN) && (use8 == 2 || (use8 == 1 && pays) || g_force_gemm8) && d.kseg == 0 && d.x_mi == 0 && g8_offsets_ok<XMODE>(d)) {
      static bool attr8 = false;
      if (!attr8) {
        (void)hipFuncSetAttribute((const void*)gemm8_kernel<L, XMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr8 = true;
      }
      hipLaunchKernelGGL((gemm8_kernel<L, XMODE>), grid, dim3(nthreads), lds, st, d, tiles_n, n_tiles, ksplit_len, vec_ok,
                         zero, gm);
      hipError_t e8 = hipGetLastError();
      if (e8 != hipSuccess) {
        ffvc_set_error("gemm8 launch failed: %s", hipGetErrorString(e8));
        return -(int)e8 - 1000;
      }
      return 1;
    }
  }
#endif
  {
    static int use_buf = -1;
    if (use_buf < 0) {
      const char* e = getenv("FFVC_DMA_BUF");
      use_buf = e ? atoi(e) : 1;
    }
    if (use_buf && dma_operand_ok<XMODE>(d, true) && dma_operand_ok<WMODE>(d, false)) {
      using namespace ffvc_gemm_detail;
      // epilogue class: convolutions never carry an activation (GroupNorm moments only), weight gradients (TRANS x TRANS)
      // neither; the K-major x K-major / K-major x TRANS kernels exist lean and complete
      static int lean_opt = -1;
      if (lean_opt < 0) {
        const char* e = getenv("FFVC_EPI_LEAN");
        lean_opt = e ? atoi(e) : 1;
      }
      const bool wants_act = d.act != FFVC_ACT_NONE || (d.flags & (FFVC_F_MUL_ACT_GRAD | FFVC_F_WRITE_PREACT | FFVC_F_COLSUM));
      const bool wants_gn = d.flags & FFVC_F_GN_SUMS;
      if constexpr (FFVC_MFMA16 && XMODE == FFVC_OP_KMAJOR && WMODE == FFVC_OP_KMAJOR && BM == 256 && BN == 256) {
        // Tail mode of the in-kernel split-K: one 256x256 workgroup per CU, so 260 tiles (ViT-L/14's 16448 rows x 1024) or 300
        // (ViT-B/32's 25600 x 768) run TWO rounds with the second one nearly empty.  The tiles of a last round that fills at
        // most half of the chip are cut along K so that it takes a fraction of a round; capped by the partial-tile traffic
        // (256 KiB per slice, written and read once) and kept to reductions of 2048 and more: the ONE workgroup that combines a
        // tile reads all its slices through the memory side (~20 us for 8 x 256 KiB), which a shorter K loop does not repay
        // (16448x1024x4096: 189 -> 167 us, 25600x768x3072: 141 -> 127; K = 1024 / 768: 4-6 us slower, left alone).
        // FFVC_SK_TAIL=0 off.
        static int tail_opt = -1;
        if (tail_opt < 0) {
          const char* e = getenv("FFVC_SK_TAIL");
          tail_opt = e ? atoi(e) : 1;
        }
        if (tail_opt && split == 1 && d.batch == 1 && d.slab_stride == 0 && d.sk_ws == nullptr && n_tiles > n_cu && ksteps >= 32 &&
            !(d.flags & FFVC_F_VQ_ARGMIN)) {
          const int full = (n_tiles / n_cu) * n_cu, tail = n_tiles - full;
          if (tail > 0 && 2 * tail <= n_cu) {
            int S = n_cu / tail;
            if (S > 8) S = 8;
            if (S > ksteps / 4) S = ksteps / 4;
            while (S > 1 && tail * S > (tail_opt > 1 ? tail_opt : 96)) --S;
            float* ws = nullptr;
            uint32_t* cnt = nullptr;
            if (S >= 2 && skfix_scratch(st, (size_t)tail * S * BM * BN * sizeof(float), (size_t)tail, &ws, &cnt)) {
              d.sk_ws = ws;
              d.sk_cnt = cnt;
              d.sk_full = full;
              d.sk_slices = S;
              grid = dim3(full + tail * S, 1, 1);
            }
          }
        }
      }
      auto go = [&](auto epi_tag) -> int {
        constexpr int EPI = decltype(epi_tag)::value;
        static bool attr_b = false;
        if (!attr_b) {
          (void)hipFuncSetAttribute((const void*)gemm2_kernel<L, XMODE, WMODE, BM, BN, true, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          attr_b = true;
        }
        hipLaunchKernelGGL((gemm2_kernel<L, XMODE, WMODE, BM, BN, true, EPI>), grid, dim3(nthreads), lds, st, d, tiles_n, n_tiles,
                           ksplit_len, vec_ok, zero, gm);
        hipError_t eb = hipGetLastError();
        if (eb != hipSuccess) {
          ffvc_set_error("gemm2 (buffer DMA) launch failed: %s", hipGetErrorString(eb));
          return -(int)eb - 1000;
        }
        return 1;
      };
      if constexpr (XMODE == FFVC_OP_CONV3X3) {
        if (!wants_act && !wants_gn && lean_opt) return go(std::integral_constant<int, EPI_LEAN>{});   // dgrad convolutions
        if (!wants_act) return go(std::integral_constant<int, EPI_GN>{});
      } else if constexpr (XMODE == FFVC_OP_TRANS && WMODE == FFVC_OP_TRANS) {
        if (!wants_act && !wants_gn && lean_opt >= 1 && (d.flags & FFVC_F_OUT_F32) && !d.residual && !d.bias &&
            !(d.flags & (FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT)))
          return go(std::integral_constant<int, EPI_LEAN | EPI_O_F32>{});          // weight-gradient slabs
        if (!wants_act && !wants_gn) return go(std::integral_constant<int, EPI_LEAN>{});
      } else if constexpr (XMODE == FFVC_OP_KMAJOR) {
        if constexpr (FFVC_MFMA16 && WMODE == FFVC_OP_KMAJOR && BM == 256 && BN == 256) {
          if (d.flags & FFVC_F_VQ_ARGMIN) return split == 1 ? go(std::integral_constant<int, EPI_K_VQ>{}) : 0;
        }
        if (d.flags & FFVC_F_VQ_ARGMIN) return 0;
        if constexpr (WMODE == FFVC_OP_KMAJOR) {
          if (lean_opt >= 1 && !wants_act && !wants_gn && !(d.flags & (FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT | FFVC_F_BIAS_ALONG_M)) &&
              d.slab_stride == 0) {
            if (!d.residual && !(d.flags & FFVC_F_OUT_F32)) return go(std::integral_constant<int, EPI_LEAN | EPI_O_T>{});
            if (d.residual && (d.flags & FFVC_F_RES_F32) && (d.flags & FFVC_F_OUT_F32))
              return go(std::integral_constant<int, EPI_LEAN | EPI_O_F32R>{});
          }
        }
        if (lean_opt && !wants_act && !wants_gn) return go(std::integral_constant<int, EPI_LEAN>{});
        if constexpr (WMODE == FFVC_OP_KMAJOR && BM == 256 && BN == 256) {
          // the MLP launches of the Mixer / ViT blocks: one activation, fixed at compile time
          // the kinds fix the rest of the epilogue too: 16-bit plain output, no residual, column bias (forward) / none (backward)
          const bool plain_out = !d.residual && !(d.flags & (FFVC_F_OUT_F32 | FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT)) &&
                                 d.slab_stride == 0 && d.alpha == 1.0f;
          const bool bwd_k = d.flags & FFVC_F_MUL_ACT_GRAD;
          const bool bias_ok = bwd_k ? d.bias == nullptr : (d.bias != nullptr && !(d.flags & FFVC_F_BIAS_ALONG_M));
          if (lean_opt >= 1 && !wants_gn && plain_out && bias_ok && (d.act == FFVC_ACT_GELU || d.act == FFVC_ACT_QUICKGELU)) {
            const bool bwd = d.flags & FFVC_F_MUL_ACT_GRAD;
            if (d.flags & FFVC_F_AUX_ACTGRAD) {        // aux carries act'(pre): specialised store / plain multiply
              if (bwd && !(d.flags & FFVC_F_WRITE_PREACT)) return go(std::integral_constant<int, EPI_K_MULAUX>{});
              if (!bwd && (d.flags & FFVC_F_WRITE_PREACT) && !(d.flags & FFVC_F_COLSUM)) {
                const int r = d.act == FFVC_ACT_GELU ? go(std::integral_constant<int, EPI_K_GELU_FWDG>{})
                                                     : go(std::integral_constant<int, EPI_K_QGELU_FWDG>{});
                return r == 1 ? 2 : r;                 // 2 = aux already holds the derivative (no conversion pass needed)
              }
            }
            if (!bwd && !(d.flags & (FFVC_F_COLSUM | FFVC_F_AUX_ACTGRAD))) {
              if (d.act == FFVC_ACT_GELU) return go(std::integral_constant<int, EPI_K_GELU_FWD>{});
              return go(std::integral_constant<int, EPI_K_QGELU_FWD>{});
            }
            if (bwd && !(d.flags & (FFVC_F_WRITE_PREACT | FFVC_F_AUX_ACTGRAD))) {
              if (d.act == FFVC_ACT_GELU) return go(std::integral_constant<int, EPI_K_GELU_BWD>{});
              return go(std::integral_constant<int, EPI_K_QGELU_BWD>{});
            }
          }
        }
        return go(std::integral_constant<int, EPI_ALL>{});
      } else {
        return go(std::integral_constant<int, EPI_ALL>{});
      }
      // a convolution / weight gradient that does ask for an activation falls through to the global-address kernel below
    }
  }
  if (d.flags & FFVC_F_VQ_ARGMIN) return 0;      // the argmin epilogue exists on the buffer-descriptor 256x256 kernel only
  hipLaunchKernelGGL((gemm2_kernel<L, XMODE, WMODE, BM, BN>), grid, dim3(nthreads), lds, st, d, tiles_n, n_tiles, ksplit_len,
                     vec_ok, zero, gm);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ffvc_set_error("gemm2 launch failed: %s", hipGetErrorString(e));
    return -(int)e - 1000;
  }
  return 1;
}

// cfg: 128 -> 128x128, 256 -> 256x128, 512 -> 256x256, 64 -> 64x128 (K-major x K-major only: the small-M linears of the
// VitGAN / x-transformer mappers, a few hundred rows against 2-8 MB of weights)
template <typename L, int XMODE, int WMODE>
int launch2_cfg_t(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg) {
  if constexpr (FFVC_MFMA16 && XMODE == FFVC_OP_KMAJOR && WMODE == FFVC_OP_KMAJOR) {
    if (cfg == 64) return launch2<L, XMODE, WMODE, 64, 128>(d, st, vec_ok, zero);
  }
  if (cfg == 512) return launch2<L, XMODE, WMODE, 256, 256>(d, st, vec_ok, zero);
  if (cfg == 256) return launch2<L, XMODE, WMODE, 256, 128>(d, st, vec_ok, zero);
  return launch2<L, XMODE, WMODE, 128, 128>(d, st, vec_ok, zero);
}
template <int XMODE, int WMODE>
int launch2_cfg(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg) {
  if (d.in_dtype == FFVC_F16) return launch2_cfg_t<f16_t, XMODE, WMODE>(d, st, vec_ok, zero, cfg);
  return launch2_cfg_t<uint16_t, XMODE, WMODE>(d, st, vec_ok, zero, cfg);
}


}  // namespace
