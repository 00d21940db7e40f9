// Weight gradient of the ResNet stem (7x7 / stride 2 / pad 3, 3 input channels stored as 4, 64 output channels) for gfx950
// (v_mfma_f32_16x16x4_f32).  It is the LAST kernel of the backward — the stem's dY exists only after the max-pool backward and
// the stem BatchNorm backward — so it runs alone at the tail of the step: 380 us on the flattened implicit GEMM, whose 128-column
// tiles compute 256 columns for the 147 real ones (49 taps x 3 channels; the zero pad channel and the tile padding are 43 %
// of its MFMAs).
//
//   dW[co][tap][c] = sum_{n,oy,ox} dY[n][oy][ox][co] * x[n][2*oy + r - 3][2*ox + s - 3][c]
//
// Here the GEMM columns are COMPACT: j = tap*3 + c, 147 -> 160 = ten 16-column MFMA blocks.  Pixels are the reduction: an MFMA
// takes four consecutive output pixels of a row; the B operand is a 4-byte gather out of the LDS input patch at a per-lane
// constant offset (its tap and channel) + the pixel offset, the A operand a 4-byte read of the dY tile (16-byte units of pixel p
// XOR-ed with (p & 3) << 2, applied on the LDS-DMA source side: conflict-free).  A wave owns 32 output channels x 5 column
// blocks (10 accumulators): 7 LDS reads per 10 MFMAs.  Persistent workgroups over 2x32-pixel tiles, patch (through registers,
// zero padded) and dY (LDS-DMA) double-buffered, one partial [64][160] per workgroup, wgrad_stem_reduce_kernel adds the
// partials in a fixed order and scatters the compact columns back to dW's [co][tap*4 + c] layout.
//
// Replaces the weight-gradient half of autograd's conv2d backward for encoder.conv1.
#include "uwm_kernels.h"
#include <cstdlib>

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
__device__ __forceinline__ void st_glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(uintptr_t)l, 16, 0, 0);
}

constexpr int kSR = 2, kSC = 32;                          // output pixels per tile: 2 rows x 32 columns
constexpr int kSPH = 2 * kSR + 5, kSPW = 2 * kSC + 5;     // 9 x 69 input patch (pixels of 4 floats)
constexpr int kSPatch = kSPH * kSPW * 4;                  // floats
constexpr int kSDy = kSR * kSC * 64;                      // floats: [64 px][64 co]
constexpr int kSBuf = kSPatch + kSDy;
constexpr int kSCols = 160;                               // compact columns: 147 real (tap*3 + c) + 13 zero
constexpr int kSPart = 64 * kSCols;

__global__ __launch_bounds__(256, 3) void wgrad_stem_kernel(const WgradArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 1, wb = wave & 1;                // 32 output channels x 5 column blocks
  const int lrow = lane & 15, lq = lane >> 4;
  const int H = a.s0.H, W = a.s0.W;                       // input image (4-channel NHWC)
  const int tilesW = a.Wo / kSC, tilesH = a.Ho / kSR;

  // per-lane constant gather offsets of the five column blocks: column j -> (r, s, c); j >= 147: a zeroed pad slot
  int boff[5]; float bmask[5];
#pragma unroll
  for (int jb = 0; jb < 5; ++jb) {
    const int j = (wb * 5 + jb) * 16 + lrow;
    const int tap = j / 3, c = j - tap * 3, r = tap / 7, s = tap - r * 7;
    const bool ok = j < 147;
    boff[jb] = ok ? ((r * kSPW + s) * 4 + c) : 0;
    bmask[jb] = ok ? 1.f : 0.f;
  }

  // ---- patch staging geometry: 9 x 69 = 621 pixels (one 16-byte unit each), 3 rounds
  int spy[3], spx[3];
#pragma unroll
  for (int rd = 0; rd < 3; ++rd) {
    const int pp = min(rd * 256 + tid, kSPH * kSPW - 1);
    spy[rd] = pp / kSPW; spx[rd] = pp - spy[rd] * kSPW;
  }
  const bool last_live = (2 * 256 + tid) < kSPH * kSPW;
  f4 pv[3];
  auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    oy0 = th * kSR; ox0 = tw * kSC;
  };
  auto patch_load = [&](int t) {
    int n, oy0, ox0; tile_origin(t, n, oy0, ox0);
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
      const int iy = 2 * oy0 - 3 + spy[rd], ix = 2 * ox0 - 3 + spx[rd];
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int yc = min(max(iy, 0), H - 1), xc = min(max(ix, 0), W - 1);
      const f4 v = *(const f4*)(a.s0.ptr + (((size_t)n * H + yc) * W + xc) * 4);
      pv[rd] = ok ? v : (f4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto patch_store = [&](int buf) {
    float* const p_ = smem + buf * kSBuf;
#pragma unroll
    for (int rd = 0; rd < 3; ++rd)
      if (rd < 2 || last_live) *(f4*)(p_ + (rd * 256 + tid) * 4) = pv[rd];
  };
  // dY tile [64 px][64 co] = 1024 units = 16 LDS-DMA instructions, 4 per wave; unit q of pixel p holds logical unit q ^ ((p&3)<<2)
  auto dy_issue = [&](int t, int buf) {
    int n, oy0, ox0; tile_origin(t, n, oy0, ox0);
    float* const d_ = smem + buf * kSBuf + kSPatch;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int L = (i * 4 + wave) * 64 + lane;
      const int p = L >> 4, q = L & 15;
      const int u = q ^ ((p & 3) << 2);
      const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
      st_glds16(a.dy + (((size_t)n * a.Ho + oy) * a.Wo + ox) * 64 + u * 4, d_ + (i * 4 + wave) * 256);
    }
  };

  f4 acc[2][5];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  int t = blockIdx.x;
  if (t < ntiles) { patch_load(t); dy_issue(t, 0); patch_store(0); }
  __syncthreads();

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    const bool more = tn < ntiles;
    if (more) { patch_load(tn); dy_issue(tn, cur ^ 1); }
    const float* const ps = smem + cur * kSBuf;
    const float* const ds = ps + kSPatch;
#pragma unroll
    for (int g = 0; g < 16; ++g) {                         // K-group: 4 consecutive output pixels of one row
      const int p = g * 4 + lq;                            // tile pixel this lane supplies
      const int oyl = p >> 5, oxl = p & 31;
      const int pbase = (oyl * 2 * kSPW + oxl * 2) * 4;
      const int sw = (p & 3) << 2;
      float af[2], bf[5];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int col = (wa * 2 + i) * 16 + lrow;
        af[i] = ds[p * 64 + (((col >> 2) ^ sw) << 2) + (col & 3)];
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) bf[j] = ps[pbase + boff[j]] * bmask[j];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) patch_store(cur ^ 1);
    __syncthreads();
  }

  // partial [64 co][160 cols] of this workgroup: lane holds rows 4*lq + e, column lrow of each block
  float* const o = a.part + (size_t)blockIdx.x * kSPart;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[((wa * 2 + i) * 16 + lq * 4 + e) * kSCols + (wb * 5 + j) * 16 + lrow] = acc[i][j][e];
}

// dw[co][tap*4 + c] += sum_wg part[wg][co][tap*3 + c]: 64 x 147 outputs; workgroup = 8 outputs x 32 partial groups, fixed order
__global__ __launch_bounds__(256) void wgrad_stem_reduce_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw, int wrows, int Kpad) {
  __shared__ float red[32][8];
  const int u8 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int ou = blockIdx.x * 8 + u8;                      // 0 .. 64*147 - 1
  const int co = ou / 147, j = ou - co * 147;
  float s = 0.f;
  if (co < 64)
    for (int k = grp; k < nparts; k += 32) s += part[(size_t)k * kSPart + co * kSCols + j];
  red[grp][u8] = s;
  __syncthreads();
  if (grp == 0 && co < wrows) {
    float tsum = red[0][u8];
#pragma unroll
    for (int g = 1; g < 32; ++g) tsum += red[g][u8];
    const int tap = j / 3, c = j - tap * 3;
    dw[(size_t)co * Kpad + tap * 4 + c] += tsum;
  }
}

// ------------------------------------------------------------------------------------------------ fp16x3 form
// The same weight gradient on v_mfma_f32_16x16x32_f16 with split products (the fp16x3 precision modes: WgradArgs::prec == 2, dY
// scaled by the power of two its maximum — WgradArgs::xmax, from bn_bwd_apply — calls for).  The fp32 kernel above is bound by its
// MFMAs (160 of 32 cycles per tile and wave = 136 of its 201 us) at the very end of the step, where nothing overlaps it.  Here
// one MFMA reduces 32 output pixels (a tile row), and both operands are read with the transposing LDS load ds_read_b64_tr_b16 out
// of pixel-major images: the A operand from dY [64 px][hi 64 co | lo 64 co] (the swizzle of wgrad_f16x3.hip), the B operand
// straight out of the input patch [px][4 channels] (8 bytes per pixel: one lane address per (pixel, tap) — the stride-2 gather
// and the tap shift are just addresses).  Columns j = tap*4 + c (196 -> 13 blocks of four taps; the stored zero channel rides
// along), which IS dW's layout.  84 MFMAs of 16 cycles per tile and wave.
typedef _Float16 s_h8 __attribute__((ext_vector_type(8)));
typedef __fp16 s_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) s_fp16x4 s_lds_fp16x4;
__device__ __forceinline__ s_h8 s_tr_pair(const char* base, int o0, int o1) {      // two transposed reads -> one 8-half operand fragment
  const s_fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((s_lds_fp16x4*)(uintptr_t)(base + o0));
  const s_fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((s_lds_fp16x4*)(uintptr_t)(base + o1));
  typedef __fp16 fp16x8 __attribute__((__vector_size__(8 * sizeof(__fp16))));
  const fp16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(s_h8, v);
}
constexpr int kHPix = kSPH * kSPW;                         // 621 patch pixels
constexpr int kHPlane = 5120;                              // bytes of one half-plane of the patch: 621 x 8 + zero bytes (the taps past 48), two planes = a multiple of 256
constexpr int kHDy = kSR * kSC * 256;                      // bytes of the dY image: [64 px][hi 128 B | lo 128 B]
constexpr int kHBuf = 2 * kHPlane + kHDy;                  // bytes per stage buffer (26 624)
static_assert(kHPix * 8 + 128 <= kHPlane, "zero bytes behind the patch plane");
constexpr int kHCols = 208;                                // 13 blocks of 16 columns: j = tap*4 + c, 196 real
constexpr int kHPart = 64 * kHCols;
__device__ __forceinline__ int s_off_dy(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__global__ __launch_bounds__(256, 3) void wgrad_stem_f16_kernel(const WgradArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(256))) char hsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 1, wb = wave & 1;                // 32 output channels x 7 (6) column blocks
  const int H = a.s0.H, W = a.s0.W;
  const int tilesW = a.Wo / kSC, tilesH = a.Ho / kSR;
  float xs = 1.f;
  if (a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }

  // ---- fragment addresses (bytes): lane = (k-group kg, row-in-group q, column quad p)
  const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  int boff[7];                                            // hi plane, k-step 0, first read (rows 8kg + q); second read: + 4 pixels = + 64 B
#pragma unroll
  for (int jb = 0; jb < 7; ++jb) {
    const int tap = (wb * 7 + jb) * 4 + p;
    const int r = tap / 7, s = tap - r * 7;
    boff[jb] = tap < 49 ? ((r * kSPW + s + 2 * (8 * kg + q)) * 8) : (kHPix * 8);      // taps past the filter: the zero bytes behind the plane
  }
  const int bstep = 8 * 8;                                // + 4 output pixels = + 8 patch pixels
  int dyo[2][2];
#pragma unroll
  for (int cf = 0; cf < 2; ++cf)
#pragma unroll
    for (int rdx = 0; rdx < 2; ++rdx) dyo[cf][rdx] = s_off_dy(8 * kg + 4 * rdx + q, 2 * (2 * wa + cf) + (p >> 1)) + 8 * (p & 1);

  // ---- staging geometry: patch 621 pixels (3 rounds), dY 64 px x 16 quads = 1024 units (4 rounds)
  int spy[3], spx[3];
#pragma unroll
  for (int rd = 0; rd < 3; ++rd) {
    const int pp = min(rd * 256 + tid, kHPix - 1);
    spy[rd] = pp / kSPW; spx[rd] = pp - spy[rd] * kSPW;
  }
  const bool last_live = (2 * 256 + tid) < kHPix;
  int dlo[4], dgo[4];
#pragma unroll
  for (int rd = 0; rd < 4; ++rd) {
    const int u = rd * 256 + tid, px = u >> 4, cq = u & 15;
    dlo[rd] = s_off_dy(px, cq >> 1) + (cq & 1) * 8;
    dgo[rd] = ((px >> 5) * a.Wo + (px & 31)) * 64 + cq * 4;
  }
  f4 pv[3], dv[4];
  auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    oy0 = th * kSR; ox0 = tw * kSC;
  };
  auto stage_load = [&](int t) {
    int n, oy0, ox0; tile_origin(t, n, oy0, ox0);
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
      const int iy = 2 * oy0 - 3 + spy[rd], ix = 2 * ox0 - 3 + spx[rd];
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int yc = min(max(iy, 0), H - 1), xc = min(max(ix, 0), W - 1);
      const f4 v = *(const f4*)(a.s0.ptr + (((size_t)n * H + yc) * W + xc) * 4);
      pv[rd] = ok ? v : (f4){0.f, 0.f, 0.f, 0.f};
    }
    const float* const db = a.dy + (((size_t)n * a.Ho + oy0) * a.Wo + ox0) * 64;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) dv[rd] = *(const f4*)(db + dgo[rd]);
  };
  auto stage_store = [&](int buf) {
    char* const b_ = hsm + buf * kHBuf;
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
      uwm_u2 hi, lo;
      uwm_split4(__builtin_amdgcn_fmed3f(pv[rd].x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(pv[rd].y, -65504.f, 65504.f),
                 __builtin_amdgcn_fmed3f(pv[rd].z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(pv[rd].w, -65504.f, 65504.f), hi, lo);
      if (rd < 2 || last_live) { *(uwm_u2*)(b_ + (rd * 256 + tid) * 8) = hi; *(uwm_u2*)(b_ + kHPlane + (rd * 256 + tid) * 8) = lo; }
    }
    char* const d_ = b_ + 2 * kHPlane;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const f4 v = dv[rd] * xs;                           // (below 2^14 by construction: no clamp)
      uwm_u2 hi, lo;
      uwm_split4(v.x, v.y, v.z, v.w, hi, lo);
      *(uwm_u2*)(d_ + dlo[rd]) = hi; *(uwm_u2*)(d_ + (dlo[rd] ^ 128)) = lo;
    }
  };
  if (tid < 64) {                                         // the zero bytes behind both half-planes of both buffers
    const int b = tid >> 5, pl = (tid >> 4) & 1, w = tid & 15;
    *(uwm_u2*)(hsm + b * kHBuf + pl * kHPlane + kHPix * 8 + w * 8) = (uwm_u2){0u, 0u};
  }

  f4 acc[2][7];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  const int nb = wb == 0 ? 7 : 6;                         // 13 column blocks

  int t = blockIdx.x;
  if (t < ntiles) { stage_load(t); stage_store(0); }
  __syncthreads();

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    const bool more = tn < ntiles;
    if (more) stage_load(tn);
    const char* const ps = hsm + cur * kHBuf;
    const char* const ds = ps + 2 * kHPlane;
#pragma unroll
    for (int kr = 0; kr < kSR; ++kr) {                     // k-step = the 32 output pixels of tile row kr
      s_h8 ah[2], al[2];
#pragma unroll
      for (int cf = 0; cf < 2; ++cf) {
        ah[cf] = s_tr_pair(ds + kr * 32 * 256, dyo[cf][0], dyo[cf][1]);
        al[cf] = s_tr_pair(ds + kr * 32 * 256, dyo[cf][0] ^ 128, dyo[cf][1] ^ 128);
      }
      const char* const pk = ps + kr * 2 * kSPW * 8;
#pragma unroll
      for (int jb = 0; jb < 7; ++jb) {
        if (jb < nb) {
          const bool z = boff[jb] == kHPix * 8;           // (zero slot: no row / pixel offsets)
          const char* const pb = z ? ps : pk;
          const int st2 = z ? 0 : bstep;
          const s_h8 bh = s_tr_pair(pb, boff[jb], boff[jb] + st2);
          const s_h8 bl = s_tr_pair(pb + kHPlane, boff[jb], boff[jb] + st2);
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) acc[cf][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cf], bl, acc[cf][jb], 0, 0, 0);
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) acc[cf][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cf], bh, acc[cf][jb], 0, 0, 0);
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) acc[cf][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cf], bh, acc[cf][jb], 0, 0, 0);
        }
      }
    }
    if (more) stage_store(cur ^ 1);
    __syncthreads();
  }

  // partial [64 co][208 cols] of this workgroup: lane holds rows 4*(lane >> 4) + e, column lane & 15 of each block
  float* const o = a.part + (size_t)blockIdx.x * kHPart;
  const float ixs = 1.f / xs;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j < nb) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[((wa * 2 + i) * 16 + (lane >> 4) * 4 + e) * kHCols + (wb * 7 + j) * 16 + (lane & 15)] = acc[i][j][e] * ixs;
      }
}

// dw[co][j] += sum_wg part[wg][co][j], j = tap*4 + c < 196: workgroup = 8 outputs x 32 partial groups, fixed order
__global__ __launch_bounds__(256) void wgrad_stem_f16_reduce_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw, int wrows, int Kpad) {
  __shared__ float red[32][8];
  const int u8 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int ou = blockIdx.x * 8 + u8;                      // 0 .. 64*196 - 1
  const int co = ou / 196, j = ou - co * 196;
  float s = 0.f;
  if (co < 64)
    for (int k = grp; k < nparts; k += 32) s += part[(size_t)k * kHPart + co * kHCols + j];
  red[grp][u8] = s;
  __syncthreads();
  if (grp == 0 && co < wrows) {
    float tsum = red[0][u8];
#pragma unroll
    for (int g = 1; g < 32; ++g) tsum += red[g][u8];
    dw[(size_t)co * Kpad + j] += tsum;
  }
}

bool wgrad_stem_applicable(const WgradArgs& a) {
  static const bool off = dbg_flag("UWM_NO_WGRAD_STEM");
  return !off && a.ntaps == 49 && a.kw == 7 && a.stride == 2 && a.pad == 3 && a.s0.up == 0 && a.C0 == a.Ctot && a.Ctot == 4 && a.s0.C == 4 &&
         a.Cout == 64 && a.wrows <= 64 && a.Kpad >= 196 && a.s0.scale == nullptr && 2 * a.Ho == a.s0.H && 2 * a.Wo == a.s0.W &&
         (a.Ho % kSR) == 0 && (a.Wo % kSC) == 0;
}

hipError_t launch_wgrad_stem(const WgradArgs& a0, hipStream_t st) {
  if (!wgrad_stem_applicable(a0)) return hipErrorInvalidValue;
  WgradArgs a = a0;
  const int ntiles = a.N * (a.Ho / kSR) * (a.Wo / kSC);
  const int slots = 3 * device_cu_count();                 // 134 VGPRs, 52.6 KB of LDS: three workgroups per CU
  const int nwg = ntiles < slots ? ntiles : slots;
  const size_t need = (size_t)nwg * kSPart;
  if (!a.part || a.part_floats < need) { a.part = wgrad_op_scratch(); a.part_floats = wgrad_wino_scratch_floats(); }
  if (!a.part || a.part_floats < need) return hipErrorOutOfMemory;
  if (a.prec == 2 && a.xmax) {                             // fp16x3 precision modes
    const size_t need16 = (size_t)nwg * kHPart;
    if (a.part_floats < need16) { a.part = wgrad_op_scratch(); a.part_floats = wgrad_wino_scratch_floats(); }
    if (!a.part || a.part_floats < need16) return hipErrorOutOfMemory;
    const size_t lds16 = (size_t)2 * kHBuf;
    static DevOnce lds_attr16;
    { hipError_t e = lds_attr16.set_max_lds((const void*)wgrad_stem_f16_kernel, lds16); if (e != hipSuccess) return e; }
    UWM_LAUNCH(50, a.flops, a.bytes, wgrad_stem_f16_kernel, dim3((unsigned)nwg), dim3(256), lds16, st, a, ntiles);
    hipLaunchKernelGGL(wgrad_stem_f16_reduce_kernel, dim3((64 * 196 + 7) / 8), dim3(256), 0, st, (const float*)a.part, nwg, a.dw, a.wrows, a.Kpad);
    return hipGetLastError();
  }
  const size_t lds = (size_t)2 * kSBuf * sizeof(float);
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_stem_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(41, a.flops, a.bytes, wgrad_stem_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
  hipLaunchKernelGGL(wgrad_stem_reduce_kernel, dim3((64 * 147 + 7) / 8), dim3(256), 0, st, (const float*)a.part, nwg, a.dw, a.wrows, a.Kpad);
  return hipGetLastError();
}

}  // namespace uwm
