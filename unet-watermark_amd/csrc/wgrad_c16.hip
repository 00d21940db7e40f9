// Weight gradients of the 16-channel full-resolution layers (decoder block 4 conv2: 16 -> 16, and the segmentation head:
// 16 -> <= 4 classes) for gfx950.  K = 9*16 = 144 is too short for the general kernels: wgrad_patch_kernel<16,16> ran these
// at 19 TF (12 % of the fp32-MFMA peak AND 21 % of HBM: 540 us per launch, 1.08 ms per step).  Here the pixel dimension is
// the MFMA reduction (k = 4 consecutive pixels of a row) and the whole 16 x 9 x 16 result lives in 9 accumulators per wave:
//
//   wgrad_c16_kernel   dW[co][tap][ci] = sum_p dY[p][co] * x~[p + tap - 1][ci]          x~ = lazy BatchNorm + ReLU, zero padded
//       tile = 8 rows x 32 columns of one image: dY tile [8][32][16] by LDS-DMA (2-KB rows, straight copy), x~ halo patch
//       [10][34][16] through registers (lazy transform applied).  Wave w takes rows 2w, 2w+1: per 4 pixels ONE read of dY
//       (MFMA A: lane = (co, pixel)) and nine shifted reads of x~ (MFMA B: lane = (ci, pixel)), 9 MFMAs.  19.3 GFLOP per
//       launch at 16 x 512^2 = 123 us at the fp32-MFMA peak against 107 us for its 536 MB at 5 TB/s: balanced.
//   wgrad_head_kernel  dW[cls][tap][ci] = sum_q x~[q][ci] * dY[q - (tap - 1)][cls]       (the same sum re-indexed by q = p + tap - 1)
//       with ONE live class the co dimension is 1: the taps take the MFMA N dimension instead (B: lane = (tap, pixel) reads the
//       shifted dY plane, zero for tap >= 9), A = x~ unshifted, one MFMA per 4 pixels per class: HBM-bound (335 MB, ~70 us).
//
// Workgroups walk tiles grid-stride with single-buffered LDS (38 KB: three workgroups per CU overlap each other's staging)
// and keep their sums in registers; at the end the four waves are added through LDS and ONE partial tile per workgroup is
// stored — wgrad_c16_reduce_kernel adds the partials in workgroup order: no atomics, bit-reproducible.
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kRows = 8, kCols = 32;                         // tile
constexpr int kPR = kRows + 2, kPC = kCols + 2;              // halo patch
constexpr int kC = 16;
constexpr int kTileF = kRows * kCols * kC;                   // 4096 floats (16 KB)
constexpr int kPatchF = kPR * kPC * kC;                      // 5440 floats (21.8 KB)
constexpr int kMaxWG = 768;                                  // 3 per CU

struct C16Geo { int tilesW, tilesH, ntiles; };

// x~ staging shared by both kernels: `rows` x `cols` pixels starting at (h0 + dh, w0 + dw) of image n -> dst [rows][cols][16],
// lazy affine + ReLU applied, zero outside the image
template <int ROWS, int COLS>
__device__ __forceinline__ void stage_x(const Src& s, int n, int hs, int ws, int H, int W, float* dst, int tid) {
  const int chu = tid & 3;
  f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  const bool has = s.scale != nullptr;
  if (has) { sc = *(const f4*)(s.scale + chu * 4); sh = *(const f4*)(s.shift + chu * 4); }
  constexpr int UNITS = ROWS * COLS * 4, RND = (UNITS + 255) / 256;
  f4 v[RND]; bool ok[RND];
#pragma unroll
  for (int rd = 0; rd < RND; ++rd) {
    const int u = rd * 256 + tid, pp = u >> 2;
    const int py = pp / COLS, px = pp - py * COLS;
    const int hh = hs + py, ww = ws + px;
    ok[rd] = u < UNITS && hh >= 0 && hh < H && ww >= 0 && ww < W;
    const int hc = min(max(hh, 0), H - 1), wc = min(max(ww, 0), W - 1);
    v[rd] = *(const f4*)(s.ptr + (((size_t)n * H + hc) * W + wc) * kC + chu * 4);
  }
#pragma unroll
  for (int rd = 0; rd < RND; ++rd) {
    const int u = rd * 256 + tid;
    f4 t = v[rd];
    if (has) {
      t = t * sc + sh;
      if (s.relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
    }
    if (!ok[rd]) t = (f4){0.f, 0.f, 0.f, 0.f};
    if (u < UNITS) *(f4*)(dst + u * 4) = t;
  }
}

__global__ __launch_bounds__(256, 3) void wgrad_c16_kernel(const WgradArgs a, const C16Geo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ys = smem;                    // [8][32][16]  dY tile (LDS-DMA, lane-linear)
  float* const Xs = smem + kTileF;           // [10][34][16] x~ halo patch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  f4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    const int tw = tile % g.tilesW, q = tile / g.tilesW;
    const int th = q % g.tilesH, n = q / g.tilesH;
    const int h0 = th * kRows, w0 = tw * kCols;
    // dY tile: row r of the tile = 32 px x 16 ch = 2 KB contiguous in HBM; 16 wave-instructions of 1 KB, 4 per wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = i * 4 + wave;                       // (row, half) = (piece >> 1, piece & 1)
      const float* src = a.dy + (((size_t)n * a.Ho + h0 + (piece >> 1)) * a.Wo + w0 + (piece & 1) * 16) * kC + lane * 4;
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(uintptr_t)(Ys + piece * 256), 16, 0, 0);
    }
    stage_x<kPR, kPC>(a.s0, n, h0 - 1, w0 - 1, a.Ho, a.Wo, Xs, tid);
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
#pragma unroll 2
      for (int gq = 0; gq < kCols / 4; ++gq) {
        const int col = gq * 4 + lq;                        // this lane's pixel (MFMA k = lq)
        const float av = Ys[(row * kCols + col) * kC + li];
        const float* xb = Xs + (row * kPC + col) * kC + li;
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
          for (int s3 = 0; s3 < 3; ++s3)
            acc[r3 * 3 + s3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xb[(r3 * kPC + s3) * kC], acc[r3 * 3 + s3], 0, 0, 0);
      }
    }
    __syncthreads();                                        // everyone is done with the tile before it is overwritten
  }
  // ---- the four waves' sums -> LDS -> one partial tile [16 co][Kpad] per workgroup (pad columns written as zeros)
  float* const red = smem;                                  // [4][9][16 co][16 ci]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[((wave * 9 + t) * 16 + lq * 4 + e) * 16 + li] = acc[t][e];
  __syncthreads();
  float* const dst = a.part + (size_t)blockIdx.x * a.wrows * a.Kpad;
  const int co = tid >> 4, ci = tid & 15;
  if (co < a.wrows) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float s = (red[((0 * 9 + t) * 16 + co) * 16 + ci] + red[((1 * 9 + t) * 16 + co) * 16 + ci]) +
                      (red[((2 * 9 + t) * 16 + co) * 16 + ci] + red[((3 * 9 + t) * 16 + co) * 16 + ci]);
      dst[(size_t)co * a.Kpad + t * kC + ci] = s;
    }
    for (int k = 9 * kC + ci; k < a.Kpad; k += 16) dst[(size_t)co * a.Kpad + k] = 0.f;
  }
}

// head: <= 4 classes (dY [px][4], a.wrows live classes), NCLS accumulators
template <int NCLS>
__global__ __launch_bounds__(256, 3) void wgrad_head_kernel(const WgradArgs a, const C16Geo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Xs = smem;                    // [8][32][16]  x~ tile (no halo)
  float* const Ds = smem + kTileF;           // [NCLS][10][34] dY halo planes
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const bool tap_ok = li < 9;
  const int r3 = li / 3, s3 = li - r3 * 3;
  f4 acc[NCLS];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    const int tw = tile % g.tilesW, q = tile / g.tilesW;
    const int th = q % g.tilesH, n = q / g.tilesH;
    const int h0 = th * kRows, w0 = tw * kCols;
    stage_x<kRows, kCols>(a.s0, n, h0, w0, a.Ho, a.Wo, Xs, tid);
    for (int u = tid; u < kPR * kPC; u += 256) {            // dY halo: 340 pixels, one 16-byte load each (4 padded classes)
      const int py = u / kPC, px = u - py * kPC;
      const int hh = h0 - 1 + py, ww = w0 - 1 + px;
      f4 d = {0.f, 0.f, 0.f, 0.f};
      if (hh >= 0 && hh < a.Ho && ww >= 0 && ww < a.Wo) d = *(const f4*)(a.dy + (((size_t)n * a.Ho + hh) * a.Wo + ww) * 4);
#pragma unroll
      for (int c = 0; c < NCLS; ++c) Ds[c * kPR * kPC + u] = d[c];
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
#pragma unroll 2
      for (int gq = 0; gq < kCols / 4; ++gq) {
        const int col = gq * 4 + lq;
        const float av = Xs[(row * kCols + col) * kC + li];              // A: (ci = li, pixel)
        // B: (tap = li, pixel): dY at q - (tap - 1) = halo coordinates (row + 1 - (r3 - 1), col + 1 - (s3 - 1))
        const int doff = (row + 2 - r3) * kPC + col + 2 - s3;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
          const float bv = tap_ok ? Ds[c * kPR * kPC + doff] : 0.f;
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  float* const red = smem;                                  // [4][NCLS][16 ci][16 tap]
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[((wave * NCLS + c) * 16 + lq * 4 + e) * 16 + li] = acc[c][e];
  __syncthreads();
  float* const dst = a.part + (size_t)blockIdx.x * a.wrows * a.Kpad;
  for (int i = tid; i < a.wrows * a.Kpad; i += 256) {
    const int cls = i / a.Kpad, k = i - cls * a.Kpad;
    float s = 0.f;
    if (k < 9 * kC && cls < NCLS) {
      const int t = k >> 4, ci = k & 15;
      s = (red[((0 * NCLS + cls) * 16 + ci) * 16 + t] + red[((1 * NCLS + cls) * 16 + ci) * 16 + t]) +
          (red[((2 * NCLS + cls) * 16 + ci) * 16 + t] + red[((3 * NCLS + cls) * 16 + ci) * 16 + t]);
    }
    dst[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------ fp16x3 form of wgrad_c16_kernel
// (the fp16x3 precision modes: WgradArgs::prec == 2, dY scaled through WgradArgs::xmax).  The fp32 kernel is bound by its MFMAs
// (9 of 32 cycles per 4 pixels: 123 of its 204 us); here one v_mfma_f32_16x16x32_f16 k-step = the 32 pixels of a tile row, three
// split products per tap (27 MFMAs of 16 cycles per row), both operands through the transposing LDS load ds_read_b64_tr_b16 out of
// pixel-major images [px][hi 16 ch | lo 16 ch] (64 B per pixel) — a tap shift is just another row address.  Persistent workgroups,
// the stage (dY tile + x~ halo patch, split while staging) double-buffered: the next tile's global loads are in flight behind this
// tile's MFMAs, one barrier per tile.
typedef _Float16 c_h8 __attribute__((ext_vector_type(8)));
typedef __fp16 c_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) c_fp16x4 c_lds_fp16x4;
__device__ __forceinline__ c_h8 c_tr_pair(const char* base, int o0, int o1) {
  const c_fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((c_lds_fp16x4*)(uintptr_t)(base + o0));
  const c_fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((c_lds_fp16x4*)(uintptr_t)(base + o1));
  typedef __fp16 fp16x8 __attribute__((__vector_size__(8 * sizeof(__fp16))));
  const fp16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(c_h8, v);
}
constexpr int kFDy = kRows * kCols * 64;                     // bytes of the dY image (16 KB)
constexpr int kFX = kPR * kPC * 64;                          // bytes of the x~ image (21 760)
constexpr int kFBuf = kFDy + kFX;

__global__ __launch_bounds__(256, 2) void wgrad_c16_f16_kernel(const WgradArgs a, const C16Geo g) {
  extern __shared__ __attribute__((aligned(256))) char fsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.Ho, W = a.Wo;
  float xs = 1.f;
  if (a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }
  // ---- staging geometry: thread = (pixel, channel quad); x~ 340 px x 4 = 1360 units (6 rounds), dY 256 px x 4 = 1024 units (4 rounds)
  const int chu = tid & 3;
  f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  const bool has = a.s0.scale != nullptr;
  if (has) { sc = *(const f4*)(a.s0.scale + chu * 4); sh = *(const f4*)(a.s0.shift + chu * 4); }
  const float vlo = (has && a.s0.relu) ? 0.f : -65504.f;
  int xpy[6], xpx[6];
#pragma unroll
  for (int rd = 0; rd < 6; ++rd) {
    const int pp = min((rd * 256 + tid) >> 2, kPR * kPC - 1);
    xpy[rd] = pp / kPC; xpx[rd] = pp - xpy[rd] * kPC;
  }
  const bool xlast = (5 * 256 + tid) < kPR * kPC * 4;
  f4 xv[6], dv[4]; unsigned xok = 0;
  auto tile_origin = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % g.tilesW; t /= g.tilesW;
    const int th = t % g.tilesH; n = t / g.tilesH;
    h0 = th * kRows; w0 = tw * kCols;
  };
  auto stage_load = [&](int t) {
    int n, h0, w0; tile_origin(t, n, h0, w0);
    xok = 0;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const int hh = h0 - 1 + xpy[rd], ww = w0 - 1 + xpx[rd];
      const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
      const int hc = min(max(hh, 0), H - 1), wc = min(max(ww, 0), W - 1);
      xv[rd] = *(const f4*)(a.s0.ptr + (((size_t)n * H + hc) * W + wc) * kC + chu * 4);
      xok |= (ok ? 1u : 0u) << rd;
    }
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const int px = (rd * 256 + tid) >> 2;
      dv[rd] = *(const f4*)(a.dy + (((size_t)n * H + h0 + (px >> 5)) * W + w0 + (px & 31)) * kC + chu * 4);
    }
  };
  auto stage_store = [&](int buf) {
    char* const yb = fsm + buf * kFBuf;
    char* const xb = yb + kFDy;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      f4 v = xv[rd];
      if (has) v = v * sc + sh;
      const bool ok = (xok >> rd) & 1u;
      const float top = ok ? 65504.f : vlo;
      v.x = __builtin_amdgcn_fmed3f(v.x, vlo, top); v.y = __builtin_amdgcn_fmed3f(v.y, vlo, top);
      v.z = __builtin_amdgcn_fmed3f(v.z, vlo, top); v.w = __builtin_amdgcn_fmed3f(v.w, vlo, top);
      if (vlo != 0.f && !ok) v = (f4){0.f, 0.f, 0.f, 0.f};
      uwm_u2 hi, lo;
      uwm_split4(v.x, v.y, v.z, v.w, hi, lo);
      const int u = rd * 256 + tid;
      if (rd < 5 || xlast) { *(uwm_u2*)(xb + (u >> 2) * 64 + chu * 8) = hi; *(uwm_u2*)(xb + (u >> 2) * 64 + 32 + chu * 8) = lo; }
    }
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const f4 v = dv[rd] * xs;                           // (below 2^14 by construction: no clamp)
      uwm_u2 hi, lo;
      uwm_split4(v.x, v.y, v.z, v.w, hi, lo);
      const int px = (rd * 256 + tid) >> 2;
      *(uwm_u2*)(yb + px * 64 + chu * 8) = hi; *(uwm_u2*)(yb + px * 64 + 32 + chu * 8) = lo;
    }
  };
  // ---- fragment addresses: lane = (k-group kg, row-in-group q, channel quad p): pixel 8 kg + q (+4) of the k-step's row
  const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int fo = (8 * kg + q) * 64 + p * 8;

  f4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};

  int t = blockIdx.x;
  if (t < g.ntiles) { stage_load(t); stage_store(0); }
  __syncthreads();
  for (int it = 0; t < g.ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    const bool more = tn < g.ntiles;
    if (more) stage_load(tn);
    const char* const yb = fsm + cur * kFBuf;
    const char* const xb = yb + kFDy;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
      const char* const ya = yb + row * kCols * 64 + fo;
      const c_h8 ah = c_tr_pair(ya, 0, 256), al = c_tr_pair(ya, 32, 256 + 32);
#pragma unroll
      for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          const char* const xa = xb + ((row + r3) * kPC + s3) * 64 + fo;
          const c_h8 bh = c_tr_pair(xa, 0, 256), bl = c_tr_pair(xa, 32, 256 + 32);
          f4 c = acc[r3 * 3 + s3];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
          acc[r3 * 3 + s3] = c;
        }
    }
    if (more) stage_store(cur ^ 1);
    __syncthreads();
  }
  // ---- the four waves' sums -> LDS -> one partial tile [16 co][Kpad] per workgroup (pad columns written as zeros)
  float* const red = (float*)fsm;                           // [4][9][16 co][16 ci]
  const int li = lane & 15, lq = lane >> 4;
  const float ixs = 1.f / xs;
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[((wave * 9 + t9) * 16 + lq * 4 + e) * 16 + li] = acc[t9][e] * ixs;
  __syncthreads();
  float* const dst = a.part + (size_t)blockIdx.x * a.wrows * a.Kpad;
  const int co = tid >> 4, ci = tid & 15;
  if (co < a.wrows) {
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const float s = (red[((0 * 9 + t9) * 16 + co) * 16 + ci] + red[((1 * 9 + t9) * 16 + co) * 16 + ci]) +
                      (red[((2 * 9 + t9) * 16 + co) * 16 + ci] + red[((3 * 9 + t9) * 16 + co) * 16 + ci]);
      dst[(size_t)co * a.Kpad + t9 * kC + ci] = s;
    }
    for (int k = 9 * kC + ci; k < a.Kpad; k += 16) dst[(size_t)co * a.Kpad + k] = 0.f;
  }
}

// dw[i] += part[0][i] + part[1][i] + ... (16-byte units).  Workgroup = 8 units x 32 partial groups: thread (unit, group) adds
// partials group, group + 32, ... in order, the 32 group sums are combined in group order through LDS (fixed association:
// bit-reproducible); hundreds of partial tiles of only 10 KB each, so the parallelism has to come from the partial index
__global__ __launch_bounds__(256) void wgrad_c16_reduce_kernel(const float* __restrict__ part, int nparts, int n4, float* __restrict__ dw) {
  __shared__ f4 red[32][8];
  const int u = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + u;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4)
    for (int k = grp; k < nparts; k += 32) s += *(const f4*)(part + ((size_t)k * n4 + i) * 4);
  red[grp][u] = s;
  __syncthreads();
  if (grp == 0 && i < n4) {
    f4 t = red[0][u];
#pragma unroll
    for (int g = 1; g < 32; ++g) t += red[g][u];
    *(f4*)(dw + i * 4) += t;
  }
}

bool wgrad_c16_applicable(const WgradArgs& a) {
  return a.ntaps == 9 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Ctot == kC && a.C0 == kC && a.s0.C == kC && a.s0.up == 0 &&
         a.Hl == a.Ho && a.Wl == a.Wo && (a.Ho % kRows) == 0 && (a.Wo % kCols) == 0 && a.Kpad >= 9 * kC &&
         ((a.Cout == kC && a.wrows == kC) || (a.Cout == 4 && a.wrows >= 1 && a.wrows <= 4));
}

hipError_t launch_wgrad_c16(const WgradArgs& a0, hipStream_t st) {
  if (!wgrad_c16_applicable(a0)) return hipErrorInvalidValue;
  WgradArgs a = a0;
  C16Geo g; g.tilesW = a.Wo / kCols; g.tilesH = a.Ho / kRows; g.ntiles = a.N * g.tilesH * g.tilesW;
  int nwg = device_cu_count() * 3; if (nwg > kMaxWG) nwg = kMaxWG; if (nwg > g.ntiles) nwg = g.ntiles;
  const size_t need = (size_t)nwg * a.wrows * a.Kpad;
  if (!a.part || a.part_floats < need) { a.part = wgrad_op_scratch(); a.part_floats = wgrad_wino_scratch_floats(); }     // single-operator entry points
  if (!a.part || a.part_floats < need) return hipErrorOutOfMemory;
  const bool head = a.Cout == 4;
  if (!head && a.prec == 2 && a.xmax) {                     // fp16x3 precision modes: two workgroups per CU (75.5 KB of LDS each)
    int nwg2 = device_cu_count() * 2; if (nwg2 > g.ntiles) nwg2 = g.ntiles;
    const size_t lds16 = (size_t)2 * kFBuf;
    static DevOnce lds_attr16;
    { hipError_t e = lds_attr16.set_max_lds((const void*)wgrad_c16_f16_kernel, lds16); if (e != hipSuccess) return e; }
    UWM_LAUNCH(49, a.flops, a.bytes, wgrad_c16_f16_kernel, dim3((unsigned)nwg2), dim3(256), lds16, st, a, g);
    const int n4h = a.wrows * a.Kpad / 4;
    hipLaunchKernelGGL(wgrad_c16_reduce_kernel, dim3((unsigned)((n4h + 7) / 8)), dim3(256), 0, st, (const float*)a.part, nwg2, n4h, a.dw);
    return hipGetLastError();
  }
  const size_t lds = (size_t)(kTileF + kPatchF) * sizeof(float);      // (the head's dY planes and the final [4][9][256] sums fit inside)
  if (!head) {
    UWM_LAUNCH(32, a.flops, a.bytes, wgrad_c16_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, g);
  } else if (a.wrows == 1) {
    UWM_LAUNCH(33, a.flops, a.bytes, (wgrad_head_kernel<1>), dim3((unsigned)nwg), dim3(256), lds, st, a, g);
  } else {
    UWM_LAUNCH(33, a.flops, a.bytes, (wgrad_head_kernel<4>), dim3((unsigned)nwg), dim3(256), lds, st, a, g);
  }
  const int n4 = a.wrows * a.Kpad / 4;
  hipLaunchKernelGGL(wgrad_c16_reduce_kernel, dim3((unsigned)((n4 + 7) / 8)), dim3(256), 0, st, (const float*)a.part, nwg, n4, a.dw);
  return hipGetLastError();
}

}  // namespace uwm
