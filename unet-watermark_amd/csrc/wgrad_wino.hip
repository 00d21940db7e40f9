// Weight gradient of a 3x3 / stride 1 / pad 1 convolution through the Winograd F(2x2,3x3) domain, for
// gfx950 on v_mfma_f32_16x16x4_f32:
//
//   Z[xi][co][c] = sum over 2x2 tiles  (A dY A^T)[xi][tile][co] * (B^T d B)[xi][tile][c]     (16 GEMMs, K = tiles)
//   dW[co][c]    = G^T Z[.][co][c] G                                                          (4x4 -> 3x3)
//
// 16 multiplies per tile instead of 36 (9 taps x 4 pixels): 2.25x fewer MFMA FLOPs than wgrad_patch.hip.
//
// A workgroup (256 threads) owns TA output channels x CW = 32 input channels and walks a range of
// 4x16-pixel stages (16 tiles each).  Per stage the dY block [64 px][TA] arrives by LDS-DMA
// (global_load_lds, channel-block swizzle applied on the SOURCE address) and the 6x18-pixel input patch
// through registers (lazy BatchNorm+ReLU, nearest x2 upsample, concat, zero padding), both double
// buffered: one barrier per stage.  Wave i owns row i of the 4x4 Winograd domain: each lane builds the
// (A dY A^T)[i][0..3] values of ITS (tile, co) and the (B^T d B)[i][0..3] values of ITS (tile, c) in
// registers from 4-byte LDS reads — MFMA k index = tile — and feeds 4 xi x (TA/16) x 2 MFMAs per 4 tiles.
// Epilogue: per-wave column transform (Z G), cross-wave row transform (G^T .) through LDS in three
// ordered phases (plain read-add-write, no LDS atomics), then coalesced stores.
//
// Round 2: TA = 64 | 32 run on wgrad_wino2_kernel below (a wave owns output-channel blocks x one input-channel block at
// ALL 16 Winograd points: 25 % fewer LDS operand reads per MFMA, the whole G^T Z G epilogue in registers, stage-boundary
// loads folded into the MFMA blocks); the row-per-wave kernel above it stays for TA = 16.  Neither uses global atomics
// any more: a workgroup stores ONE partial dW tile per pixel split and wgrad_reduce_kernel adds the splits in order —
// the weight gradient is bit-reproducible from run to run (float atomics were not), and the write traffic is the same
// bytes as plain stores instead of 4-byte atomic requests.  Blocks of one pixel split are dealt to ONE XCD, so the dY /
// input range they all stream is fetched from HBM once per XCD L2 instead of once per block.
#include "uwm_kernels.h"
#include <cstdlib>

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kSH = 4, kSW = 16, kPW = kSW + 2, kPP = (kSH + 2) * kPW;    // 108 patch pixels per stage
constexpr int kCW = 32;
constexpr int kPUnits = kPP * (kCW / 4);                                  // 864 16-byte units
constexpr int kPRounds = (kPUnits + 255) / 256;                           // 4

template <int TA>
__global__ __launch_bounds__(256, 2) void wgrad_wino_kernel(const WgradArgs a) {
  constexpr int CA = TA / 16;              // output-channel MFMA tiles
  constexpr int UPP = TA / 4;              // 16-byte units per dY pixel
  constexpr int YI = 64 * UPP / 64 / 4;    // LDS-DMA instructions per wave per stage (TA=64: 4)
  constexpr int kYs = 64 * TA;             // floats per dY buffer
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ys = smem;                        // [2][64 px][TA]       (16-channel blocks swizzled by tile parity)
  float* const Ps = smem + 2 * kYs;              // [2][108 px][32]      (same swizzle)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;

  const int nchunk = a.Ctot / kCW;
  const int tilesA = a.Cout / TA;
  const int tilesW = a.Wo / kSW, tilesH = a.Ho / kSH;
  const int nstages = a.N * tilesH * tilesW;
  int b = blockIdx.x;
  const int cc = b % nchunk; b /= nchunk;
  const int ta = b % tilesA; const int split = b / tilesA;
  const int a0 = ta * TA;
  const int t0 = split * a.msplit, t1 = min(nstages, t0 + a.msplit);

  // ---- patch loader constants (channel unit fixed per thread: 256 % 8 == 0)
  const int chu = tid & 7;
  const int c = cc * kCW + chu * 4;
  const bool first = c < a.C0;
  const float* sp = first ? a.s0.ptr : a.s1.ptr;
  const float* ssc = first ? a.s0.scale : a.s1.scale;
  const float* ssh = first ? a.s0.shift : a.s1.shift;
  const int sC = first ? a.s0.C : a.s1.C, sH = first ? a.s0.H : a.s1.H, sW = first ? a.s0.W : a.s1.W;
  const int sup = first ? a.s0.up : a.s1.up;
  const int trelu = first ? a.s0.relu : a.s1.relu;
  const int cl = first ? c : c - a.C0;
  const bool thas = ssc != nullptr;
  f4 tsc = {1.f, 1.f, 1.f, 1.f}, tsh = {0.f, 0.f, 0.f, 0.f};
  if (thas) { tsc = *(const f4*)(ssc + cl); tsh = *(const f4*)(ssh + cl); }
  int ppy[kPRounds], ppx[kPRounds], ppos[kPRounds]; bool pact[kPRounds];
#pragma unroll
  for (int rd = 0; rd < kPRounds; ++rd) {
    const int u = rd * 256 + tid;
    pact[rd] = u < kPUnits;
    const int pp = pact[rd] ? (u >> 3) : 0;
    ppy[rd] = pp / kPW; ppx[rd] = pp - ppy[rd] * kPW;
    ppos[rd] = pp * kCW + ((chu ^ (((ppx[rd] >> 1) & 1) << 2)) << 2);
  }
  // ---- dY DMA constants: instruction (i, wave) covers 16-byte units [(i*4+wave)*64, +64) of the [64 px][UPP] image
  int ypix[YI], ycu[YI];
#pragma unroll
  for (int i = 0; i < YI; ++i) {
    const int L = (i * 4 + wave) * 64 + lane;
    const int px = L / UPP, su = L - px * UPP;
    ypix[i] = px;
    ycu[i] = (UPP >= 8) ? (su ^ (((px >> 1) & 1) << 2)) : su;       // global unit stored at LDS unit su
  }

  f4 acc[4][CA][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ca = 0; ca < CA; ++ca) { acc[j][ca][0] = (f4){0.f, 0.f, 0.f, 0.f}; acc[j][ca][1] = (f4){0.f, 0.f, 0.f, 0.f}; }

  f4 pv[kPRounds]; unsigned pok = 0;
  auto stage_geo = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % tilesW; const int q = t / tilesW;
    h0 = (q % tilesH) * kSH; n = q / tilesH; w0 = tw * kSW;
  };
  auto y_dma = [&](int t, int buf) {
    int n, h0, w0; stage_geo(t, n, h0, w0);
#pragma unroll
    for (int i = 0; i < YI; ++i) {
      const float* g = a.dy + ((size_t)((size_t)n * a.Ho + h0 + (ypix[i] >> 4)) * a.Wo + w0 + (ypix[i] & 15)) * a.Cout + a0 + ycu[i] * 4;
      __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(uintptr_t)(Ys + buf * kYs + (i * 4 + wave) * 256), 16, 0, 0);
    }
  };
  auto patch_load = [&](int t) {
    int n, h0, w0; stage_geo(t, n, h0, w0);
    pok = 0;
#pragma unroll
    for (int rd = 0; rd < kPRounds; ++rd) {
      const int hl = h0 - 1 + ppy[rd], wl = w0 - 1 + ppx[rd];
      const bool v = pact[rd] && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
      const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
      pv[rd] = *(const f4*)(sp + ((size_t)((size_t)n * sH + (hc >> sup)) * sW + (wc >> sup)) * sC + cl);
      pok |= (v ? 1u : 0u) << rd;
    }
  };
  auto patch_store = [&](int buf) {
    float* ps = Ps + buf * kPP * kCW;
#pragma unroll
    for (int rd = 0; rd < kPRounds; ++rd) {
      f4 v = pv[rd];
      if (thas) {
        v = v * tsc + tsh;
        if (trelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      if (!((pok >> rd) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
      if (pact[rd]) *(f4*)(ps + ppos[rd]) = v;
    }
  };

  // ---- wave-row selectors:  s_b = al*dY[0][b] + be*dY[1][b]   (row `wave` of A);   r_c = d[ra][c] + sg*d[rb][c]   (row of B^T)
  const float al = wave == 3 ? 0.f : 1.f;
  const float be = wave == 0 ? 0.f : (wave == 1 ? 1.f : -1.f);
  const int ra = (wave == 0) ? 0 : (wave == 2 ? 2 : 1);
  const int rb = (wave == 3) ? 3 : (wave == 2 ? 1 : 2);
  const float sg = (wave == 1) ? 1.f : -1.f;

  if (t0 < t1) {
    y_dma(t0, 0);
    patch_load(t0);
    patch_store(0);
  }
  __syncthreads();

  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    const int tn = t + 1 < t1 ? t + 1 : t;          // last stage: harmless re-fetch into the dead buffers
    y_dma(tn, cur ^ 1);
    patch_load(tn);
    __builtin_amdgcn_sched_barrier(0);
    const float* ys = Ys + cur * kYs;
    const float* ps = Ps + cur * kPP * kCW;
#pragma unroll 2
    for (int ks = 0; ks < 4; ++ks) {
      const int tile = ks * 4 + lq;                 // this lane's tile (MFMA k index = lq)
      const int ty = tile >> 3, tx = tile & 7;
      const int sw = (tx & 1) << 4;                 // 16-channel block swizzle of every pixel of this tile
      // A side: P[ca][j] = (A dY A^T)[wave][j] for (tile, co = ca*16 + li)
      float P[CA][4];
      const float* yb = ys + ((2 * ty) * 16 + 2 * tx) * TA;
#pragma unroll
      for (int ca = 0; ca < CA; ++ca) {
        const int co = (UPP >= 8) ? ((ca * 16 + li) ^ sw) : (ca * 16 + li);
        const float y00 = yb[co], y01 = yb[TA + co], y10 = yb[16 * TA + co], y11 = yb[17 * TA + co];
        const float s0 = al * y00 + be * y10, s1 = al * y01 + be * y11;
        P[ca][0] = s0; P[ca][1] = s0 + s1; P[ca][2] = s0 - s1; P[ca][3] = -s1;
      }
      // B side: V[cb][j] = (B^T d B)[wave][j] for (tile, c = cb*16 + li)
      float V[2][4];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        // pixel columns 2tx, 2tx+1 carry this tile's swizzle parity, columns 2tx+2, 2tx+3 the other one
        const int ch = (cb * 16 + li) ^ sw, ch2 = ch ^ 16;
        const float* pa = ps + ((2 * ty + ra) * kPW + 2 * tx) * kCW;
        const float* pb = ps + ((2 * ty + rb) * kPW + 2 * tx) * kCW;
        const float r0 = pa[ch] + sg * pb[ch], r1 = pa[kCW + ch] + sg * pb[kCW + ch];
        const float r2 = pa[2 * kCW + ch2] + sg * pb[2 * kCW + ch2], r3 = pa[3 * kCW + ch2] + sg * pb[3 * kCW + ch2];
        V[cb][0] = r0 - r2; V[cb][1] = r1 + r2; V[cb][2] = r2 - r1; V[cb][3] = r1 - r3;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ca = 0; ca < CA; ++ca) {
          acc[j][ca][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(P[ca][j], V[0][j], acc[j][ca][0], 0, 0, 0);
          acc[j][ca][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(P[ca][j], V[1][j], acc[j][ca][1], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    patch_store(cur ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue: dW = G^T Z G ----------------
  // per wave (row i): y[s] = sum_j Z[i][j] G[j][s]   (recomputed per use: keeping all 3 x CA x 2 tiles would spill)
  auto ycalc = [&](int s, int ca, int cb) -> f4 {
    const f4 z1 = acc[1][ca][cb], z2 = acc[2][ca][cb];
    if (s == 0) return acc[0][ca][cb] + 0.5f * (z1 + z2);
    if (s == 1) return 0.5f * (z1 - z2);
    return 0.5f * (z1 + z2) + acc[3][ca][cb];
  };
  // R[r][s][TA co][32 c] in LDS; rows of G: wave 0 -> r0 ; wave 1 -> (r0,r1,r2)/2 ; wave 2 -> (r0,-r1,r2)/2 ; wave 3 -> r2
  float* const R = smem;
  constexpr int RS = TA * kCW;                      // floats per (r, s) plane
  auto upd = [&](int r, float coef, bool init) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int ca = 0; ca < CA; ++ca)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const f4 yv = ycalc(s, ca, cb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float* p = R + (r * 3 + s) * RS + (ca * 16 + lq * 4 + e) * kCW + cb * 16 + li;
            const float v = coef * yv[e];
            *p = init ? v : *p + v;
          }
        }
  };
  // phase A: first writer of each r plane
  if (wave == 0) upd(0, 1.f, true);
  if (wave == 1) upd(1, 0.5f, true);
  if (wave == 3) upd(2, 1.f, true);
  __syncthreads();
  if (wave == 1) upd(0, 0.5f, false);
  if (wave == 2) { upd(1, -0.5f, false); upd(2, 0.5f, false); }
  __syncthreads();
  if (wave == 1) upd(2, 0.5f, false);
  if (wave == 2) upd(0, 0.5f, false);
  __syncthreads();
  // one partial tile per pixel split (plain stores; wgrad_reduce_kernel adds the splits in order), or straight into dW
  float* const dst = a.nsplit > 1 ? a.part + (size_t)split * a.wrows * a.Kpad : a.dw;
  for (int i = tid; i < 9 * RS; i += 256) {
    const int t = i / RS, rem = i - t * RS;
    const int co = rem / kCW, cch = rem - co * kCW;
    const int row = a0 + co;
    if (row < a.wrows) {
      float* q = dst + (size_t)row * a.Kpad + t * a.Ctot + cc * kCW + cch;
      *q = a.nsplit > 1 ? R[i] : *q + R[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad_wino2: TA = 64 | 32.  Same workgroup tile (TA output x 32 input channels), same 4x16-pixel stages, same LDS
// images and loaders as above; what changes is who multiplies what:
//   wave w owns output-channel blocks ca in [(w>>1)*NCA, +NCA) (NCA = TA/32) x input-channel block cb = w&1 at ALL 16
//   Winograd points.  Per 4 tiles it reads its raw dY (4 pixels per ca) and its raw 4x4 input patch (16 values) once,
//   forms the full A dY A^T and B^T d B in registers and issues 16*NCA MFMAs: 8 + 16 = 24 LDS reads per 32 MFMAs
//   instead of 32, and nothing a lane reads is read again by another wave's lane for the same product.
//   The accumulators of a lane are Z[xi] of ITS (co, ci) elements for every xi, so dW = G^T Z G needs no cross-wave
//   exchange: no LDS epilogue, no extra barriers.
//   The next stage's dY DMA and patch loads are issued inside MFMA block 0 / 1, the patch is written to LDS inside
//   block 3: the stage boundary is the barrier alone.
template <int TA>
__global__ __launch_bounds__(256, 2) void wgrad_wino2_kernel(const WgradArgs a) {
  static_assert(TA == 64 || TA == 32, "wgrad_wino2: TA = 64 | 32");
  constexpr int NCA = TA / 32;             // output-channel MFMA tiles per wave
  constexpr int UPP = TA / 4;              // 16-byte units per dY pixel
  constexpr int YI = UPP / 4;              // LDS-DMA instructions per wave per stage (TA=64: 4, TA=32: 2)
  constexpr int kYs = 64 * TA;             // floats per dY buffer
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ys = smem;                        // [2][64 px][TA]       (16-channel blocks swizzled by tile parity)
  float* const Ps = smem + 2 * kYs;              // [2][108 px][32]      (same swizzle)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int wca = (wave >> 1) * NCA, wcb = wave & 1;

  const int nchunk = (a.Ctot + kCW - 1) / kCW;      // (the last chunk of the second source may be partial: Ctot % 8 == 0)
  const int tilesA = a.Cout / TA;
  const int tilesW = a.Wo / kSW, tilesH = a.Ho / kSH;
  const int nstages = a.N * tilesH * tilesW;
  const int pairs = nchunk * tilesA;
  // block -> (pair, split).  With nsplit % 8 == 0 every block of a pixel split lands on the same XCD (blocks b and b + 8
  // share one): XCD x serves splits x, x + 8, ... one after the other, its `pairs` blocks of a split side by side.
  int pair, split;
  {
    const int b = blockIdx.x;
    if ((a.nsplit & 7) == 0) { const int x = b & 7, idx = b >> 3; split = (idx / pairs) * 8 + x; pair = idx % pairs; }
    else { pair = b % pairs; split = b / pairs; }
  }
  const int cc = pair % nchunk, ta = pair / nchunk;
  const int a0 = ta * TA;
  const int t0 = split * a.msplit, t1 = min(nstages, t0 + a.msplit);

  // ---- patch loader: thread = (pixel of the 6x18 halo patch, 4-channel unit), rounds of 256 threads.  Every address is a
  // wave-uniform stage origin (scalar registers) + a per-thread offset fixed for the whole launch; only the zero-padding
  // predicate depends on the stage (which image borders it touches), as 4 scalar bits against 4 static bits per round.
  const int chu = tid & 7;
  const bool first = cc * kCW < a.C0;              // block-uniform: C0 is a multiple of the 32-channel chunk
  const float* sp = first ? a.s0.ptr : a.s1.ptr;
  const float* ssc = first ? a.s0.scale : a.s1.scale;
  const float* ssh = first ? a.s0.shift : a.s1.shift;
  const int sC = first ? a.s0.C : a.s1.C, sH = first ? a.s0.H : a.s1.H, sW = first ? a.s0.W : a.s1.W;
  const int sup = first ? a.s0.up : a.s1.up;
  const int trelu = first ? a.s0.relu : a.s1.relu;
  const int cl_raw = (first ? cc * kCW : cc * kCW - a.C0) + chu * 4;
  // channel tail (Ctot % 32 != 0: EfficientNet decoder concats 64 + 48, 256 + 56): units past the source's channel count are
  // read from its last unit (a valid address) and stored as zeros
  const bool cvalid = cl_raw + 4 <= sC;
  const int cl = cvalid ? cl_raw : sC - 4;
  const bool thas = ssc != nullptr;
  f4 tsc = {1.f, 1.f, 1.f, 1.f}, tsh = {0.f, 0.f, 0.f, 0.f};
  if (thas) { tsc = *(const f4*)(ssc + cl); tsh = *(const f4*)(ssh + cl); }
  // offsets are taken from the pixel one row up / one column left of the stage origin (two when upsampled), so they are >= 0
  unsigned poff[kPRounds]; int ppos[kPRounds]; unsigned pflags = 0;
  unsigned pcenter;                                // a pixel that always exists: the stage's own first pixel
  {
    const int sh1 = sup ? 1 : 0;
    pcenter = (unsigned)((((1 + sh1) >> sh1) * sW + ((1 + sh1) >> sh1)) * sC + cl);
#pragma unroll
    for (int rd = 0; rd < kPRounds; ++rd) {
      const int u = rd * 256 + tid;
      const bool act = u < kPUnits;
      const int pp = act ? (u >> 3) : 0;
      const int py = pp / kPW, px = pp - py * kPW;
      ppos[rd] = act ? pp * kCW + ((chu ^ (((px >> 1) & 1) << 2)) << 2) : -1;
      poff[rd] = (unsigned)((((py + sh1) >> sh1) * sW + ((px + sh1) >> sh1)) * sC + cl);
      const unsigned fl = (py == 0 ? 1u : 0u) | (py == kSH + 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == kSW + 1 ? 8u : 0u);
      pflags |= fl << (4 * rd);
    }
  }
  // ---- dY DMA: instruction (i, wave) covers 16-byte units [(i*4+wave)*64, +64) of the [64 px][UPP] image
  unsigned yoff[YI];
#pragma unroll
  for (int i = 0; i < YI; ++i) {
    const int L = (i * 4 + wave) * 64 + lane;
    const int px = L / UPP, su = L - px * UPP;
    const int cu = su ^ (((px >> 1) & 1) << 2);      // global unit stored at LDS unit su (UPP >= 8 for both TA)
    yoff[i] = (unsigned)(((px >> 4) * a.Wo + (px & 15)) * a.Cout + cu * 4);
  }

  f4 acc[4][4][NCA];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < NCA; ++q) acc[i][j][q] = (f4){0.f, 0.f, 0.f, 0.f};

  f4 pv[kPRounds]; unsigned pinv = 0;
  auto stage_geo = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % tilesW; const int q = t / tilesW;
    h0 = (q % tilesH) * kSH; n = q / tilesH; w0 = tw * kSW;
  };
  auto y_dma = [&](int t, int buf, int i) {
    int n, h0, w0; stage_geo(t, n, h0, w0);
    const float* g0 = a.dy + ((size_t)((size_t)n * a.Ho + h0) * a.Wo + w0) * a.Cout + a0;     // wave-uniform
    __builtin_amdgcn_global_load_lds((gbl_void*)(g0 + yoff[i]), (lds_void*)(uintptr_t)(Ys + buf * kYs + (i * 4 + wave) * 256), 16, 0, 0);
  };
  auto patch_load = [&](int t) {
    int n, h0, w0; stage_geo(t, n, h0, w0);
    const int sh1 = sup ? 1 : 0;
    // origin one (upsampled: two) pixels up-left of the stage: may lie before the tensor for the first stage of an image;
    // lanes whose pixel is outside the image are redirected to the stage's own first pixel before the load
    const long long org = ((long long)((long long)n * sH + ((h0 - 1 - sh1) >> sh1)) * sW + ((w0 - 1 - sh1) >> sh1)) * sC;
    const float* g0 = sp + org;
    const unsigned smask = ((h0 == 0 ? 1u : 0u) | (h0 + kSH == a.Hl ? 2u : 0u) | (w0 == 0 ? 4u : 0u) | (w0 + kSW == a.Wl ? 8u : 0u)) * 0x1111u;
    pinv = pflags & smask;
#pragma unroll
    for (int rd = 0; rd < kPRounds; ++rd) {
      const bool inv = ((pinv >> (4 * rd)) & 0xFu) != 0;
      pv[rd] = *(const f4*)(g0 + (inv ? pcenter : poff[rd]));
    }
  };
  auto patch_store = [&](int buf) {
    float* ps = Ps + buf * kPP * kCW;
#pragma unroll
    for (int rd = 0; rd < kPRounds; ++rd) {
      f4 v = pv[rd];
      if (thas) {
        v = v * tsc + tsh;
        if (trelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      if (((pinv >> (4 * rd)) & 0xFu) || !cvalid) v = (f4){0.f, 0.f, 0.f, 0.f};
      if (ppos[rd] >= 0) *(f4*)(ps + ppos[rd]) = v;
    }
  };

  // ---- software-pipelined operand path.  Raw LDS values of the NEXT 4-tile block are read one MFMA group (16 MFMAs,
  // >= 512 cycles) ahead of their first use, so no MFMA ever waits for an LDS read it has just issued (round 1's kernel
  // read, waited and multiplied in place: the matrix pipe idled for the LDS latency ~6 times per block).
  //   dB[16]: raw 4x4 input patch of (tile, ci) ; dA[q][4]: raw 2x2 dY of (tile, co_q) ; V / Vn: B^T d B of this / the next block
  float dB[16], dA[NCA][4], V[4][4];
  // tile = ks*4 + lq  =>  ty = ks >> 1, tx = (ks & 1)*4 + lq: everything but lq is a compile-time constant of the block, and
  // the swizzle parity (tx & 1) = (lq & 1) is a lane constant, so every operand read is ONE lane base + an immediate offset
  const int lsw = (lq & 1) << 4;
  const int bch = (wcb * 16 + li) ^ lsw;                       // input channel slot of pixel columns 2tx, 2tx+1 (2tx+2, 2tx+3: ^16)
  const int pbase0 = (2 * lq) * kCW + bch, pbase1 = (2 * lq) * kCW + (bch ^ 16);
  const int ybase = (2 * lq) * TA + ((wca * 16 + li) ^ lsw);   // co block q sits 16 floats further, on either side of the swizzle
  const int yq1 = (((wca + 1) * 16 + li) ^ lsw) - ((wca * 16 + li) ^ lsw);
  auto read_B = [&](const float* ps, int ks) {
    const int koff = ((2 * (ks >> 1)) * kPW + 8 * (ks & 1)) * kCW;      // compile-time per unrolled block
    const float* p0 = ps + pbase0 + koff; const float* p1 = ps + pbase1 + koff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dB[r * 4 + 0] = p0[(r * kPW + 0) * kCW]; dB[r * 4 + 1] = p0[(r * kPW + 1) * kCW];
      dB[r * 4 + 2] = p1[(r * kPW + 2) * kCW]; dB[r * 4 + 3] = p1[(r * kPW + 3) * kCW];
    }
  };
  auto read_A = [&](const float* ys, int ks, int q) {
    const int koff = ((2 * (ks >> 1)) * 16 + 8 * (ks & 1)) * TA;
    const float* yb = ys + ybase + koff + (q ? yq1 : 0);
    dA[q][0] = yb[0]; dA[q][1] = yb[TA]; dA[q][2] = yb[16 * TA]; dA[q][3] = yb[17 * TA];
  };
  auto transform_B = [&](float (&o)[4][4]) {         // o = B^T d B
    float r_[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      r_[0][x] = dB[0 + x] - dB[8 + x]; r_[1][x] = dB[4 + x] + dB[8 + x]; r_[2][x] = dB[8 + x] - dB[4 + x]; r_[3][x] = dB[4 + x] - dB[12 + x];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i][0] = r_[i][0] - r_[i][2]; o[i][1] = r_[i][1] + r_[i][2]; o[i][2] = r_[i][2] - r_[i][1]; o[i][3] = r_[i][1] - r_[i][3];
    }
  };
  auto mfma_group = [&](int q) {                     // 16 MFMAs: (A dY A^T)[i][j] of co block q  x  V[i][j]
    const float y00 = dA[q][0], y01 = dA[q][1], y10 = dA[q][2], y11 = dA[q][3];
    const float ta_[4] = {y00, y00 + y10, y00 - y10, -y10};
    const float tb_[4] = {y01, y01 + y11, y01 - y11, -y11};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float p0 = ta_[i], p1 = ta_[i] + tb_[i], p2 = ta_[i] - tb_[i], p3 = -tb_[i];
      acc[i][0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(p0, V[i][0], acc[i][0][q], 0, 0, 0);
      acc[i][1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1, V[i][1], acc[i][1][q], 0, 0, 0);
      acc[i][2][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(p2, V[i][2], acc[i][2][q], 0, 0, 0);
      acc[i][3][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(p3, V[i][3], acc[i][3][q], 0, 0, 0);
    }
  };
#define UWM_FENCE() __builtin_amdgcn_sched_barrier(0)

  if (t0 < t1) {
#pragma unroll
    for (int i = 0; i < YI; ++i) y_dma(t0, 0, i);
    patch_load(t0);
    patch_store(0);
  }
  __syncthreads();

  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    const int tn = t + 1 < t1 ? t + 1 : t;          // last stage: harmless re-fetch into the dead buffers
    const float* ys = Ys + cur * kYs;
    const float* ps = Ps + cur * kPP * kCW;
    if constexpr (NCA == 1) {
      // TA = 32 (3 workgroups per CU, registers to spare): software-pipelined operand path.  Raw LDS values of the NEXT
      // 4-tile block are read one MFMA group ahead of their first use, so no MFMA waits for a read it has just issued.
      read_B(ps, 0);
      read_A(ys, 0, 0);
      transform_B(V);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) read_B(ps, ks + 1);             // R0: raw input patch of the next block
        UWM_FENCE();
        // R1: the MFMA group.  Stage-boundary work rides on it: block 0 carries the next stage's patch loads, blocks 1 / 2
        // its dY DMA, block 3 the patch stores (their loads are >= 2 blocks old by then)
        if (ks == 0) patch_load(tn);
        if (ks == 1) y_dma(tn, cur ^ 1, 0);
        if (ks == 2) y_dma(tn, cur ^ 1, 1);
        mfma_group(0);
        if (ks == 3) patch_store(cur ^ 1);
        UWM_FENCE();
        if (ks < 3) { read_A(ys, ks + 1, 0); transform_B(V); }     // R2: next block's dY + input transform
      }
    } else {
      // TA = 64 (2 workgroups per CU, ~220 registers): the fully fenced pipeline above spills here (measured 133 vs 123 us
      // on layer3).  Lighter form: the raw input patch of the NEXT block is read right after this block's transform has
      // consumed the registers, i.e. a whole block (32 MFMAs) ahead of its use; this block's dY reads sit above the input
      // transform, whose 32 VALU ops cover their latency.  Inside a block the compiler interleaves transforms and MFMAs.
      read_B(ps, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks == 0) patch_load(tn);
        if (ks == 1) { y_dma(tn, cur ^ 1, 0); y_dma(tn, cur ^ 1, 1); }
        if (ks == 2) { y_dma(tn, cur ^ 1, 2); y_dma(tn, cur ^ 1, 3); }
        read_A(ys, ks, 0);
        read_A(ys, ks, 1);
        transform_B(V);
        if (ks < 3) read_B(ps, ks + 1);
        mfma_group(0);
        mfma_group(1);
        if (ks == 3) patch_store(cur ^ 1);
        UWM_FENCE();
      }
    }
    __syncthreads();
  }
#undef UWM_FENCE

  // ---------------- epilogue: dW = G^T Z G, all in registers ----------------
  // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]: over i then over j: (z0 + h(z1+z2), h(z1-z2), h(z1+z2) + z3), h = 1/2
  float* const dst = a.nsplit > 1 ? a.part + (size_t)split * a.wrows * a.Kpad : a.dw;
  const bool direct = a.nsplit <= 1;
#pragma unroll
  for (int q = 0; q < NCA; ++q) {
    f4 R[3][4];                                      // rows r = 0..2 (i transformed), columns j = 0..3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 sp_ = 0.5f * (acc[1][j][q] + acc[2][j][q]), sm_ = 0.5f * (acc[1][j][q] - acc[2][j][q]);
      R[0][j] = acc[0][j][q] + sp_; R[1][j] = sm_; R[2][j] = sp_ + acc[3][j][q];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const f4 sp_ = 0.5f * (R[r][1] + R[r][2]), sm_ = 0.5f * (R[r][1] - R[r][2]);
      const f4 w3[3] = {R[r][0] + sp_, sm_, sp_ + R[r][3]};
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = a0 + (wca + q) * 16 + lq * 4 + e;
          if (row < a.wrows && cc * kCW + wcb * 16 + li < a.Ctot) {
            float* p = dst + (size_t)row * a.Kpad + (r * 3 + s_) * a.Ctot + cc * kCW + wcb * 16 + li;
            *p = direct ? *p + w3[s_][e] : w3[s_][e];
          }
        }
    }
  }
  // K padding of a partial image (Kpad > 9 * Ctot only when Ctot % 32 != 0): the reduce adds whole images, so it must read zeros there
  if (!direct && cc == nchunk - 1 && a.Kpad > 9 * a.Ctot) {
    const int padw = a.Kpad - 9 * a.Ctot;
    for (int i = tid; i < TA * padw; i += 256) {
      const int row = a0 + i / padw;
      if (row < a.wrows) dst[(size_t)row * a.Kpad + 9 * a.Ctot + i % padw] = 0.f;
    }
  }
}

// dw[i] += part[0][i] + part[1][i] + ... (16-byte units; n4 = wrows * Kpad / 4).  Workgroup = 32 units x 8 split groups:
// thread (unit, group) adds splits group, group + 8, ... in order, the 8 group sums are combined in group order through
// LDS — a fixed association for a given nsplit, so the result is bit-reproducible; the first version walked all splits
// serially per thread (36 workgroups of 256 dependent loads on a 64x64 layer: 35-80 us per launch, 1.35 ms per step).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, size_t n4, float* __restrict__ dw) {
  __shared__ f4 red[8][32];
  const int u = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + u;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4)
    for (int k = grp; k < nsplit; k += 8) s += *(const f4*)(part + ((size_t)k * n4 + i) * 4);
  red[grp][u] = s;
  __syncthreads();
  if (grp == 0 && i < n4) {
    f4 t = red[0][u];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][u];
    *(f4*)(dw + i * 4) += t;
  }
}

// the same reduce for every job of a queue in one launch: a workgroup finds its job by its block range.  A thread owns ONE 16-byte
// unit of dW and walks all the split images itself, eight loads in flight (a workgroup reads 4 KB contiguous per split: the
// 8-group / 512-byte form above ran at 2 TB/s on the step's 0.7 GB of partial images); the order of the additions depends on
// nsplit alone: bit-reproducible
constexpr int kRedUnits = 256;                 // 16-byte units of dW per workgroup
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const ReduceJobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && blockIdx.x >= jobs.j[k + 1].block0) ++k;
  const float* __restrict__ part = jobs.j[k].part; float* __restrict__ dw = jobs.j[k].dw;
  const size_t n4 = jobs.j[k].n4; const int nsplit = jobs.j[k].nsplit;
  const size_t i = (size_t)(blockIdx.x - jobs.j[k].block0) * kRedUnits + threadIdx.x;
  if (i >= n4) return;
  const float* __restrict__ p0 = part + i * 4;
  const size_t st = n4 * 4;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  int q = 0;
  for (; q + 8 <= nsplit; q += 8) {
    const f4 a0 = *(const f4*)(p0 + (size_t)q * st), a1 = *(const f4*)(p0 + (size_t)(q + 1) * st);
    const f4 a2 = *(const f4*)(p0 + (size_t)(q + 2) * st), a3 = *(const f4*)(p0 + (size_t)(q + 3) * st);
    const f4 a4 = *(const f4*)(p0 + (size_t)(q + 4) * st), a5 = *(const f4*)(p0 + (size_t)(q + 5) * st);
    const f4 a6 = *(const f4*)(p0 + (size_t)(q + 6) * st), a7 = *(const f4*)(p0 + (size_t)(q + 7) * st);
    s += ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  }
  for (; q < nsplit; ++q) s += *(const f4*)(p0 + (size_t)q * st);
  *(f4*)(dw + i * 4) += s;
}

hipError_t launch_wgrad_reduce(const float* part, int nsplit, size_t n4, float* dw, hipStream_t st, ReduceQueue* rq) {      // shared with wgrad_gemm.hip
  if (rq) {
    if (rq->n >= ReduceQueue::kMax) return hipErrorInvalidValue;       // the owner flushes before the queue is full
    ReduceJob& j = rq->j[rq->n++];
    j.part = part; j.dw = dw; j.n4 = n4; j.nsplit = nsplit; j.block0 = rq->blocks;
    rq->blocks += (unsigned)((n4 + kRedUnits - 1) / kRedUnits);
    rq->used_floats += ((size_t)nsplit * n4 * 4 + 63) & ~(size_t)63;
    return hipSuccess;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 31) / 32)), dim3(256), 0, st, part, nsplit, n4, dw);
  return hipGetLastError();
}
hipError_t launch_wgrad_reduce_multi(ReduceQueue& q, hipStream_t st) {
  if (q.n == 0) { q.used_floats = 0; q.blocks = 0; return hipSuccess; }
  ReduceJobs jobs; jobs.n = q.n;
  for (int i = 0; i < q.n; ++i) jobs.j[i] = q.j[i];
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(q.blocks), dim3(256), 0, st, jobs);
  q.n = 0; q.blocks = 0; q.used_floats = 0;
  return hipGetLastError();
}

constexpr int kMaxPartBlocks = 1024;          // workgroups of a split launch: bounds the partial-sum scratch
size_t wgrad_wino_scratch_floats() { return (size_t)kMaxPartBlocks * 64 * kCW * 9; }

float* wgrad_op_scratch() {                   // single-operator entry points (no workspace): one cached buffer per device
  static float* buf[64] = {nullptr};
  int dev = 0; (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return nullptr;
  if (!buf[dev] && hipMalloc((void**)&buf[dev], wgrad_wino_scratch_floats() * sizeof(float)) != hipSuccess) { buf[dev] = nullptr; (void)hipGetLastError(); }
  return buf[dev];
}

template <int TA>
static hipError_t launch_ww(const WgradArgs& a, hipStream_t st, int cls, int nblocks) {
  size_t lds = (size_t)(2 * 64 * TA + 2 * kPP * kCW) * sizeof(float);
  static const bool v1 = dbg_flag("UWM_WGRAD_V1");        // experiments: the round-1 row-per-wave kernel for every TA
  if (TA == 16 || v1) {
    const size_t rl = (size_t)9 * TA * kCW * sizeof(float);
    if (lds < rl) lds = rl;
    static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
    { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_wino_kernel<TA>, lds); if (e != hipSuccess) return e; }
    UWM_LAUNCH(cls, a.flops, a.bytes, (wgrad_wino_kernel<TA>), dim3((unsigned)nblocks), dim3(256), lds, st, a);
  } else if constexpr (TA != 16) {
    static DevOnce lds_attr;
    { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_wino2_kernel<TA>, lds); if (e != hipSuccess) return e; }
    UWM_LAUNCH(cls, a.flops, a.bytes, (wgrad_wino2_kernel<TA>), dim3((unsigned)nblocks), dim3(256), lds, st, a);
  }
  if (a.nsplit > 1) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_wgrad_reduce(a.part, a.nsplit, (size_t)a.wrows * a.Kpad / 4, a.dw, st, a.rq);
  }
  return hipGetLastError();
}

bool wgrad_wino_applicable(const WgradArgs& a) {
  const int TA = a.Cout >= 64 ? 64 : a.Cout;
  // channel tail (wgrad_wino2 only, TA >= 32): Ctot % 8 == 0 with the concat boundary on a 32-channel chunk
  static const bool v1 = dbg_flag("UWM_WGRAD_V1");
  const bool tail_ok = TA >= 32 && !v1 && (a.Ctot & 7) == 0 && (a.Ctot - a.C0 == 0 || (a.Ctot - a.C0) >= 4);
  return a.ntaps == 9 && a.kw == 3 && a.stride == 1 && a.pad == 1 && ((a.Ctot & 31) == 0 || tail_ok) && ((a.C0 & 31) == 0 || (tail_ok && a.C0 == a.Ctot)) &&
         (TA == 64 || TA == 32 || TA == 16) && a.Cout % TA == 0 && a.wrows <= a.Cout && a.Kpad >= 9 * a.Ctot && a.Kpad < 9 * a.Ctot + 32 &&
         a.Hl == a.Ho && a.Wl == a.Wo && (a.Ho % kSH) == 0 && (a.Wo % kSW) == 0;
}

hipError_t launch_wgrad_wino(const WgradArgs& a0, hipStream_t st) {
  WgradArgs a = a0;
  if (!wgrad_wino_applicable(a)) return hipErrorInvalidValue;
  int TA = a.Cout >= 64 ? 64 : a.Cout;
  static const int force_ta = dbg_int("UWM_WW_TA", 0);      // experiments
  if (force_ta && TA > force_ta && a.Cout % force_ta == 0) TA = force_ta;
  const int nchunk = (a.Ctot + kCW - 1) / kCW, tilesA = a.Cout / TA;
  const int nstages = a.N * (a.Ho / kSH) * (a.Wo / kSW);
  // cost model: rounds x (stages per workgroup + epilogue worth E stages); a split launch is capped at kMaxPartBlocks
  // workgroups (the partial-sum scratch) and, from 8 splits on, uses a multiple of 8 of them (one XCD per split)
  const int cus = device_cu_count();
  const int pairs = nchunk * tilesA;
  const int slots_per_cu = TA == 64 ? 2 : 3;
  const double E = TA == 64 ? 2.0 : 1.0;
  const int slots = cus * slots_per_cu;
  int nsplit = 1; double best = 1e30;
  for (int ns = 1; ns <= nstages && ns <= 2048; ++ns) {
    const int tp = (nstages + ns - 1) / ns;
    int nsr = (nstages + tp - 1) / tp;
    if (nsr >= 8) nsr = (nsr + 7) & ~7;
    if (nsr > nstages) continue;
    const long blocks = (long)pairs * nsr;
    if (nsr > 1 && blocks > kMaxPartBlocks) break;
    const int tpe = (nstages + nsr - 1) / nsr;
    const long rounds = (blocks + slots - 1) / slots;
    const double cost = (double)rounds * (tpe + E) + 0.02 * nsr;     // (+ the reduce pass reads nsr partial tiles)
    if (cost < best - 1e-9) { best = cost; nsplit = nsr; }
  }
  int tps = (nstages + nsplit - 1) / nsplit;
  if (nsplit < 8) nsplit = (nstages + tps - 1) / tps;               // (a multiple of 8 keeps its empty tail splits: t0 >= t1)
  a.nsplit = nsplit; a.msplit = tps;
  if (nsplit > 1) {
    if (!a.part || a.part_floats < (size_t)nsplit * a.wrows * a.Kpad) { a.part = wgrad_op_scratch(); a.part_floats = wgrad_wino_scratch_floats(); }
    if (!a.part || a.part_floats < (size_t)nsplit * a.wrows * a.Kpad) return hipErrorOutOfMemory;
  }
  const int nblocks = nsplit * tilesA * nchunk;
  switch (TA) {
    case 64: return launch_ww<64>(a, st, 22, nblocks);
    case 32: return launch_ww<32>(a, st, 23, nblocks);
    default: return launch_ww<16>(a, st, 24, nblocks);
  }
}

}  // namespace uwm
