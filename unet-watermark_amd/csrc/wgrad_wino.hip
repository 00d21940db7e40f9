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
// ordered phases (plain read-add-write, no LDS atomics), then coalesced fp32 global atomics.
#include "uwm_kernels.h"

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
  for (int i = tid; i < 9 * RS; i += 256) {
    const int t = i / RS, rem = i - t * RS;
    const int co = rem / kCW, cch = rem - co * kCW;
    const int row = a0 + co;
    if (row < a.wrows) atomicAdd(a.dw + (size_t)row * a.Kpad + t * a.Ctot + cc * kCW + cch, R[i]);
  }
}

template <int TA>
static hipError_t launch_ww(const WgradArgs& a, hipStream_t st, int cls, int nblocks) {
  size_t lds = (size_t)(2 * 64 * TA + 2 * kPP * kCW) * sizeof(float);
  const size_t rl = (size_t)9 * TA * kCW * sizeof(float);
  if (lds < rl) lds = rl;
  static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
  { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_wino_kernel<TA>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(cls, a.flops, a.bytes, (wgrad_wino_kernel<TA>), dim3((unsigned)nblocks), dim3(256), lds, st, a);
  return hipGetLastError();
}

bool wgrad_wino_applicable(const WgradArgs& a) {
  const int TA = a.Cout >= 64 ? 64 : a.Cout;
  return a.ntaps == 9 && a.kw == 3 && a.stride == 1 && a.pad == 1 && (a.Ctot & 31) == 0 && (a.C0 & 31) == 0 &&
         (TA == 64 || TA == 32 || TA == 16) && a.Cout % TA == 0 && a.wrows <= a.Cout &&
         a.Hl == a.Ho && a.Wl == a.Wo && (a.Ho % kSH) == 0 && (a.Wo % kSW) == 0;
}

hipError_t launch_wgrad_wino(const WgradArgs& a0, hipStream_t st) {
  WgradArgs a = a0;
  if (!wgrad_wino_applicable(a)) return hipErrorInvalidValue;
  const int TA = a.Cout >= 64 ? 64 : a.Cout;
  const int nchunk = a.Ctot / kCW, tilesA = a.Cout / TA;
  const int nstages = a.N * (a.Ho / kSH) * (a.Wo / kSW);
  // same cost model as wgrad_patch: rounds x (stages per workgroup + epilogue worth E stages)
  const int cus = device_cu_count();
  const int pairs = nchunk * tilesA;
  const int slots = cus * 2;
  const double E = TA == 64 ? 4.0 : (TA == 32 ? 2.0 : 1.0);
  int nsplit = 1; double best = 1e30;
  for (int ns = 1; ns <= nstages && ns <= 2048; ++ns) {
    const int tp = (nstages + ns - 1) / ns;
    const int nsr = (nstages + tp - 1) / tp;
    const long blocks = (long)pairs * nsr;
    const long rounds = (blocks + slots - 1) / slots;
    const double cost = (double)rounds * (tp + E);
    if (cost < best - 1e-9) { best = cost; nsplit = nsr; }
  }
  int tps = (nstages + nsplit - 1) / nsplit;
  nsplit = (nstages + tps - 1) / tps;
  a.nsplit = nsplit; a.msplit = tps;
  const int nblocks = nsplit * tilesA * nchunk;
  switch (TA) {
    case 64: return launch_ww<64>(a, st, 22, nblocks);
    case 32: return launch_ww<32>(a, st, 23, nblocks);
    default: return launch_ww<16>(a, st, 24, nblocks);
  }
}

}  // namespace uwm
