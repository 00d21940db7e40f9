// fp16x3 weight gradient of the 3x3 / stride-1 convolutions for gfx950 on v_mfma_f32_16x16x32_f16 (the arithmetic of
// conv_f16x3.hip: every fp32 operand split into two fp16 halves, three MFMAs per product block, fp32 accumulation; dY scaled
// by the power of two that puts its maximum — tracked by bn_bwd_apply — into [2^13, 2^14), X staged as it is):
//
//   dW[co][tap][ci] = sum over pixels p of dY[p][co] * X[p + tap][ci]         (direct form, 9 taps, pixels = the MFMA K dimension)
//
// The reduction runs over PIXELS, so both MFMA operands are needed k-major (8 consecutive pixels per lane) while NHWC memory
// and its LDS image are channel-major: the fragments are read with gfx950's transposing LDS load ds_read_b64_tr_b16 (a 4 x 16
// block of halfs per 16 lanes, delivered column-major) — no transposing stores, and a tap shift is just another row address.
//
// Work split: one 768-thread workgroup per CU owns 64 output channels x 32 input channels x 9 taps of dW (72 accumulator VGPRs
// per MMA lane) and walks its share of the image in stages of 4 rows x 32 pixels:
//   * waves 4-11 (LOADERS; twice the MMA waves: staging — index arithmetic, the lazy transform, the split — is what bounds
//     the kernel with four of them: 87 -> 43 us on layer1 by the compile-time ablation) stage dY [128 px][hi 64 co | lo 64 co] (256-byte rows, 16-byte chunks XOR-swizzled so that the
//     transposed reads are conflict-free) and the 6 x 34-pixel halo patch of X [204 px][hi 32 ci | lo 32 ci] (136-byte rows)
//     with the lazy BatchNorm + ReLU / nearest x2 upsample / concat transform and the hi / lo split, double-buffered, with the
//     global loads of the stage after next already in flight;
//   * waves 0-3 (MMA) = (output-channel pair, input-channel fragment): per pixel row (one k-step of 32 pixels) 2 dY fragments
//     and, per tap, one X fragment (2 transposed reads per half) feeding 6 MFMAs; 54 MFMAs per k-step;
//   * one barrier per stage.  Pixel splits store dW-shaped partial images, wgrad_reduce adds them in a fixed order
//     (deterministic; same queue as the Winograd weight gradient).
//
// Replaces the weight-gradient half of autograd's conv2d backward for these layers.
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4 lds_fp16x4;

#ifndef UWM_WG16_ABL
#define UWM_WG16_ABL 0      // compile-time timing ablations (scripts/ablate_f16x3.sh wgrad ...): 1 no MFMA, 2 no X fragment reads, 4 no loader work after the first stage, 8 no partial-image stores; 0 in the product build
#endif
// Stage geometry: 128 pixels = kWR rows x WX pixels (WX = 32: 4 x 32, one image row per 32-pixel k-step; WX = 16, the 16-pixel-wide
// maps of the deepest encoder stage: 8 x 16, two image rows per k-step).  Halo patch (kWR + 2) x (WX + 2).
#ifndef UWM_WG16_XCD
#define UWM_WG16_XCD 1
#endif
template <int WX> struct WG16 {
  static constexpr int kWR = 128 / WX, kWX = WX, kWPW = WX + 2, kWPH = kWR + 2;
  static constexpr int kWXB = kWPH * kWPW * 136;                         // bytes of an X stage image (27 744 / 24 480)
  static constexpr int kWStage = 32768 + ((kWXB + 255) & ~255);          // bytes per stage buffer
  static constexpr int kWXRounds = (kWPH * kWPW * 8 + 511) / 512;        // X units per loader thread: 4 / 3
};
constexpr int kWDyB = 128 * 256;                                       // bytes of a dY stage image (32 768)
constexpr int kWXS = 136;                                              // bytes per X patch pixel (128 + 8 pad)
constexpr int kWLT = 512;                                              // loader threads (8 waves) next to the 4 MMA waves
constexpr int kWDyRounds = 128 * 16 / kWLT;                            // dY 16-byte units per loader thread: 4

__device__ __forceinline__ int off_dy(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__device__ __forceinline__ float clamp_hw(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }      // (one v_med3_f32; fminf(fmaxf()) adds a canonicalising v_max per value)
__device__ __forceinline__ h8 tr_pair(const char* base, int o0, int o1) {      // two transposed reads -> one 8-half operand fragment
  const fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(uintptr_t)(base + o0));
  const fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(uintptr_t)(base + o1));
  typedef __fp16 fp16x8 __attribute__((__vector_size__(8 * sizeof(__fp16))));
  const fp16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(h8, v);
}

// NP = split products per tile (WgradArgs::nprod): 3 = dy_hi*x_hi + dy_hi*x_lo + dy_lo*x_hi; 2 = dY as ONE fp16 (no dy_lo*x_hi); 1 = dy_hi*x_hi only
template <int WX, int NP = 3>
__global__ __launch_bounds__(768, 1) void wgrad_f16x3_kernel(const WgradArgs a, int stages_per_split, int nstages) {
  constexpr int kWR = WG16<WX>::kWR, kWX = WX, kWPW = WG16<WX>::kWPW, kWPH = WG16<WX>::kWPH, kWStage = WG16<WX>::kWStage, kWXRounds = WG16<WX>::kWXRounds;
  constexpr int kKS = 4, kRK = 32 / WX;                // k-steps (32 pixels) per stage; image rows per k-step (1 / 2)
  extern __shared__ __attribute__((aligned(256))) char wsm[];       // [2][dY image | X image]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_mma = wave < 4;
  const int ltid = tid - 256;                          // loader thread index (waves 4-11)

  const int tilesA = (a.wrows + 63) / 64, tilesB = a.Ctot / 32;
  const int pairs = tilesA * tilesB;
  // XCD-aware walk (round 4): workgroups are dealt round-robin over the 8 XCDs, so consecutive blockIdx values — the input-channel
  // tiles that stream the SAME dY block — used to land on eight different L2s and fetch it eight times (PMC: 235 MB fetched per
  // launch against 83 MB algorithmic).  The bijective remap gives every XCD a contiguous range of the (split, co tile, ci tile)
  // order: the siblings that share a dY block (and, next, the co tiles that share an X patch) meet in one L2.
#if UWM_WG16_XCD
  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int lin = (int)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3));
#else
  const int lin = blockIdx.x;
#endif
  const int split = lin / pairs, pr = lin - split * pairs;
  const int ta = pr / tilesB, tb = pr - ta * tilesB;
  const int a0 = ta * 64, b0 = tb * 32;
  const int st0 = split * stages_per_split, st1 = min(nstages, st0 + stages_per_split);
  const int tilesX = a.Wo / kWX, tilesY = a.Ho / kWR;

  float xs = 1.f;                                      // power-of-two scale of dY (conv_f16x3.hip)
  if (a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }

  f4 acc[2][9];
#pragma unroll
  for (int cf = 0; cf < 2; ++cf)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[cf][t] = (f4){0.f, 0.f, 0.f, 0.f};

  if (!is_mma) {
    // ================= loader waves =================
    const bool first = b0 < a.C0;                      // the workgroup's 32 input channels lie in one source of the concat
    const Src& s = first ? a.s0 : a.s1;
    const int cl = first ? b0 : b0 - a.C0;
    // dY unit (round rd): pixel = (rd * 256 + ltid) >> 4, channel quad = ltid & 15
    const int dcu = ltid & 15;
    const bool dok = a0 + dcu * 4 < a.Cout;
    // X unit (round rd): patch pixel = (rd * 256 + ltid) >> 3, channel quad = ltid & 7
    const int xcu = ltid & 7;
    f4 xsc = {1.f, 1.f, 1.f, 1.f}, xsh = {0.f, 0.f, 0.f, 0.f};
    const bool xhas = s.scale != nullptr;
    if (xhas) { xsc = *(const f4*)(s.scale + cl + xcu * 4); xsh = *(const f4*)(s.shift + cl + xcu * 4); }
    const int xrelu = s.relu;
    struct Stage { f4 dv[kWDyRounds]; f4 xv[kWXRounds]; unsigned xok; };
    // stage-invariant pieces of the unit addresses (the loaders are the critical role: every integer operation taken out of
    // stage_load / stage_store counts): dY pixel offsets and LDS positions, X patch coordinates and LDS positions
    int dgo[kWDyRounds], dlo[kWDyRounds], xpy[kWXRounds], xlo[kWXRounds];
#pragma unroll
    for (int rd = 0; rd < kWDyRounds; ++rd) {
      const int px = (rd * kWLT + ltid) >> 4;
      dgo[rd] = ((px / kWX) * a.Wo + (px % kWX)) * a.Cout;
      dlo[rd] = off_dy(px, dcu >> 1) + (dcu & 1) * 8;
    }
#pragma unroll
    for (int rd = 0; rd < kWXRounds; ++rd) {
      const int pp = (rd * kWLT + ltid) >> 3;
      const bool act = pp < kWPH * kWPW;
      const int py = act ? pp / kWPW : 0, px = act ? pp - py * kWPW : 0;
      xpy[rd] = act ? ((py << 8) | px) : -1;
      xlo[rd] = pp * kWXS + xcu * 8;
    }
    const float* const dyp = a.dy + a0 + dcu * 4;
    const float* const xp = s.ptr + cl + xcu * 4;
    auto stage_load = [&](int st, Stage& sg) {
      int q = st;
      const int tx = q % tilesX; q /= tilesX;
      const int ty = q % tilesY; const int n = q / tilesY;
      const int y0 = ty * kWR, x0 = tx * kWX;
      const float* const dbase = dyp + (((size_t)n * a.Ho + y0) * a.Wo + x0) * a.Cout;
#pragma unroll
      for (int rd = 0; rd < kWDyRounds; ++rd) sg.dv[rd] = dok ? *(const f4*)(dbase + dgo[rd]) : (f4){0.f, 0.f, 0.f, 0.f};
      sg.xok = 0;
      const int nb = n * s.H;
#pragma unroll
      for (int rd = 0; rd < kWXRounds; ++rd) {
        const int hl = y0 - 1 + (xpy[rd] >> 8), wl = x0 - 1 + (xpy[rd] & 255);
        const bool ok = xpy[rd] >= 0 && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
        const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
        sg.xv[rd] = *(const f4*)(xp + (size_t)(((nb + (hc >> s.up)) * s.W + (wc >> s.up))) * s.C);
        sg.xok |= (ok ? 1u : 0u) << rd;
      }
    };
    auto split_store = [&](char* dst_hi, char* dst_lo, f4 v, bool clamp) {      // (dY times xs is below 2^14 by construction: no clamp)
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = clamp ? clamp_hw(v[e]) : v[e];
        const _Float16 h = (_Float16)x;
        hi[e] = h; lo[e] = (_Float16)(x - (float)h);
      }
      *(h4*)dst_hi = hi; *(h4*)dst_lo = lo;
    };
    auto stage_store = [&](int buf, const Stage& sg) {
      char* const dyb = wsm + buf * kWStage;
      char* const xb = dyb + kWDyB;
#pragma unroll
      for (int rd = 0; rd < kWDyRounds; ++rd) split_store(dyb + dlo[rd], dyb + (dlo[rd] ^ 128), sg.dv[rd] * xs, false);      // lo plane: chunk + 8 (bit 7 of the swizzled offset)
#pragma unroll
      for (int rd = 0; rd < kWXRounds; ++rd) {
        if (xpy[rd] >= 0) {
          f4 v = sg.xv[rd];
          if (xhas) {
            v = v * xsc + xsh;
            if (xrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          }
          if (!((sg.xok >> rd) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
          split_store(xb + xlo[rd], xb + xlo[rd] + 64, v, true);
        }
      }
    };
    Stage sA, sB;
    const int ns = st1 - st0;
    if (ns > 0) {
      stage_load(st0, sA);
      stage_load(ns > 1 ? st0 + 1 : st0, sB);
      stage_store(0, sA);
    }
    __syncthreads();
    constexpr bool skip = (UWM_WG16_ABL & 4) != 0;
    for (int i = 0; i < ns; i += 2) {
      {                                                  // stage i is being multiplied; stage i+1 (in sB) goes to buffer 1
        if (i + 1 < ns && !(skip && i > 0)) stage_store(1, sB);
        if (!skip) stage_load(st0 + (i + 2 < ns ? i + 2 : ns - 1), sA);
        __syncthreads();
      }
      if (i + 1 < ns) {
        if (i + 2 < ns && !skip) stage_store(0, sA);
        if (!skip) stage_load(st0 + (i + 3 < ns ? i + 3 : ns - 1), sB);
        __syncthreads();
      }
    }
  } else {
    // ================= MMA waves: (output-channel pair cp, input-channel fragment bf) =================
    const int cp = wave & 1, bf = wave >> 1;
    const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    // dY fragment addresses (bytes inside a dY image) for pixel row 0 of the stage: rows 8*kg + q (+4), chunks 2*(2cp+cf) + (p>>1) (+8 lo)
    int dyo[2][2];                                       // [cf][read]; the lo plane is the same address with bit 7 flipped (chunk + 8)
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int rdx = 0; rdx < 2; ++rdx)
        dyo[cf][rdx] = off_dy(8 * kg + 4 * rdx + q, 2 * (2 * cp + cf) + (p >> 1)) + 8 * (p & 1);
    // X fragment base (bytes inside an X image): patch pixel of k index 8*kg + q of k-step 0 (row (8kg) / WX, column (8kg) % WX + q), chunk 2*bf + (p>>1)
    const int xo = (((8 * kg) / kWX) * kWPW + (8 * kg) % kWX + q) * kWXS + (2 * bf + (p >> 1)) * 16 + 8 * (p & 1);
    const int ns = st1 - st0;
    __syncthreads();
    for (int i = 0; i < ns; ++i) {
      const char* const dyb = wsm + (i & 1) * kWStage;
      const char* const xb = dyb + kWDyB;
#pragma unroll
      for (int kr = 0; kr < kKS; ++kr) {
        h8 ah[2], al[2];
#pragma unroll
        for (int cf = 0; cf < 2; ++cf) {
          ah[cf] = tr_pair(dyb + kr * 32 * 256, dyo[cf][0], dyo[cf][1]);
          al[cf] = tr_pair(dyb + kr * 32 * 256, dyo[cf][0] ^ 128, dyo[cf][1] ^ 128);
        }
        // Iteration it: the X fragments of tap it+1 are read (one tap ahead of their first use), hh of tap it and hl, lh of tap
        // it-1 are issued, interleaved over the two channel fragments so that the three products of a tile — they update the SAME
        // accumulator — sit at least two issue slots apart.  The fence keeps hipcc from hoisting every tap's reads to the top
        // of the k-step (144 VGPRs of fragments: spills).
        h8 xh[3], xl[3];
        auto x_read = [&](int t) {
          const int r = t / 3, sx = t % 3;
          const char* xp = xb + ((kr * kRK + r) * kWPW + sx) * kWXS + xo;
          xh[t % 3] = (UWM_WG16_ABL & 2) ? ah[0] : tr_pair(xp, 0, 4 * kWXS);
          xl[t % 3] = (UWM_WG16_ABL & 2) ? al[1] : tr_pair(xp, 64, 4 * kWXS + 64);
        };
        x_read(0);
#pragma unroll
        for (int it = 0; it < 10; ++it) {
          if (it + 1 < 9) x_read(it + 1);
          if (UWM_WG16_ABL & 1) { if (it < 9) { acc[0][it][0] += (float)xh[it % 3][0] + (float)xl[it % 3][1]; acc[1][it][1] += (float)ah[1][2] + (float)al[0][3]; } continue; }
          const int pt = it - 1, ps = (it + 2) % 3, cs = it % 3;
          if (it < 9) acc[0][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], xh[cs], acc[0][it], 0, 0, 0);
          if (NP >= 2 && it >= 1) acc[0][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0], xl[ps], acc[0][pt], 0, 0, 0);
          if (it < 9) acc[1][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], xh[cs], acc[1][it], 0, 0, 0);
          if (NP >= 2 && it >= 1) acc[1][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1], xl[ps], acc[1][pt], 0, 0, 0);
          if (NP >= 3 && it >= 1) acc[0][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[0], xh[ps], acc[0][pt], 0, 0, 0);
          if (NP >= 3 && it >= 1) acc[1][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[1], xh[ps], acc[1][pt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }
    // ---- D[row = co 4*(lane>>4) + reg][col = ci lane & 15] of tile (cf, tap) -> this split's partial image (or dW itself)
    const float ixs = 1.f / xs;
    float* const dst = a.nsplit > 1 ? a.part + (size_t)split * a.wrows * a.Kpad : a.dw;
    const int ci = b0 + 16 * bf + (lane & 15);
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = a0 + 16 * (2 * cp + cf) + 4 * (lane >> 4) + e;
          if (row < a.wrows && ci < a.Ctot) {
            float* pd = dst + (size_t)row * a.Kpad + t * a.Ctot + ci;
            const float v = acc[cf][t][e] * ixs;
            if (UWM_WG16_ABL & 8) { if (v == 123.456f) *pd = v; }      // (timing ablation: no partial-image stores)
            else if (a.nsplit > 1) *pd = v; else *pd += v;
          }
        }
  }
}

// 3x3 / stride 1 / pad 1, input channels in whole 32-channel tiles on either side of the concat, the image tiled by whole
// 4 x 32-pixel or 8 x 16-pixel stages, dY scaled through a.xmax
static int wgrad_f16x3_wx(const WgradArgs& a) {
  if ((a.Wo % 32) == 0 && (a.Ho % 4) == 0) return 32;
  if ((a.Wo % 16) == 0 && (a.Ho % 8) == 0) return 16;
  return 0;
}
bool wgrad_f16x3_applicable(const WgradArgs& a) {
  return a.xmax != nullptr && a.ntaps == 9 && a.kw == 3 && a.stride == 1 && a.pad == 1 && (a.Ctot & 31) == 0 && (a.C0 & 31) == 0 &&
         (a.Cout & 3) == 0 && a.wrows <= a.Cout && a.Kpad == 9 * a.Ctot && a.Hl == a.Ho && a.Wl == a.Wo &&
         wgrad_f16x3_wx(a) != 0 && (a.s0.C & 3) == 0 && (a.s1.C & 3) == 0 && a.Wo < 256 * 128 &&
         (size_t)a.N * a.s0.H * a.s0.W < (1ull << 31) && (size_t)a.N * a.s1.H * a.s1.W < (1ull << 31) && (size_t)8 * a.Wo * a.Cout < (1ull << 31);
}

template <int WX, int NP>
static hipError_t launch_wg16p(WgradArgs& a, hipStream_t st, int pairs, int sps, int nstages) {
  const size_t lds = (size_t)2 * WG16<WX>::kWStage;
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_f16x3_kernel<WX, NP>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(43, a.flops, a.bytes, (wgrad_f16x3_kernel<WX, NP>), dim3((unsigned)(pairs * a.nsplit)), dim3(768), lds, st, a, sps, nstages);
  return hipGetLastError();
}
template <int WX>
static hipError_t launch_wg16(WgradArgs& a, hipStream_t st, int pairs, int sps, int nstages) {
  const int np = (a.nprod >= 1 && a.nprod <= 3) ? a.nprod : 3;
  return np == 3 ? launch_wg16p<WX, 3>(a, st, pairs, sps, nstages) : np == 2 ? launch_wg16p<WX, 2>(a, st, pairs, sps, nstages) : launch_wg16p<WX, 1>(a, st, pairs, sps, nstages);
}

hipError_t launch_wgrad_f16x3(const WgradArgs& a0, hipStream_t st) {
  WgradArgs a = a0;
  if (!wgrad_f16x3_applicable(a)) return hipErrorInvalidValue;
  const int wx = wgrad_f16x3_wx(a);
  const int pairs = ((a.wrows + 63) / 64) * (a.Ctot / 32);
  const int nstages = a.N * (a.Ho / (128 / wx)) * (a.Wo / wx);
  // one 768-thread workgroup per CU; at least 4 stages per split (the two-stage prefetch needs a few to pay)
  // one 768-thread workgroup holds a CU's registers almost alone (3 x 138 of 512 per SIMD): beside the dependent chain of the
  // backward — whose HBM-bound BatchNorm pass then gets one wave per SIMD instead of six — the launch is sized for 3/4 of the
  // CUs (a.cu_share = 3; measured 1051 -> 1066 img/s; 7/8: 1065, 1/2: 1052)
  static const int cu_cap = dbg_int("UWM_WG16_CUS", 0);
  const int cus = cu_cap > 0 ? cu_cap : (a.cu_share > 0 ? device_cu_count() * a.cu_share / 4 : device_cu_count());
  int nsplit = (cus + pairs - 1) / pairs;
  if (nsplit > nstages / 4) nsplit = nstages / 4;
  if (nsplit < 1) nsplit = 1;
  const size_t image = (size_t)a.wrows * a.Kpad;
  if (nsplit > 1 && (!a.part || a.part_floats < 2 * image)) { a.part = wgrad_op_scratch(); a.part_floats = wgrad_wino_scratch_floats(); a.rq = nullptr; }
  if (nsplit > 1 && (!a.part || (size_t)nsplit * image > a.part_floats)) nsplit = a.part ? (int)(a.part_floats / image) : 1;
  if (nsplit < 1) nsplit = 1;
  const int sps = (nstages + nsplit - 1) / nsplit;
  nsplit = (nstages + sps - 1) / sps;
  a.nsplit = nsplit; a.msplit = sps;
  hipError_t e = wx == 32 ? launch_wg16<32>(a, st, pairs, sps, nstages) : launch_wg16<16>(a, st, pairs, sps, nstages);
  if (e != hipSuccess) return e;
  if (nsplit > 1) return launch_wgrad_reduce(a.part, nsplit, image / 4, a.dw, st, a.rq);
  return hipSuccess;
}

}  // namespace uwm
