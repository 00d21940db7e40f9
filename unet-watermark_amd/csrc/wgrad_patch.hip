// Weight gradient of a 3x3 / stride 1 / pad 1 convolution for gfx950 on v_mfma_f32_16x16x4_f32 with
// the input PATCH and the dY tile resident in LDS.
//
//   dW[co][tap][c] = sum over pixels  dY[pixel][co] * X[pixel + tap][c]
//
// A workgroup (512 threads, 8 waves) owns TA output channels x one 32-channel input chunk x ALL nine
// taps and walks a range of 8x16-pixel tiles.  Per tile it stages dY[128 px][TA] and the 10x18-pixel
// input patch (lazy BatchNorm+ReLU, nearest x2 upsample, concat, zero padding applied once) in LDS,
// double-buffered, and every wave accumulates 9 taps x (16 co x 32 c) in registers (72 VGPRs): a
// dY fragment is read once per 4 pixels and reused by the 18 MFMAs of the nine taps, the tap shift is
// only an LDS address offset.  Versus the flattened wgrad (wgrad_igemm.hip) the gathered-X traffic and
// the loader's index/transform work per MFMA drop ~9x, and small Cout (16/32) wastes nothing:
// waves split the output channels (TA/16 ways) and the tile's pixels (8/(TA/16) ways).
// Partial sums are combined across workgroups (and pixel-split waves) with fp32 atomics.
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kTH = 8, kTW = 16, kPW = kTW + 2, kPP = (kTH + 2) * kPW;   // 180 patch pixels
constexpr int kPUnits = kPP * 8;                                         // 1440 16-byte units
constexpr int kPRounds = (kPUnits + 511) / 512;                          // 3

// CW = input channels per workgroup chunk: 32 (two MFMA column tiles: even / odd channels) or 16 (one tile,
// the 16-channel layers at full resolution)
template <int TA, int CW>
__global__ __launch_bounds__(512, 2) void wgrad_patch_kernel(const WgradArgs a) {
  constexpr int WA = TA / 16;            // waves across output channels
  constexpr int WK = 8 / WA;             // waves across the tile's pixels
  constexpr int K4W = 32 / WK;           // 4-pixel steps per wave per tile
  constexpr int YUNITS = 128 * TA / 4;   // 16-byte units of a dY tile
  constexpr int YU = (YUNITS + 511) / 512;
  constexpr int UPR = TA / 4;            // units per dY pixel row
  constexpr int NJ = CW / 16;            // MFMA column tiles (input-channel tiles)
  constexpr int PU = CW / 4;             // 16-byte units per patch pixel
  constexpr int PUNITS = kPP * PU, PROUNDS = (PUNITS + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ys = smem;                      // [2][128][TA]
  float* const Ps = smem + 2 * 128 * TA;       // [2][180][CW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave % WA, wk = wave / WA;
  const int li = lane & 15, lq = lane >> 4;

  // block -> (pixel-tile split, output-channel tile, input-channel chunk)
  const int nchunk = a.Ctot / CW;
  const int tilesA = (a.wrows + TA - 1) / TA;
  const int tilesW = (a.Wo + kTW - 1) / kTW, tilesH = (a.Ho + kTH - 1) / kTH;
  const int ntiles = a.N * tilesH * tilesW;
  int b = blockIdx.x;
  const int cc = b % nchunk; b /= nchunk;
  const int ta = b % tilesA; const int split = b / tilesA;
  const int a0 = ta * TA;
  const int t0 = split * a.msplit, t1 = min(ntiles, t0 + a.msplit);   // msplit = tiles per split here

  // ---- per-thread loader constants
  const int chu = tid % PU;                      // patch: channel unit is fixed per thread (512 % PU == 0)
  const int c = cc * CW + chu * 4;
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

  f4 acc[9][NJ];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[t][j] = (f4){0.f, 0.f, 0.f, 0.f};

  f4 yv[YU], pv[PROUNDS];
  unsigned pok = 0;

  auto tile_load = [&](int t, bool enable) {
    const int tw = t % tilesW; const int q = t / tilesW;
    const int th = q % tilesH; const int n = q / tilesH;
    const int h0 = th * kTH, w0 = tw * kTW;
#pragma unroll
    for (int i = 0; i < YU; ++i) {
      const int uu = tid + 512 * i;
      const int pix = uu / UPR, cu = uu - pix * UPR;
      const int ho = h0 + (pix >> 4), wo = w0 + (pix & 15);
      const int co = a0 + cu * 4;
      const bool v = enable && uu < YUNITS && ho < a.Ho && wo < a.Wo && co < a.Cout;
      yv[i] = v ? *(const f4*)(a.dy + ((size_t)((size_t)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    pok = 0;
#pragma unroll
    for (int rd = 0; rd < PROUNDS; ++rd) {
      const int u = rd * 512 + tid;
      const int pp = u / PU;
      const int py = pp / kPW, px = pp - py * kPW;
      const int hl = h0 - 1 + py, wl = w0 - 1 + px;
      const bool v = enable && u < PUNITS && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
      const float* p = sp + ((size_t)((size_t)n * sH + (hl >> sup)) * sW + (wl >> sup)) * sC + cl;
      pv[rd] = v ? *(const f4*)p : (f4){0.f, 0.f, 0.f, 0.f};
      pok |= (v ? 1u : 0u) << rd;
    }
  };
  auto tile_store = [&](int buf) {
    float* ys = Ys + buf * 128 * TA;
    float* ps = Ps + buf * kPP * CW;
#pragma unroll
    for (int i = 0; i < YU; ++i) {
      const int uu = tid + 512 * i;
      if (uu < YUNITS) {
        const int pix = uu / UPR, cu = uu - pix * UPR;
        const int su = (TA >= 32) ? (cu ^ ((pix & 1) << 2)) : cu;
        *(f4*)(ys + pix * TA + su * 4) = yv[i];
      }
    }
#pragma unroll
    for (int rd = 0; rd < PROUNDS; ++rd) {
      const int u = rd * 512 + tid;
      if (u < PUNITS) {
        const int pp = u / PU;
        f4 v = pv[rd];
        if (thas && ((pok >> rd) & 1u)) {
          v = v * tsc + tsh;
          if (trelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        *(f4*)(ps + pp * CW + (chu << 2)) = v;      // natural layout: b64 (CW=32) / b32 (CW=16) fragment reads are conflict-free
      }
    }
  };

  if (t0 < t1) {
    tile_load(t0, true);
    tile_store(0);
  }
  __syncthreads();

  const int dbg = a.force_igemm >> 8;                        // timing-only ablation bits (results are garbage when set)
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    tile_load(t + 1 < t1 ? t + 1 : t, (t + 1 < t1) && !(dbg & 1));          // unconditional, lane-masked prefetch
    const float* ys = Ys + cur * 128 * TA;
    const float* ps = Ps + cur * kPP * CW;
    // Fragment reads run ONE k4-step ahead of the MFMAs (explicit register double buffer + scheduling
    // fences): hipcc otherwise emits read -> lgkmcnt(0) -> 2 MFMAs per tap and exposes the LDS latency
    // nine times per step.  MFMA tile 0 takes the EVEN channels (column li <-> channel 2*li), tile 1 the
    // ODD ones, so both B fragments of a tap are one 8-byte read at an immediate offset from one base
    // (lanes 0-15 cover a pixel's 128 B, lanes 16-31 the next pixel: conflict-free, no swizzle).
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 bb[2][9]; float af[2];
    auto frag_read = [&](int kk, int slot) {
      const int k4 = wk * K4W + kk;
      const int ty = k4 >> 2, tx = (k4 & 3) * 4 + lq;       // this lane's pixel (k index = lq)
      const int pix = ty * 16 + tx;
      const int cof = wa * 16 + li;
      af[slot] = ys[pix * TA + ((TA >= 32) ? (cof ^ ((pix & 1) << 4)) : cof)];
      if (NJ == 2) {
        const float* P0 = ps + (ty * kPW + tx) * CW + 2 * li;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s = 0; s < 3; ++s) bb[slot][r * 3 + s] = *(const f2*)(P0 + (r * kPW + s) * CW);
      } else {                                   // CW = 16: column li <-> channel li, one 4-byte read per tap
        const float* P0 = ps + (ty * kPW + tx) * CW + li;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s = 0; s < 3; ++s) bb[slot][r * 3 + s].x = P0[(r * kPW + s) * CW];
      }
    };
    frag_read(0, 0);
#pragma unroll
    for (int kk = 0; kk < K4W; ++kk) {
      const int cb = kk & 1;
      if (kk + 1 < K4W && !(dbg & 8)) frag_read(kk + 1, cb ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        acc[t9][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cb], bb[cb][t9].x, acc[t9][0], 0, 0, 0);
        if (NJ == 2) acc[t9][NJ - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cb], bb[cb][t9].y, acc[t9][NJ - 1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);   // consumers of the prefetch stay below the MFMA block
    if (!(dbg & 2)) tile_store(cur ^ 1);
    if (!(dbg & 4)) __syncthreads();
  }
  if (dbg & 16) return;

  // ---- combine the 8 waves' partial tiles in LDS (ds_add_f32), then ONE coalesced pass of global atomics
  // tile layout in LDS: [TA co][9 taps][32 c]   (the staging buffers are dead after the last barrier)
  float* const Rt = smem;
  constexpr int RT = TA * 9 * CW;
  // pixel-split wave 0 stores its tile, the other WK-1 waves add theirs (no zeroing pass)
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = wa * 16 + lq * 4 + e;                 // D row = output channel
        if (wk == 0) Rt[co * 9 * CW + t * CW + NJ * li + j] = acc[t][j][e];   // D column li -> channel NJ*li + j
      }
  __syncthreads();
  if (WK == 2) {
    // exactly one other wave owns the same (co, tap, c) elements: plain read-add-write, no LDS atomics
    if (wk == 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int co = wa * 16 + lq * 4 + e;
            Rt[co * 9 * CW + t * CW + NJ * li + j] += acc[t][j][e];
          }
    }
  } else if (wk != 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = wa * 16 + lq * 4 + e;
          atomicAdd(Rt + co * 9 * CW + t * CW + NJ * li + j, acc[t][j][e]);
        }
  }
  __syncthreads();
  for (int i = tid; i < RT; i += 512) {
    const int co = i / (9 * CW), rem = i - co * (9 * CW);
    const int t = rem / CW, cch = rem - t * CW;
    const int row = a0 + co;
    if (row < a.wrows) atomicAdd(a.dw + (size_t)row * a.Kpad + t * a.Ctot + cc * CW + cch, Rt[i]);
  }
}

template <int TA, int CW>
static hipError_t launch_wp(const WgradArgs& a, hipStream_t st, int cls, int nblocks) {
  size_t lds = (size_t)(2 * 128 * TA + 2 * kPP * CW) * sizeof(float);
  if (lds < (size_t)TA * 9 * CW * sizeof(float)) lds = (size_t)TA * 9 * CW * sizeof(float);
  static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
  { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_patch_kernel<TA, CW>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(cls, a.flops, a.bytes, (wgrad_patch_kernel<TA, CW>), dim3((unsigned)nblocks), dim3(512), lds, st, a);
  return hipGetLastError();
}

bool wgrad_patch_applicable(const WgradArgs& a) {
  const bool c32 = (a.Ctot & 31) == 0 && (a.C0 & 31) == 0;
  const bool c16 = a.Ctot == 16 && a.C0 == 16;                  // single 16-channel source
  return a.ntaps == 9 && a.kw == 3 && a.stride == 1 && a.pad == 1 && (c32 || c16) &&
         a.Hl == a.Ho && a.Wl == a.Wo && a.Ho >= kTH && a.Wo >= kTW;
}

hipError_t launch_wgrad_patch(const WgradArgs& a0, hipStream_t st) {
  WgradArgs a = a0;
  if (!wgrad_patch_applicable(a) || (a.Cout & 3)) return hipErrorInvalidValue;
  const int TA = a.wrows <= 16 ? 16 : (a.wrows <= 32 ? 32 : 64);
  const bool c16 = a.Ctot == 16;
  const int nchunk = c16 ? 1 : a.Ctot >> 5, tilesA = (a.wrows + TA - 1) / TA;
  const int ntiles = a.N * ((a.Ho + kTH - 1) / kTH) * ((a.Wo + kTW - 1) / kTW);
  // Pixel-tile split: every workgroup ends with a TAx288 LDS-reduce + global-atomic epilogue worth about E
  // tiles of work, and workgroups run in rounds of `slots` (1 resident per CU for TA=64, 2 otherwise).
  // Pick the split that minimises rounds * (tiles_per_workgroup + E).
  const int cus = device_cu_count();
  const int pairs = nchunk * tilesA;
  const int slots = cus * (TA == 64 ? 1 : 2);
  const double E = TA == 64 ? 2.0 : (TA == 32 ? 1.0 : 0.5);
  int nsplit = 1; double best = 1e30;
  for (int ns = 1; ns <= ntiles && ns <= 1024; ++ns) {
    const int tp = (ntiles + ns - 1) / ns;
    const int nsr = (ntiles + tp - 1) / tp;
    const long blocks = (long)pairs * nsr;
    const long rounds = (blocks + slots - 1) / slots;
    const double cost = (double)rounds * (tp + E);
    if (cost < best - 1e-9) { best = cost; nsplit = nsr; }
  }
  if (nsplit > ntiles) nsplit = ntiles;
  if (nsplit < 1) nsplit = 1;
  int tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  a.nsplit = nsplit; a.msplit = tps;
  const int nblocks = nsplit * tilesA * nchunk;
  if (c16) return TA == 16 ? launch_wp<16, 16>(a, st, 14, nblocks) : (TA == 32 ? launch_wp<32, 16>(a, st, 15, nblocks) : launch_wp<64, 16>(a, st, 16, nblocks));
  switch (TA) {
    case 16: return launch_wp<16, 32>(a, st, 14, nblocks);
    case 32: return launch_wp<32, 32>(a, st, 15, nblocks);
    default: return launch_wp<64, 32>(a, st, 16, nblocks);
  }
}

}  // namespace uwm
