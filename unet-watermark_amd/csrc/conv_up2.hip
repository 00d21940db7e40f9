// 3x3 / stride 1 / pad 1 convolution over a NEAREST-x2-UPSAMPLED 32-channel input with 16 output channels, for gfx950
// (v_mfma_f32_16x16x4_f32): decoder block 4 conv1 of every smp.Unet / UnetPlusPlus (32 -> 16 channels at full
// resolution, no skip), its dgrad and its wgrad.
//
// Sub-pixel decomposition.  With U[y][x] = X[y>>1][x>>1], output pixel (2i+a, 2j+b) only ever sees the 2x2 low-resolution
// neighbourhood rows {i-1+a, i+a} x cols {j-1+b, j+b}: three taps collapse onto two source rows,
//     a = 0:  row i-1 <- W[0]          row i   <- W[1] + W[2]
//     a = 1:  row i   <- W[0] + W[1]   row i+1 <- W[2]                  (same for columns with b)
// so each of the four output parity classes is a 2x2 convolution on X with pre-summed filters: 16 tap-MACs per four
// outputs instead of 36 — the 2.25x of Winograd F(2x2,3x3) with NO input/output transform, which is what the Winograd
// kernel cannot amortise over 16 output channels (conv_wino_kernel<1>: 375 us, 46 TFLOP/s executed on this layer).
// The zero padding of the upsampled image coincides with zero padding of X (U[-1] = 0 <=> X[-1] = 0).
//
// Forward (conv_up2_kernel): a tile = 8x16 low-resolution pixels (16x32 outputs) of one image: its 10x18x32 patch (lazy
// BatchNorm+ReLU applied while staging) sits in LDS next to the raw [16][288] filter; per class the wave sums its A
// fragments out of the raw filter (9 ds_read_b128 x 2 channel groups — every tap exactly once) and runs
// 2 pixel rows x 4 class taps x 8 MFMAs.  Persistent workgroups, double-buffered patch, 64.8 KB of LDS: two per CU.
//
// Reference semantics replaced: F.interpolate(scale_factor=2, mode="nearest") + Conv2dReLU's conv of smp's
// DecoderBlock (/root/reference/src/models/unet_model.py:64-71 -> smp).
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kUH = 8, kUW = 16;                                   // low-resolution pixels per workgroup
constexpr int kUPH = kUH + 2, kUPW = kUW + 2, kUPP = kUPH * kUPW;  // 10 x 18 = 180 patch pixels
constexpr int kUC = 32;                                            // input channels
constexpr int kUWLd = 9 * kUC + 4;                                 // raw filter row stride (73 16-byte units: odd)

// tap sets of (parity a, class tap d): r in [kR0[a][d], kR1[a][d]]
__device__ __forceinline__ constexpr int up2_r0(int a, int d) { return a == 0 ? (d == 0 ? 0 : 1) : (d == 0 ? 0 : 2); }
__device__ __forceinline__ constexpr int up2_r1(int a, int d) { return a == 0 ? (d == 0 ? 0 : 2) : (d == 0 ? 1 : 2); }

// Persistent: a workgroup walks tiles bid, bid + grid, ... with the patch double-buffered — the next tile's global loads are
// issued before this tile's MFMAs and land in registers behind them (one barrier per tile; the one-tile-per-workgroup form
// measured 241 us: load -> barrier -> MFMA -> store with nothing of its own to overlap).  The filter is staged once per
// workgroup and the BatchNorm statistics leave as ONE set of atomics per workgroup.
__global__ __launch_bounds__(256, 2) void conv_up2_kernel(const ConvArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Wr = smem;                          // [16 co][292]: the raw 3x3 filter, k = tap*32 + c
  float* const Ps = smem + 16 * kUWLd;             // [2][180 px][32 ch], 16-byte units XOR-swizzled by (px >> 1) & 7

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int Hs = a.s0.H, Wsrc = a.s0.W;
  const int tilesW = Wsrc / kUW, tilesH = Hs / kUH;

  // ---- per-thread staging geometry (tile-invariant): 180 pixels x 8 units = 1440 units, 6 rounds
  const int unit = tid & 7;
  const bool has = a.s0.scale != nullptr;
  f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (has) { sc = *(const f4*)(a.s0.scale + unit * 4); sh = *(const f4*)(a.s0.shift + unit * 4); }
  int spy[6], spx[6], spos[6];
#pragma unroll
  for (int rd = 0; rd < 6; ++rd) {
    const int pp = min((rd * 256 + tid) >> 3, kUPP - 1);
    spy[rd] = pp / kUPW; spx[rd] = pp - spy[rd] * kUPW;
    spos[rd] = pp * kUC + ((unit ^ ((pp >> 1) & 7)) << 2);
  }
  const bool last_live = (5 * 256 + tid) < kUPP * 8;        // round 5 covers only 160 units
  f4 pv[6]; unsigned pok = 0;
  auto tile_origin = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    h0 = th * kUH; w0 = tw * kUW;
  };
  auto patch_load = [&](int t) {
    int n, h0, w0; tile_origin(t, n, h0, w0);
    pok = 0;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const int hl = h0 - 1 + spy[rd], wl = w0 - 1 + spx[rd];
      const bool ok = hl >= 0 && hl < Hs && wl >= 0 && wl < Wsrc;
      const int hc = min(max(hl, 0), Hs - 1), wc = min(max(wl, 0), Wsrc - 1);
      pv[rd] = *(const f4*)(a.s0.ptr + (((size_t)n * Hs + hc) * Wsrc + wc) * kUC + unit * 4);
      pok |= (ok ? 1u : 0u) << rd;
    }
  };
  auto patch_store = [&](int buf) {
    float* const pb_ = Ps + buf * kUPP * kUC;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      f4 v = pv[rd];
      if (has) {
        v = v * sc + sh;
        if (a.s0.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      if (!((pok >> rd) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
      if (rd < 5 || last_live) *(f4*)(pb_ + spos[rd]) = v;
    }
  };

  // ---- stage the raw filter once: 16 rows x 72 units
  for (int u = tid; u < 16 * 72; u += 256) {
    const int row = u / 72, ku = u - row * 72;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < a.wrows) v = *(const f4*)(a.w + (size_t)row * a.Kpad + ku * 4);
    *(f4*)(Wr + row * kUWLd + ku * 4) = v;
  }
  int t = blockIdx.x;
  patch_load(t);
  patch_store(0);
  __syncthreads();

  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = {0.f, 0.f, 0.f, 0.f};
  const int co = lq * 4;
  f4 bias = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias = *(const f4*)(a.bias + co);
  const float* const wl_ = Wr + lrow * kUWLd + lq * 4;

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    patch_load(tn < ntiles ? tn : t);                  // (last tile: harmless re-read)
    int n, h0, w0; tile_origin(t, n, h0, w0);
    const float* const pc = Ps + cur * kUPP * kUC;
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
      f4 acc[2][2];                                    // [pb][rb]
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        acc[pb][0] = acc[pb][1] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dr = 0; dr < 2; ++dr)
#pragma unroll
          for (int ds = 0; ds < 2; ++ds)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              // class filter fragment: the raw taps that land on source offset (dr, ds)
              f4 A = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int r = up2_r0(pa, dr); r <= up2_r1(pa, dr); ++r)
#pragma unroll
                for (int s2 = up2_r0(pb, ds); s2 <= up2_r1(pb, ds); ++s2) A += *(const f4*)(wl_ + (r * 3 + s2) * kUC + g * 16);
              f4 xf[2];
#pragma unroll
              for (int rb = 0; rb < 2; ++rb) {
                const int pp = (wave * 2 + rb + pa + dr) * kUPW + lrow + pb + ds;
                xf[rb] = *(const f4*)(pc + pp * kUC + (((g * 4 + lq) ^ ((pp >> 1) & 7)) << 2));
              }
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[pb][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[e], xf[rb][e], acc[pb][rb], 0, 0, 0);
            }
      }
      // both column parities of this row parity: a wave writes 32 consecutive pixels x 64 bytes per row
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const int ho = 2 * (h0 + wave * 2 + rb) + pa, wo = 2 * (w0 + lrow) + pb;
          const f4 v = acc[pb][rb] + bias;
          *(f4*)(a.out + (((size_t)n * a.Ho + ho) * a.Wo + wo) * 16 + co) = v;
          ps_ += v; pq_ += v * v;
        }
    }
    patch_store(cur ^ 1);
    __syncthreads();
  }

  if (a.ssum != nullptr) {            // BatchNorm statistics: 16 pixel lanes -> 4 waves (LDS) -> fp64 atomics on one replica
    const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sv = ps_[e], qv = pq_[e];
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) { sv += __shfl_xor(sv, d); qv += __shfl_xor(qv, d); }
      ps_[e] = sv; pq_[e] = qv;
    }
    float* red = Ps;                  // [4 waves][16][2]  (the last barrier of the loop has passed)
    if (lrow == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[(wave * 16 + co + e) * 2] = ps_[e]; red[(wave * 16 + co + e) * 2 + 1] = pq_[e]; }
    }
    __syncthreads();
    if (tid < 16 && tid < a.Cout) {
      double sv = 0.0, qv = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { sv += (double)red[(w * 16 + tid) * 2]; qv += (double)red[(w * 16 + tid) * 2 + 1]; }
      atomicAdd(a.ssum + srep_off + tid, sv);
      atomicAdd(a.ssq + srep_off + tid, qv);
    }
  }
}

bool conv_up2_applicable(const ConvArgs& a) {
  return a.rmul == 1 && a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && a.off == -1 && a.s0.up == 1 &&
         a.C0 == a.Ctot && a.Ctot == kUC && a.s0.C == kUC && a.Cout == 16 && a.wrows <= 16 && a.Kpad >= 9 * kUC &&
         a.Ho == 2 * a.s0.H && a.Wo == 2 * a.s0.W && a.Hl == a.Ho && a.Wl == a.Wo && (a.s0.H % kUH) == 0 && (a.s0.W % kUW) == 0 &&
         !a.addend && !a.mask && !a.out_up && !a.bnb_mean;
}

hipError_t launch_conv_up2(const ConvArgs& a, hipStream_t st) {
  if (!conv_up2_applicable(a)) return hipErrorInvalidValue;
  if (a.ig16) return launch_conv_up2_f16(a, st);
  const size_t lds = (size_t)(2 * kUPP * kUC + 16 * kUWLd) * sizeof(float);
  const int ntiles = a.N * (a.s0.H / kUH) * (a.s0.W / kUW);
  const int nwg = ntiles < 2 * device_cu_count() ? ntiles : 2 * device_cu_count();
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_up2_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(34, a.flops, a.bytes, conv_up2_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- dgrad
// Backward of the same layer wrt the LOW-RESOLUTION input (upsample backward fused): a 4x4 / stride-2 / pad-1
// convolution over dY,
//     dX[p][q][c] = sum_{ty,tx in 0..3} sum_co dY[2p-1+ty][2q-1+tx][co] * W4[ty][tx][co][c]
//     W4[t] = sum of the raw taps r in R(t):  R(0) = {2}, R(1) = {1,2}, R(2) = {0,1}, R(3) = {0}   (rows and columns alike)
// — 16 tap-MACs per low-resolution pixel where dgrad-at-full-resolution + 2x2 pooling spends 36 (Winograd: 16 + transforms).
// Epilogue = the fused concat-split contract of conv_wino.hip (ConvArgs::out_up): ReLU mask of the low-resolution producer,
// optional accumulate, optional fused BatchNorm-backward sums.  Tile = 4x16 low-resolution pixels (wave = one row); the
// 10x34x16 dY patch is split into even / odd column planes so the 16 pixel lanes of an MFMA read consecutive 64-byte
// entries (unit index XOR-swizzled by (entry >> 2) & 3); persistent workgroups, double-buffered patch.
constexpr int kDH = 4, kDW = 16;                         // low-resolution pixels per tile
constexpr int kDPH = 2 * kDH + 2, kDPJ = kDW + 1;        // 10 dY rows, 17 entries per column-parity plane
constexpr int kDPlane = kDPJ * 16, kDBuf = kDPH * 2 * kDPlane;      // floats
constexpr int kDWLd = 148;                               // packed filter row stride (37 16-byte units: odd)
__device__ __forceinline__ constexpr int up2_t0(int t) { return t == 0 ? 2 : (t == 1 ? 1 : 0); }
__device__ __forceinline__ constexpr int up2_t1(int t) { return t == 0 ? 2 : (t == 1 ? 2 : (t == 2 ? 1 : 0)); }

__global__ __launch_bounds__(256, 2) void conv_up2_dgrad_kernel(const ConvArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Wd = smem;                          // [32 c][148]: packed dgrad filter, k = tap*16 + co
  float* const Ds = smem + 32 * kDWLd;             // [2][10 rows][2 planes][17][16 ch]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int Hs = a.Ho >> 1, Wsrc = a.Wo >> 1;       // low-resolution dims (a.Ho x a.Wo is the dY / full-resolution grid)
  const int tilesW = Wsrc / kDW, tilesH = Hs / kDH;

  // ---- staging geometry: 10 rows x 34 columns x 4 units = 1360 units, 6 rounds
  const int unit = tid & 3;
  int scy[6], scx[6], spos[6];
#pragma unroll
  for (int rd = 0; rd < 6; ++rd) {
    const int px = min((rd * 256 + tid) >> 2, kDPH * 34 - 1);
    scy[rd] = px / 34; scx[rd] = px - scy[rd] * 34;
    const int j = scx[rd] >> 1;
    spos[rd] = ((scy[rd] * 2 + (scx[rd] & 1)) * kDPJ + j) * 16 + ((unit ^ ((j >> 2) & 3)) << 2);
  }
  const bool last_live = (5 * 256 + tid) < kDPH * 34 * 4;
  f4 pv[6];
  auto tile_origin = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    h0 = th * kDH; w0 = tw * kDW;
  };
  auto patch_load = [&](int t) {
    int n, h0, w0; tile_origin(t, n, h0, w0);
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const int y = 2 * h0 - 1 + scy[rd], x = 2 * w0 - 1 + scx[rd];
      const bool ok = y >= 0 && y < a.Ho && x >= 0 && x < a.Wo;
      const int yc = min(max(y, 0), a.Ho - 1), xc = min(max(x, 0), a.Wo - 1);
      const f4 v = *(const f4*)(a.s0.ptr + (((size_t)n * a.Ho + yc) * a.Wo + xc) * 16 + unit * 4);
      pv[rd] = ok ? v : (f4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto patch_store = [&](int buf) {
    float* const pb_ = Ds + buf * kDBuf;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd)
      if (rd < 5 || last_live) *(f4*)(pb_ + spos[rd]) = pv[rd];
  };

  // ---- stage the packed filter once: 32 rows x 36 units
  for (int u = tid; u < 32 * 36; u += 256) {
    const int row = u / 36, ku = u - row * 36;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < a.wrows) v = *(const f4*)(a.w + (size_t)row * a.Kpad + ku * 4);
    *(f4*)(Wd + row * kDWLd + ku * 4) = v;
  }
  int t = blockIdx.x;
  patch_load(t);
  patch_store(0);
  __syncthreads();

  const bool bnb = a.bnb_mean != nullptr;
  f4 ps_[2], pq_[2], bmu[2], brs[2], msc[2], msh[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int c = cb * 16 + lq * 4;
    ps_[cb] = pq_[cb] = bmu[cb] = brs[cb] = msh[cb] = (f4){0.f, 0.f, 0.f, 0.f};
    msc[cb] = (f4){1.f, 1.f, 1.f, 1.f};
    if (bnb) { bmu[cb] = *(const f4*)(a.bnb_mean + c); brs[cb] = *(const f4*)(a.bnb_rstd + c); }
    if (a.up_mask && a.up_mscale) { msc[cb] = *(const f4*)(a.up_mscale + c); msh[cb] = *(const f4*)(a.up_mshift + c); }
  }

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    patch_load(tn < ntiles ? tn : t);
    int n, h0, w0; tile_origin(t, n, h0, w0);
    const float* const pc = Ds + cur * kDBuf;
    f4 acc[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ty = 0; ty < 4; ++ty)
#pragma unroll
      for (int tx = 0; tx < 4; ++tx) {
        const int j = lrow + (tx >> 1);
        const f4 yf = *(const f4*)(pc + (((2 * wave + ty) * 2 + (tx & 1)) * kDPJ + j) * 16 + ((lq ^ ((j >> 2) & 3)) << 2));
        f4 A[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          A[cb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = up2_t0(ty); r <= up2_t1(ty); ++r)
#pragma unroll
            for (int s2 = up2_t0(tx); s2 <= up2_t1(tx); ++s2)
              A[cb] += *(const f4*)(Wd + (cb * 16 + lrow) * kDWLd + (r * 3 + s2) * 16 + lq * 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cb][e], yf[e], acc[cb], 0, 0, 0);
      }
    // epilogue: pixel (h0 + wave, w0 + lrow), channels cb*16 + 4*lq ..
    const size_t o2 = (((size_t)n * Hs + h0 + wave) * Wsrc + w0 + lrow) * 32 + lq * 4;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f4 v = acc[cb];
      if (a.up_mask) {
        const f4 yr = *(const f4*)(a.up_mask + o2 + cb * 16);
        const f4 mk = yr * msc[cb] + msh[cb];
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        if (bnb) { ps_[cb] += v; pq_[cb] += v * ((yr - bmu[cb]) * brs[cb]); }
      }
      if (a.up_accum) v += *(const f4*)(a.out_up + o2 + cb * 16);
      *(f4*)(a.out_up + o2 + cb * 16) = v;
    }
    patch_store(cur ^ 1);
    __syncthreads();
  }

  if (bnb) {                          // fused BatchNorm-backward sums: 16 pixel lanes -> 4 waves (LDS) -> fp64 atomics on one replica
    const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sv = ps_[cb][e], qv = pq_[cb][e];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { sv += __shfl_xor(sv, d); qv += __shfl_xor(qv, d); }
        ps_[cb][e] = sv; pq_[cb][e] = qv;
      }
    float* red = Ds;                  // [4 waves][32][2]
    if (lrow == 0) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = cb * 16 + lq * 4 + e;
          red[(wave * 32 + c) * 2] = ps_[cb][e]; red[(wave * 32 + c) * 2 + 1] = pq_[cb][e];
        }
    }
    __syncthreads();
    if (tid < 32) {
      double sv = 0.0, qv = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { sv += (double)red[(w * 32 + tid) * 2]; qv += (double)red[(w * 32 + tid) * 2 + 1]; }
      atomicAdd(a.ssum + srep_off + tid, sv);
      atomicAdd(a.ssq + srep_off + tid, qv);
    }
  }
}

bool conv_up2_dgrad_applicable(const ConvArgs& a) {
  return a.rmul == -1 && a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && a.off == 1 && a.out_up != nullptr &&
         a.up_c0 == 32 && a.Cout == 32 && a.wrows <= 32 && a.Ctot == 16 && a.C0 == 16 && a.s0.C == 16 && a.s0.up == 0 && a.Kpad >= 144 &&
         a.Hl == a.Ho && a.Wl == a.Wo && a.s0.H == a.Ho && a.s0.W == a.Wo && (a.Ho % (2 * kDH)) == 0 && (a.Wo % (2 * kDW)) == 0 &&
         !a.addend && !a.mask && !a.bias && !a.s0.scale;
}

hipError_t launch_conv_up2_dgrad(const ConvArgs& a, hipStream_t st) {
  if (!conv_up2_dgrad_applicable(a)) return hipErrorInvalidValue;
  if (a.ig16) return launch_conv_up2_dgrad_f16(a, st);
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !a.up_mask || a.up_accum)) return hipErrorInvalidValue;
  const size_t lds = (size_t)(32 * kDWLd + 2 * kDBuf) * sizeof(float);
  const int ntiles = a.N * ((a.Ho >> 1) / kDH) * ((a.Wo >> 1) / kDW);
  const int nwg = ntiles < 2 * device_cu_count() ? ntiles : 2 * device_cu_count();
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_up2_dgrad_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(35, a.flops, a.bytes, conv_up2_dgrad_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- wgrad
// dW[co][r][s][c] = sum dY[y][x][co] * U[y+r-1][x+s-1][c] in the same sub-pixel form: with y = 2i+a the row y+r-1 of the
// upsampled image is low-resolution row i-1+py, py = a + dpy, dpy = (a == 0 ? r >= 1 : r == 2) (columns alike), so
//     P[a][dpy][b][dpx][co][c] = sum_{n,i,j} dY[2i+a][2j+b][co] * X~[i-1+a+dpy][j-1+b+dpx][c]        (16 products)
//     dW[r][s] = sum_{a,b} P[a][dpy(a,r)][b][dpx(b,s)]
// — 16 pixel-MACs per four output pixels instead of 36.  Pixels are the MFMA reduction dimension: a wave owns one
// low-resolution row of the 4x16 tile (4 K-groups of 4 pixels), reads 4 dY operands (one per class) and 9 x 2 X~ operands
// (3x3 neighbourhood x two 16-channel blocks) per group with conflict-free ds_read_b32 and feeds 32 MFMAs; the 32
// accumulators (128 VGPRs) live across ALL tiles of the persistent workgroup.  Waves are summed through LDS in a fixed
// order, one partial per workgroup goes to the scratch, wgrad_up2_reduce_kernel folds partials and classes into dW in a
// fixed order (bit-reproducible).
constexpr int kGH = 4, kGW = 16;
constexpr int kGXP = (kGH + 2) * (kGW + 2);              // 108 patch pixels
constexpr int kGXPlane = kGXP * 16;                      // floats per 16-channel plane of the X~ patch
constexpr int kGYJ = kGW;                                // entries per dY column-parity plane
constexpr int kGYBuf = 2 * kGH * 2 * kGYJ * 16;          // [8 rows][2 parities][16][16 co]
constexpr int kGBuf = 2 * kGXPlane + kGYBuf;             // floats per stage buffer
constexpr int kGPart = 16 * 2 * 64 * 4;                  // floats per workgroup partial: [16 products][2 cb][64 lanes][4]

__global__ __launch_bounds__(256, 2) void wgrad_up2_kernel(const WgradArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int Hs = a.s0.H, Wsrc = a.s0.W;
  const int tilesW = Wsrc / kGW, tilesH = Hs / kGH;

  // ---- staging geometry.  X~: 108 px x 8 units = 864 units (4 rounds); dY: 8 rows x 32 cols x 4 units = 1024 units (4 rounds)
  const int xunit = tid & 7, yunit = tid & 3;
  const bool has = a.s0.scale != nullptr;
  f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (has) { sc = *(const f4*)(a.s0.scale + xunit * 4); sh = *(const f4*)(a.s0.shift + xunit * 4); }
  int xpy[4], xpx[4], xpos[4], ypos[4], yy[4], yx[4];
#pragma unroll
  for (int rd = 0; rd < 4; ++rd) {
    const int pp = min((rd * 256 + tid) >> 3, kGXP - 1);
    xpy[rd] = pp / (kGW + 2); xpx[rd] = pp - xpy[rd] * (kGW + 2);
    xpos[rd] = (xunit >> 2) * kGXPlane + pp * 16 + (xunit & 3) * 4;
    const int q = (rd * 256 + tid) >> 2;                 // dY pixel of the 8 x 32 tile
    yy[rd] = q >> 5; yx[rd] = q & 31;
    ypos[rd] = 2 * kGXPlane + ((yy[rd] * 2 + (yx[rd] & 1)) * kGYJ + (yx[rd] >> 1)) * 16 + yunit * 4;
  }
  const bool xlast = (3 * 256 + tid) < kGXP * 8;
  f4 xv[4], yv[4]; unsigned xok = 0;
  auto tile_origin = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    h0 = th * kGH; w0 = tw * kGW;
  };
  auto stage_load = [&](int t) {
    int n, h0, w0; tile_origin(t, n, h0, w0);
    xok = 0;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const int hl = h0 - 1 + xpy[rd], wl = w0 - 1 + xpx[rd];
      const bool ok = hl >= 0 && hl < Hs && wl >= 0 && wl < Wsrc;
      const int hc = min(max(hl, 0), Hs - 1), wc = min(max(wl, 0), Wsrc - 1);
      xv[rd] = *(const f4*)(a.s0.ptr + (((size_t)n * Hs + hc) * Wsrc + wc) * 32 + xunit * 4);
      xok |= (ok ? 1u : 0u) << rd;
      yv[rd] = *(const f4*)(a.dy + (((size_t)n * a.Ho + 2 * h0 + yy[rd]) * a.Wo + 2 * w0 + yx[rd]) * 16 + yunit * 4);
    }
  };
  auto stage_store = [&](int buf) {
    float* const b_ = smem + buf * kGBuf;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      f4 v = xv[rd];
      if (has) {
        v = v * sc + sh;
        if (a.s0.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      if (!((xok >> rd) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
      if (rd < 3 || xlast) *(f4*)(b_ + xpos[rd]) = v;
      *(f4*)(b_ + ypos[rd]) = yv[rd];
    }
  };

  f4 acc[16][2];
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p][0] = acc[p][1] = (f4){0.f, 0.f, 0.f, 0.f};

  int t = blockIdx.x;
  stage_load(t);
  stage_store(0);
  __syncthreads();

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    stage_load(tn < ntiles ? tn : t);
    const float* const xs = smem + cur * kGBuf;
    const float* const ys = xs + 2 * kGXPlane;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {                     // 4 low-resolution pixels (wave, 4*kg + lq) per MFMA
      const int j = kg * 4 + lq;
      float yf[2][2];                                    // [a][b]: dY[2*wave + a][2*j + b][co = lrow]
#pragma unroll
      for (int pa = 0; pa < 2; ++pa)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) yf[pa][pb] = ys[(((2 * wave + pa) * 2 + pb) * kGYJ + j) * 16 + lrow];
      float xf[3][3][2];                                 // [py][px][cb]: X~[wave - 1 + py][j - 1 + px][c = cb*16 + lrow]
#pragma unroll
      for (int py = 0; py < 3; ++py)
#pragma unroll
        for (int px = 0; px < 3; ++px)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) xf[py][px][cb] = xs[cb * kGXPlane + ((wave + py) * (kGW + 2) + j + px) * 16 + lrow];
#pragma unroll
      for (int pa = 0; pa < 2; ++pa)
#pragma unroll
        for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int dx_ = 0; dx_ < 2; ++dx_)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) {
                const int p = ((pa * 2 + dy_) * 2 + pb) * 2 + dx_;
                acc[p][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(yf[pa][pb], xf[pa + dy_][pb + dx_][cb], acc[p][cb], 0, 0, 0);
              }
    }
    stage_store(cur ^ 1);
    __syncthreads();
  }

  // ---- waves 1, 3 -> LDS -> waves 0, 2 ; wave 2 -> LDS -> wave 0 ; wave 0 stores the workgroup partial
  f4* const red = (f4*)smem;                             // [2][32][64] f4 = 64 KB
  if (wave & 1) {
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) red[((wave >> 1) * 32 + p * 2 + cb) * 64 + lane] = acc[p][cb];
  }
  __syncthreads();
  if (!(wave & 1)) {
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[p][cb] += red[((wave >> 1) * 32 + p * 2 + cb) * 64 + lane];
  }
  __syncthreads();
  if (wave == 2) {
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) red[(p * 2 + cb) * 64 + lane] = acc[p][cb];
  }
  __syncthreads();
  if (wave == 0) {
    f4* const out = (f4*)(a.part + (size_t)blockIdx.x * kGPart);
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) out[(p * 2 + cb) * 64 + lane] = acc[p][cb] + red[(p * 2 + cb) * 64 + lane];
  }
}

// dW[co][(r*3+s)*32 + c] += sum_wg sum_{a,b} part[wg][P(a, dpy(a,r), b, dpx(b,s))][cb][lane][e],  co = 4*(lane>>4) + e,
// c = cb*16 + (lane & 15).  Workgroup = 8 output units (tap, cb, lane) x 32 partial groups, fixed order throughout.
__global__ __launch_bounds__(256) void wgrad_up2_reduce_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw, int wrows, int Kpad) {
  __shared__ f4 red[32][8];
  const int u8 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int ou = blockIdx.x * 8 + u8;                    // 0 .. 9*2*64-1
  const int tap = ou >> 7, cb = (ou >> 6) & 1, lane = ou & 63;
  const int r = tap / 3, s_ = tap - r * 3;
  f4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int k = grp; k < nparts; k += 32) {
    const f4* const pk = (const f4*)(part + (size_t)k * kGPart);
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const int dy_ = pa == 0 ? (r >= 1) : (r == 2), dx_ = pb == 0 ? (s_ >= 1) : (s_ == 2);
        const int p = ((pa * 2 + dy_) * 2 + pb) * 2 + dx_;
        sum += pk[(p * 2 + cb) * 64 + lane];
      }
  }
  red[grp][u8] = sum;
  __syncthreads();
  if (grp == 0) {
    f4 tsum = red[0][u8];
#pragma unroll
    for (int g = 1; g < 32; ++g) tsum += red[g][u8];
    const int c = cb * 16 + (lane & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = 4 * (lane >> 4) + e;
      if (co < wrows) dw[(size_t)co * Kpad + tap * 32 + c] += tsum[e];
    }
  }
}

bool wgrad_up2_applicable(const WgradArgs& a) {
  return a.ntaps == 9 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.s0.up == 1 && a.C0 == a.Ctot && a.Ctot == 32 && a.s0.C == 32 &&
         a.Cout == 16 && a.wrows <= 16 && a.Kpad >= 288 && a.Ho == 2 * a.s0.H && a.Wo == 2 * a.s0.W && a.Hl == a.Ho && a.Wl == a.Wo &&
         (a.s0.H % kGH) == 0 && (a.s0.W % kGW) == 0;
}

hipError_t launch_wgrad_up2(const WgradArgs& a0, hipStream_t st) {
  if (!wgrad_up2_applicable(a0)) return hipErrorInvalidValue;
  WgradArgs a = a0;
  const bool f16 = a.prec == 2 && a.xmax && wgrad_up2_f16_shape(a);      // fp16x3 precision modes: conv_up2_f16.hip's kernel, same partial layout
  const int ntiles = a.N * (a.s0.H / kGH) * (a.s0.W / kGW);
  const int nwg = f16 ? wgrad_up2_f16_parts(a) : (ntiles < 2 * device_cu_count() ? ntiles : 2 * device_cu_count());
  const size_t need = (size_t)nwg * kGPart;
  if (!a.part || a.part_floats < need) { a.part = wgrad_op_scratch(); a.part_floats = wgrad_wino_scratch_floats(); }     // single-operator entry points
  if (!a.part || a.part_floats < need) return hipErrorOutOfMemory;
  if (f16) {
    hipError_t e = launch_wgrad_up2_f16(a, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wgrad_up2_reduce_kernel, dim3(9 * 2 * 64 / 8), dim3(256), 0, st, (const float*)a.part, nwg, a.dw, a.wrows, a.Kpad);
    return hipGetLastError();
  }
  size_t lds = (size_t)2 * kGBuf * sizeof(float);
  if (lds < 64 * 1024) lds = 64 * 1024;                  // the cross-wave sum needs [2][32][64] f4
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_up2_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(36, a.flops, a.bytes, wgrad_up2_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
  hipLaunchKernelGGL(wgrad_up2_reduce_kernel, dim3(9 * 2 * 64 / 8), dim3(256), 0, st, (const float*)a.part, nwg, a.dw, a.wrows, a.Kpad);
  return hipGetLastError();
}

}  // namespace uwm
