// 3x3 / stride 1 / pad 1 convolution from 16 to 16 channels at full resolution in the fp16x3 arithmetic, forward AND dgrad, for gfx950
// (v_mfma_f32_16x16x32_f16): decoder block 4 conv2 of every smp.Unet / UnetPlusPlus (DecoderBlock.conv2,
// /root/reference/src/models/unet_model.py:64-71 -> smp).
//
// Why its own kernel: 19 GFLOP over 536 MB — an HBM-bound layer (94 us at 5.7 TB/s) that the general fp16x3 kernel ran in 158 us
// with its 64-channel tile machinery three quarters empty (conv_f16x3_kernel<1>).  Here a k-step of the MFMA is a PAIR of taps x 16
// channels (9 taps -> 5 k-steps, the last one half zero), the 16 x 144 filter sits in LDS as ten ready A fragments (hi | lo), the
// halo patch [10 x 34 px][hi 16 ch | lo 16 ch] is split while it is staged, and persistent workgroups double-buffer it: the next
// tile's global loads are in flight behind this tile's 60 MFMAs per wave; one barrier per tile.
//
//   out[p][m] = sum_{tap} sum_k in[p + d(tap)][k] * W[m][tap*16 + k]      d(tap) = (r - 1, s - 1) * rmul
// forward: rmul = +1, W = the layer's weights [co][tap*16 + ci], lazy BatchNorm + ReLU source, BatchNorm statistics of the output;
// dgrad:   rmul = -1, W = the packed dgrad filter [ci][tap*16 + co], dY scaled by the power of two its maximum calls for
//          (ConvArgs::xmax), ReLU mask of the producer (+ addend), fused BatchNorm-backward sums (ConvArgs::bnb_*).
// Arithmetic as conv_up2_f16.hip: operands = hi + lo fp16 halves, hi*hi' + hi*lo' + lo*hi', fp32 accumulation, filter times 2^12.
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {
constexpr float kWScale = 4096.f;
constexpr int kTH = 8, kTW = 32, kPW = kTW + 2, kPix = (kTH + 2) * kPW;      // tile 8 x 32, halo patch 10 x 34 = 340 pixels
constexpr int kPatchB = kPix * 64;                                            // bytes: [px][4 units of 16 B: hi 0-7, hi 8-15, lo 0-7, lo 8-15], units XOR-swizzled by (px >> 2) & 3
constexpr int kBankB = 5 * 2 * 64 * 16;                                       // [5 k-steps][hi | lo][64 lanes][16 B]
__device__ __forceinline__ int px_off(int px, int unit) { return px * 64 + ((unit ^ ((px >> 2) & 3)) << 4); }
}  // namespace

__global__ __launch_bounds__(256, 2) void conv_c16_f16_kernel(const ConvArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  char* const bank = smem_;
  char* const Ps = smem_ + kBankB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int H = a.Ho, W = a.Wo;
  const int tilesW = W / kTW, tilesH = H / kTH;
  const bool dg = a.rmul < 0;

  float xs = 1.f;                                         // dgrad: power-of-two scale of dY
  if (dg && a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }

  // ---- staging: 340 px x 4 quads = 1360 units, 6 rounds
  const int unit = tid & 3;
  const bool has = a.s0.scale != nullptr;
  f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (has) { sc = *(const f4*)(a.s0.scale + unit * 4); sh = *(const f4*)(a.s0.shift + unit * 4); }
  sc = sc * xs; sh = sh * xs;                              // (xs = 1 forward; a dgrad's dY has no lazy transform: sc = xs)
  const float vlo = (has && a.s0.relu) ? 0.f : -65504.f;
  int spy[6], spx[6], spos[6];
#pragma unroll
  for (int rd = 0; rd < 6; ++rd) {
    const int pp = min((rd * 256 + tid) >> 2, kPix - 1);
    spy[rd] = pp / kPW; spx[rd] = pp - spy[rd] * kPW;
    spos[rd] = px_off(pp, unit >> 1) + (unit & 1) * 8;     // hi half; lo half: unit + 2 -> byte ^ 32
  }
  const bool last_live = (5 * 256 + tid) < kPix * 4;
  f4 pv[6]; unsigned pok = 0;
  auto tile_origin = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    h0 = th * kTH; w0 = tw * kTW;
  };
  auto patch_load = [&](int t) {
    int n, h0, w0; tile_origin(t, n, h0, w0);
    pok = 0;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const int hl = h0 - 1 + spy[rd], wl = w0 - 1 + spx[rd];
      const bool ok = hl >= 0 && hl < H && wl >= 0 && wl < W;
      const int hc = min(max(hl, 0), H - 1), wc = min(max(wl, 0), W - 1);
      pv[rd] = *(const f4*)(a.s0.ptr + (((size_t)n * H + hc) * W + wc) * 16 + unit * 4);
      pok |= (ok ? 1u : 0u) << rd;
    }
  };
  auto patch_store = [&](int buf) {
    char* const pb_ = Ps + buf * kPatchB;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      f4 v = pv[rd] * sc + sh;
      const bool ok = (pok >> rd) & 1u;
      const float top = ok ? 65504.f : vlo;
      v.x = __builtin_amdgcn_fmed3f(v.x, vlo, top); v.y = __builtin_amdgcn_fmed3f(v.y, vlo, top);
      v.z = __builtin_amdgcn_fmed3f(v.z, vlo, top); v.w = __builtin_amdgcn_fmed3f(v.w, vlo, top);
      if (vlo != 0.f && !ok) v = (f4){0.f, 0.f, 0.f, 0.f};
      uwm_u2 hi, lo;
      uwm_split4(v.x, v.y, v.z, v.w, hi, lo);
      if (rd < 5 || last_live) { *(uwm_u2*)(pb_ + spos[rd]) = hi; *(uwm_u2*)(pb_ + (spos[rd] ^ 32)) = lo; }
    }
  };

  // ---- the filter as MFMA A fragments: k-step j = taps 2j, 2j+1; lane (m = lane & 15, k-group kg): tap 2j + (kg >> 1), channels (kg & 1)*8 ..
  for (int slot = tid; slot < 5 * 64; slot += 256) {
    const int j = slot >> 6, L = slot & 63, m = L & 15, kg = L >> 4;
    const int tap = 2 * j + (kg >> 1);
    f4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
    if (m < a.wrows && tap < 9) {
      const float* p = a.w + (size_t)m * a.Kpad + tap * 16 + (kg & 1) * 8;
      w0 = *(const f4*)p; w1 = *(const f4*)(p + 4);
    }
    w0 = w0 * kWScale; w1 = w1 * kWScale;
    uwm_u2 h0, l0, h1, l1;
    uwm_split4(__builtin_amdgcn_fmed3f(w0.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w0.y, -65504.f, 65504.f),
               __builtin_amdgcn_fmed3f(w0.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w0.w, -65504.f, 65504.f), h0, l0);
    uwm_split4(__builtin_amdgcn_fmed3f(w1.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w1.y, -65504.f, 65504.f),
               __builtin_amdgcn_fmed3f(w1.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w1.w, -65504.f, 65504.f), h1, l1);
    uwm_u2* const ph = (uwm_u2*)(bank + ((j * 2 + 0) * 64 + L) * 16);
    uwm_u2* const pl = (uwm_u2*)(bank + ((j * 2 + 1) * 64 + L) * 16);
    ph[0] = h0; ph[1] = h1; pl[0] = l0; pl[1] = l1;
  }
  // ---- B fragment geometry: lane (pixel column lrow of a 16-column block, k-group lq): tap 2j + (lq >> 1), hi unit lq & 1
  int toff[5];                                             // patch pixel offset (rows * kPW + cols) of this lane's tap in k-step j
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int tap = min(2 * j + (lq >> 1), 8);             // (tap 9: zero weights; read tap 8 again)
    const int r = tap / 3, s = tap - r * 3;
    toff[j] = dg ? ((2 - r) * kPW + (2 - s)) : (r * kPW + s);      // dgrad: in[p + 1 - r][q + 1 - s]
  }

  int t = blockIdx.x;
  patch_load(t);
  patch_store(0);
  __syncthreads();

  const int co = lq * 4;
  const bool bnb = a.bnb_mean != nullptr;
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = ps_, bmu = ps_, brs = ps_, msc = {1.f, 1.f, 1.f, 1.f}, msh = ps_, bias = ps_;
  if (bnb) { bmu = *(const f4*)(a.bnb_mean + co); brs = *(const f4*)(a.bnb_rstd + co); }
  if (a.mask && a.mscale) { msc = *(const f4*)(a.mscale + co); msh = *(const f4*)(a.mshift + co); }
  if (a.bias) bias = *(const f4*)(a.bias + co);
  const float unscale = 1.f / (kWScale * xs);

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    patch_load(tn < ntiles ? tn : t);                      // (last tile: harmless re-read)
    int n, h0, w0; tile_origin(t, n, h0, w0);
    const char* const pc = Ps + cur * kPatchB;
    f4 acc[2][2];                                          // [row of the wave's two][16-column block]
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) acc[rr][0] = acc[rr][1] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const h8 Ah = *(const h8*)(bank + ((j * 2 + 0) * 64 + lane) * 16);
      const h8 Al = *(const h8*)(bank + ((j * 2 + 1) * 64 + lane) * 16);
      h8 Bh[2][2], Bl[2][2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int pp = (wave * 2 + rr) * kPW + cb * 16 + lrow + toff[j];
          const int o = px_off(pp, lq & 1);
          Bh[rr][cb] = *(const h8*)(pc + o); Bl[rr][cb] = *(const h8*)(pc + (o ^ 32));
        }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rr][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl[rr][cb], acc[rr][cb], 0, 0, 0);
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rr][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh[rr][cb], acc[rr][cb], 0, 0, 0);
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[rr][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh[rr][cb], acc[rr][cb], 0, 0, 0);
    }
    // epilogue: pixel (h0 + 2 wave + rr, w0 + 16 cb + lrow), channels 4 lq ..: a wave-level store covers 16 px x 64 B contiguous
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const size_t o = (((size_t)n * H + h0 + wave * 2 + rr) * W + w0 + cb * 16 + lrow) * 16 + co;
        f4 v = acc[rr][cb] * unscale + bias;
        if (a.addend) v += *(const f4*)(a.addend + o);
        if (a.mask) {
          const f4 yr = *(const f4*)(a.mask + o);
          const f4 mk = yr * msc + msh;
          v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
          if (bnb) { ps_ += v; pq_ += v * ((yr - bmu) * brs); }
        }
        *(f4*)(a.out + o) = v;
        if (!bnb) { ps_ += v; pq_ += v * v; }
      }
    patch_store(cur ^ 1);
    __syncthreads();
  }

  if (a.ssum != nullptr) {            // BatchNorm statistics (forward) / BatchNorm-backward sums (dgrad): 16 pixel lanes -> 4 waves (LDS) -> fp64 atomics on one replica
    const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sv = ps_[e], qv = pq_[e];
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) { sv += __shfl_xor(sv, d); qv += __shfl_xor(qv, d); }
      ps_[e] = sv; pq_[e] = qv;
    }
    float* red = (float*)Ps;          // [4 waves][16][2]  (the last barrier of the loop has passed)
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

bool conv_c16_f16_applicable(const ConvArgs& a) {
  return a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && (a.rmul == 1 ? a.off == -1 : (a.rmul == -1 && a.off == 1)) &&
         a.Ctot == 16 && a.C0 == 16 && a.s0.C == 16 && a.s0.up == 0 && a.Cout == 16 && a.wrows <= 16 && a.Kpad >= 144 &&
         a.Hl == a.Ho && a.Wl == a.Wo && a.s0.H == a.Ho && a.s0.W == a.Wo && (a.Ho % kTH) == 0 && (a.Wo % kTW) == 0 &&
         !a.out_up && !a.bnb_y && (a.rmul == 1 || !a.s0.scale) && (!a.bnb_mean || (a.mask && a.ssum && a.ssq && a.bnb_rstd));
}

hipError_t launch_conv_c16_f16(const ConvArgs& a, hipStream_t st) {
  if (!conv_c16_f16_applicable(a)) return hipErrorInvalidValue;
  const size_t lds = (size_t)kBankB + 2 * kPatchB;
  const int ntiles = a.N * (a.Ho / kTH) * (a.Wo / kTW);
  const int nwg = ntiles < 2 * device_cu_count() ? ntiles : 2 * device_cu_count();
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_c16_f16_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(48, a.flops, a.bytes, conv_c16_f16_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- 32 -> 32 channels
// The same layer shape one decoder level up (decoder block 3 conv2: 32 -> 32 at half resolution, forward and dgrad): a k-step = ONE
// tap x 32 channels (9 k-steps), two 16-row fragments of output channels, 36 KB of filter fragments + 2 x 43.5 KB of patch: one
// 512-thread workgroup per CU (wave = one row of the 8 x 32 tile, 2 x 2 accumulators, 108 MFMAs per tile), persistent and
// double-buffered like the 16-channel kernel.  Patch entries of 128 B = [hi 32 ch | lo 32 ch], 16-byte units XOR-swizzled by
// (px >> 1) & 7.  It replaces conv_f16x3v2s_kernel<1> on this layer: 107 us forward / 133 us dgrad for 268 / 402 MB.
namespace {
constexpr int kQBankB = 9 * 2 * 2 * 64 * 16;                                   // [9 taps][2 fragments][hi | lo][64 lanes][16 B]
constexpr int kQPatchB = kPix * 128;
__device__ __forceinline__ int qx_off(int px, int unit) { return px * 128 + ((unit ^ ((px >> 1) & 7)) << 4); }
}  // namespace

__global__ __launch_bounds__(512, 1) void conv_c32_f16_kernel(const ConvArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  char* const bank = smem_;
  char* const Ps = smem_ + kQBankB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int H = a.Ho, W = a.Wo;
  const int tilesW = W / kTW, tilesH = H / kTH;
  const bool dg = a.rmul < 0;

  float xs = 1.f;
  if (dg && a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }

  // ---- staging: 340 px x 8 quads = 2720 units, 6 rounds of 512 threads
  const int unit = tid & 7;
  const bool has = a.s0.scale != nullptr;
  f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (has) { sc = *(const f4*)(a.s0.scale + unit * 4); sh = *(const f4*)(a.s0.shift + unit * 4); }
  sc = sc * xs; sh = sh * xs;
  const float vlo = (has && a.s0.relu) ? 0.f : -65504.f;
  int spy[6], spx[6], spos[6];
#pragma unroll
  for (int rd = 0; rd < 6; ++rd) {
    const int pp = min((rd * 512 + tid) >> 3, kPix - 1);
    spy[rd] = pp / kPW; spx[rd] = pp - spy[rd] * kPW;
    spos[rd] = qx_off(pp, unit >> 1) + (unit & 1) * 8;     // hi half; lo half: unit + 4 -> byte ^ 64
  }
  const bool last_live = (5 * 512 + tid) < kPix * 8;
  f4 pv[6]; unsigned pok = 0;
  auto tile_origin = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    h0 = th * kTH; w0 = tw * kTW;
  };
  auto patch_load = [&](int t) {
    int n, h0, w0; tile_origin(t, n, h0, w0);
    pok = 0;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const int hl = h0 - 1 + spy[rd], wl = w0 - 1 + spx[rd];
      const bool ok = hl >= 0 && hl < H && wl >= 0 && wl < W;
      const int hc = min(max(hl, 0), H - 1), wc = min(max(wl, 0), W - 1);
      pv[rd] = *(const f4*)(a.s0.ptr + (((size_t)n * H + hc) * W + wc) * 32 + unit * 4);
      pok |= (ok ? 1u : 0u) << rd;
    }
  };
  auto patch_store = [&](int buf) {
    char* const pb_ = Ps + buf * kQPatchB;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      f4 v = pv[rd] * sc + sh;
      const bool ok = (pok >> rd) & 1u;
      const float top = ok ? 65504.f : vlo;
      v.x = __builtin_amdgcn_fmed3f(v.x, vlo, top); v.y = __builtin_amdgcn_fmed3f(v.y, vlo, top);
      v.z = __builtin_amdgcn_fmed3f(v.z, vlo, top); v.w = __builtin_amdgcn_fmed3f(v.w, vlo, top);
      if (vlo != 0.f && !ok) v = (f4){0.f, 0.f, 0.f, 0.f};
      uwm_u2 hi, lo;
      uwm_split4(v.x, v.y, v.z, v.w, hi, lo);
      if (rd < 5 || last_live) { *(uwm_u2*)(pb_ + spos[rd]) = hi; *(uwm_u2*)(pb_ + (spos[rd] ^ 64)) = lo; }
    }
  };

  // ---- the filter as MFMA A fragments: (tap, fragment mb): lane (m = mb*16 + (lane & 15), k-group kg): channels kg*8 ..
  for (int slot = tid; slot < 18 * 64; slot += 512) {
    const int tm = slot >> 6, L = slot & 63, tap = tm >> 1, mb = tm & 1, m = mb * 16 + (L & 15), kg = L >> 4;
    f4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
    if (m < a.wrows) {
      const float* p = a.w + (size_t)m * a.Kpad + tap * 32 + kg * 8;
      w0 = *(const f4*)p; w1 = *(const f4*)(p + 4);
    }
    w0 = w0 * kWScale; w1 = w1 * kWScale;
    uwm_u2 h0, l0, h1, l1;
    uwm_split4(__builtin_amdgcn_fmed3f(w0.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w0.y, -65504.f, 65504.f),
               __builtin_amdgcn_fmed3f(w0.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w0.w, -65504.f, 65504.f), h0, l0);
    uwm_split4(__builtin_amdgcn_fmed3f(w1.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w1.y, -65504.f, 65504.f),
               __builtin_amdgcn_fmed3f(w1.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w1.w, -65504.f, 65504.f), h1, l1);
    uwm_u2* const ph = (uwm_u2*)(bank + ((tm * 2 + 0) * 64 + L) * 16);
    uwm_u2* const pl = (uwm_u2*)(bank + ((tm * 2 + 1) * 64 + L) * 16);
    ph[0] = h0; ph[1] = h1; pl[0] = l0; pl[1] = l1;
  }

  int t = blockIdx.x;
  patch_load(t);
  patch_store(0);
  __syncthreads();

  const bool bnb = a.bnb_mean != nullptr;
  f4 ps_[2], pq_[2], bmu[2], brs[2], msc[2], msh[2], bias[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int c = mb * 16 + lq * 4;
    ps_[mb] = pq_[mb] = bmu[mb] = brs[mb] = msh[mb] = bias[mb] = (f4){0.f, 0.f, 0.f, 0.f};
    msc[mb] = (f4){1.f, 1.f, 1.f, 1.f};
    if (bnb) { bmu[mb] = *(const f4*)(a.bnb_mean + c); brs[mb] = *(const f4*)(a.bnb_rstd + c); }
    if (a.mask && a.mscale) { msc[mb] = *(const f4*)(a.mscale + c); msh[mb] = *(const f4*)(a.mshift + c); }
    if (a.bias) bias[mb] = *(const f4*)(a.bias + c);
  }
  const float unscale = 1.f / (kWScale * xs);

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    patch_load(tn < ntiles ? tn : t);
    int n, h0, w0; tile_origin(t, n, h0, w0);
    const char* const pc = Ps + cur * kQPatchB;
    f4 acc[2][2];                                          // [fragment mb][16-column block]
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc[mb][0] = acc[mb][1] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s3 = tap - r * 3;
      const int toff = dg ? ((2 - r) * kPW + (2 - s3)) : (r * kPW + s3);
      h8 Ah[2], Al[2], Bh[2], Bl[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        Ah[mb] = *(const h8*)(bank + (((tap * 2 + mb) * 2 + 0) * 64 + lane) * 16);
        Al[mb] = *(const h8*)(bank + (((tap * 2 + mb) * 2 + 1) * 64 + lane) * 16);
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int pp = wave * kPW + cb * 16 + lrow + toff;
        const int o = qx_off(pp, lq);
        Bh[cb] = *(const h8*)(pc + o); Bl[cb] = *(const h8*)(pc + (o ^ 64));
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[mb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mb], Bl[cb], acc[mb][cb], 0, 0, 0);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[mb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[mb], Bh[cb], acc[mb][cb], 0, 0, 0);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[mb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[mb], Bh[cb], acc[mb][cb], 0, 0, 0);
    }
    // epilogue: pixel (h0 + wave, w0 + 16 cb + lrow), channels mb*16 + 4 lq ..
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const size_t o = (((size_t)n * H + h0 + wave) * W + w0 + cb * 16 + lrow) * 32 + mb * 16 + lq * 4;
        f4 v = acc[mb][cb] * unscale + bias[mb];
        if (a.addend) v += *(const f4*)(a.addend + o);
        if (a.mask) {
          const f4 yr = *(const f4*)(a.mask + o);
          const f4 mk = yr * msc[mb] + msh[mb];
          v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
          if (bnb) { ps_[mb] += v; pq_[mb] += v * ((yr - bmu[mb]) * brs[mb]); }
        }
        *(f4*)(a.out + o) = v;
        if (!bnb) { ps_[mb] += v; pq_[mb] += v * v; }
      }
    patch_store(cur ^ 1);
    __syncthreads();
  }

  if (a.ssum != nullptr) {            // statistics / BatchNorm-backward sums: 16 pixel lanes -> 8 waves (LDS) -> fp64 atomics on one replica
    const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
    float* red = (float*)Ps;          // [8 waves][32][2]  (the last barrier of the loop has passed)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sv = ps_[mb][e], qv = pq_[mb][e];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { sv += __shfl_xor(sv, d); qv += __shfl_xor(qv, d); }
        if (lrow == 0) { red[(wave * 32 + mb * 16 + lq * 4 + e) * 2] = sv; red[(wave * 32 + mb * 16 + lq * 4 + e) * 2 + 1] = qv; }
      }
    __syncthreads();
    if (tid < 32 && tid < a.Cout) {
      double sv = 0.0, qv = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) { sv += (double)red[(w * 32 + tid) * 2]; qv += (double)red[(w * 32 + tid) * 2 + 1]; }
      atomicAdd(a.ssum + srep_off + tid, sv);
      atomicAdd(a.ssq + srep_off + tid, qv);
    }
  }
}

bool conv_c32_f16_applicable(const ConvArgs& a) {
  return a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && (a.rmul == 1 ? a.off == -1 : (a.rmul == -1 && a.off == 1)) &&
         a.Ctot == 32 && a.C0 == 32 && a.s0.C == 32 && a.s0.up == 0 && a.Cout == 32 && a.wrows <= 32 && a.Kpad >= 288 &&
         a.Hl == a.Ho && a.Wl == a.Wo && a.s0.H == a.Ho && a.s0.W == a.Wo && (a.Ho % kTH) == 0 && (a.Wo % kTW) == 0 &&
         !a.out_up && !a.bnb_y && (a.rmul == 1 || !a.s0.scale) && (!a.bnb_mean || (a.mask && a.ssum && a.ssq && a.bnb_rstd));
}

hipError_t launch_conv_c32_f16(const ConvArgs& a, hipStream_t st) {
  if (!conv_c32_f16_applicable(a)) return hipErrorInvalidValue;
  const size_t lds = (size_t)kQBankB + 2 * kQPatchB;
  const int ntiles = a.N * (a.Ho / kTH) * (a.Wo / kTW);
  const int nwg = ntiles < device_cu_count() ? ntiles : device_cu_count();
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_c32_f16_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(48, a.flops, a.bytes, conv_c32_f16_kernel, dim3((unsigned)nwg), dim3(512), lds, st, a, ntiles);
  return hipGetLastError();
}

}  // namespace uwm
