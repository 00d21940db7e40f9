// conv_up2.hip's layer (3x3 conv over a nearest-x2-upsampled 32-channel input, 16 output channels: decoder block 4 conv1 of every
// smp.Unet / UnetPlusPlus) and its dgrad in the fp16x3 arithmetic of the fp16x3 precision modes, for gfx950: the same sub-pixel
// decomposition (four 2x2 class convolutions forward, a 4x4 / stride-2 convolution backward), the products on
// v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_16x16x4_f32.
//
// Why: the fp32 kernels run 256 (forward) / 128 (dgrad) MFMAs of 32 cycles per tile and wave and are bound by that pipe (109 of
// 185 us forward); one fp16 MFMA covers all 32 input channels of a class tap (forward) or two taps x 16 channels (dgrad) in 16
// cycles, three of them per product (hi*hi' + hi*lo' + lo*hi'), which leaves the layer with its HBM stream.
//
// Arithmetic: every fp32 operand = hi + lo, two fp16 halves (22 mantissa bits), fp32 accumulation.  The class filters
// are SUMS of up to four raw taps, made in fp32, times 2^12 (raw smp weights sit far inside fp16's range; the scale keeps the low
// halves normal), split once per workgroup into an LDS bank of ready MFMA A fragments; the activations are split while the patch is
// staged (lazy BatchNorm + ReLU first, clamped to fp16's range); a dgrad's dY is scaled by the power of two that puts max|dY| into
// [2^13, 2^14) (ConvArgs::xmax, from bn_bwd_apply).  All scales are powers of two and leave in the epilogue: exact.
//
// Reference semantics replaced: as conv_up2.hip (F.interpolate(scale_factor=2, mode="nearest") + Conv2dReLU's conv of smp's
// DecoderBlock, /root/reference/src/models/unet_model.py:64-71 -> smp).
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {
constexpr float kWScale = 4096.f;
constexpr int kBankBytes = 16 * 2 * 64 * 16;                       // [16 fragments][hi | lo][64 lanes][16 B] = 32 KB
// forward tile: 8 x 16 low-resolution pixels, 10 x 18 patch; entries of 128 B = [hi: 32 ch | lo: 32 ch], 16-byte units XOR-swizzled
constexpr int kFH = 8, kFW = 16, kFPW = kFW + 2, kFPP = (kFH + 2) * kFPW;
// dgrad tile: 4 x 16 low-resolution pixels, dY patch of 10 rows x 17 column PAIRS; entries of 128 B = [hi: 2 px x 16 co | lo: same]
constexpr int kDH = 4, kDW = 16, kDRows = 2 * kDH + 2, kDCP = kDW + 1, kDEnt = kDRows * kDCP;

__device__ __forceinline__ int r0f(int a, int d) { return a == 0 ? (d == 0 ? 0 : 1) : (d == 0 ? 0 : 2); }      // forward: raw taps of (parity a, class tap d)
__device__ __forceinline__ int r1f(int a, int d) { return a == 0 ? (d == 0 ? 0 : 2) : (d == 0 ? 1 : 2); }
__device__ __forceinline__ int t0d(int t) { return t == 0 ? 2 : (t == 1 ? 1 : 0); }                            // dgrad: raw taps of 4x4 tap t
__device__ __forceinline__ int t1d(int t) { return t == 0 ? 2 : (t == 1 ? 2 : (t == 2 ? 1 : 0)); }

__device__ __forceinline__ void put_frag(char* bank, int frag, int lane, const float* v) {
  uwm_u2 h0, l0, h1, l1;
  uwm_split4(__builtin_amdgcn_fmed3f(v[0] * kWScale, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[1] * kWScale, -65504.f, 65504.f),
             __builtin_amdgcn_fmed3f(v[2] * kWScale, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[3] * kWScale, -65504.f, 65504.f), h0, l0);
  uwm_split4(__builtin_amdgcn_fmed3f(v[4] * kWScale, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[5] * kWScale, -65504.f, 65504.f),
             __builtin_amdgcn_fmed3f(v[6] * kWScale, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[7] * kWScale, -65504.f, 65504.f), h1, l1);
  uwm_u2* const ph = (uwm_u2*)(bank + ((frag * 2 + 0) * 64 + lane) * 16);
  uwm_u2* const pl = (uwm_u2*)(bank + ((frag * 2 + 1) * 64 + lane) * 16);
  ph[0] = h0; ph[1] = h1; pl[0] = l0; pl[1] = l1;
}
__device__ __forceinline__ float dy_scale(const float* xmax, int lane) {
  float xs = 1.f;
  if (xmax) {
    float mx = xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }
  return xs;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------- forward
// Persistent workgroups, double-buffered patch (as conv_up2_kernel).  LDS: bank 32 KB + 2 x 180 x 128 B = 78 KB: two per CU.
// A wave owns two low-resolution rows of the tile: per output parity (pa, pb) and class tap (dr, ds) one A fragment pair (hi, lo)
// and, per row, one B fragment pair; 96 MFMAs per tile and wave.
__global__ __launch_bounds__(256, 2) void conv_up2_f16_kernel(const ConvArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  char* const bank = smem_;
  char* const Ps = smem_ + kBankBytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int Hs = a.s0.H, Wsrc = a.s0.W;
  const int tilesW = Wsrc / kFW, tilesH = Hs / kFH;

  const int unit = tid & 7;
  const bool has = a.s0.scale != nullptr;
  f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (has) { sc = *(const f4*)(a.s0.scale + unit * 4); sh = *(const f4*)(a.s0.shift + unit * 4); }
  const float vlo = (has && a.s0.relu) ? 0.f : -65504.f;
  int spy[6], spx[6], spos[6];
#pragma unroll
  for (int rd = 0; rd < 6; ++rd) {
    const int pp = min((rd * 256 + tid) >> 3, kFPP - 1);
    spy[rd] = pp / kFPW; spx[rd] = pp - spy[rd] * kFPW;
    spos[rd] = pp * 128 + (((unit >> 1) ^ ((pp >> 1) & 7)) << 4) + (unit & 1) * 8;      // hi half; the lo half: unit index ^ 4 -> byte ^ 64
  }
  const bool last_live = (5 * 256 + tid) < kFPP * 8;
  f4 pv[6]; unsigned pok = 0;
  auto tile_origin = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    h0 = th * kFH; w0 = tw * kFW;
  };
  auto patch_load = [&](int t) {
    int n, h0, w0; tile_origin(t, n, h0, w0);
    pok = 0;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const int hl = h0 - 1 + spy[rd], wl = w0 - 1 + spx[rd];
      const bool ok = hl >= 0 && hl < Hs && wl >= 0 && wl < Wsrc;
      const int hc = min(max(hl, 0), Hs - 1), wc = min(max(wl, 0), Wsrc - 1);
      pv[rd] = *(const f4*)(a.s0.ptr + (((size_t)n * Hs + hc) * Wsrc + wc) * 32 + unit * 4);
      pok |= (ok ? 1u : 0u) << rd;
    }
  };
  auto patch_store = [&](int buf) {
    char* const pb_ = Ps + buf * kFPP * 128;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      f4 v = pv[rd];
      if (has) v = v * sc + sh;
      const float top = ((pok >> rd) & 1u) ? 65504.f : vlo;          // out of the image: clamp to [vlo, vlo] = 0 under ReLU, else zeroed below
      v.x = __builtin_amdgcn_fmed3f(v.x, vlo, top); v.y = __builtin_amdgcn_fmed3f(v.y, vlo, top);
      v.z = __builtin_amdgcn_fmed3f(v.z, vlo, top); v.w = __builtin_amdgcn_fmed3f(v.w, vlo, top);
      if (vlo != 0.f && !((pok >> rd) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
      uwm_u2 hi, lo;
      uwm_split4(v.x, v.y, v.z, v.w, hi, lo);
      if (rd < 5 || last_live) { *(uwm_u2*)(pb_ + spos[rd]) = hi; *(uwm_u2*)(pb_ + (spos[rd] ^ 64)) = lo; }
    }
  };

  // ---- the 16 class filters as MFMA A fragments: fragment ((pa*2 + pb)*2 + dr)*2 + ds, lane (co = lane & 15, k-group = lane >> 4)
  for (int slot = tid; slot < 16 * 64; slot += 256) {
    const int frag = slot >> 6, L = slot & 63, co = L & 15, kg = L >> 4;
    const int pa = frag >> 3, pb = (frag >> 2) & 1, dr = (frag >> 1) & 1, ds = frag & 1;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (co < a.wrows)
      for (int r = r0f(pa, dr); r <= r1f(pa, dr); ++r)
        for (int s2 = r0f(pb, ds); s2 <= r1f(pb, ds); ++s2) {
          const float* p = a.w + (size_t)co * a.Kpad + (r * 3 + s2) * 32 + kg * 8;
          const f4 w0 = *(const f4*)p, w1 = *(const f4*)(p + 4);
          v[0] += w0.x; v[1] += w0.y; v[2] += w0.z; v[3] += w0.w; v[4] += w1.x; v[5] += w1.y; v[6] += w1.z; v[7] += w1.w;
        }
    put_frag(bank, frag, L, v);
  }
  int t = blockIdx.x;
  patch_load(t);
  patch_store(0);
  __syncthreads();

  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = {0.f, 0.f, 0.f, 0.f};
  const int co = lq * 4;
  f4 bias = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias = *(const f4*)(a.bias + co);
  const float unscale = 1.f / kWScale;

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    patch_load(tn < ntiles ? tn : t);                  // (last tile: harmless re-read)
    int n, h0, w0; tile_origin(t, n, h0, w0);
    const char* const pc = Ps + cur * kFPP * 128;
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
      f4 acc[2][2];                                    // [pb][rb]
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        acc[pb][0] = acc[pb][1] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dr = 0; dr < 2; ++dr)
#pragma unroll
          for (int ds = 0; ds < 2; ++ds) {
            const int frag = ((pa * 2 + pb) * 2 + dr) * 2 + ds;
            const h8 Ah = *(const h8*)(bank + ((frag * 2 + 0) * 64 + lane) * 16);
            const h8 Al = *(const h8*)(bank + ((frag * 2 + 1) * 64 + lane) * 16);
            h8 Bh[2], Bl[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
              const int pp = (wave * 2 + rb + pa + dr) * kFPW + lrow + pb + ds;
              const int o = pp * 128 + ((lq ^ ((pp >> 1) & 7)) << 4);
              Bh[rb] = *(const h8*)(pc + o); Bl[rb] = *(const h8*)(pc + (o ^ 64));
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[pb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl[rb], acc[pb][rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[pb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh[rb], acc[pb][rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[pb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh[rb], acc[pb][rb], 0, 0, 0);
          }
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const int ho = 2 * (h0 + wave * 2 + rb) + pa, wo = 2 * (w0 + lrow) + pb;
          const f4 v = acc[pb][rb] * unscale + bias;
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

hipError_t launch_conv_up2_f16(const ConvArgs& a, hipStream_t st) {
  if (!conv_up2_applicable(a)) return hipErrorInvalidValue;
  const size_t lds = (size_t)kBankBytes + 2 * kFPP * 128;
  const int ntiles = a.N * (a.s0.H / kFH) * (a.s0.W / kFW);
  const int nwg = ntiles < 2 * device_cu_count() ? ntiles : 2 * device_cu_count();
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_up2_f16_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(45, a.flops, a.bytes, conv_up2_f16_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- dgrad
// dX[p][q][c] = sum_{ty,tx in 0..3} sum_co dY[2p-1+ty][2q-1+tx][co] * W4[ty][tx][co][c] (conv_up2.hip): one MFMA k-step = the two
// taps tx = 2j, 2j+1 x 16 co = two ADJACENT dY pixels, which the patch keeps side by side (a column pair).  Fragment
// (ty*2 + j)*2 + cb holds W4[ty][2j + (k >> 4)][co = k & 15][c = cb*16 + row].  A wave owns one low-resolution row of the 4 x 16
// tile: 48 MFMAs per tile and wave.  Epilogue: the fused concat-split contract of conv_up2_dgrad_kernel.
__global__ __launch_bounds__(256, 2) void conv_up2_dgrad_f16_kernel(const ConvArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  char* const bank = smem_;
  char* const Ds = smem_ + kBankBytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int Hs = a.Ho >> 1, Wsrc = a.Wo >> 1;
  const int tilesW = Wsrc / kDW, tilesH = Hs / kDH;
  const float xs = dy_scale(a.xmax, lane);

  // ---- staging geometry: 10 rows x 34 columns x 4 units (4 co each) = 1360 units, 6 rounds
  const int unit = tid & 3;
  int scy[6], scx[6], spos[6];
#pragma unroll
  for (int rd = 0; rd < 6; ++rd) {
    const int px = min((rd * 256 + tid) >> 2, kDRows * 34 - 1);
    scy[rd] = px / 34; scx[rd] = px - scy[rd] * 34;
    const int ent = scy[rd] * kDCP + (scx[rd] >> 1);
    const int u16 = (scx[rd] & 1) * 2 + (unit >> 1);                 // 16-byte unit of the hi half: [px0 co 0-7 | px0 co 8-15 | px1 co 0-7 | px1 co 8-15]
    spos[rd] = ent * 128 + ((u16 ^ ((ent >> 1) & 7)) << 4) + (unit & 1) * 8;
  }
  const bool last_live = (5 * 256 + tid) < kDRows * 34 * 4;
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
    char* const pb_ = Ds + buf * kDEnt * 128;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const f4 v = pv[rd] * xs;
      uwm_u2 hi, lo;
      uwm_split4(__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f),
                 __builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f), hi, lo);
      if (rd < 5 || last_live) { *(uwm_u2*)(pb_ + spos[rd]) = hi; *(uwm_u2*)(pb_ + (spos[rd] ^ 64)) = lo; }
    }
  };

  // ---- the 16 tap-pair filters as MFMA A fragments (a.w = the packed dgrad filter [c][tap*16 + co], conv_up2_dgrad_kernel's)
  for (int slot = tid; slot < 16 * 64; slot += 256) {
    const int frag = slot >> 6, L = slot & 63, crow = L & 15, kg = L >> 4;
    const int ty = frag >> 2, j = (frag >> 1) & 1, cb = frag & 1;
    const int tx = 2 * j + (kg >> 1), co0 = (kg & 1) * 8, c = cb * 16 + crow;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < a.wrows)
      for (int r = t0d(ty); r <= t1d(ty); ++r)
        for (int s2 = t0d(tx); s2 <= t1d(tx); ++s2) {
          const float* p = a.w + (size_t)c * a.Kpad + (r * 3 + s2) * 16 + co0;
          const f4 w0 = *(const f4*)p, w1 = *(const f4*)(p + 4);
          v[0] += w0.x; v[1] += w0.y; v[2] += w0.z; v[3] += w0.w; v[4] += w1.x; v[5] += w1.y; v[6] += w1.z; v[7] += w1.w;
        }
    put_frag(bank, frag, L, v);
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
  const float unscale = 1.f / (kWScale * xs);

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    patch_load(tn < ntiles ? tn : t);
    int n, h0, w0; tile_origin(t, n, h0, w0);
    const char* const pc = Ds + cur * kDEnt * 128;
    f4 acc[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ty = 0; ty < 4; ++ty)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ent = (2 * wave + ty) * kDCP + lrow + j;
        const int o = ent * 128 + ((lq ^ ((ent >> 1) & 7)) << 4);
        const h8 Bh = *(const h8*)(pc + o), Bl = *(const h8*)(pc + (o ^ 64));
        h8 Ah[2], Al[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int frag = (ty * 2 + j) * 2 + cb;
          Ah[cb] = *(const h8*)(bank + ((frag * 2 + 0) * 64 + lane) * 16);
          Al[cb] = *(const h8*)(bank + ((frag * 2 + 1) * 64 + lane) * 16);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[cb], Bl, acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[cb], Bh, acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[cb], Bh, acc[cb], 0, 0, 0);
      }
    // epilogue: pixel (h0 + wave, w0 + lrow), channels cb*16 + 4*lq ..
    const size_t o2 = (((size_t)n * Hs + h0 + wave) * Wsrc + w0 + lrow) * 32 + lq * 4;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f4 v = acc[cb] * unscale;
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
    float* red = (float*)Ds;          // [4 waves][32][2]
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

hipError_t launch_conv_up2_dgrad_f16(const ConvArgs& a, hipStream_t st) {
  if (!conv_up2_dgrad_applicable(a)) return hipErrorInvalidValue;
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !a.up_mask || a.up_accum)) return hipErrorInvalidValue;
  const size_t lds = (size_t)kBankBytes + 2 * kDEnt * 128;
  const int ntiles = a.N * ((a.Ho >> 1) / kDH) * ((a.Wo >> 1) / kDW);
  const int nwg = ntiles < 2 * device_cu_count() ? ntiles : 2 * device_cu_count();
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_up2_dgrad_f16_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(46, a.flops, a.bytes, conv_up2_dgrad_f16_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- wgrad
// wgrad_up2_kernel's sixteen class products P[a][dpy][b][dpx][co][c] = sum dY[2i+a][2j+b][co] * X~[i-1+a+dpy][j-1+b+dpx][c]
// (conv_up2.hip) with the pixels as the k dimension of v_mfma_f32_16x16x32_f16: a k-step = the 32 low-resolution pixels of a
// tile row, both operands through the transposing LDS load ds_read_b64_tr_b16 out of pixel-major images (dY [4 rows][64 px][hi 16
// co | lo], X~ [4][34][hi 32 ch | lo]; the stride-2 walk over dY and the neighbourhood shifts are lane addresses).  Tile = 2 x 32
// low-resolution pixels; wave = (tile row, 16-channel block cb): 16 accumulators, 48 MFMAs per tile.  Persistent, stages
// double-buffered.  The workgroup partial has wgrad_up2_kernel's layout: wgrad_up2_reduce_kernel folds both.
typedef __fp16 u_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) u_fp16x4 u_lds_fp16x4;
__device__ __forceinline__ h8 u_tr_pair(const char* base, int o0, int o1) {
  const u_fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((u_lds_fp16x4*)(uintptr_t)(base + o0));
  const u_fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((u_lds_fp16x4*)(uintptr_t)(base + o1));
  typedef __fp16 fp16x8 __attribute__((__vector_size__(8 * sizeof(__fp16))));
  const fp16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(h8, v);
}
namespace {
constexpr int kWH = 2, kWW = 32;                                   // low-resolution pixels per tile
constexpr int kWPW = kWW + 2, kWPix = (kWH + 2) * kWPW;            // 4 x 34 = 136 patch pixels
constexpr int kWX = kWPix * 128;                                   // bytes of the X~ image (17 408)
constexpr int kWDy = 2 * kWH * 2 * kWW * 64;                       // bytes of the dY image: 4 rows x 64 px x 64 B (16 384)
constexpr int kWBuf = kWX + kWDy;
constexpr int kWPart = 16 * 2 * 64 * 4;                            // floats per workgroup partial (= conv_up2.hip's kGPart)
}  // namespace

__global__ __launch_bounds__(256, 2) void wgrad_up2_f16_kernel(const WgradArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(256))) char smem_[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kr = wave >> 1, cb = wave & 1;
  const int Hs = a.s0.H, Wsrc = a.s0.W;
  const int tilesW = Wsrc / kWW, tilesH = Hs / kWH;
  const float xs = dy_scale(a.xmax, lane);

  // ---- staging: X~ 136 px x 8 quads = 1088 units (5 rounds), dY 256 px x 4 quads = 1024 units (4 rounds)
  const int xunit = tid & 7, yunit = tid & 3;
  const bool has = a.s0.scale != nullptr;
  f4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (has) { sc = *(const f4*)(a.s0.scale + xunit * 4); sh = *(const f4*)(a.s0.shift + xunit * 4); }
  const float vlo = (has && a.s0.relu) ? 0.f : -65504.f;
  int xpy[5], xpx[5];
#pragma unroll
  for (int rd = 0; rd < 5; ++rd) {
    const int pp = min((rd * 256 + tid) >> 3, kWPix - 1);
    xpy[rd] = pp / kWPW; xpx[rd] = pp - xpy[rd] * kWPW;
  }
  const bool xlast = (4 * 256 + tid) < kWPix * 8;
  f4 xv[5], yv[4]; unsigned xok = 0;
  auto tile_origin = [&](int t, int& n, int& h0, int& w0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    h0 = th * kWH; w0 = tw * kWW;
  };
  auto stage_load = [&](int t) {
    int n, h0, w0; tile_origin(t, n, h0, w0);
    xok = 0;
#pragma unroll
    for (int rd = 0; rd < 5; ++rd) {
      const int hl = h0 - 1 + xpy[rd], wl = w0 - 1 + xpx[rd];
      const bool ok = hl >= 0 && hl < Hs && wl >= 0 && wl < Wsrc;
      const int hc = min(max(hl, 0), Hs - 1), wc = min(max(wl, 0), Wsrc - 1);
      xv[rd] = *(const f4*)(a.s0.ptr + (((size_t)n * Hs + hc) * Wsrc + wc) * 32 + xunit * 4);
      xok |= (ok ? 1u : 0u) << rd;
    }
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const int px = (rd * 256 + tid) >> 2;                // dY pixel of the 4 x 64 tile
      yv[rd] = *(const f4*)(a.dy + (((size_t)n * a.Ho + 2 * h0 + (px >> 6)) * a.Wo + 2 * w0 + (px & 63)) * 16 + yunit * 4);
    }
  };
  auto stage_store = [&](int buf) {
    char* const xb = smem_ + buf * kWBuf;
    char* const yb = xb + kWX;
#pragma unroll
    for (int rd = 0; rd < 5; ++rd) {
      f4 v = xv[rd];
      if (has) v = v * sc + sh;
      const bool ok = (xok >> rd) & 1u;
      const float top = ok ? 65504.f : vlo;
      v.x = __builtin_amdgcn_fmed3f(v.x, vlo, top); v.y = __builtin_amdgcn_fmed3f(v.y, vlo, top);
      v.z = __builtin_amdgcn_fmed3f(v.z, vlo, top); v.w = __builtin_amdgcn_fmed3f(v.w, vlo, top);
      if (vlo != 0.f && !ok) v = (f4){0.f, 0.f, 0.f, 0.f};
      uwm_u2 hi, lo;
      uwm_split4(v.x, v.y, v.z, v.w, hi, lo);
      const int pp = (rd * 256 + tid) >> 3;
      if (rd < 4 || xlast) { *(uwm_u2*)(xb + pp * 128 + xunit * 8) = hi; *(uwm_u2*)(xb + pp * 128 + 64 + xunit * 8) = lo; }
    }
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      const f4 v = yv[rd] * xs;                           // (below 2^14 by construction: no clamp)
      uwm_u2 hi, lo;
      uwm_split4(v.x, v.y, v.z, v.w, hi, lo);
      const int px = (rd * 256 + tid) >> 2;
      *(uwm_u2*)(yb + px * 64 + yunit * 8) = hi; *(uwm_u2*)(yb + px * 64 + 32 + yunit * 8) = lo;
    }
  };
  // ---- fragment addresses: lane = (k-group kg, row-in-group q, quad p): low-resolution column 8 kg + q (+4)
  const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int yo = (2 * kr * 64 + 2 * (8 * kg + q)) * 64 + p * 8;                    // + (pa * 64 + pb) * 64; second read + 8 px = + 512 B; lo + 32
  const int xo = (kr * kWPW + 8 * kg + q) * 128 + cb * 32 + p * 8;                 // + (py * kWPW + px) * 128; second read + 4 px = + 512 B; lo + 64

  f4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};

  int t = blockIdx.x;
  if (t < ntiles) { stage_load(t); stage_store(0); }
  __syncthreads();
  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    const bool more = tn < ntiles;
    if (more) stage_load(tn);
    const char* const xb = smem_ + cur * kWBuf;
    const char* const yb = xb + kWX;
    h8 ah[2][2], al[2][2];
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const char* const ya = yb + yo + (pa * 64 + pb) * 64;
        ah[pa][pb] = u_tr_pair(ya, 0, 512); al[pa][pb] = u_tr_pair(ya, 32, 512 + 32);
      }
#pragma unroll
    for (int py = 0; py < 3; ++py)
#pragma unroll
      for (int px = 0; px < 3; ++px) {
        const char* const xa = xb + xo + (py * kWPW + px) * 128;
        const h8 bh = u_tr_pair(xa, 0, 512), bl = u_tr_pair(xa, 64, 512 + 64);
        // (py, px) = (pa + dy_, pb + dx_): the class products this neighbourhood pixel feeds
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            const int dy_ = py - pa, dx_ = px - pb;
            if (dy_ >= 0 && dy_ < 2 && dx_ >= 0 && dx_ < 2) {
              const int pi = ((pa * 2 + dy_) * 2 + pb) * 2 + dx_;
              f4 c = acc[pi];
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[pa][pb], bl, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[pa][pb], bh, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[pa][pb], bh, c, 0, 0, 0);
              acc[pi] = c;
            }
          }
      }
    if (more) stage_store(cur ^ 1);
    __syncthreads();
  }

  // ---- the two tile rows of a channel block: waves 2, 3 -> LDS -> waves 0, 1, which store the workgroup partial [16][2 cb][64][4]
  f4* const red = (f4*)smem_;                             // [2 cb][16][64] f4 = 32 KB
  const float ixs = 1.f / xs;
  if (kr == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(cb * 16 + i) * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (kr == 0) {
    f4* const out = (f4*)(a.part + (size_t)blockIdx.x * kWPart);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[(i * 2 + cb) * 64 + lane] = (acc[i] + red[(cb * 16 + i) * 64 + lane]) * ixs;
  }
}

// the main launch (conv_up2.hip's launch_wgrad_up2 sizes the scratch and runs wgrad_up2_reduce_kernel behind it): returns the
// number of workgroup partials
bool wgrad_up2_f16_shape(const WgradArgs& a) { return (a.s0.H % kWH) == 0 && (a.s0.W % kWW) == 0; }
int wgrad_up2_f16_parts(const WgradArgs& a) {
  const int ntiles = a.N * (a.s0.H / kWH) * (a.s0.W / kWW);
  return ntiles < 2 * device_cu_count() ? ntiles : 2 * device_cu_count();
}
hipError_t launch_wgrad_up2_f16(const WgradArgs& a, hipStream_t st) {
  const int ntiles = a.N * (a.s0.H / kWH) * (a.s0.W / kWW);
  const int nwg = wgrad_up2_f16_parts(a);
  const size_t lds = (size_t)2 * kWBuf;
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_up2_f16_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(47, a.flops, a.bytes, wgrad_up2_f16_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
  return hipGetLastError();
}

}  // namespace uwm
