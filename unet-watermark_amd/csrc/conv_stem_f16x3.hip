// The ResNet stem — 7x7 / stride 2 / pad 3 convolution of the 3(+1 pad)-channel image to 64 channels — as an fp16x3 kernel for gfx950
// (the arithmetic of conv_f16x3.hip: fp32 operands split into two fp16 halves, a*b = ah*bh + ah*bl + al*bh on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation, filter rows scaled by an exact power of two; the image is taken as it is).
//
// Why its own kernel: K = 7*7*3 = 147.  The flattened implicit GEMM pads that to 224 fp32 columns and is bound by its matrix pipe
// (345 us at batch 16 x 512^2: 30 GFLOP executed on the 157-TFLOP/s fp32 pipe, against 268 MB of output = a 65-us HBM floor).  Here one
// MFMA k-step is one KERNEL ROW: 7 taps x 4 channels = 28 of the 32 k indices (k = 8g + e: tap 2g + (e >> 2), channel e & 3; tap 7 is
// zero), so a lane's B fragment — 2 adjacent input pixels x 4 channels — is ONE 16-byte LDS read of the [row][pixel][4 ch] patch,
// and the stride of 2 between output pixels makes the 16 lanes of a fragment read 256 contiguous bytes.  7 k-steps x 3 products:
// 21 half-precision MFMAs per 16 pixels x 16 channels against 56 fp32 ones at twice the cycles each.
//
// Workgroup = 16 x 16 output pixels x 64 channels, 4 waves (wave w = output rows 4w .. 4w+3, all four channel fragments): the
// 37 x 38-pixel input patch is loaded once (six 16-byte loads per thread, all in flight together), split, and kept in LDS as a hi
// and a lo plane; the filter fragments come from L2 in MFMA lane order (stem_f16x3_weights_kernel packs
// [7 rows][4 channel fragments][hi | lo][64 lanes][8 halfs] + rinv[64] once per step), one kernel row ahead.  Epilogue =
// conv_f16x3_kernel's plain form: through LDS, 256 contiguous bytes per pixel, row un-scale, optional bias, BatchNorm statistics.
//
// Reference semantics replaced: encoder.conv1 of smp's ResNet encoders (/root/reference/src/models/unet_model.py:64-71 ->
// SURVEY.md §8 a3).
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int kSTile = 16;                              // output pixels per tile side
constexpr int kSPH = 2 * (kSTile - 1) + 7;              // patch rows: 37
constexpr int kSPWu = 2 * (kSTile - 1) + 8;             // patch pixels read per row: 38 (tap slot 7 reads one past the 7th tap)
constexpr int kSPW = 40;                                // row pitch in pixels (320 bytes per plane row: 16-byte aligned fragments)
constexpr int kSPlane = kSPH * kSPW * 4;                // halfs per plane
constexpr int kSRounds = (kSPH * kSPWu + 255) / 256;    // 16-byte units (pixels) per thread: 6

size_t stem_f16x3_bank_floats() { return (size_t)(7 * 4 * 2 * 64 * 8) / 2 + 64; }
static inline __host__ __device__ size_t stem_rinv_off_floats() { return (size_t)(7 * 4 * 2 * 64 * 8) / 2; }

// one workgroup per 16-channel fragment: row maxima -> power-of-two row scales -> the fragments of the 7 kernel rows
__global__ __launch_bounds__(256) void stem_f16x3_weights_kernel(const float* __restrict__ w, int Kpad, int cin_p, float* __restrict__ bank) {
  __shared__ float sc[16];
  const int j = blockIdx.x, tid = threadIdx.x;
  {
    const int r = tid >> 4, q = tid & 15;
    const float* wr = w + (size_t)(j * 16 + r) * Kpad;
    float mx = 0.f;
    for (int i = q; i < 49 * cin_p; i += 16) mx = fmaxf(mx, fabsf(wr[i]));
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    float s = 1.f;
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); s = ldexpf(1.f, 14 - e); }
    if (q == 0) { sc[r] = s; bank[stem_rinv_off_floats() + j * 16 + r] = 1.f / s; }
  }
  __syncthreads();
  const int lane = tid & 63, row = lane & 15, g = lane >> 4;
  const float s = sc[row];
  const float* wr = w + (size_t)(j * 16 + row) * Kpad;
  _Float16* hb = (_Float16*)bank;
  for (int kh = tid >> 6; kh < 7; kh += 4) {
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kw = 2 * g + (e >> 2), c = e & 3;
      const float v = (kw < 7 && c < cin_p) ? wr[(kh * 7 + kw) * cin_p + c] * s : 0.f;
      const _Float16 h = (_Float16)v;
      hi[e] = h; lo[e] = (_Float16)(v - (float)h);
    }
    h8* dst = (h8*)(hb + ((size_t)(kh * 4 + j) * 2) * 512 + lane * 8);
    dst[0] = hi;
    dst[64] = lo;                                       // plane 1: + 512 halfs
  }
}
hipError_t launch_stem_f16x3_weights(const float* w, int Kpad, int cin_p, float* bank, hipStream_t st) {
  hipLaunchKernelGGL(stem_f16x3_weights_kernel, dim3(4), dim3(256), 0, st, w, Kpad, cin_p, bank);
  return hipGetLastError();
}

__device__ __forceinline__ float clamp_hs(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }

__global__ __launch_bounds__(256, 2) void conv_stem_f16x3_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 ssm[];      // [hi | lo][37][40][4]; the epilogue's [4][64][68] floats alias it
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px16 = lane & 15, g = lane >> 4;
  const int tilesW = (a.Wo + kSTile - 1) / kSTile, tilesH = (a.Ho + kSTile - 1) / kSTile;
  unsigned tile = blockIdx.x;
  const int tw = tile % tilesW; tile /= tilesW;
  const int th = tile % tilesH; const int n = tile / tilesH;
  const int h0 = th * kSTile, w0 = tw * kSTile;
  const int H = a.s0.H, W = a.s0.W;

  // ---- the patch: input rows 2 h0 - 3 + py, columns 2 w0 - 3 + px; all loads first, then the split
  f4 pv[kSRounds];
  const float* const img = a.s0.ptr + (size_t)n * H * W * 4;
#pragma unroll
  for (int rd = 0; rd < kSRounds; ++rd) {
    const int u = rd * 256 + tid;
    const int py = u / kSPWu, px = u - py * kSPWu;
    const int hi_ = 2 * h0 - 3 + py, wi = 2 * w0 - 3 + px;
    const bool ok = u < kSPH * kSPWu && hi_ >= 0 && hi_ < H && wi >= 0 && wi < W;
    pv[rd] = ok ? *(const f4*)(img + ((size_t)hi_ * W + wi) * 4) : (f4){0.f, 0.f, 0.f, 0.f};
  }
  f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  const _Float16* const wb = (const _Float16*)a.wu + lane * 8;
  auto w_load = [&](int kh, h8 (&whi)[4], h8 (&wlo)[4]) {
    const _Float16* p = wb + (size_t)kh * 4 * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) { whi[j] = *(const h8*)(p + j * 1024); wlo[j] = *(const h8*)(p + j * 1024 + 512); }
  };
  h8 wA_hi[4], wA_lo[4], wB_hi[4], wB_lo[4];
  w_load(0, wA_hi, wA_lo);
#pragma unroll
  for (int rd = 0; rd < kSRounds; ++rd) {
    const int u = rd * 256 + tid;
    if (u < kSPH * kSPWu) {
      const int py = u / kSPWu, px = u - py * kSPWu;
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = clamp_hs(pv[rd][e]);
        const _Float16 h = (_Float16)x;
        hi[e] = h; lo[e] = (_Float16)(x - (float)h);
      }
      _Float16* d = ssm + (py * kSPW + px) * 4;
      *(h4*)d = hi;
      *(h4*)(d + kSPlane) = lo;
    }
  }
  __syncthreads();

  // ---- 7 k-steps = 7 kernel rows.  B fragment of output pixel (4 wave + i, px16), lane group g: input row 2 (4 wave + i) + kh,
  // pixels 2 px16 + 2g, + 1
  const int xbase = ((2 * (wave * 4)) * kSPW + 2 * px16 + 2 * g) * 4;
  auto mma_row = [&](int kh, const h8 (&whi)[4], const h8 (&wlo)[4]) {
    h8 xh[4], xl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const _Float16* pp = ssm + xbase + ((2 * i + kh) * kSPW) * 4;
      xh[i] = *(const h8*)pp; xl[i] = *(const h8*)(pp + kSPlane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xl[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[j], xh[i], acc[i][j], 0, 0, 0);
  };
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    if ((kh & 1) == 0) { if (kh + 1 < 7) w_load(kh + 1, wB_hi, wB_lo); mma_row(kh, wA_hi, wA_lo); }
    else { if (kh + 1 < 7) w_load(kh + 1, wA_hi, wA_lo); mma_row(kh, wB_hi, wB_lo); }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                                      // the patch is dead: its LDS becomes the epilogue's

  // ---------------- epilogue (conv_f16x3_kernel's plain form)
  constexpr int kQLd = 68;
  float* const R = (float*)ssm + wave * 64 * kQLd;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f4*)(R + (i * 16 + px16) * kQLd + j * 16 + g * 4) = acc[i][j];
  __syncthreads();
  const float* rinv = (const float*)a.wu + stem_rinv_off_floats();
  const bool do_stats = a.ssum != nullptr;
  const int cq = lane & 15, sub = lane >> 4;
  const int co = cq * 4;
  const bool cok = co < a.Cout;
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = ps_;
  f4 rs = {0.f, 0.f, 0.f, 0.f}, bia = rs;
  if (cok) rs = *(const f4*)(rinv + co);
  if (a.bias && cok) bia = *(const f4*)(a.bias + co);
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int p = r * 4 + sub;
    const int ho = h0 + wave * 4 + (p >> 4), wo = w0 + (p & 15);
    if (ho < a.Ho && wo < a.Wo && cok) {
      const size_t o = (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co;
      const f4 v = *(const f4*)(R + p * kQLd + cq * 4) * rs + bia;
      *(f4*)(a.out + o) = v;
      ps_ += v; pq_ += v * v;
    }
  }
  if (do_stats) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sv = ps_[e], qv = pq_[e];
      sv += __shfl_xor(sv, 16); qv += __shfl_xor(qv, 16);
      sv += __shfl_xor(sv, 32); qv += __shfl_xor(qv, 32);
      ps_[e] = sv; pq_[e] = qv;
    }
    __syncthreads();                                    // every wave is done with its block
    float* red = (float*)ssm;                           // [4 waves][64][2]
    if (sub == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[(wave * 64 + cq * 4 + e) * 2] = ps_[e]; red[(wave * 64 + cq * 4 + e) * 2 + 1] = pq_[e]; }
    }
    __syncthreads();
    if (tid < 64 && tid < a.Cout) {
      double sv = 0.0, qv = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { sv += (double)red[(w * 64 + tid) * 2]; qv += (double)red[(w * 64 + tid) * 2 + 1]; }
      const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
      atomicAdd(a.ssum + srep_off + tid, sv);
      atomicAdd(a.ssq + srep_off + tid, qv);
    }
  }
}

// 7x7 / stride 2 / pad 3 over one 4-channel (3 + pad) plain source, 64 outputs, no mask / addend / fused backward sums
bool conv_stem_f16x3_applicable(const ConvArgs& a) {
  return a.wu != nullptr && a.ntaps == 49 && a.kw == 7 && a.smul == 2 && a.sdiv == 1 && a.rmul == 1 && a.off == -3 &&
         a.Ctot == 4 && a.C0 == 4 && a.s0.C == 4 && a.s0.up == 0 && a.s0.scale == nullptr && a.Cout == 64 && a.wrows == 64 &&
         !a.mask && !a.addend && !a.out_up && !a.bnb_mean && !a.bnb_y && a.Hl == a.s0.H && a.Wl == a.s0.W &&
         a.Ho == (a.Hl + 6 - 7) / 2 + 1 && a.Wo == (a.Wl + 6 - 7) / 2 + 1 && (size_t)a.N * a.Hl * a.Wl * 4 < (1ull << 31);
}

hipError_t launch_conv_stem_f16x3(const ConvArgs& a, hipStream_t st) {
  if (!conv_stem_f16x3_applicable(a)) return hipErrorInvalidValue;
  const int tilesW = (a.Wo + kSTile - 1) / kSTile, tilesH = (a.Ho + kSTile - 1) / kSTile;
  const size_t patch = (size_t)2 * kSPlane * sizeof(_Float16), q_lds = (size_t)4 * 64 * 68 * sizeof(float);
  const size_t lds = patch > q_lds ? patch : q_lds;
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_stem_f16x3_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(44, a.flops, a.bytes, conv_stem_f16x3_kernel, dim3((unsigned)(a.N * tilesH * tilesW)), dim3(256), lds, st, a);
  return hipGetLastError();
}

}  // namespace uwm
