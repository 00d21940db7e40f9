// Segmentation head: 3x3 / stride 1 / pad 1 convolution from a few channels (8 | 16 | 32) to <= 4 classes at full
// resolution, with bias — smp SegmentationHead(16 -> classes, k=3) (/root/reference/src/models/
// unet_model.py:64-71 -> smp).  0.04 GMAC per image: pure HBM streaming (read C floats, write 4 per pixel), which the
// MFMA implicit GEMM served badly (a 128x16 tile for 1 live output channel, nine separate tap gathers: 486 us for the
// 16x16x512x512 layer against ~75 us of HBM time).  Here a lane owns one channel quad of an output column and walks the
// image in bands of TH = 4 rows: per input row it loads its three columns once (lanes run along (W, quad): contiguous
// memory, the neighbours' columns hit in L1), applies the producer's lazy BatchNorm scale/shift + ReLU, and feeds up to
// three output rows; the 9*C weights per class sit in LDS.  The quads' partial dot products are added across adjacent lanes;
// one 16-byte store per pixel (classes padded to 4).
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));

// Lane mapping (both kernels): consecutive lanes = the CQ channel quads of one pixel, then the next pixel along W, so
// every wave-level load / store covers contiguous memory (with one pixel per lane each instruction touched 64 B of every
// other 128-byte line: 4x the line requests; 189 -> ~100 us forward).
template <int CQ, int NCO>
__global__ __launch_bounds__(256) void conv_head_kernel(const ConvArgs a) {
  constexpr int TH = 4, C = CQ * 4;
  __shared__ f4 ws[NCO * 9 * CQ];
  for (int i = threadIdx.x; i < NCO * 9 * CQ; i += 256) {
    const int co = i / (9 * CQ), r = i - co * 9 * CQ;                // r = tap * CQ + cq
    ws[i] = co < a.wrows ? *(const f4*)(a.w + (size_t)co * a.Kpad + r * 4) : (f4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const int H = a.Ho, W = a.Wo;
  const int nbands = (H + TH - 1) / TH;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int pix = idx / CQ, q = idx - pix * CQ;          // CQ divides 256: the quads of a pixel sit in one wave
  const bool live = pix < nbands * W;
  const int band = live ? pix / W : 0, wo = live ? pix - band * W : 0;
  const int n = blockIdx.y, ho0 = band * TH;
  const bool lazy = a.s0.scale != nullptr;
  const int relu = a.s0.relu;
  const f4 sc = lazy ? *(const f4*)(a.s0.scale + q * 4) : (f4){1.f, 1.f, 1.f, 1.f};
  const f4 sh = lazy ? *(const f4*)(a.s0.shift + q * 4) : (f4){0.f, 0.f, 0.f, 0.f};
  float acc[TH][NCO];
#pragma unroll
  for (int j = 0; j < TH; ++j)
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[j][co] = 0.f;
  const float* xn = a.s0.ptr + (size_t)n * H * W * C + q * 4;
#pragma unroll
  for (int rr = 0; rr < TH + 2; ++rr) {
    const int hi = ho0 - 1 + rr;
    if (!live || hi < 0 || hi >= H) continue;
    const float* row = xn + (size_t)hi * W * C;
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      const int wi = wo - 1 + s_;
      if (wi < 0 || wi >= W) continue;                  // zero padding applies AFTER the producer's activation
      f4 v = *(const f4*)(row + (size_t)wi * C);
      if (lazy) {
        v = v * sc + sh;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
#pragma unroll
      for (int j = 0; j < TH; ++j) {
        const int r = rr - j;                           // compile-time after unrolling
        if (r >= 0 && r < 3) {
#pragma unroll
          for (int co = 0; co < NCO; ++co) {
            const f4 wv = ws[(co * 9 + r * 3 + s_) * CQ + q];
            acc[j][co] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
          }
        }
      }
    }
  }
  // add the CQ channel-quad partials of each pixel (adjacent lanes), lane q == 0 stores
#pragma unroll
  for (int j = 0; j < TH; ++j)
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      float v = acc[j][co];
#pragma unroll
      for (int d = 1; d < CQ; d <<= 1) v += __shfl_xor(v, d);
      acc[j][co] = v;
    }
  if (live && q == 0) {
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int ho = ho0 + j;
      if (ho < H) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int co = 0; co < NCO; ++co) o[co] = acc[j][co] + ((a.bias && co < a.wrows) ? a.bias[co] : 0.f);
        *(f4*)(a.out + (((size_t)n * H + ho) * W + wo) * 4) = (f4){o[0], o[1], o[2], o[3]};
      }
    }
  }
}

// dgrad of the same layer: dx[h][w][c] = sum over (r, s, class) dy[h+1-r][w+1-s][class] * W[class][c][r][s], then the ReLU
// mask of the tensor the head read.  dy has 4 (padded) channels, dx has C = 8 | 16 | 32: read 16 B, write C*4 B (+ mask
// C*4 B) per pixel.  Same walk and lane mapping: a lane owns one channel quad of a pixel column and bands of 4 rows, loads
// its three columns of dy once per row (the quad's lanes share the address) and feeds up to three output rows; the
// repacked filter wd[c][tap*4 + class] sits in LDS as [tap][class][channel quad].
template <int CQ, int NCO>
__global__ __launch_bounds__(256) void conv_head_dgrad_kernel(const ConvArgs a) {
  constexpr int TH = 4, C = CQ * 4;
  __shared__ f4 ws[9 * NCO * CQ];
  for (int i = threadIdx.x; i < 9 * NCO * CQ; i += 256) {
    const int q = i % CQ, tc = i / CQ, co = tc % NCO, tap = tc / NCO;
    f4 v;
    v.x = a.w[(size_t)(q * 4 + 0) * a.Kpad + tap * 4 + co]; v.y = a.w[(size_t)(q * 4 + 1) * a.Kpad + tap * 4 + co];
    v.z = a.w[(size_t)(q * 4 + 2) * a.Kpad + tap * 4 + co]; v.w = a.w[(size_t)(q * 4 + 3) * a.Kpad + tap * 4 + co];
    ws[i] = v;
  }
  __syncthreads();
  const int H = a.Ho, W = a.Wo;
  const int nbands = (H + TH - 1) / TH;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int pix = idx / CQ, q = idx - pix * CQ;
  const bool live = pix < nbands * W;             // (no early return: the fused BatchNorm-backward sums end in a workgroup barrier)
  const int band = live ? pix / W : 0, wo = live ? pix - band * W : 0;
  const int n = blockIdx.y, ho0 = band * TH;
  f4 acc[TH];
#pragma unroll
  for (int j = 0; j < TH; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
  const float* gn = a.s0.ptr + (size_t)n * H * W * 4;
#pragma unroll
  for (int rr = 0; rr < TH + 2; ++rr) {
    const int hh = ho0 - 1 + rr;
    if (hh < 0 || hh >= H || !live) continue;
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      const int ww = wo - 1 + s2;
      if (ww < 0 || ww >= W) continue;
      const f4 g = *(const f4*)(gn + ((size_t)hh * W + ww) * 4);
      const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int j = 0; j < TH; ++j) {
        const int r = j + 2 - rr;                       // filter row that maps output row ho0+j to dy row hh (compile-time)
        if (r >= 0 && r < 3) {
          const int tap = r * 3 + (2 - s2);
#pragma unroll
          for (int co = 0; co < NCO; ++co) acc[j] += ws[(tap * NCO + co) * CQ + q] * gv[co];
        }
      }
    }
  }
  // fused BatchNorm-backward sums (uwm_kernels.h ConvArgs::bnb_*): the masked output IS the gradient wrt the BatchNorm output
  // whose raw input is the mask tensor: dbeta += v, dgamma += v * yhat, per workgroup -> one of srep fp64 replicas
  const bool bnb = a.bnb_mean != nullptr;
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = {0.f, 0.f, 0.f, 0.f}, bmu = ps_, brs = ps_;
  if (bnb) { bmu = *(const f4*)(a.bnb_mean + q * 4); brs = *(const f4*)(a.bnb_rstd + q * 4); }
#pragma unroll
  for (int j = 0; j < TH; ++j) {
    const int ho = ho0 + j;
    if (ho >= H || !live) continue;
    const size_t o = (((size_t)n * H + ho) * W + wo) * C + q * 4;
    f4 v = acc[j];
    if (a.addend) v += *(const f4*)(a.addend + o);
    if (a.mask) {
      f4 mk = *(const f4*)(a.mask + o);
      const f4 yr = mk;
      if (a.mscale) mk = mk * *(const f4*)(a.mscale + q * 4) + *(const f4*)(a.mshift + q * 4);
      v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      if (bnb) { ps_ += v; pq_ += v * ((yr - bmu) * brs); }
    }
    *(f4*)(a.out + o) = v;
  }
  if (bnb) {              // threads = (pixel column, channel quad q = idx % CQ): sum the 256 / CQ columns of a quad through LDS
    __shared__ float red[256 * 8];
    float* r = red + threadIdx.x * 8;
    r[0] = ps_.x; r[1] = ps_.y; r[2] = ps_.z; r[3] = ps_.w; r[4] = pq_.x; r[5] = pq_.y; r[6] = pq_.z; r[7] = pq_.w;
    __syncthreads();
    if (threadIdx.x < CQ * 8) {
      const int qq = threadIdx.x / 8, e = threadIdx.x % 8;
      // thread t holds quad (blockIdx.x * 256 + t) % CQ = (t + base) % CQ with base = (blockIdx.x * 256) % CQ = 0 (256 % CQ == 0)
      double sacc = 0.0;
      for (int k = qq; k < 256; k += CQ) sacc += (double)red[k * 8 + e];
      const size_t srep_off = a.srep > 1 ? (size_t)((blockIdx.x + blockIdx.y * gridDim.x) & (unsigned)(a.srep - 1)) * a.sstride : 0;
      if (e < 4) atomicAdd(a.ssum + srep_off + qq * 4 + e, sacc); else atomicAdd(a.ssq + srep_off + qq * 4 + (e - 4), sacc);
    }
  }
}

bool conv_head_dgrad_applicable(const ConvArgs& a) {
  return a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && a.rmul == -1 && a.off == 1 && a.Ctot == 4 && a.C0 == 4 && a.s0.C == 4 &&
         (a.Cout == 8 || a.Cout == 16 || a.Cout == 32) && a.wrows == a.Cout && a.s0.up == 0 && !a.s0.scale && a.Ho == a.Hl && a.Wo == a.Wl &&
         (!a.ssum || a.bnb_mean) && !a.bias && !a.out_up;
}
template <int CQ>
static hipError_t launch_head_dgrad(const ConvArgs& a, hipStream_t st) {
  const int nbands = (a.Ho + 3) / 4;
  const dim3 g((unsigned)(((size_t)nbands * a.Wo * CQ + 255) / 256), (unsigned)a.N);
  if (a.live_ch == 1) UWM_LAUNCH(30, a.flops, a.bytes, (conv_head_dgrad_kernel<CQ, 1>), g, dim3(256), 0, st, a);
  else UWM_LAUNCH(30, a.flops, a.bytes, (conv_head_dgrad_kernel<CQ, 4>), g, dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_conv_head_dgrad(const ConvArgs& a, hipStream_t st) {
  if (!conv_head_dgrad_applicable(a)) return hipErrorInvalidValue;
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !a.mask)) return hipErrorInvalidValue;
  switch (a.Cout) {
    case 8: return launch_head_dgrad<2>(a, st);
    case 16: return launch_head_dgrad<4>(a, st);
    default: return launch_head_dgrad<8>(a, st);
  }
}

bool conv_head_applicable(const ConvArgs& a) {
  return a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && a.rmul == 1 && a.off == -1 && a.Cout == 4 && a.wrows >= 1 && a.wrows <= 4 &&
         (a.Ctot == 8 || a.Ctot == 16 || a.Ctot == 32) && a.C0 == a.Ctot && a.s0.C == a.Ctot && a.s0.up == 0 && a.Ho == a.Hl && a.Wo == a.Wl &&
         !a.ssum && !a.addend && !a.mask && !a.out_up;
}

template <int CQ>
static hipError_t launch_head(const ConvArgs& a, hipStream_t st) {
  const int nbands = (a.Ho + 3) / 4;
  const dim3 g((unsigned)(((size_t)nbands * a.Wo * CQ + 255) / 256), (unsigned)a.N);
  if (a.wrows == 1) UWM_LAUNCH(30, a.flops, a.bytes, (conv_head_kernel<CQ, 1>), g, dim3(256), 0, st, a);
  else UWM_LAUNCH(30, a.flops, a.bytes, (conv_head_kernel<CQ, 4>), g, dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_conv_head(const ConvArgs& a, hipStream_t st) {
  if (!conv_head_applicable(a) || a.bnb_mean) return hipErrorInvalidValue;      // (the forward head has no BatchNorm-backward epilogue)
  switch (a.Ctot) {
    case 8: return launch_head<2>(a, st);
    case 16: return launch_head<4>(a, st);
    default: return launch_head<8>(a, st);
  }
}

}  // namespace uwm
