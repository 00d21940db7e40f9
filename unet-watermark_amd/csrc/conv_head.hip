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
//
// Round 4: both kernels stage their 3x3 neighbourhood in LDS.  Before, a lane loaded its three columns of every input row itself
// (the neighbours' columns through L1): 18 16-byte global loads with their bounds tests per 4 outputs, the lazy BatchNorm + ReLU
// evaluated three times per element (forward), and 94 VGPRs in the dgrad (5 waves per SIMD): 126 / 227 us for the 16 x 512 x 512
// head against 50 / 95 us of HBM time.  Now a workgroup = 256 / CQ columns x TH rows: the (TH+2) x (cols+2) halo tile is loaded once
// with contiguous rows, activated once, zero padded in LDS, and the lanes read it back with compile-time offsets (lane-linear
// ds_read_b128: conflict-free).
// Forward: PERSISTENT workgroups with the tile double-buffered (the next tile's global loads are issued before this tile's dot
// products and land in registers behind them; one barrier per tile): the one-tile-per-workgroup form measured 116-124 us whatever
// the tile height and slower with fewer workgroups per CU, i.e. bound by its own load -> barrier -> compute -> exit chain.
constexpr int kHeadTH = 4;              // forward: output rows per tile
template <int CQ, int NCO, int TH = kHeadTH>
__global__ __launch_bounds__(256, 3) void conv_head_kernel(const ConvArgs a, int tilesW, int tilesH, int ntiles) {
  constexpr int C = CQ * 4, COLS = 256 / CQ, PW = COLS + 2;
  constexpr int NV = (TH + 2) * PW * CQ, NR = (NV + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f4* const ws = (f4*)smem;                              // [NCO][9][CQ]
  f4* const xs0 = ws + NCO * 9 * CQ;                     // [2][(TH+2)][PW][CQ]
  for (int i = threadIdx.x; i < NCO * 9 * CQ; i += 256) {
    const int co = i / (9 * CQ), r = i - co * 9 * CQ;                // r = tap * CQ + cq
    ws[i] = co < a.wrows ? *(const f4*)(a.w + (size_t)co * a.Kpad + r * 4) : (f4){0.f, 0.f, 0.f, 0.f};
  }
  const int H = a.Ho, W = a.Wo;
  const bool lazy = a.s0.scale != nullptr;
  const int relu = a.s0.relu;
  // ---- staging: (TH+2) x PW pixels x CQ quads, consecutive threads along (pixel, quad) = contiguous memory per row
  static_assert(256 % CQ == 0, "a thread stages ONE channel quad");
  const int q = threadIdx.x % CQ, col = threadIdx.x / CQ;      // (rows are PW * CQ units long and 256 % CQ == 0: the quad of a thread's units never changes)
  f4 lsc = {1.f, 1.f, 1.f, 1.f}, lsh = {0.f, 0.f, 0.f, 0.f};
  if (lazy) { lsc = *(const f4*)(a.s0.scale + q * 4); lsh = *(const f4*)(a.s0.shift + q * 4); }
  int srow[NR], spx[NR];
#pragma unroll
  for (int rd = 0; rd < NR; ++rd) {
    const int it = rd * 256 + threadIdx.x;
    srow[rd] = it / (PW * CQ); spx[rd] = (it - srow[rd] * (PW * CQ)) / CQ;
  }
  f4 pv[NR]; unsigned pok = 0;
  auto tile_origin = [&](int t, int& n, int& ho0, int& w0) {
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    ho0 = th * TH; w0 = tw * COLS;
  };
  auto tile_load = [&](int t) {
    int n, ho0, w0; tile_origin(t, n, ho0, w0);
    const float* xn = a.s0.ptr + (size_t)n * H * W * C + q * 4;
    pok = 0;
#pragma unroll
    for (int rd = 0; rd < NR; ++rd) {
      const int hi = ho0 - 1 + srow[rd], wi = w0 - 1 + spx[rd];
      const bool ok = (rd * 256 + (int)threadIdx.x) < NV && hi >= 0 && hi < H && wi >= 0 && wi < W;
      const int hc = min(max(hi, 0), H - 1), wc = min(max(wi, 0), W - 1);
      pv[rd] = *(const f4*)(xn + ((size_t)hc * W + wc) * C);
      pok |= (ok ? 1u : 0u) << rd;
    }
  };
  auto tile_store = [&](int buf) {
    f4* const xb = xs0 + buf * NV;
#pragma unroll
    for (int rd = 0; rd < NR; ++rd) {
      f4 v = pv[rd];
      if (lazy) {
        v = v * lsc + lsh;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      if (!((pok >> rd) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};          // zero padding applies AFTER the producer's activation
      if (rd * 256 + (int)threadIdx.x < NV) xb[rd * 256 + threadIdx.x] = v;
    }
  };
  int t = blockIdx.x;
  tile_load(t);
  tile_store(0);
  __syncthreads();
  f4 wr[NCO == 1 ? 9 : 1];                               // one class (the reference's watermark mask): its nine filter quads in registers
  if (NCO == 1) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = ws[k * CQ + q];
  }
  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tn = t + (int)gridDim.x;
    tile_load(tn < ntiles ? tn : t);                     // (last tile: harmless re-read)
    int n, ho0, w0; tile_origin(t, n, ho0, w0);
    const f4* const xs = xs0 + cur * NV;
    // one class at a time (the class loop is NOT unrolled: with 4 classes the unrolled form held every filter quad in registers: 256 VGPRs)
    f4 res[TH];
#pragma unroll
    for (int j = 0; j < TH; ++j) res[j] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int co = 0; co < NCO; ++co) {
      float acc[TH];
#pragma unroll
      for (int j = 0; j < TH; ++j) acc[j] = 0.f;
      const f4* const wc = ws + co * 9 * CQ + q;
#pragma unroll
      for (int rr = 0; rr < TH + 2; ++rr)
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
          const f4 v = xs[(rr * PW + col + s_) * CQ + q];
#pragma unroll
          for (int j = 0; j < TH; ++j) {
            const int r = rr - j;                         // compile-time after unrolling
            if (r >= 0 && r < 3) {
              const f4 wv = NCO == 1 ? wr[r * 3 + s_] : wc[(r * 3 + s_) * CQ];
              acc[j] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
            }
          }
        }
      // add the CQ channel-quad partials of each pixel (adjacent lanes: every lane ends with the sum)
      const float b = (a.bias && co < a.wrows) ? a.bias[co] : 0.f;
#pragma unroll
      for (int j = 0; j < TH; ++j) {
        float v = acc[j];
#pragma unroll
        for (int d = 1; d < CQ; d <<= 1) v += __shfl_xor(v, d);
        v += b;
        if (co == 0) res[j].x = v; else if (co == 1) res[j].y = v; else if (co == 2) res[j].z = v; else res[j].w = v;
      }
    }
    // lane q stores rows q, q + CQ, ...: one 16-byte store per pixel (classes padded to 4)
    const int wo = w0 + col;
    if (wo < W) {
#pragma unroll
      for (int j = 0; j < TH; ++j) {
        const int ho = ho0 + j;
        if ((j % CQ) == q && ho < H) *(f4*)(a.out + (((size_t)n * H + ho) * W + wo) * 4) = res[j];
      }
    }
    tile_store(cur ^ 1);
    __syncthreads();
  }
}

// dgrad of the same layer: dx[h][w][c] = sum over (r, s, class) dy[h+1-r][w+1-s][class] * W[class][c][r][s], then the ReLU
// mask of the tensor the head read.  dy has 4 (padded) channels, dx has C = 8 | 16 | 32: read 16 B, write C*4 B (+ mask
// C*4 B) per pixel.  A workgroup = 256 / CQ columns x TH rows (measured on the 16 x 512 x 512 head: 222 / 171 / 117 / 107 us with
// 1 / 2 / 4 / 8 rows; the 4-row kernel without the LDS tile: 227); the (TH+2) x (cols+2) tile of dy (its live classes) sits zero padded
// in LDS, the repacked filter wd[c][tap*4 + class] as [tap][class][channel quad] (one live class: in registers).
constexpr int kHeadDTH = 8;
template <int CQ, int NCO, int TH = kHeadDTH>
__global__ __launch_bounds__(256, TH >= 8 ? 2 : (TH >= 4 ? 3 : 4)) void conv_head_dgrad_kernel(const ConvArgs a) {
  constexpr int C = CQ * 4, COLS = 256 / CQ, PW = COLS + 2;
  __shared__ f4 ws[9 * NCO * CQ];
  __shared__ float gs[(TH + 2) * PW * NCO];
  for (int i = threadIdx.x; i < 9 * NCO * CQ; i += 256) {
    const int q = i % CQ, tc = i / CQ, co = tc % NCO, tap = tc / NCO;
    f4 v;
    v.x = a.w[(size_t)(q * 4 + 0) * a.Kpad + tap * 4 + co]; v.y = a.w[(size_t)(q * 4 + 1) * a.Kpad + tap * 4 + co];
    v.z = a.w[(size_t)(q * 4 + 2) * a.Kpad + tap * 4 + co]; v.w = a.w[(size_t)(q * 4 + 3) * a.Kpad + tap * 4 + co];
    ws[i] = v;
  }
  const int H = a.Ho, W = a.Wo;
  const int n = blockIdx.z, ho0 = blockIdx.y * TH, w0 = blockIdx.x * COLS;
  const float* gn = a.s0.ptr + (size_t)n * H * W * 4;
  for (int i = threadIdx.x; i < (TH + 2) * PW * NCO; i += 256) {
    const int co = i % NCO, px = i / NCO, row = px / PW, cx = px - row * PW;
    const int hh = ho0 - 1 + row, ww = w0 - 1 + cx;
    gs[i] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? gn[((size_t)hh * W + ww) * 4 + co] : 0.f;
  }
  __syncthreads();
  const int col = threadIdx.x / CQ, q = threadIdx.x % CQ;
  const int wo = w0 + col;
  const bool live = wo < W;                       // (no early return: the fused BatchNorm-backward sums end in a workgroup barrier)
  f4 mk[TH];
#pragma unroll
  for (int j = 0; j < TH; ++j) {                  // every row's mask load in flight behind the tap loop
    const size_t o = (((size_t)n * H + min(ho0 + j, H - 1)) * W + min(wo, W - 1)) * C + q * 4;
    mk[j] = a.mask ? *(const f4*)(a.mask + o) : (f4){0.f, 0.f, 0.f, 0.f};
  }
  f4 acc[TH];
#pragma unroll
  for (int j = 0; j < TH; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int co = 0; co < NCO; ++co) {              // (not unrolled over the classes: register pressure)
    const f4* const wc = ws + co * CQ + q;
    const float* const gc = gs + col * NCO + co;
#pragma unroll
    for (int rr = 0; rr < TH + 2; ++rr)
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        const float gv = gc[(rr * PW + s2) * NCO];
#pragma unroll
        for (int j = 0; j < TH; ++j) {
          const int r = j + 2 - rr;                       // filter row that maps output row ho0+j to dy row ho0-1+rr (compile-time)
          if (r >= 0 && r < 3) acc[j] += wc[(r * 3 + (2 - s2)) * NCO * CQ] * gv;
        }
      }
  }
  // fused BatchNorm-backward sums (uwm_kernels.h ConvArgs::bnb_*): the masked output IS the gradient wrt the BatchNorm output
  // whose raw input is the mask tensor: dbeta += v, dgamma += v * yhat, per workgroup -> one of srep fp64 replicas
  const bool bnb = a.bnb_mean != nullptr;
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = {0.f, 0.f, 0.f, 0.f}, bmu = ps_, brs = ps_, msc = {1.f, 1.f, 1.f, 1.f}, msh = ps_;
  if (bnb) { bmu = *(const f4*)(a.bnb_mean + q * 4); brs = *(const f4*)(a.bnb_rstd + q * 4); }
  if (a.mask && a.mscale) { msc = *(const f4*)(a.mscale + q * 4); msh = *(const f4*)(a.mshift + q * 4); }
#pragma unroll
  for (int j = 0; j < TH; ++j) {
    const int ho = ho0 + j;
    if (ho >= H || !live) continue;
    const size_t o = (((size_t)n * H + ho) * W + wo) * C + q * 4;
    f4 v = acc[j];
    if (a.addend) v += *(const f4*)(a.addend + o);
    if (a.mask) {
      const f4 yr = mk[j];
      const f4 m = yr * msc + msh;
      v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
      if (bnb) { ps_ += v; pq_ += v * ((yr - bmu) * brs); }
    }
    *(f4*)(a.out + o) = v;
  }
  if (bnb) {              // lanes = (pixel column, channel quad q = lane % CQ): the 64 / CQ columns of a quad across the wave, the four waves through LDS
    float v8[8] = {ps_.x, ps_.y, ps_.z, ps_.w, pq_.x, pq_.y, pq_.z, pq_.w};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int d = CQ; d < 64; d <<= 1) v8[i] += __shfl_xor(v8[i], d);
    __shared__ float red[4][CQ * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < CQ) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = v8[i];
    }
    __syncthreads();
    if (threadIdx.x < CQ * 8) {
      const int qq = threadIdx.x / 8, e = threadIdx.x % 8;
      const double sacc = ((double)red[0][threadIdx.x] + (double)red[1][threadIdx.x]) + ((double)red[2][threadIdx.x] + (double)red[3][threadIdx.x]);
      const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const size_t srep_off = a.srep > 1 ? (size_t)(wg & (unsigned)(a.srep - 1)) * a.sstride : 0;
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
  const dim3 g((unsigned)((a.Wo + 256 / CQ - 1) / (256 / CQ)), (unsigned)((a.Ho + kHeadDTH - 1) / kHeadDTH), (unsigned)a.N);
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

template <int CQ, int NCO>
static hipError_t launch_head_(const ConvArgs& a, hipStream_t st) {
  constexpr int COLS = 256 / CQ, TH = kHeadTH;
  const size_t lds = sizeof(float) * 4 * ((size_t)NCO * 9 * CQ + (size_t)2 * (TH + 2) * (COLS + 2) * CQ);
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_head_kernel<CQ, NCO>, lds); if (e != hipSuccess) return e; }
  const int tilesW = (a.Wo + COLS - 1) / COLS, tilesH = (a.Ho + TH - 1) / TH;
  const long ntiles = (long)a.N * tilesH * tilesW;
  static const int per_cu = dbg_int("UWM_HEADF_WGS", 3);
  const long cap = (long)per_cu * device_cu_count();
  UWM_LAUNCH(30, a.flops, a.bytes, (conv_head_kernel<CQ, NCO>), dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256), lds, st, a, tilesW, tilesH, (int)ntiles);
  return hipGetLastError();
}
template <int CQ>
static hipError_t launch_head(const ConvArgs& a, hipStream_t st) {
  return a.wrows == 1 ? launch_head_<CQ, 1>(a, st) : launch_head_<CQ, 4>(a, st);
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
