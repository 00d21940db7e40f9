// 3x3 / stride 1 / pad 1 convolution (forward and dgrad) for inputs with exactly 16 channels on gfx950
// (v_mfma_f32_16x16x4_f32): the full-resolution tail of the decoder (decoder block 4 conv2, the
// segmentation head, and every dgrad whose incoming gradient has 16 channels).
//
// K = 9 taps x 16 channels = 144 is so small that nothing needs a K loop: a workgroup stages the WHOLE
// weight panel [BN][144] (9 KB for BN=16) and one 16-channel 18x18-pixel halo patch (20 KB) in LDS once,
// passes ONE barrier, and runs 9 taps x (4 pixel rows x NI channel tiles) x 4 MFMAs per wave straight out
// of LDS; tap (r,s) is an LDS address offset.  ~30 KB of LDS and < 100 VGPRs per workgroup put 5 workgroups
// on a CU, so the loads of one overlap the MFMAs of the others.  The flattened implicit GEMM needed a
// barrier every 8-16 MFMAs on these layers (20 TFLOP/s, profiles/r01_d).
// Loader / epilogue contracts are those of conv_igemm.hip (lazy BatchNorm+ReLU input, bias, residual
// addend, ReLU mask, BatchNorm statistics).
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kT = 16, kP = kT + 2, kPPix = kP * kP;      // 16x16 output pixels, 18x18 = 324 patch pixels
constexpr int kWLd = 148;                                  // weight row stride in floats (37 16-B units: odd)

// Persistent since round 2: a workgroup walks tiles bid, bid + grid, ... with the patch double-buffered (the next tile's
// global loads are issued before this tile's MFMAs and land in registers behind them; one barrier per tile), the weight
// panel is staged once, and the BatchNorm statistics leave as ONE set of atomics per workgroup.  The one-tile-per-workgroup
// form (load -> barrier -> 144 MFMAs -> store, five workgroups per CU to overlap) measured 266 us on decoder block 4 conv2.
template <int BN>
__global__ __launch_bounds__(256, 2) void conv_patch16_kernel(const ConvArgs a, int ntiles) {
  constexpr int NI = BN / 16;
  constexpr int MI = 4;                          // 4 waves x 4 pixel rows
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ws = smem;                        // [BN][148]
  float* const Pbuf = smem + BN * kWLd;          // [2][324][16]

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  const int tilesN = (a.Cout + BN - 1) / BN;
  const int tilesW = (a.Wo + kT - 1) / kT, tilesH = (a.Ho + kT - 1) / kT;
  const bool mirror = a.rmul < 0;                // dgrad: tap (r,s) reads patch (ty+2-r, tx+2-s)

  // ---- per-thread staging geometry (tile-invariant): 324 pixels x 4 units = 1296 units, 6 rounds
  const int chu = tid & 3;
  const bool has = a.s0.scale != nullptr;
  f4 lsc = {1.f, 1.f, 1.f, 1.f}, lsh = {0.f, 0.f, 0.f, 0.f};
  if (has) { lsc = *(const f4*)(a.s0.scale + chu * 4); lsh = *(const f4*)(a.s0.shift + chu * 4); }
  int spy[6], spx[6], spos[6];
#pragma unroll
  for (int rd = 0; rd < 6; ++rd) {
    const int pp = min((rd * 256 + tid) >> 2, kPPix - 1);
    spy[rd] = pp / kP; spx[rd] = pp - spy[rd] * kP;
    spos[rd] = pp * 16 + ((chu ^ ((pp >> 2) & 3)) << 2);
  }
  const bool last_live = (5 * 256 + tid) < kPPix * 4;
  f4 pv[6]; unsigned pok = 0;
  auto tile_origin = [&](int t, int& n, int& n0, int& h0, int& w0) {
    const int tn = t % tilesN; t /= tilesN;
    const int tw = t % tilesW; t /= tilesW;
    const int th = t % tilesH; n = t / tilesH;
    n0 = tn * BN; h0 = th * kT; w0 = tw * kT;
  };
  auto patch_load = [&](int t) {
    int n, n0, h0, w0; tile_origin(t, n, n0, h0, w0);
    pok = 0;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const int hl = h0 - 1 + spy[rd], wl = w0 - 1 + spx[rd];
      const bool ok = hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
      const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
      pv[rd] = *(const f4*)(a.s0.ptr + ((size_t)((size_t)n * a.s0.H + hc) * a.s0.W + wc) * a.s0.C + chu * 4);
      pok |= (ok ? 1u : 0u) << rd;
    }
  };
  auto patch_store = [&](int buf) {
    float* const pb_ = Pbuf + buf * kPPix * 16;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      f4 v = pv[rd];
      if (has) {
        v = v * lsc + lsh;
        if (a.s0.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      if (!((pok >> rd) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
      if (rd < 5 || last_live) *(f4*)(pb_ + spos[rd]) = v;
    }
  };

  // ---- stage the weight panel of the FIRST tile's channel block (tilesN == 1 in every model layer; else re-staged per tile)
  int t = blockIdx.x;
  int wn0 = -1;
  auto stage_w = [&](int n0) {
    for (int u = tid; u < BN * 36; u += 256) {
      const int row = u / 36, ku = u - row * 36;
      const int co = n0 + row;
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (co < a.wrows) v = *(const f4*)(a.w + (size_t)co * a.Kpad + ku * 4);
      *(f4*)(Ws + row * kWLd + ku * 4) = v;
    }
    wn0 = n0;
  };
  { int n, n0, h0, w0; tile_origin(t, n, n0, h0, w0); stage_w(n0); }
  patch_load(t);
  patch_store(0);
  __syncthreads();

  const bool do_stats = a.ssum != nullptr;
  f4 ps_[NI], pq_[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) { ps_[j] = (f4){0.f, 0.f, 0.f, 0.f}; pq_[j] = ps_[j]; }
  // fused BatchNorm-backward sums (uwm_kernels.h ConvArgs::bnb_*): second sum = v * yhat of the mask tensor
  const bool bnb = a.bnb_mean != nullptr;
  int stat_n0 = -1;                              // channel block the running sums belong to (flushed when it changes)

  auto flush_stats = [&](int n0) {               // 16 pixel lanes -> 4 waves (LDS, the buffer the NEXT tile does not use yet) -> fp64 atomics
    const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sv = ps_[j][e], qv = pq_[j][e];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { sv += __shfl_xor(sv, d); qv += __shfl_xor(qv, d); }
        ps_[j][e] = sv; pq_[j][e] = qv;
      }
    __syncthreads();
    float* red = Ws;                             // [4 waves][BN][2]  (re-staged afterwards if another tile follows)
    if (lrow == 0) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cl = j * 16 + lq * 4 + e;
          red[(wm * BN + cl) * 2 + 0] = ps_[j][e];
          red[(wm * BN + cl) * 2 + 1] = pq_[j][e];
        }
    }
    __syncthreads();
    if (tid < BN) {
      const int co = n0 + tid;
      if (co < a.Cout) {
        double sv = 0.0, qv = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { sv += (double)red[(w * BN + tid) * 2]; qv += (double)red[(w * BN + tid) * 2 + 1]; }
        atomicAdd(a.ssum + srep_off + co, sv);
        atomicAdd(a.ssq + srep_off + co, qv);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) { ps_[j] = (f4){0.f, 0.f, 0.f, 0.f}; pq_[j] = ps_[j]; }
  };

  for (int it = 0; t < ntiles; ++it, t += gridDim.x) {
    const int cur = it & 1;
    const int tnx = t + (int)gridDim.x;
    patch_load(tnx < ntiles ? tnx : t);          // (last tile: harmless re-read)
    int n, n0, h0, w0; tile_origin(t, n, n0, h0, w0);
    if (n0 != wn0) {                             // another channel block (BN < Cout: not a model shape): flush its sums, re-stage the panel
      if (do_stats && stat_n0 >= 0) flush_stats(stat_n0);
      __syncthreads();
      stage_w(n0);
      __syncthreads();
    }
    stat_n0 = n0;
    const float* const Ps = Pbuf + cur * kPPix * 16;

  f4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int r = tap / 3, s = tap - r * 3;
    const int rr = mirror ? 2 - r : r, ss = mirror ? 2 - s : s;
    f4 xf[MI], wf[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int pp = (wm * MI + i + rr) * kP + lrow + ss;
      xf[i] = *(const f4*)(Ps + pp * 16 + ((lq ^ ((pp >> 2) & 3)) << 2));
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) wf[j] = *(const f4*)(Ws + (j * 16 + lrow) * kWLd + tap * 16 + lq * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xf[i][e], acc[i][j], 0, 0, 0);
  }

  // ---------------- epilogue (same contract as conv_igemm_kernel) ----------------
  f4 bmu[NI], brs[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + j * 16 + lq * 4;
    bmu[j] = brs[j] = (f4){0.f, 0.f, 0.f, 0.f};
    if (bnb && co < a.Cout) { bmu[j] = *(const f4*)(a.bnb_mean + co); brs[j] = *(const f4*)(a.bnb_rstd + co); }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ho = h0 + wm * MI + i, wo = w0 + lrow;
    const bool pin = ho < a.Ho && wo < a.Wo;
    const size_t m = ((size_t)n * a.Ho + ho) * a.Wo + wo;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int co = n0 + j * 16 + lq * 4;
      if (pin && co < a.Cout) {
        f4 v = acc[i][j];
        const size_t o = m * a.Cout + co;
        if (a.bias) v += *(const f4*)(a.bias + co);
        if (a.addend) v += *(const f4*)(a.addend + o);
        f4 yr = {0.f, 0.f, 0.f, 0.f};
        if (a.mask) {
          f4 mk = *(const f4*)(a.mask + o);
          yr = mk;
          if (a.mscale) mk = mk * *(const f4*)(a.mscale + co) + *(const f4*)(a.mshift + co);
          v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
          v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        *(f4*)(a.out + o) = v;
        ps_[j] += v; pq_[j] += bnb ? v * ((yr - bmu[j]) * brs[j]) : v * v;
      }
    }
  }
    patch_store(cur ^ 1);
    __syncthreads();
  }
  if (do_stats && stat_n0 >= 0) flush_stats(stat_n0);
}

template <int BN>
static hipError_t launch_p16(const ConvArgs& a, hipStream_t st, int cls) {
  const int tilesN = (a.Cout + BN - 1) / BN;
  const int tilesW = (a.Wo + kT - 1) / kT, tilesH = (a.Ho + kT - 1) / kT;
  const size_t lds = (size_t)(2 * kPPix * 16 + BN * kWLd) * sizeof(float);
  const int ntiles = a.N * tilesH * tilesW * tilesN;
  const int nwg = ntiles < 2 * device_cu_count() ? ntiles : 2 * device_cu_count();      // 2 resident workgroups per CU (191-220 VGPRs)
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_patch16_kernel<BN>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(cls, a.flops, a.bytes, (conv_patch16_kernel<BN>), dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
  return hipGetLastError();
}

bool conv_patch16_applicable(const ConvArgs& a) {
  return a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && (a.rmul == 1 ? a.off == -1 : a.off == 1) &&
         a.Ctot == 16 && a.C0 == 16 && a.s0.C == 16 && a.s0.up == 0 && a.Kpad >= 144 &&
         a.Hl == a.Ho && a.Wl == a.Wo && a.Ho >= kT && a.Wo >= kT;
}

hipError_t launch_conv_patch16(const ConvArgs& a, hipStream_t st) {
  if (!conv_patch16_applicable(a) || (a.Cout & 3)) return hipErrorInvalidValue;
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !a.mask)) return hipErrorInvalidValue;
  return a.Cout > 16 ? launch_p16<32>(a, st, 18) : launch_p16<16>(a, st, 17);
}

}  // namespace uwm
