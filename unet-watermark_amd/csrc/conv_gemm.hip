// 1x1 / stride-1 convolution (forward and dgrad) as a persistent, LDS-DMA-fed GEMM for gfx950 (v_mfma_f32_16x16x4_f32):
// the Bottleneck 1x1s of resnet50 and the MBConv expand / project convs of EfficientNet (input channels a multiple of 4: a
// partial last K chunk reads clamped, duplicate units of X against the zero padding of W's rows).
//
//   D[pixel][co] = sum_c X[pixel][c] * W[co][c]        X: NHWC rows = pixels (no gather: a 1x1 / stride-1 conv IS a GEMM)
//
// Why a second kernel beside conv_igemm.hip: the short-K layers (64..256 input channels at 128^2..64^2) are two to eight
// K-steps long, so a one-tile-per-workgroup kernel is all prologue and epilogue (2.0-2.7 TB/s measured, profiles/r02_n_time_1x1_r50.txt).
// Here a workgroup walks a contiguous range of (channel tile, pixel tile) pairs and the (tile, K-chunk) sequence is ONE software
// pipeline: chunk it+1 — the next tile's first chunk at a tile boundary — is in flight (global_load_lds_dwordx4, no registers,
// no ds_write) while chunk it feeds the MFMAs and the finished tile's epilogue drains.  X and W chunks land in the exact
// 32-float-row, (row>>1)&7-swizzled LDS image conv_igemm uses: the swizzle is applied to the SOURCE address of each lane.
// The producer's lazy BatchNorm + ReLU cannot ride an LDS-DMA, so it is applied to the B fragments after ds_read (per-channel
// scale / shift prefetched one chunk ahead).  Epilogue contract = conv_igemm_kernel's (bias, addend, ReLU mask, BatchNorm
// statistics or fused BatchNorm-backward sums with bnb_y); statistics are accumulated across the tiles of one channel tile and
// leave as one set of atomics per (workgroup, channel tile).
//
// F16 = true (ConvArgs::ig16: the fp16x3 precision modes, layers whose channel counts are multiples of 32): the SAME pipeline and LDS
// image — raw fp32 chunks by LDS-DMA — and the operands are split into fp16 hi / lo halves in registers after ds_read (the place the
// lazy BatchNorm + ReLU already sits): a 32-channel chunk is ONE k-step of v_mfma_f32_16x16x32_f16, three products per tile (24 MFMAs
// of 16 cycles per wave and chunk where the fp32 form issues 64 of 32 cycles); weights times 2^12, a dgrad's dY by the power of two
// its maximum calls for (ConvArgs::xmax), both leave in the epilogue.
//
// Reference semantics replaced: the 1x1 nn.Conv2d layers of torchvision's Bottleneck / efficientnet_pytorch's MBConvBlock as smp
// wraps them (/root/reference/src/models/unet_model.py:64-71 -> smp encoders).
#include "uwm_kernels.h"
#include <cstdlib>

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
__device__ __forceinline__ void gemm_glds16(const float* g, float* l) {      // async 16 B/lane global -> LDS (wave-uniform l + lane*16)
  __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(uintptr_t)l, 16, 0, 0);
}

constexpr int kGM = 128;                         // pixels per tile
constexpr float kGWScale = 4096.f;
typedef _Float16 g_h8 __attribute__((ext_vector_type(8)));
typedef unsigned g_u4 __attribute__((ext_vector_type(4)));
// 8 consecutive channels (two f4) -> the hi and lo fp16 fragments of one v_mfma_f32_16x16x32_f16 operand lane
__device__ __forceinline__ void gemm_split8(f4 a0, f4 a1, g_h8& hi, g_h8& lo) {
  uwm_u2 h0, l0, h1, l1;
  uwm_split4(__builtin_amdgcn_fmed3f(a0.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(a0.y, -65504.f, 65504.f),
             __builtin_amdgcn_fmed3f(a0.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(a0.w, -65504.f, 65504.f), h0, l0);
  uwm_split4(__builtin_amdgcn_fmed3f(a1.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(a1.y, -65504.f, 65504.f),
             __builtin_amdgcn_fmed3f(a1.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(a1.w, -65504.f, 65504.f), h1, l1);
  hi = __builtin_bit_cast(g_h8, (g_u4){h0.x, h0.y, h1.x, h1.y});
  lo = __builtin_bit_cast(g_h8, (g_u4){l0.x, l0.y, l1.x, l1.y});
}

template <int BN, bool F16 = false>
__global__ __launch_bounds__(256, (BN == 64 ? 3 : 2)) void conv_gemm_kernel(const ConvArgs a, int tilesM, int ntiles) {
  constexpr int NI = BN / 32;                    // 16-channel MFMA tiles per wave (2 x 2 waves)
  constexpr int MI = 4;
  constexpr int XI = 4, WI = BN / 32;            // 1-KB LDS-DMA instructions per wave per chunk (X: 128 rows, W: BN rows)
  constexpr int kStage = (kGM + BN) * 32;        // floats per stage
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 15, lq = lane >> 4;
  const int nk = (a.Ctot + 31) >> 5;              // a partial last chunk reads clamped (valid, duplicate) units against zero weight columns
  const int xunits = a.Ctot >> 2;
  const int ldx = a.s0.C;

  // contiguous, balanced range of tiles for this workgroup; tile t = tn * tilesM + tm (pixel tiles fastest: the channel tile's
  // weights stay hot and the statistics of one channel tile leave once)
  const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
  const int per = ntiles / nwg, rem = ntiles - per * nwg;
  const int t0 = bid * per + (bid < rem ? bid : rem), t1 = t0 + per + (bid < rem ? 1 : 0);
  if (t0 >= t1) return;

  // per-lane LDS-DMA geometry: instruction i of this wave covers 16-byte units [(i*4 + wave)*64, +64) of the [rows][8] chunk image
  const int urow = lane >> 3, upos = lane & 7;
  auto issue = [&](int t, int kc, int buf) {
    const int tn = t / tilesM, tm = t - tn * tilesM;
    const int m0 = tm * kGM, n0 = tn * BN;
    float* const xs = smem + buf * kStage;
    float* const ws = xs + kGM * 32;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int row = (i * 4 + wave) * 8 + urow;
      const int u = upos ^ ((row >> 1) & 7);
      const int m = min(m0 + row, a.M - 1);                      // rows past M: a valid duplicate, discarded by the epilogue
      gemm_glds16(a.s0.ptr + (size_t)m * ldx + min(kc * 8 + u, xunits - 1) * 4, xs + (i * 4 + wave) * 256);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int row = (i * 4 + wave) * 8 + urow;
      const int u = upos ^ ((row >> 1) & 7);
      const int co = min(n0 + row, a.wrows - 1);                 // rows past wrows: duplicate, zeroed by the epilogue
      gemm_glds16(a.w + (size_t)co * a.Kpad + kc * 32 + u * 4, ws + (i * 4 + wave) * 256);
    }
  };

  const bool lazy = a.s0.scale != nullptr;
  const int relu = a.s0.relu;
  f4 lsc[2][2], lsh[2][2];                        // [chunk parity][k16]: producer's scale / shift of this lane's 4 channels
  auto lazy_load = [&](int kc, int par) {
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) {
      // fp32 form: this lane's k of MFMA k16 = channels (k16*4 + lq)*4 ..; F16: the lane's 8 channels lq*8 .. as two quads
      const int k = min(kc * 32 + (F16 ? (lq * 2 + k16) : (k16 * 4 + lq)) * 4, a.Ctot - 4);
      lsc[par][k16] = *(const f4*)(a.s0.scale + k); lsh[par][k16] = *(const f4*)(a.s0.shift + k);
    }
  };
  float dys = 1.f;                                // F16: power-of-two scale of a dgrad's dY
  if (F16 && a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); dys = ldexpf(1.f, 14 - e); }
  }
  const float unscale = F16 ? 1.f / (kGWScale * dys) : 1.f;

  f4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  // statistics carried across the tiles of one channel tile
  const bool do_stats = a.ssum != nullptr;
  const bool bnb = a.bnb_mean != nullptr;
  f4 ps[NI], pq[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) { ps[j] = (f4){0.f, 0.f, 0.f, 0.f}; pq[j] = ps[j]; }
  int stat_tn = -1;
  float* const red = smem + 2 * kStage;           // [2 wm][BN][2] behind the stages
  auto flush_stats = [&](int tn) {
    const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
    const int n0 = tn * BN;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = ps[j][e], q = pq[j][e];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
        ps[j][e] = s; pq[j][e] = q;
      }
    if (lrow == 0) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cl = (wn * NI + j) * 16 + lq * 4 + e;
          red[(wm * BN + cl) * 2 + 0] = ps[j][e];
          red[(wm * BN + cl) * 2 + 1] = pq[j][e];
        }
    }
    __syncthreads();
    if (tid < BN) {
      const int co = n0 + tid;
      if (co < a.Cout) {
        const double s = (double)red[tid * 2] + (double)red[(BN + tid) * 2];
        const double q = (double)red[tid * 2 + 1] + (double)red[(BN + tid) * 2 + 1];
        atomicAdd(a.ssum + srep_off + co, s);
        atomicAdd(a.ssq + srep_off + co, q);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j) { ps[j] = (f4){0.f, 0.f, 0.f, 0.f}; pq[j] = ps[j]; }
  };

  issue(t0, 0, 0);
  if (lazy) lazy_load(0, 0);
  __syncthreads();                                // (vmcnt(0): the first chunk has landed)

  int t = t0, kc = 0;
  for (int it = 0; t < t1; ++it) {
    const int cur = it & 1;
    // next chunk of the flattened (tile, chunk) sequence
    int tnx = t, kcn = kc + 1;
    if (kcn == nk) { kcn = 0; tnx = t + 1; }
    if (tnx < t1) {
      issue(tnx, kcn, cur ^ 1);
      if (lazy) lazy_load(kcn, cur ^ 1);
    }
    const float* const xs = smem + cur * kStage;
    const float* const ws = xs + kGM * 32;
    if (F16) {
      g_h8 xh[MI], xl[MI], wh[NI], wl[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = (wm * MI + i) * 16 + lrow;
        const int sw = (row >> 1) & 7;
        f4 a0 = *(const f4*)(xs + row * 32 + (((2 * lq) ^ sw) << 2)), a1 = *(const f4*)(xs + row * 32 + (((2 * lq + 1) ^ sw) << 2));
        if (lazy) {
          a0 = a0 * lsc[cur][0] + lsh[cur][0]; a1 = a1 * lsc[cur][1] + lsh[cur][1];
          if (relu) {
            a0.x = fmaxf(a0.x, 0.f); a0.y = fmaxf(a0.y, 0.f); a0.z = fmaxf(a0.z, 0.f); a0.w = fmaxf(a0.w, 0.f);
            a1.x = fmaxf(a1.x, 0.f); a1.y = fmaxf(a1.y, 0.f); a1.z = fmaxf(a1.z, 0.f); a1.w = fmaxf(a1.w, 0.f);
          }
        } else { a0 = a0 * dys; a1 = a1 * dys; }
        gemm_split8(a0, a1, xh[i], xl[i]);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = (wn * NI + j) * 16 + lrow;
        const int sw = (row >> 1) & 7;
        const f4 a0 = *(const f4*)(ws + row * 32 + (((2 * lq) ^ sw) << 2)) * kGWScale, a1 = *(const f4*)(ws + row * 32 + (((2 * lq + 1) ^ sw) << 2)) * kGWScale;
        gemm_split8(a0, a1, wh[j], wl[j]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
    } else
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) {
      f4 xf[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = (wm * MI + i) * 16 + lrow;
        xf[i] = *(const f4*)(xs + row * 32 + (((k16 * 4 + lq) ^ ((row >> 1) & 7)) << 2));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = (wn * NI + j) * 16 + lrow;
        wf[j] = *(const f4*)(ws + row * 32 + (((k16 * 4 + lq) ^ ((row >> 1) & 7)) << 2));
      }
      if (lazy) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          xf[i] = xf[i] * lsc[cur][k16] + lsh[cur][k16];
          if (relu) { xf[i].x = fmaxf(xf[i].x, 0.f); xf[i].y = fmaxf(xf[i].y, 0.f); xf[i].z = fmaxf(xf[i].z, 0.f); xf[i].w = fmaxf(xf[i].w, 0.f); }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xf[i][e], acc[i][j], 0, 0, 0);
    }

    if (kc == nk - 1) {
      // ---------------- epilogue of tile t: lane (p = lane&15 -> pixel, q = lane>>4 -> 4 channels) --------------
      const int tn = t / tilesM, tm = t - tn * tilesM;
      const int m0 = tm * kGM, n0 = tn * BN;
      if (do_stats && stat_tn >= 0 && stat_tn != tn) flush_stats(stat_tn);
      stat_tn = tn;
      f4 bmu[NI], brs[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int co = n0 + (wn * NI + j) * 16 + lq * 4;
        bmu[j] = brs[j] = (f4){0.f, 0.f, 0.f, 0.f};
        if (bnb && co < a.Cout) { bmu[j] = *(const f4*)(a.bnb_mean + co); brs[j] = *(const f4*)(a.bnb_rstd + co); }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + (wm * MI + i) * 16 + lrow;
        const bool mv = m < a.M;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int co = n0 + (wn * NI + j) * 16 + lq * 4;
          if (mv && co < a.Cout) {
            f4 v = acc[i][j] * unscale;
            if (co + 3 >= a.wrows) {                               // padded output channels: rows past wrows were duplicates
              if (co + 0 >= a.wrows) v.x = 0.f;
              if (co + 1 >= a.wrows) v.y = 0.f;
              if (co + 2 >= a.wrows) v.z = 0.f;
              if (co + 3 >= a.wrows) v.w = 0.f;
            }
            const size_t o = (size_t)m * a.Cout + co;
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
            if (a.bnb_y) yr = *(const f4*)(a.bnb_y + o);
            ps[j] += v; pq[j] += bnb ? v * ((yr - bmu[j]) * brs[j]) : v * v;
          }
          acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    t = tnx; kc = kcn;
    __syncthreads();                              // chunk it+1 (LDS-DMA) has landed; everyone is done with `cur`
  }
  if (do_stats && stat_tn >= 0) flush_stats(stat_tn);
}

// Where the implicit GEMM stays ahead (profiles/r02_*_time_1x1*.txt, both kernels per shape): 32 output channels (half of a
// 64-channel tile wasted: 69 vs 52 us on 192 -> 32 at 256^2), and launches that fill fewer than two 128x64 tiles per CU unless K is
// long enough for the pipeline to pay anyway (2048 -> 512 at 16^2: 87 vs 106 us; 960 -> 160 at 64^2: 76 vs 72)
bool conv_gemm_preferred(const ConvArgs& a) {
  if (!conv_gemm_applicable(a) || a.Cout < 48) return false;
  const long tiles64 = (long)((route_M(a) + kGM - 1) / kGM) * ((a.Cout + 63) / 64);
  const long cus = device_cu_count();
  return tiles64 >= 2 * cus || (a.Ctot >= 1024 && tiles64 >= cus);
}

bool conv_gemm_applicable(const ConvArgs& a) {
  static const bool off = dbg_flag("UWM_NO_CONV_GEMM");
  return !off && a.ntaps == 1 && a.kw == 1 && a.smul == 1 && a.sdiv == 1 && a.off == 0 && a.s0.up == 0 && a.C0 == a.Ctot && a.s0.C == a.Ctot &&
         (a.Ctot & 3) == 0 && a.Ctot >= 16 && a.Kpad >= ((a.Ctot + 31) & ~31) && a.Hl == a.Ho && a.Wl == a.Wo && a.s0.H == a.Ho && a.s0.W == a.Wo && !a.out_up &&
         a.M >= kGM && a.Cout >= 32 && (a.Cout & 3) == 0 && a.wrows >= 1 && a.prec == 0;
}

template <int BN, bool F16>
static hipError_t launch_gemm_(const ConvArgs& a, hipStream_t st, int cls) {
  const int tilesM = (a.M + kGM - 1) / kGM, tilesN = (a.Cout + BN - 1) / BN;
  const int ntiles = tilesM * tilesN;
  const size_t lds = (size_t)(2 * (kGM + BN) * 32 + 2 * BN * 2) * sizeof(float);
  const int slots = (BN == 64 ? 3 : 2) * device_cu_count();      // resident workgroups: 148 VGPRs / 48 KB LDS at BN = 64, 215 / 64 KB at 128
  const int nwg = ntiles < slots ? ntiles : slots;
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_gemm_kernel<BN, F16>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(F16 ? (BN == 128 ? 51 : 52) : cls, a.flops, a.bytes, (conv_gemm_kernel<BN, F16>), dim3((unsigned)nwg), dim3(256), lds, st, a, tilesM, ntiles);
  return hipGetLastError();
}
template <int BN>
static hipError_t launch_gemm(const ConvArgs& a, hipStream_t st, int cls) {
  // fp16x3 form: whole 32-channel chunks (a partial last chunk reads duplicate units against zero weight columns in the fp32 form only)
  return (a.ig16 && (a.Ctot & 31) == 0) ? launch_gemm_<BN, true>(a, st, cls) : launch_gemm_<BN, false>(a, st, cls);
}

// bn: 0 auto | 128 | 64 output channels per tile
hipError_t launch_conv_gemm(const ConvArgs& a, hipStream_t st, int bn) {
  if (!conv_gemm_applicable(a)) return hipErrorInvalidValue;
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !(a.bnb_y ? a.bnb_y : a.mask))) return hipErrorInvalidValue;
  if (bn <= 0) {
    // 64-channel tiles run three workgroups per CU (148 VGPRs, 48 KB): better latency hiding for the short-K, HBM-bound layers
    // and for launches with few tiles; 128-channel tiles read X half as often: the long-K layers with plenty of tiles
    // (profiles/r02_*_time_1x1_r50.txt: cfg 864 vs 928 per shape)
    const int pad128 = ((a.Cout + 127) / 128) * 128, pad64 = ((a.Cout + 63) / 64) * 64;
    const long tilesM = (route_M(a) + kGM - 1) / kGM;
    bn = 128;
    if (a.Cout <= 64 || pad128 * 100 > pad64 * 115 || a.Ctot <= 128 || tilesM * (pad128 / 128) < 4L * device_cu_count()) bn = 64;
  }
  return bn == 64 ? launch_gemm<64>(a, st, 38) : launch_gemm<128>(a, st, 37);
}

}  // namespace uwm
