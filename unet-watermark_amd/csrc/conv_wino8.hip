// Winograd F(2x2,3x3) convolution, 8-wave / 16x16-pixel / 64-channel variant of conv_wino.hip for the
// layers with >= 64 output channels and >= 256 such tiles (forward AND dgrad; same contracts).
//
// profiles/r01 ablations of conv_wino_kernel<4> put the largest single cost (41 of ~100 us that are not MFMA
// on a 280 us launch) on streaming the transformed weights L2 -> LDS: 32 KB per 8-channel chunk per
// 32 tiles.  Here ONE 512-thread workgroup per CU covers 64 tiles (16x16 pixels) with the same 32-KB weight
// chunk — half the weight bytes per MFMA — and, having no second workgroup to hide behind, pipelines itself:
//   chunk k:  LDS-DMA U(k+1) ; global-load patch(k+2) -> registers
//             ds_read patch(k+1) for the NEXT chunk's input transform   | interleaved by the scheduler
//             64 MFMAs on V(k) (registers) x U(k) (LDS fragments)       | with the reads' latency
//             finish V(k+1) = B^T d B ; store patch(k+2) ; ONE barrier
// Wave w: Winograd-domain row (w & 3), tile half (w >> 2): tiles of pixel rows 8*(w>>2) .. +7.
// Patch LDS image: four 2-channel planes [324 px][2] (+ pairwise pixel swizzle keyed on the tile-row
// parity) so that both ds_read_b64 and the ds_read2_b64 pairs hipcc fuses them into are conflict-free.
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kT8 = 16, kP8 = kT8 + 2, kPP8 = kP8 * kP8;      // 16x16 output pixels, 18x18 = 324 patch pixels
constexpr int kPl8 = 672;                                      // floats per 2-channel plane (648 + pad: == 32 mod 64)
constexpr int kPb8 = 4 * kPl8;                                 // floats per patch buffer
constexpr int kUs8 = 16 * 4 * 128;                             // floats per U buffer (16 xi x 64 co x 8 c)
constexpr int kQLD8 = 68;

__global__ __launch_bounds__(512, 1) void conv_wino8_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Us = smem;                         // [2][16][4][4][16][2]
  float* const Ps = smem + 2 * kUs8;              // [2][4 planes][324 px][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave & 3, half = wave >> 2;
  const int t16 = lane & 15, lq = lane >> 4;

  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tilesN = (a.Cout + 63) / 64;
  const int tilesW = (a.Wo + kT8 - 1) / kT8, tilesH = (a.Ho + kT8 - 1) / kT8;
  const int tn = tile % tilesN; tile /= tilesN;
  const int tw = tile % tilesW; tile /= tilesW;
  const int th = tile % tilesH; const int n = tile / tilesH;
  const int n0 = tn * 64, h0 = th * kT8, w0 = tw * kT8;
  const int nCb = a.wu_ncb;

  f4 acc[4][2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[j][tb][cb] = (f4){0.f, 0.f, 0.f, 0.f};

  // ---- patch staging through registers: 648 16-byte units = 2 rounds of 512 threads
  f4 pv[2], psc, psh; int prelu = 0; bool phas = false;
  int ppos[2]; bool pok[2], pact[2]; int poff0[2], poff1[2];
  {
    const int chu = tid & 1;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int u = rd * 512 + tid;
      pact[rd] = u < kPP8 * 2;
      const int pp = pact[rd] ? (u >> 1) : 0;
      const int py = pp / kP8, px = pp - py * kP8;
      const int hl = h0 - 1 + py, wl = w0 - 1 + px;
      pok[rd] = pact[rd] && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
      const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
      poff0[rd] = (int)(((size_t)n * a.s0.H + (hc >> a.s0.up)) * a.s0.W + (wc >> a.s0.up)) * a.s0.C + chu * 4;
      poff1[rd] = (int)(((size_t)n * a.s1.H + (hc >> a.s1.up)) * a.s1.W + (wc >> a.s1.up)) * a.s1.C + chu * 4;
      ppos[rd] = (2 * chu) * kPl8 + ((pp ^ ((py >> 1) & 1)) << 1);       // plane 2*chu (channels 0,1 of the unit); +kPl8 for 2,3
    }
  }
  auto patch_load = [&](int cc) {
    const int c = cc * 8;
    const bool first = c < a.C0;
    const float* sp = first ? a.s0.ptr : a.s1.ptr;
    const float* ssc = first ? a.s0.scale : a.s1.scale;
    const float* ssh = first ? a.s0.shift : a.s1.shift;
    prelu = first ? a.s0.relu : a.s1.relu;
    const int cl = (first ? c : c - a.C0);
    phas = ssc != nullptr;
    if (phas) { psc = *(const f4*)(ssc + cl + (tid & 1) * 4); psh = *(const f4*)(ssh + cl + (tid & 1) * 4); }
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) pv[rd] = *(const f4*)(sp + (first ? poff0[rd] : poff1[rd]) + cl);
  };
  auto patch_store = [&](int buf) {
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      f4 v = pv[rd];
      if (phas) {
        v = v * psc + psh;
        if (prelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      if (!pok[rd]) v = (f4){0.f, 0.f, 0.f, 0.f};
      if (pact[rd]) {
        float* d = Ps + buf * kPb8 + ppos[rd];
        *(f2*)d = (f2){v.x, v.y};
        *(f2*)(d + kPl8) = (f2){v.z, v.w};
      }
    }
  };
  // ---- U chunk: 32 LDS-DMA wave-instructions of 1 KB, 4 per wave
  const float* const ug = a.wu + (size_t)(n0 / 16) * 128;
  int uoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int L = (i * 8 + wave) * 64 + lane;       // f4 unit in the [16][128] chunk image
    const int xi = L >> 7, within = L & 127;
    uoff[i] = xi * nCb * 128 + within * 4;
  }
  auto u_dma = [&](int cc, int buf) {
    const float* const uc = ug + (size_t)cc * 16 * nCb * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(uc + uoff[i]), (lds_void*)(uintptr_t)(Us + buf * kUs8 + (i * 8 + wave) * 256), 16, 0, 0);
  };

  // ---- this wave's B^T row pair: r = d[ra] + sg * d[rb]
  const int ra = (wrow == 0) ? 0 : (wrow == 2 ? 2 : 1);
  const int rb = (wrow == 3) ? 3 : (wrow == 2 ? 1 : 2);
  const float sg = (wrow == 1) ? 1.f : -1.f;
  int adA[2][2], adB[2][2];                         // [tb][row sel]: columns {0,2} / {1,3}
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      const int ty = half * 4 + tb * 2 + (t16 >> 3), tx = t16 & 7;
      const int prow = 2 * ty + (rs ? rb : ra);
      const int f = (prow >> 1) & 1;
      const int base = lq * kPl8 + ((prow * kP8 + 2 * tx) << 1);
      adA[tb][rs] = base + (f << 1);
      adB[tb][rs] = base + ((f ^ 1) << 1);
    }
  const int ufrag = wrow * 4 * 4 * 128 + lq * 32 + t16 * 2;

  f2 ra_[2][8];                                     // raw patch values of the NEXT chunk: [tb][row sel * 4 + col]
  auto v_read = [&](int buf) {
    const float* const pc = Ps + buf * kPb8;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int rs = 0; rs < 2; ++rs) {
        ra_[tb][rs * 4 + 0] = *(const f2*)(pc + adA[tb][rs]);
        ra_[tb][rs * 4 + 1] = *(const f2*)(pc + adB[tb][rs]);
        ra_[tb][rs * 4 + 2] = *(const f2*)(pc + adA[tb][rs] + 4);
        ra_[tb][rs * 4 + 3] = *(const f2*)(pc + adB[tb][rs] + 4);
      }
  };
  f2 V[2][4];
  auto v_finish = [&]() {
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const f2 r0 = ra_[tb][0] + sg * ra_[tb][4], r1 = ra_[tb][1] + sg * ra_[tb][5];
      const f2 r2 = ra_[tb][2] + sg * ra_[tb][6], r3 = ra_[tb][3] + sg * ra_[tb][7];
      V[tb][0] = r0 - r2; V[tb][1] = r1 + r2; V[tb][2] = r2 - r1; V[tb][3] = r1 - r3;
    }
  };

  const int nchunk = a.Ctot >> 3;
  // prologue: U(0), patch(0) -> buffer 0, patch(1) -> buffer 1, V(0)
  u_dma(0, 0);
  patch_load(0);
  patch_store(0);
  patch_load(nchunk > 1 ? 1 : 0);
  patch_store(1);
  __syncthreads();
  v_read(0);
  v_finish();
  __syncthreads();                                  // nobody may overwrite buffer 0 (patch(2), chunk 0) before every wave has read patch(0)

  for (int cc = 0; cc < nchunk; ++cc) {
    const int cur = cc & 1, nxt = cur ^ 1;
    const int c1 = cc + 1 < nchunk ? cc + 1 : cc, c2 = cc + 2 < nchunk ? cc + 2 : cc;   // tail: harmless re-fetches
    u_dma(c1, nxt);
    patch_load(c2);
    v_read(nxt);                                    // patch(cc+1) sits in buffer (cc+1)&1
    __builtin_amdgcn_sched_barrier(0);              // prefetches stay above the MFMA block
    const float* const uc = Us + cur * kUs8 + ufrag;
    f2 wf[2][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) wf[0][cb] = *(const f2*)(uc + cb * 128);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < 3) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) wf[(j + 1) & 1][cb] = *(const f2*)(uc + ((j + 1) * 4 + cb) * 128);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
            acc[j][tb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j & 1][cb][e], V[tb][j][e], acc[j][tb][cb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    v_finish();                                     // V(cc+1)
    patch_store(cur);                               // patch(cc+2) -> the buffer patch(cc) lived in (consumed one chunk ago)
    __syncthreads();
  }

  // ---------------- epilogue, one output column parity (b) per pass: q_b -> LDS -> Y[.][b] ----------------
  float* const Q = smem;                 // [4 rows][64 tiles][kQLD8]
  const bool do_stats = a.ssum != nullptr;
  // BatchNorm statistics go to one of a.srep copies (few-channel layers launch tens of thousands of workgroups:
  // fp64 atomics on the same 2*C addresses serialise), bn_finalize adds the copies
  const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = {0.f, 0.f, 0.f, 0.f};
  const int cq = tid & 15;
  const int co = n0 + cq * 4;
#pragma unroll
  for (int xb = 0; xb < 2; ++xb) {
    if (xb) __syncthreads();             // pass 0's readers are done
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const f4 q = xb == 0 ? acc[0][tb][cb] + acc[1][tb][cb] + acc[2][tb][cb] : acc[1][tb][cb] - acc[2][tb][cb] - acc[3][tb][cb];
        const int t = half * 32 + tb * 16 + t16;
        *(f4*)(Q + (wrow * 64 + t) * kQLD8 + cb * 16 + lq * 4) = q;
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int t = (it * 512 + tid) >> 4;
      f4 q[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) q[w] = *(const f4*)(Q + (w * 64 + t) * kQLD8 + cq * 4);
      const int ty = t >> 3, tx = t & 7;
#pragma unroll
      for (int ya = 0; ya < 2; ++ya) {
        f4 v = ya == 0 ? q[0] + q[1] + q[2] : q[1] - q[2] - q[3];
        const int ho = h0 + 2 * ty + ya, wo = w0 + 2 * tx + xb;
        if (ho < a.Ho && wo < a.Wo && co < a.Cout) {
          const size_t o = (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co;
          if (a.bias) v += *(const f4*)(a.bias + co);
          if (a.addend) v += *(const f4*)(a.addend + o);
          if (a.mask) {
            f4 mk = *(const f4*)(a.mask + o);
            if (a.mscale) mk = mk * *(const f4*)(a.mscale + co) + *(const f4*)(a.mshift + co);
            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
            v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
          }
          *(f4*)(a.out + o) = v;
          ps_ += v; pq_ += v * v;
        }
      }
    }
  }
  if (do_stats) {
    __syncthreads();                       // Q is dead
    float* red = smem;                     // [32 groups][64][2]
    const int grp = tid >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[((grp * 64) + cq * 4 + e) * 2 + 0] = ps_[e];
      red[((grp * 64) + cq * 4 + e) * 2 + 1] = pq_[e];
    }
    __syncthreads();
    if (tid < 64) {
      const int c1 = n0 + tid;
      if (c1 < a.Cout) {
        double sv = 0.0, qv = 0.0;
        for (int g = 0; g < 32; ++g) { sv += (double)red[(g * 64 + tid) * 2]; qv += (double)red[(g * 64 + tid) * 2 + 1]; }
        atomicAdd(a.ssum + srep_off + c1, sv);
        atomicAdd(a.ssq + srep_off + c1, qv);
      }
    }
  }
}

bool conv_wino8_applicable(const ConvArgs& a) {
  if (!conv_wino_applicable(a) || a.Cout < 64 || a.Ho < kT8 || a.Wo < kT8) return false;
  // Measured (r01): 3-5 % faster than conv_wino_kernel<4> launch for launch, but its 87-KB / 512-thread workgroups
  // co-reside worse with the weight-gradient kernels of the side stream: with EVERY eligible launch on it the
  // overlapped train step is 2 % slower (580 vs 590 img/s).  So: forward convolutions only (nothing runs beside them),
  // dgrads stay on the 4-wave kernel; Winograd mode 2 / force_cfg 308 select it wherever the shape allows.
  if (wino_mode_of(a.wino) == 2) return true;
  const long wgs = (long)route_N(a) * ((a.Ho + kT8 - 1) / kT8) * ((a.Wo + kT8 - 1) / kT8) * ((a.Cout + 63) / 64);
  return a.rmul == 1 && wgs >= 256;      // one 512-thread workgroup per CU: fewer would leave CUs idle
}

hipError_t launch_conv_wino8(const ConvArgs& a, hipStream_t st) {
  if (!conv_wino_applicable(a) || a.Cout < 64 || a.Ho < kT8 || a.Wo < kT8) return hipErrorInvalidValue;
  if (a.bnb_mean) return hipErrorInvalidValue;      // no fused BatchNorm-backward sums in this epilogue (conv_wino_kernel<NI> carries them)
  const int tilesN = (a.Cout + 63) / 64;
  const int tilesW = (a.Wo + kT8 - 1) / kT8, tilesH = (a.Ho + kT8 - 1) / kT8;
  const size_t main_lds = (size_t)(2 * kUs8 + 2 * kPb8) * sizeof(float);
  const size_t q_lds = (size_t)4 * 64 * kQLD8 * sizeof(float);
  const size_t lds = main_lds > q_lds ? main_lds : q_lds;
  static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_wino8_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(25, a.flops, a.bytes, conv_wino8_kernel, dim3((unsigned)(a.N * tilesH * tilesW * tilesN)), dim3(512), lds, st, a);
  return hipGetLastError();
}

}  // namespace uwm
