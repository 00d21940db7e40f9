// Winograd F(2x2,3x3) convolution (forward AND dgrad) in "bf16x3" arithmetic for gfx950 — the opt-in second precision
// mode of the library (uwm_set_precision(h, UWM_PREC_BF16X3); the reference's own GPU path is reduced precision:
// fp16 autocast + GradScaler, /root/reference/src/train.py:75,89-98).
//
// Every fp32 operand x of the 16 Winograd-domain GEMMs is split x = hi + lo, hi = bf16(x), lo = bf16(x - hi), and each
// product a*b is taken as   a_hi*b_hi + a_hi*b_lo + a_lo*b_hi   on v_mfma_f32_16x16x16_bf16 with fp32 accumulation:
// ~16 mantissa bits per operand (the dropped lo*lo term is 2^-16 relative) at 3 bf16 MFMAs per product instead of one
// fp32 MFMA at 1/16 the rate.  The filter bank U = G g G^T is split once per step by the transform kernel; the input
// transform V = B^T d B is formed in fp32 in registers and split there.  Everything around the products — lazy
// BatchNorm + ReLU on load, transforms, accumulation, epilogue, BatchNorm statistics — is fp32 exactly as in
// conv_wino.hip, whose work split this kernel keeps:
//   workgroup = 8x16 output pixels (32 tiles) x 32 output channels, 256 threads; wave i = row i of the 4x4 domain;
//   per 16-channel chunk: the 32-KB slice of U (16 xi x 32 rows x 16 channels x {hi,lo}) arrives by LDS-DMA (double
//   buffered, straight copy of the global image, one conflict-free ds_read_b128 per MFMA A fragment pair), the 10x18
//   halo patch goes through registers into ONE LDS buffer (lazy transform applied), each lane builds V[i][0..3] of its
//   tile for ITS 4 channels (MFMA k group = lane >> 4) from 16 ds_read_b128, splits it and issues 48 MFMAs.
//   Two barriers per chunk (patch consumed / next chunk landed); 75 KB of LDS: two workgroups per CU.
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kTH = 8, kTW = 16, kPH = kTH + 2, kPW = kTW + 2, kPP = kPH * kPW;   // 180 patch pixels
constexpr int kPlane = kPP * 4;                                                  // floats per 4-channel plane
constexpr int kNI = 2, kBN = 32;                                                 // output-channel MFMA tiles / channels per workgroup
constexpr int kUs = 16 * kNI * 256;                                              // floats per U buffer (16 xi x 2 blocks x 1 KB)
constexpr int kQPad = 4;

// x = hi + lo in bf16 (round to nearest even both times): two fp32 values -> packed hi pair, packed lo pair
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const b2 h = {(__bf16)x0, (__bf16)x1};
  hi = __builtin_bit_cast(unsigned, h);
  const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
  const b2 l = {(__bf16)(x0 - h0), (__bf16)(x1 - h1)};
  lo = __builtin_bit_cast(unsigned, l);
}

// ---------------------------------------------------------------- filter transform U = G g G^T, split into bf16 hi | lo
// Ux[Ctot/16][16 xi][nCb][64 lanes][hi c0..c3 | lo c0..c3]   lane = (channel group kq = (c & 15) >> 2) * 16 + (row & 15)
// = the MFMA A fragment of v_mfma_f32_16x16x16_bf16 (row = lane & 15, k = 4*(lane >> 4) + j): 16 bytes per lane, the LDS
// image is a straight copy.  Same float count as the fp32 bank (conv_wino.hip), so the two share their workspace slots.
// mode 0: U from [rows][Kpad] weights (k = tap*chans + c); mode 2: the dgrad bank straight from the FORWARD weights
// (rows = input channels, chans = output channels, g'[r][s] = w[c][(2-r)*3 + (2-s)][row]).
__global__ void wino_weights_x3_multi_kernel(const WinoJobs jobs) {
  const WinoJob jb = jobs.j[blockIdx.y];
  const int nCb = ((jb.rows + 63) / 64) * 4;
  const size_t total = (size_t)(jb.chans >> 4) * nCb * 64;        // one thread = one row x 4 channels
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int r16 = (int)(i & 15), kq = (int)((i >> 4) & 3);
  const int cb = (int)((i >> 6) % nCb), chunk = (int)((i >> 6) / nCb);
  const int row = cb * 16 + r16, c = chunk * 16 + kq * 4;
  f4 g[9];
  if ((jb.mode & 3) == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
      g[t] = row < jb.rows ? *(const f4*)(jb.w + (size_t)row * jb.Kpad + (size_t)t * jb.chans + c) : (f4){0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const size_t o = (size_t)(8 - t) * jb.rows + row;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[t][e] = (row < jb.rows && c + e < jb.src_rows) ? jb.w[(size_t)(c + e) * jb.Kpad + o] : 0.f;
    }
  }
  f4 t4[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t4[0][s] = g[s];
    t4[1][s] = 0.5f * (g[s] + g[3 + s] + g[6 + s]);
    t4[2][s] = 0.5f * (g[s] - g[3 + s] + g[6 + s]);
    t4[3][s] = g[6 + s];
  }
  float* const base = jb.ut + ((size_t)chunk * 16 * nCb + cb) * 256 + (kq * 16 + r16) * 4;
  const size_t xs = (size_t)nCb * 256;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f4 u[4] = {t4[r][0], 0.5f * (t4[r][0] + t4[r][1] + t4[r][2]), 0.5f * (t4[r][0] - t4[r][1] + t4[r][2]), t4[r][2]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned h01, l01, h23, l23;
      split2(u[j].x, u[j].y, h01, l01);
      split2(u[j].z, u[j].w, h23, l23);
      // (stored through the buffer's own float type: a u4 store / load through a float pointer is a strict-aliasing
      //  violation that hipcc exploits — the first version's fragment loads were folded to ONE dword)
      *(f4*)(base + (size_t)(r * 4 + j) * xs) = (f4){__builtin_bit_cast(float, h01), __builtin_bit_cast(float, h23),
                                                      __builtin_bit_cast(float, l01), __builtin_bit_cast(float, l23)};
    }
  }
}
hipError_t launch_wino_weights_x3_multi(const WinoJobs& jobs, hipStream_t st) {
  if (jobs.n <= 0) return hipSuccess;
  size_t mx = 0;
  for (int i = 0; i < jobs.n; ++i) {
    if (jobs.j[i].chans & 15) return hipErrorInvalidValue;
    const size_t t = (size_t)(jobs.j[i].chans >> 4) * (((jobs.j[i].rows + 63) / 64) * 4) * 64;
    if (t > mx) mx = t;
  }
  hipLaunchKernelGGL(wino_weights_x3_multi_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)jobs.n), dim3(256), 0, st, jobs);
  return hipGetLastError();
}

// ---------------------------------------------------------------- main kernel
__global__ __launch_bounds__(256, 2) void conv_wino_x3_kernel(const ConvArgs a) {
  constexpr int NI = kNI, BN = kBN;
  constexpr int UR = 16 * NI / 4;                 // 1-KB LDS-DMA wave-instructions of U per wave per chunk (8)
  constexpr int QLD = BN + kQPad;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Us = smem;                         // [2][16 xi][NI][64 lanes][4]   (LDS-DMA destination, lane-linear)
  float* const Ps = smem + 2 * kUs;               // [4 planes][180 px (pairwise swizzled)][4]   ONE buffer

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t16 = lane & 15, lq = lane >> 4;

  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tilesN = (a.Cout + BN - 1) / BN;
  const int tilesW = (a.Wo + kTW - 1) / kTW, tilesH = (a.Ho + kTH - 1) / kTH;
  const int tn = tile % tilesN; tile /= tilesN;
  const int tw = tile % tilesW; tile /= tilesW;
  const int th = tile % tilesH; const int n = tile / tilesH;
  const int n0 = tn * BN, h0 = th * kTH, w0 = tw * kTW;
  const int nCb = a.wu_ncb;

  f4 acc[4][2][NI];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int cb = 0; cb < NI; ++cb) acc[j][tb][cb] = (f4){0.f, 0.f, 0.f, 0.f};

  // ---- patch staging through registers: 180 px x 4 channel units = 720 16-byte units = 3 rounds; loads are unconditional
  // (clamped addresses, zero-selected afterwards)
  constexpr int PR = 3;
  f4 pv[PR], psc, psh; int prelu = 0; bool phas = false;
  int ppos[PR]; bool pok[PR], pact[PR]; int poff0[PR], poff1[PR];     // chunk-invariant per-thread geometry
  const int chu = tid & 3;
#pragma unroll
  for (int rd = 0; rd < PR; ++rd) {
    const int u = rd * 256 + tid;
    pact[rd] = u < kPP * 4;
    const int pp = pact[rd] ? (u >> 2) : 0;
    const int py = pp / kPW, px = pp - py * kPW;
    const int hl = h0 - 1 + py, wl = w0 - 1 + px;
    pok[rd] = pact[rd] && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
    const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
    poff0[rd] = (int)(((size_t)n * a.s0.H + (hc >> a.s0.up)) * a.s0.W + (wc >> a.s0.up)) * a.s0.C + chu * 4;
    poff1[rd] = (int)(((size_t)n * a.s1.H + (hc >> a.s1.up)) * a.s1.W + (wc >> a.s1.up)) * a.s1.C + chu * 4;
    ppos[rd] = chu * kPlane + ((pp ^ ((py >> 1) & 1)) << 2);
  }
  auto patch_load = [&](int cc) {
    const int c = cc * 16;                          // chunk base channel (chunks never straddle the two sources: C0 % 16 == 0)
    const bool first = c < a.C0;
    const float* sp = first ? a.s0.ptr : a.s1.ptr;
    const float* ssc = first ? a.s0.scale : a.s1.scale;
    const float* ssh = first ? a.s0.shift : a.s1.shift;
    prelu = first ? a.s0.relu : a.s1.relu;
    const int cl = (first ? c : c - a.C0);
    phas = ssc != nullptr;
    if (phas) { psc = *(const f4*)(ssc + cl + chu * 4); psh = *(const f4*)(ssh + cl + chu * 4); }
#pragma unroll
    for (int rd = 0; rd < PR; ++rd) pv[rd] = *(const f4*)(sp + (first ? poff0[rd] : poff1[rd]) + cl);
  };
  auto patch_store = [&]() {
#pragma unroll
    for (int rd = 0; rd < PR; ++rd) {
      f4 v = pv[rd];
      if (phas) {
        v = v * psc + psh;
        if (prelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      if (!pok[rd]) v = (f4){0.f, 0.f, 0.f, 0.f};
      if (pact[rd]) *(f4*)(Ps + ppos[rd]) = v;
    }
  };
  // ---- U chunk: UR LDS-DMA instructions per wave, 1 KB each = one (xi, row block) fragment set, straight copy
  const float* const ug = a.wu + (size_t)(n0 / 16) * 256;
  int uoff[UR];
#pragma unroll
  for (int i = 0; i < UR; ++i) {
    const int piece = i * 4 + wave;                 // (xi, cb) = (piece / NI, piece % NI)
    uoff[i] = ((piece / NI) * nCb + (piece % NI)) * 256 + lane * 4;
  }
  auto u_dma = [&](int cc, int buf) {
    const float* const uc = ug + (size_t)cc * 16 * nCb * 256;
#pragma unroll
    for (int i = 0; i < UR; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(uc + uoff[i]), (lds_void*)(uintptr_t)(Us + buf * kUs + (i * 4 + wave) * 256), 16, 0, 0);
  };

  // ---- per-lane addresses of the B^T row pair this wave combines: r = d[ra] + sg * d[rb]; plane = this lane's channel group
  const int ra = (wave == 0) ? 0 : (wave == 2 ? 2 : 1);
  const int rb = (wave == 3) ? 3 : (wave == 2 ? 1 : 2);
  const float sg = (wave == 1) ? 1.f : -1.f;
  int adA[2][2], adB[2][2];                         // [tb][row sel]: columns {0,2} / {1,3} (pairwise pixel swizzle)
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      const int ty = tb * 2 + (t16 >> 3), tx = t16 & 7;
      const int prow = 2 * ty + (rs ? rb : ra);
      const int f = (prow >> 1) & 1;
      const int base = lq * kPlane + ((prow * kPW + 2 * tx) << 2);
      adA[tb][rs] = base + (f << 2);
      adB[tb][rs] = base + ((f ^ 1) << 2);
    }
  const int ufrag = wave * 4 * NI * 256 + lane * 4;

  const int nchunk = a.Ctot >> 4;
  u_dma(0, 0);
  patch_load(0);
  patch_store();
  patch_load(nchunk > 1 ? 1 : 0);
  __syncthreads();

  for (int cc = 0; cc < nchunk; ++cc) {
    const int cur = cc & 1, nxt = cur ^ 1;
    // (B^T d B)[wave][0..3] of this lane's tile and 4 channels, both tile blocks, fp32 -> split to bf16 hi / lo pairs
    unsigned Vh[2][4][2], Vl[2][4][2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const f4 a0 = *(const f4*)(Ps + adA[tb][0]), a1 = *(const f4*)(Ps + adB[tb][0]);
      const f4 a2 = *(const f4*)(Ps + adA[tb][0] + 8), a3 = *(const f4*)(Ps + adB[tb][0] + 8);
      const f4 b0 = *(const f4*)(Ps + adA[tb][1]), b1 = *(const f4*)(Ps + adB[tb][1]);
      const f4 b2 = *(const f4*)(Ps + adA[tb][1] + 8), b3 = *(const f4*)(Ps + adB[tb][1] + 8);
      const f4 r0 = a0 + sg * b0, r1 = a1 + sg * b1, r2 = a2 + sg * b2, r3 = a3 + sg * b3;
      const f4 v[4] = {r0 - r2, r1 + r2, r2 - r1, r1 - r3};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        split2(v[j].x, v[j].y, Vh[tb][j][0], Vl[tb][j][0]);
        split2(v[j].z, v[j].w, Vh[tb][j][1], Vl[tb][j][1]);
      }
    }
    __syncthreads();                                // every wave has taken its patch values: the buffer may be overwritten
    const int cn = cc + 1 < nchunk ? cc + 1 : cc;   // last chunk: harmless re-fetch into the dead buffer
    u_dma(cn, nxt);
    if (cc + 1 < nchunk) patch_store();             // patch(cc+1), loaded one chunk ago
    patch_load(cc + 2 < nchunk ? cc + 2 : cc);
    __builtin_amdgcn_sched_barrier(0);              // keep the prefetch ABOVE the MFMA block
    const float* const uc = Us + cur * kUs + ufrag;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int cb = 0; cb < NI; ++cb) {
        const f4 w = *(const f4*)(uc + (j * NI + cb) * 256);      // bits {hi01, hi23, lo01, lo23} of (xi = 4*wave + j, rows cb)
        const s4 wh = __builtin_bit_cast(s4, (f2){w.x, w.y});
        const s4 wl = __builtin_bit_cast(s4, (f2){w.z, w.w});
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          const s4 vh = __builtin_bit_cast(s4, (f2){__builtin_bit_cast(float, Vh[tb][j][0]), __builtin_bit_cast(float, Vh[tb][j][1])});
          const s4 vl = __builtin_bit_cast(s4, (f2){__builtin_bit_cast(float, Vl[tb][j][0]), __builtin_bit_cast(float, Vl[tb][j][1])});
          acc[j][tb][cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wl, vh, acc[j][tb][cb], 0, 0, 0);   // small terms first
          acc[j][tb][cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh, vl, acc[j][tb][cb], 0, 0, 0);
          acc[j][tb][cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh, vh, acc[j][tb][cb], 0, 0, 0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                // next chunk's U (LDS-DMA) and patch have landed; everyone is done with `cur`
  }

  // ---------------- epilogue (the fp32 kernel's, conv_wino.hip): q_b = sum_j M[wave][j] A[j][b] -> LDS -> Y = sum_i A^T[a][i] q_b
  float* const Q = smem;                 // [4 waves][2][32 tiles][QLD]  (main-loop LDS is dead: last barrier passed)
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int cb = 0; cb < NI; ++cb) {
      const f4 q0 = acc[0][tb][cb] + acc[1][tb][cb] + acc[2][tb][cb];
      const f4 q1 = acc[1][tb][cb] - acc[2][tb][cb] - acc[3][tb][cb];
      const int t = tb * 16 + t16;
      *(f4*)(Q + ((wave * 2 + 0) * 32 + t) * QLD + cb * 16 + lq * 4) = q0;
      *(f4*)(Q + ((wave * 2 + 1) * 32 + t) * QLD + cb * 16 + lq * 4) = q1;
    }
  __syncthreads();

  constexpr int CQ = BN / 4;             // channel quads per tile (8): 32 tiles x 8 quads = 256 items, one per thread
  const bool do_stats = a.ssum != nullptr;
  const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = {0.f, 0.f, 0.f, 0.f};
  const int cq = tid % CQ;
  const int co = n0 + cq * 4;
  // fused BatchNorm-backward sums, as in conv_wino.hip
  const bool bnb = a.bnb_mean != nullptr;
  const int stat_c = a.out_up != nullptr ? a.up_c0 : a.Cout;
  f4 bmu = {0.f, 0.f, 0.f, 0.f}, brs = {0.f, 0.f, 0.f, 0.f};
  if (bnb && co < stat_c) { bmu = *(const f4*)(a.bnb_mean + co); brs = *(const f4*)(a.bnb_rstd + co); }
  {
    const int t = tid / CQ;
    f4 q[4][2];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int b = 0; b < 2; ++b) q[w][b] = *(const f4*)(Q + ((w * 2 + b) * 32 + t) * QLD + cq * 4);
    const int ty = t >> 3, tx = t & 7;
    if (a.out_up != nullptr) {               // fused concat split of a decoder dgrad (no addend / mask / stats here)
      const int ho = h0 + 2 * ty, wo = w0 + 2 * tx;
      if (ho < a.Ho && wo < a.Wo && co < a.Cout) {
        if (co < a.up_c0) {
          f4 v = q[0][0] + 2.f * q[1][0] - q[3][0] + q[0][1] + 2.f * q[1][1] - q[3][1];       // Y00 + Y10 + Y01 + Y11
          const size_t o2 = (((size_t)n * (a.Ho >> 1) + (ho >> 1)) * (a.Wo >> 1) + (wo >> 1)) * a.up_c0 + co;
          if (a.up_mask) {
            f4 mk = *(const f4*)(a.up_mask + o2);
            const f4 yr = mk;
            if (a.up_mscale) mk = mk * *(const f4*)(a.up_mscale + co) + *(const f4*)(a.up_mshift + co);
            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
            v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
            if (bnb) { ps_ += v; pq_ += v * ((yr - bmu) * brs); }
          }
          if (a.up_accum) v += *(const f4*)(a.out_up + o2);
          *(f4*)(a.out_up + o2) = v;
        } else {
          const int c1n = a.Cout - a.up_c0;
#pragma unroll
          for (int ya = 0; ya < 2; ++ya)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
              const f4 v = ya == 0 ? q[0][xb] + q[1][xb] + q[2][xb] : q[1][xb] - q[2][xb] - q[3][xb];
              *(f4*)(a.out + (((size_t)n * a.Ho + ho + ya) * a.Wo + wo + xb) * c1n + (co - a.up_c0)) = v;
            }
        }
      }
    } else {
#pragma unroll
      for (int ya = 0; ya < 2; ++ya)
#pragma unroll
        for (int xb = 0; xb < 2; ++xb) {
          f4 v = ya == 0 ? q[0][xb] + q[1][xb] + q[2][xb] : q[1][xb] - q[2][xb] - q[3][xb];
          const int ho = h0 + 2 * ty + ya, wo = w0 + 2 * tx + xb;
          if (ho < a.Ho && wo < a.Wo && co < a.Cout) {
            const size_t o = (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co;
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
            ps_ += v; pq_ += bnb ? v * ((yr - bmu) * brs) : v * v;
          }
        }
    }
  }
  if (do_stats) {
    __syncthreads();                       // Q is dead
    float* red = smem;                     // [32 groups][BN][2]
    constexpr int G = 256 / CQ;
    const int grp = tid / CQ;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[((grp * BN) + cq * 4 + e) * 2 + 0] = ps_[e];
      red[((grp * BN) + cq * 4 + e) * 2 + 1] = pq_[e];
    }
    __syncthreads();
    if (tid < BN) {
      const int c1 = n0 + tid;
      if (c1 < stat_c) {
        double sv = 0.0, qv = 0.0;
        for (int g = 0; g < G; ++g) { sv += (double)red[(g * BN + tid) * 2]; qv += (double)red[(g * BN + tid) * 2 + 1]; }
        atomicAdd(a.ssum + srep_off + c1, sv);
        atomicAdd(a.ssq + srep_off + c1, qv);
      }
    }
  }
}

// the fp32 Winograd rules plus whole 16-channel chunks on either side of the concat
bool conv_wino_x3_applicable(const ConvArgs& a) {
  return conv_wino_applicable(a) && (a.Ctot & 15) == 0 && (a.C0 & 15) == 0;
}

hipError_t launch_conv_wino_x3(const ConvArgs& a, hipStream_t st) {
  if (!conv_wino_x3_applicable(a)) return hipErrorInvalidValue;
  if (a.out_up && ((a.Ho | a.Wo) & 1 || (a.up_c0 & 3) || a.up_c0 > a.Cout || a.addend || a.mask || a.bias || (a.ssum && !a.bnb_mean) ||
                   (a.up_c0 < a.Cout && !a.out)))
    return hipErrorInvalidValue;
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !(a.out_up ? a.up_mask : a.mask) || a.up_accum)) return hipErrorInvalidValue;
  const int tilesN = (a.Cout + kBN - 1) / kBN;
  const int tilesW = (a.Wo + kTW - 1) / kTW, tilesH = (a.Ho + kTH - 1) / kTH;
  const size_t main_lds = (size_t)(2 * kUs + 4 * kPlane) * sizeof(float);
  const size_t q_lds = (size_t)4 * 2 * 32 * (kBN + kQPad) * sizeof(float);
  const size_t lds = main_lds > q_lds ? main_lds : q_lds;
  static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_wino_x3_kernel, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(31, a.flops, a.bytes, conv_wino_x3_kernel, dim3((unsigned)(a.N * tilesH * tilesW * tilesN)), dim3(256), lds, st, a);
  return hipGetLastError();
}

}  // namespace uwm
