// Winograd F(2x2, 3x3) convolution (forward AND dgrad) for gfx950 on v_mfma_f32_16x16x4_f32.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input tile, g: 3x3 filter, Y: 2x2 outputs
//
// 16 multiplies per 4 outputs instead of 36: 2.25x fewer MFMA FLOPs than the direct form for every
// 3x3 / stride 1 / pad 1 layer (fp32 throughout; the transforms only add and halve, so the error stays at a
// few fp32 ulps of the direct result — tests/test_ops_gpu.py bounds it against an fp64 reference).
//
// Work split (one workgroup = 8x16 output pixels = 4x8 Winograd tiles of one image, BN output channels):
//   * the 10x18 halo patch of an 8-channel chunk is staged in LDS exactly as conv_patch.hip does (lazy
//     BatchNorm+ReLU, nearest x2 upsample, channel concat, zero padding applied while staging);
//   * wave i (0..3) owns row i of the 4x4 Winograd domain: it builds (B^T d B)[i][0..3] for its 32 tiles
//     straight from the patch INTO REGISTERS in MFMA B-operand layout (lane = tile, 2 channels) — the
//     transformed input never goes through LDS — and multiplies with the pre-transformed weights
//     U[xi][cout][c] of its four xi (LDS image is a straight copy of the global layout, conflict-free
//     ds_read_b64);  acc[4 xi][2 tile blocks][BN/16] = 128 VGPRs for BN = 64;
//   * epilogue: each wave reduces its row over the columns (q_b = sum_j M[i][j] A[j][b], two values), the
//     four waves' q go through LDS once and every thread finishes Y = sum_i A^T[a][i] q_b for a 2x2 pixel
//     block x 4 channels, then bias / residual addend / ReLU mask / BatchNorm statistics as in conv_igemm.hip.
//
// Pre-transformed weights (wino_weights_kernel): Ut[Ctot/8][16 xi][ceil(rows/16)][4][16][2]
// (k-pair-major fragments: lane (co, lq) reads channels 2*lq, 2*lq+1 of its chunk).
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef UWM_WINO_ABL
#define UWM_WINO_ABL 0
#endif

constexpr int kTH = 8, kTW = 16, kPH = kTH + 2, kPW = kTW + 2, kPP = kPH * kPW;   // 180 patch pixels
constexpr int kPlane = kPP * 4;                                                  // floats per 4-channel plane
constexpr int kQPad = 4;

// ---------------------------------------------------------------- filter transform U = G g G^T
__global__ void wino_weights_kernel(const float* __restrict__ w, int wrows, int Kpad, int Ctot, int mirror,
                                    float* __restrict__ ut, int nCb, size_t total) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % Ctot), co = (int)(i / Ctot);
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = co < wrows ? w[(size_t)co * Kpad + (size_t)(mirror ? 8 - t : t) * Ctot + c] : 0.f;
  float t4[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t4[0][s] = g[s];
    t4[1][s] = 0.5f * (g[s] + g[3 + s] + g[6 + s]);
    t4[2][s] = 0.5f * (g[s] - g[3 + s] + g[6 + s]);
    t4[3][s] = g[6 + s];
  }
  const size_t base = ((size_t)(c >> 3) * 16 * nCb + (co >> 4)) * 128 + ((c & 7) >> 1) * 32 + (co & 15) * 2 + (c & 1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float u0 = t4[r][0], u1 = 0.5f * (t4[r][0] + t4[r][1] + t4[r][2]), u2 = 0.5f * (t4[r][0] - t4[r][1] + t4[r][2]),
                u3 = t4[r][2];
    ut[base + (size_t)(r * 4 + 0) * nCb * 128] = u0;
    ut[base + (size_t)(r * 4 + 1) * nCb * 128] = u1;
    ut[base + (size_t)(r * 4 + 2) * nCb * 128] = u2;
    ut[base + (size_t)(r * 4 + 3) * nCb * 128] = u3;
  }
}

// all layers of a model in one launch: blockIdx.y = job.  mode 0: U from [rows][Kpad] weights (k = tap*chans + c);
// mode 2: the dgrad filter bank straight from the FORWARD weights: rows = input channels, chans = output
// channels, g'[r][s] = w[c][(2-r)*3 + (2-s)][row]  (transposed + mirrored), no intermediate repack
__global__ void wino_weights_multi_kernel(const WinoJobs jobs) {
  const WinoJob jb = jobs.j[blockIdx.y];
  const int nCb = ((jb.rows + 63) / 64) * 4;
  // one thread = one output row x one channel PAIR, rows fastest: a wave writes 512 contiguous bytes per xi
  const size_t total = (size_t)(jb.chans >> 3) * nCb * 64;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int r16 = (int)(i & 15), lq = (int)((i >> 4) & 3);
  const int cb = (int)((i >> 6) % nCb), chunk = (int)((i >> 6) / nCb);
  const int row = cb * 16 + r16, c = chunk * 8 + lq * 2;
  f2 g[9];
  if (jb.mode == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
      g[t] = row < jb.rows ? *(const f2*)(jb.w + (size_t)row * jb.Kpad + (size_t)t * jb.chans + c) : (f2){0.f, 0.f};
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const size_t o = (size_t)(8 - t) * jb.rows + row;
      g[t].x = (row < jb.rows && c < jb.src_rows) ? jb.w[(size_t)c * jb.Kpad + o] : 0.f;
      g[t].y = (row < jb.rows && c + 1 < jb.src_rows) ? jb.w[(size_t)(c + 1) * jb.Kpad + o] : 0.f;
    }
  }
  f2 t4[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t4[0][s] = g[s];
    t4[1][s] = 0.5f * (g[s] + g[3 + s] + g[6 + s]);
    t4[2][s] = 0.5f * (g[s] - g[3 + s] + g[6 + s]);
    t4[3][s] = g[6 + s];
  }
  float* const base = jb.ut + ((size_t)chunk * 16 * nCb + cb) * 128 + lq * 32 + r16 * 2;
  const size_t xs = (size_t)nCb * 128;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    *(f2*)(base + (size_t)(r * 4 + 0) * xs) = t4[r][0];
    *(f2*)(base + (size_t)(r * 4 + 1) * xs) = 0.5f * (t4[r][0] + t4[r][1] + t4[r][2]);
    *(f2*)(base + (size_t)(r * 4 + 2) * xs) = 0.5f * (t4[r][0] - t4[r][1] + t4[r][2]);
    *(f2*)(base + (size_t)(r * 4 + 3) * xs) = t4[r][2];
  }
}
hipError_t launch_wino_weights_multi(const WinoJobs& jobs, hipStream_t st) {
  if (jobs.n <= 0) return hipSuccess;
  size_t mx = 0;
  for (int i = 0; i < jobs.n; ++i) {
    const size_t t = (size_t)(jobs.j[i].chans >> 3) * (((jobs.j[i].rows + 63) / 64) * 4) * 64;
    if (t > mx) mx = t;
  }
  hipLaunchKernelGGL(wino_weights_multi_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)jobs.n), dim3(256), 0, st, jobs);
  return hipGetLastError();
}

int wino_ncb(int wrows) { return ((wrows + 63) / 64) * 4; }     // 16-row blocks per xi, padded to whole 64-row tiles (LDS-DMA cannot zero-fill)
size_t wino_weights_floats(int wrows, int Ctot) { return (size_t)(Ctot / 8) * 16 * wino_ncb(wrows) * 128; }

hipError_t launch_wino_weights(const float* w, int wrows, int Kpad, int Ctot, int mirror, float* ut, hipStream_t st) {
  if (Ctot & 7) return hipErrorInvalidValue;
  const int nCb = wino_ncb(wrows);
  const size_t total = (size_t)nCb * 16 * Ctot;
  hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, wrows, Kpad, Ctot, mirror, ut,
                     nCb, total);
  return hipGetLastError();
}

// ---------------------------------------------------------------- main kernel
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
__device__ __forceinline__ void glds16(const float* g, float* l) {      // async 16 B/lane global -> LDS (wave-uniform l + lane*16)
  __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(uintptr_t)l, 16, 0, 0);
}

template <int NI>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const ConvArgs a) {
  constexpr int BN = NI * 16;
  constexpr int UR = NI * 2;                      // 1-KB wave-instructions of U per wave per chunk
  constexpr int QLD = BN + kQPad;
  constexpr int kUs = 16 * NI * 128;              // floats per U buffer
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Us = smem;                         // [2][16][NI][4][16][2]   (LDS-DMA destination, lane-linear)
  float* const Ps = smem + 2 * kUs;               // [2][2 planes][180 px (pairwise swizzled)][4]

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
  constexpr int dbg = UWM_WINO_ABL;       // compile-time timing ablations (scripts/ablate_wino.sh); 0 in the product build

  f4 acc[4][2][NI];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int cb = 0; cb < NI; ++cb) acc[j][tb][cb] = (f4){0.f, 0.f, 0.f, 0.f};

  // ---- patch staging through registers: 360 16-byte units = 2 rounds; loads are unconditional (clamped
  // addresses, zero-selected afterwards) so the loop body has no branches
  f4 pv[2], psc, psh; int prelu = 0; bool phas = false;
  int ppos[2]; bool pok[2], pact[2]; int poff0[2], poff1[2];     // chunk-invariant per-thread geometry
  {
    const int chu = tid & 1;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int u = rd * 256 + tid;
      pact[rd] = u < kPP * 2;
      const int pp = pact[rd] ? (u >> 1) : 0;
      const int py = pp / kPW, px = pp - py * kPW;
      const int hl = h0 - 1 + py, wl = w0 - 1 + px;
      pok[rd] = pact[rd] && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
      const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
      poff0[rd] = (int)(((size_t)n * a.s0.H + (hc >> a.s0.up)) * a.s0.W + (wc >> a.s0.up)) * a.s0.C + chu * 4;
      poff1[rd] = (int)(((size_t)n * a.s1.H + (hc >> a.s1.up)) * a.s1.W + (wc >> a.s1.up)) * a.s1.C + chu * 4;
      ppos[rd] = chu * kPlane + ((pp ^ ((py >> 1) & 1)) << 2);
    }
  }
  auto patch_load = [&](int cc) {
    const int c = cc * 8;                           // chunk base channel (chunks never straddle the two sources)
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
      if (pact[rd]) *(f4*)(Ps + buf * 2 * kPlane + ppos[rd]) = v;
    }
  };
  // ---- U chunk: UR LDS-DMA instructions per wave, 1 KB each, straight copy of the global image
  const float* const ug = a.wu + (size_t)(n0 / 16) * 128;
  int uoff[UR];                                     // per-lane source offset of each wave-instruction (chunk 0)
#pragma unroll
  for (int i = 0; i < UR; ++i) {
    const int L = (i * 4 + wave) * 64 + lane;       // f4 unit in the [16][NI*32] chunk image
    const int xi = L / (NI * 32), within = L - xi * (NI * 32);
    uoff[i] = xi * nCb * 128 + within * 4;
  }
  auto u_dma = [&](int cc, int buf, int i0, int i1) {      // pieces [i0, i1) of the chunk's UR wave-instructions
    const float* const uc = ug + (size_t)cc * 16 * nCb * 128;
#pragma unroll
    for (int i = 0; i < UR; ++i)
      if (i >= i0 && i < i1) glds16(uc + uoff[i], Us + buf * kUs + (i * 4 + wave) * 256);
  };
  // NI <= 2: the next chunk's U arrives in 1-KB LDS-DMA pieces issued BETWEEN the MFMA groups of this chunk (layer3-sized
  // launches: 130 -> 120 us, conv_wino<32> 193 -> 177 us in the step); the last group carries none, so every piece has
  // >= 16 MFMAs to land.  NI = 4 keeps all 8 pieces with group 0: spreading them measured 2-4 % SLOWER there (282 -> 293 us
  // on decoder block 0 conv1; 6.2 -> 6.4 ms per step) — its 16-MFMA groups are long enough that the issue cost was
  // already hidden by the second resident workgroup.
  constexpr int kD0 = NI == 4 ? UR : (UR * 3 + 7) / 8, kD1 = NI == 4 ? UR : (UR * 6 + 7) / 8;      // pieces [0,kD0) with group 0, [kD0,kD1) with 1, rest with 2

  // ---- per-lane addresses of the B^T row pair this wave combines: r = d[ra] + sg * d[rb]
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
      const int base = (lq >> 1) * kPlane + ((prow * kPW + 2 * tx) << 2) + (lq & 1) * 2;
      adA[tb][rs] = base + (f << 2);
      adB[tb][rs] = base + ((f ^ 1) << 2);
    }
  const int ufrag = wave * 4 * NI * 128 + lq * 32 + t16 * 2;

  const int nchunk = a.Ctot >> 3;
  u_dma(0, 0, 0, UR);
  patch_load(0);
  patch_store(0);
  __syncthreads();

  for (int cc = 0; cc < nchunk; ++cc) {
    const int cur = cc & 1, nxt = cur ^ 1;
    const int cn = cc + 1 < nchunk ? cc + 1 : cc;   // last chunk: harmless re-fetch into the dead buffer
    if (!(dbg & 4)) patch_load(cn);                 // (fenced into group 0 below: hipcc otherwise sinks it to the end)

    // (B^T d B)[wave][0..3] for both tile blocks, in registers
    const float* const pc = Ps + cur * 2 * kPlane;
    f2 V[2][4];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      if (dbg & 8) { V[tb][0] = V[tb][1] = V[tb][2] = V[tb][3] = (f2){1.f, 2.f}; continue; }
      const f2 a0 = *(const f2*)(pc + adA[tb][0]), a1 = *(const f2*)(pc + adB[tb][0]);
      const f2 a2 = *(const f2*)(pc + adA[tb][0] + 8), a3 = *(const f2*)(pc + adB[tb][0] + 8);
      const f2 b0 = *(const f2*)(pc + adA[tb][1]), b1 = *(const f2*)(pc + adB[tb][1]);
      const f2 b2 = *(const f2*)(pc + adA[tb][1] + 8), b3 = *(const f2*)(pc + adB[tb][1] + 8);
      const f2 r0 = a0 + sg * b0, r1 = a1 + sg * b1, r2 = a2 + sg * b2, r3 = a3 + sg * b3;
      V[tb][0] = r0 - r2; V[tb][1] = r1 + r2; V[tb][2] = r2 - r1; V[tb][3] = r1 - r3;
    }
    const float* const uc = Us + cur * kUs + ufrag;
    f2 wf[2][NI];
#pragma unroll
    for (int cb = 0; cb < NI; ++cb) wf[0][cb] = (dbg & 16) ? (f2){1.f, 1.f} : *(const f2*)(uc + cb * 128);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!(dbg & 2)) {
        if (j == 0) u_dma(cn, nxt, 0, kD0);
        if (j == 1) u_dma(cn, nxt, kD0, kD1);
        if (j == 2) u_dma(cn, nxt, kD1, UR);
      }
      if (j < 3) {
#pragma unroll
        for (int cb = 0; cb < NI; ++cb) wf[(j + 1) & 1][cb] = (dbg & 16) ? (f2){1.f, 1.f} : *(const f2*)(uc + ((j + 1) * NI + cb) * 128);
      }
#pragma unroll
      for (int cb = 0; cb < NI; ++cb)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
            if (!(dbg & 1)) acc[j][tb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j & 1][cb][e], V[tb][j][e], acc[j][tb][cb], 0, 0, 0);
            else acc[j][tb][cb][0] += wf[j & 1][cb][e] * V[tb][j][e];
      __builtin_amdgcn_sched_barrier(0);            // one fence per MFMA group: pins the DMA pieces / prefetch to their group
    }
    if (!(dbg & 4)) patch_store(nxt);
    if (!(dbg & 64)) __syncthreads();                    // next chunk's U (LDS-DMA) and patch have landed; everyone is done with `cur`
  }

  // ---------------- epilogue: q_b = sum_j M[wave][j] A[j][b]  ->  LDS  ->  Y = sum_i A^T[a][i] q_b
  if (dbg & 32) { if (acc[0][0][0][0] + acc[1][1][0][1] + acc[3][0][NI - 1][2] == 123.456f) a.out[0] = 1.f; return; }
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

  constexpr int CQ = BN / 4;             // channel quads per tile
  constexpr int ITEMS = 32 * CQ / 256;   // (tile, quad) items per thread (BN=64: 2, 32: 1)
  static_assert(32 * CQ % 256 == 0 || 32 * CQ < 256, "item split");
  const bool do_stats = a.ssum != nullptr;
  // BatchNorm statistics go to one of a.srep copies (few-channel layers launch tens of thousands of workgroups:
  // fp64 atomics on the same 2*C addresses serialise), bn_finalize adds the copies
  const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = {0.f, 0.f, 0.f, 0.f};
  const int cq = tid % CQ;
  const int co = n0 + cq * 4;
  // fused BatchNorm-backward sums (a dgrad whose masked output is the gradient wrt a BatchNorm output): second sum = v * yhat
  const bool bnb = a.bnb_mean != nullptr;
  const int stat_c = a.out_up != nullptr ? a.up_c0 : a.Cout;      // channels the sums cover
  f4 bmu = {0.f, 0.f, 0.f, 0.f}, brs = {0.f, 0.f, 0.f, 0.f};
  if (bnb && co < stat_c) { bmu = *(const f4*)(a.bnb_mean + co); brs = *(const f4*)(a.bnb_rstd + co); }
#pragma unroll
  for (int it = 0; it < (ITEMS > 0 ? ITEMS : 1); ++it) {
    const int item = it * 256 + tid;
    const int t = item / CQ;
    if (t >= 32) break;
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
      continue;
    }
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
          if (a.bnb_y) yr = *(const f4*)(a.bnb_y + o);
          ps_ += v; pq_ += bnb ? v * ((yr - bmu) * brs) : v * v;
        }
      }
  }
  if (do_stats) {
    __syncthreads();                       // Q is dead
    float* red = smem;                     // [256 / CQ groups][BN][2]
    constexpr int G = (256 / CQ) < 1 ? 1 : 256 / CQ;
    const int grp = tid / CQ;
    if (grp < G) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[((grp * BN) + cq * 4 + e) * 2 + 0] = ps_[e];
        red[((grp * BN) + cq * 4 + e) * 2 + 1] = pq_[e];
      }
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

template <int NI>
static hipError_t launch_w(const ConvArgs& a, hipStream_t st, int cls) {
  constexpr int BN = NI * 16;
  const int tilesN = (a.Cout + BN - 1) / BN;
  const int tilesW = (a.Wo + kTW - 1) / kTW, tilesH = (a.Ho + kTH - 1) / kTH;
  const size_t main_lds = (size_t)2 * (2 * kPlane + 16 * NI * 128) * sizeof(float);
  const size_t q_lds = (size_t)4 * 2 * 32 * (BN + kQPad) * sizeof(float);
  const size_t lds = main_lds > q_lds ? main_lds : q_lds;
  static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_wino_kernel<NI>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(cls, a.flops, a.bytes, (conv_wino_kernel<NI>), dim3((unsigned)(a.N * tilesH * tilesW * tilesN)), dim3(256), lds, st, a);
  return hipGetLastError();
}

bool conv_wino_applicable(const ConvArgs& a) {
  return a.wu != nullptr && a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 &&
         (a.rmul == 1 ? a.off == -1 : a.off == 1) && (a.Ctot & 7) == 0 && (a.C0 & 7) == 0 && (a.Cout & 3) == 0 &&
         a.Hl == a.Ho && a.Wl == a.Wo && a.Ho >= kTH && a.Wo >= kTW &&
         (size_t)a.N * a.s0.H * a.s0.W * a.s0.C < 0x7fffffffull && (size_t)a.N * a.s1.H * a.s1.W * a.s1.C < 0x7fffffffull;   // 32-bit patch offsets
}

// bn: 0 auto | 64 | 32 | 16 output channels per workgroup
hipError_t launch_conv_wino(const ConvArgs& a, hipStream_t st, int bn) {
  if (!conv_wino_applicable(a)) return hipErrorInvalidValue;
  if (a.out_up && ((a.Ho | a.Wo) & 1 || (a.up_c0 & 3) || a.up_c0 > a.Cout || a.addend || a.mask || a.bias || (a.ssum && !a.bnb_mean) ||
                   (a.up_c0 < a.Cout && !a.out) || bn == 8))
    return hipErrorInvalidValue;
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !(a.out_up ? a.up_mask : (a.bnb_y ? a.bnb_y : a.mask)) || a.up_accum || bn == 8 || (a.out_up && a.bnb_y))) return hipErrorInvalidValue;
  if (bn == 8) return launch_conv_wino8(a, st);
  if (bn <= 0 && !a.bnb_mean && conv_wino8_applicable(a)) return launch_conv_wino8(a, st);
  if (bn <= 0) {
    bn = a.Cout > 32 ? 64 : (a.Cout > 16 ? 32 : 16);
    // deep, spatially small layers: 64-channel tiles leave half the workgroup slots empty -> 32-channel tiles
    const long sp = (long)route_N(a) * ((a.Ho + kTH - 1) / kTH) * ((a.Wo + kTW - 1) / kTW);
    if (bn == 64 && sp * ((a.Cout + 63) / 64) < 512) bn = 32;
  }
  switch (bn) {
    case 64: return launch_w<4>(a, st, 19);
    case 32: return launch_w<2>(a, st, 20);
    case 16: return launch_w<1>(a, st, 21);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace uwm
