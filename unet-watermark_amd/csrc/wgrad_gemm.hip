// Weight gradient of the 1x1 / stride-1 convolutions as a persistent, LDS-DMA-fed GEMM for gfx950 (v_mfma_f32_16x16x4_f32):
//
//   dW[co][c] = sum_m dY[m][co] * X~[m][c]          m = pixel (the reduction), X~ = lazily normalised input
//
// Both operands lie in memory as [pixel][channel] rows, the reduction runs down the rows: a stage is 32 pixels x TA output
// channels of dY and 32 pixels x TB input channels of X, both brought in by LDS-DMA (no registers, no ds_write), double
// buffered, one barrier per stage (TA, TB in {64, 128}: the 64-wide tiles for layers with <= 64 channels on that side).  The MFMA fragments are 4-byte LDS reads (lane = channel, MFMA k = pixel); a 512-byte row
// stride would put the four pixels of a fragment on the same banks, so the 16-byte units of row p are stored XOR-ed with
// (p & 3) << 2 — applied to the SOURCE address of each LDS-DMA lane — which makes every fragment read conflict-free.  The
// producer's BatchNorm + ReLU is applied to the B fragments after the read: a lane's four channels are fixed for the whole
// kernel, so its scale / shift are eight registers loaded once.
// The pixel range is split over workgroups; every split writes its TA x 128 partial tile into its own dW-shaped image and
// wgrad_reduce_kernel (wgrad_wino.hip) adds the images in a fixed order: no float atomics, bit-reproducible, and the write
// amplification of the atomic version (wgrad_igemm.hip: every split read-modify-writes dW through L2) is gone.
//
// Replaces the weight-gradient half of autograd's conv2d backward for the Bottleneck / MBConv 1x1 layers.
#include "uwm_kernels.h"
#include <cstdlib>

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
__device__ __forceinline__ void wg_glds16(const float* g, float* l) {      // async 16 B/lane global -> LDS (wave-uniform l + lane*16)
  __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(uintptr_t)l, 16, 0, 0);
}


template <int TA, int TB>
__global__ __launch_bounds__(256, (TA + TB <= 128 ? 4 : (TA + TB <= 192 ? 3 : 2))) void wgrad_gemm_kernel(const WgradArgs a, int tilesB, int ntiles, int msplit, float* __restrict__ out, size_t split_stride) {
  constexpr int MI = TA / 32;                     // 16-row (output channel) MFMA tiles per wave (2 x 2 waves)
  constexpr int NI = TB / 32;                     // 16-column (input channel) tiles per wave
  constexpr int AI = TA / 32, BI = TB / 32;       // 1-KB LDS-DMA instructions per wave per stage (dY: 32 x TA, X: 32 x TB floats)
  constexpr int kStage = 32 * (TA + TB);          // floats per stage
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 1, wb = wave & 1;
  const int lrow = lane & 15, lq = lane >> 4;

  const int bid = (int)blockIdx.x;
  const int split = bid / ntiles, tile = bid - split * ntiles;
  const int ta = tile / tilesB, tb = tile - ta * tilesB;
  const int a0 = ta * TA, b0 = tb * TB;
  const int mbeg = split * msplit, mend = min(a.M, mbeg + msplit);
  const int nst = (mend - mbeg) >> 5;             // (msplit and M are multiples of 32)

  // per-lane LDS-DMA geometry.  dY stage [32][TA]: a row is TA/4 units; X stage [32][TB]: TB/4 units per row.
  constexpr int UA = TA / 4, UB = TB / 4;         // units per row (32 or 16)
  auto issue = [&](int st_, int buf) {
    const int m0 = mbeg + st_ * 32;
    float* const as = smem + buf * kStage;
    float* const bs = as + 32 * TA;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int L = (i * 4 + wave) * 64 + lane;                   // unit index in the [32][UA] image
      const int row = L / UA, q = L - row * UA;
      const int u = q ^ ((row & 3) << 2);
      const int cu = min(a0 / 4 + u, a.Cout / 4 - 1);              // columns past Cout: duplicates, never stored
      wg_glds16(a.dy + (size_t)(m0 + row) * a.Cout + cu * 4, as + (i * 4 + wave) * 256);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int L = (i * 4 + wave) * 64 + lane;
      const int row = L / UB, q = L - row * UB;
      const int u = q ^ ((row & 3) << 2);
      const int cu = min(b0 / 4 + u, a.Ctot / 4 - 1);
      wg_glds16(a.s0.ptr + (size_t)(m0 + row) * a.s0.C + cu * 4, bs + (i * 4 + wave) * 256);
    }
  };

  // lazy BatchNorm + ReLU of the input: this lane's columns are b0 + (wb*4 + j)*16 + lrow for the whole kernel
  const bool lazy = a.s0.scale != nullptr;
  float lsc[NI], lsh[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int c = min(b0 + (wb * NI + j) * 16 + lrow, a.Ctot - 1);
    lsc[j] = lazy ? a.s0.scale[c] : 1.f; lsh[j] = lazy ? a.s0.shift[c] : 0.f;
  }
  const int relu = a.s0.relu;

  f4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  if (nst > 0) issue(0, 0);
  __syncthreads();
  for (int st_ = 0; st_ < nst; ++st_) {
    const int cur = st_ & 1;
    if (st_ + 1 < nst) issue(st_ + 1, cur ^ 1);
    const float* const as = smem + cur * kStage;
    const float* const bs = as + 32 * TA;
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      const int p = k4 * 4 + lq;                                   // pixel of the stage this lane supplies (MFMA k index)
      const int sw = (p & 3) << 2;
      float af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int col = (wa * MI + i) * 16 + lrow;
        af[i] = as[p * TA + (((col >> 2) ^ sw) << 2) + (col & 3)];
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = (wb * NI + j) * 16 + lrow;
        float v = bs[p * TB + (((col >> 2) ^ sw) << 2) + (col & 3)];
        if (lazy) { v = v * lsc[j] + lsh[j]; if (relu) v = fmaxf(v, 0.f); }
        bf[j] = v;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                               // stage st_+1 (LDS-DMA) has landed; everyone is done with `cur`
  }

  // D[m = co][n = c]: lane holds rows 4*lq + e, column lrow.  Partial image `out + split*split_stride` has dW's [wrows][Kpad] layout.
  float* const o = out + (size_t)split * split_stride;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int c = b0 + (wb * NI + j) * 16 + lrow;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = a0 + (wa * MI + i) * 16 + lq * 4 + e;
        if (co < a.wrows && c < a.Kpad) {                // columns in [Ctot, Kpad): zeros in a partial image (the reduce adds whole images), untouched in dW
          float* const d = o + (size_t)co * a.Kpad + c;
          if (split_stride) *d = c < a.Ctot ? acc[i][j][e] : 0.f; else if (c < a.Ctot) *d += acc[i][j][e];
        }
      }
    }
}

// many pixels under a tiny dW (<= 64 channels on one side at >= 64k pixels: layer1 of resnet50, the first MBConv stages): one or two
// output tiles, hundreds of pixel splits — the register-staged kernel of wgrad_igemm.hip streams those at 2.8-3.1 TB/s, this one at
// 2.0-2.4 (profiles/r02_*_time_1x1*.txt); everywhere else this kernel is 1.2-1.5x faster
bool wgrad_gemm_preferred(const WgradArgs& a) {
  return wgrad_gemm_applicable(a) && !(a.M >= 65536 && (a.wrows <= 64 || a.Ctot <= 64));
}

bool wgrad_gemm_applicable(const WgradArgs& a) {
  static const bool off = dbg_flag("UWM_NO_WGRAD_GEMM");
  return !off && a.ntaps == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && a.s0.up == 0 && a.C0 == a.Ctot && a.s0.C == a.Ctot &&
         (a.Ctot & 3) == 0 && a.Kpad == ((a.Ctot + 31) & ~31) && (a.Cout & 3) == 0 && a.Cout >= 32 && a.Ctot >= 16 && (a.M & 31) == 0 && a.M >= 256 &&
         a.Hl == a.Ho && a.Wl == a.Wo && a.s0.H == a.Ho && a.s0.W == a.Wo;
}

template <int TA, int TB>
static hipError_t launch_wg(const WgradArgs& a0, hipStream_t st, int cls) {
  WgradArgs a = a0;
  const int tilesA = (a.wrows + TA - 1) / TA, tilesB = (a.Kpad + TB - 1) / TB;      // (Kpad, not Ctot: the pad columns of the partial images must be written)
  const int ntiles = tilesA * tilesB;
  // pixel splits: fill 2 workgroups per CU, at least 8 stages (256 pixels) per split, bounded by the partial-sum scratch
  const int slots = (TA + TB <= 128 ? 4 : (TA + TB <= 192 ? 3 : 2)) * device_cu_count();      // resident workgroups by LDS (32 / 48 / 64 KB)
  int nsplit = (slots + ntiles - 1) / ntiles;
  const int max_by_m = a.M / 256 > 0 ? a.M / 256 : 1;
  if (nsplit > max_by_m) nsplit = max_by_m;
  const size_t image = (size_t)a.wrows * a.Kpad;
  if (!a.part || a.part_floats < image * 2) { a.part = wgrad_op_scratch(); a.part_floats = wgrad_wino_scratch_floats(); }
  if (a.part && (size_t)nsplit * image > a.part_floats) nsplit = (int)(a.part_floats / image);
  if (nsplit < 1) nsplit = 1;
  int msplit = ((a.M / 32 + nsplit - 1) / nsplit) * 32;
  nsplit = (a.M + msplit - 1) / msplit;
  if (nsplit > 1 && !a.part) return hipErrorOutOfMemory;
  const size_t lds = (size_t)2 * 32 * (TA + TB) * sizeof(float);
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_gemm_kernel<TA, TB>, lds); if (e != hipSuccess) return e; }
  float* const out = nsplit > 1 ? a.part : a.dw;
  const size_t stride = nsplit > 1 ? image : 0;
  UWM_LAUNCH(cls, a.flops, a.bytes, (wgrad_gemm_kernel<TA, TB>), dim3((unsigned)(ntiles * nsplit)), dim3(256), lds, st, a, tilesB, ntiles, msplit, out, stride);
  if (nsplit > 1) return launch_wgrad_reduce(a.part, nsplit, image / 4, a.dw, st, a.rq);
  return hipGetLastError();
}

hipError_t launch_wgrad_gemm(const WgradArgs& a, hipStream_t st) {
  if (!wgrad_gemm_applicable(a)) return hipErrorInvalidValue;
  if (a.Kpad <= 64) return a.wrows <= 64 ? launch_wg<64, 64>(a, st, 40) : launch_wg<128, 64>(a, st, 39);
  return a.wrows <= 64 ? launch_wg<64, 128>(a, st, 40) : launch_wg<128, 128>(a, st, 39);
}

}  // namespace uwm
