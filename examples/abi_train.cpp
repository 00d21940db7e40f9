// Torch-free use of the C ABI (include/uwm.h): a host program that owns every device buffer through the HIP runtime,
// builds Unet / UnetPlusPlus, and runs a few complete train steps (forward, Dice loss, staged backward, Adam) on a
// synthetic batch.  Prints one line per step; the loss must fall.  Built by __graft_entry__.build()
// (hipcc examples/abi_train.cpp -Iinclude -Lunet-watermark_amd -luwm) and run by tests/test_abi_example_gpu.py.
//
//   usage: abi_train [arch 0|1] [encoder 18|34|50] [N] [H] [W] [steps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "uwm.h"

#define HCHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define UCHK(e) do { if ((e) != 0) { fprintf(stderr, "uwm error: %s (line %d)\n", uwm_last_error(), __LINE__); return 3; } } while (0)

static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
static float urand() { g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull; return (float)((g_rng >> 40) & 0xFFFFFF) / 16777216.f; }
static float nrand() { const float u = urand() + 1e-7f, v = urand(); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }

int main(int argc, char** argv) {
  const int arch = argc > 1 ? atoi(argv[1]) : 0, enc = argc > 2 ? atoi(argv[2]) : 18;
  const int N = argc > 3 ? atoi(argv[3]) : 4, H = argc > 4 ? atoi(argv[4]) : 128, W = argc > 5 ? atoi(argv[5]) : 128;
  const int steps = argc > 6 ? atoi(argv[6]) : 4;
  uwm_unet_desc d; memset(&d, 0, sizeof(d));
  d.encoder = enc; d.in_channels = 3; d.classes = 1; d.bn_eps = 1e-5f; d.bn_momentum = 0.1f; d.arch = arch;
  const int dc[5] = {256, 128, 64, 32, 16};
  for (int i = 0; i < 5; ++i) d.decoder_channels[i] = dc[i];
  uwm_handle h = nullptr;
  UCHK(uwm_create(&d, &h));
  const long long np = uwm_param_arena_floats(h), nb = uwm_buffer_arena_floats(h);
  const int CP = uwm_logits_channels(h);

  // host-side init through the tensor table: He-normal conv weights, gamma 1, running_var 1, everything else 0
  std::vector<float> hp((size_t)np, 0.f), hb((size_t)nb, 0.f);
  for (int i = 0; i < uwm_num_tensors(h); ++i) {
    uwm_tensor_info t; UCHK(uwm_tensor_info_get(h, i, &t));
    float* base = (t.arena == UWM_ARENA_PARAM ? hp.data() : hb.data()) + t.offset;
    if (t.kind == UWM_KIND_CONV_W) {
      const float sd = sqrtf(2.f / (float)(t.shape[1] * t.shape[2] * t.shape[3]));
      for (long long o = 0; o < t.shape[0]; ++o) for (long long c = 0; c < t.shape[1]; ++c)
        for (long long r = 0; r < t.shape[2]; ++r) for (long long s = 0; s < t.shape[3]; ++s)
          base[o * t.stride[0] + c * t.stride[1] + r * t.stride[2] + s * t.stride[3]] = sd * nrand();
    } else if (t.kind == UWM_KIND_BN_GAMMA || t.kind == UWM_KIND_BN_VAR) {
      for (long long c = 0; c < t.shape[0]; ++c) base[c] = 1.f;
    }
  }
  float *params, *grads, *buffers, *m1, *m2;
  HCHK(hipMalloc((void**)&params, np * sizeof(float))); HCHK(hipMalloc((void**)&grads, np * sizeof(float)));
  HCHK(hipMalloc((void**)&m1, np * sizeof(float))); HCHK(hipMalloc((void**)&m2, np * sizeof(float)));
  HCHK(hipMalloc((void**)&buffers, nb * sizeof(float)));
  HCHK(hipMemcpy(params, hp.data(), np * sizeof(float), hipMemcpyHostToDevice));
  HCHK(hipMemcpy(buffers, hb.data(), nb * sizeof(float), hipMemcpyHostToDevice));
  HCHK(hipMemset(grads, 0, np * sizeof(float))); HCHK(hipMemset(m1, 0, np * sizeof(float))); HCHK(hipMemset(m2, 0, np * sizeof(float)));
  UCHK(uwm_bind(h, params, grads, buffers));

  // synthetic batch: noise image with a brighter rectangle, the rectangle is the mask
  const size_t npix = (size_t)N * H * W;
  std::vector<float> hx(npix * 3); std::vector<unsigned char> ht(npix, 0);
  for (int n = 0; n < N; ++n) {
    const int y0 = H / 4 + 3 * n, x0 = W / 5 + 5 * n, hh = H / 3, ww = W / 2;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
      const bool in = y >= y0 && y < y0 + hh && x >= x0 && x < x0 + ww;
      ht[((size_t)n * H + y) * W + x] = in ? 1 : 0;
      for (int c = 0; c < 3; ++c) hx[(((size_t)n * 3 + c) * H + y) * W + x] = 0.5f * nrand() + (in ? 1.5f : 0.f);
    }
  }
  float *x, *logits, *dlogits, *loss3; unsigned char* t; void *ws, *scr;
  const size_t wsb = uwm_workspace_bytes(h, N, H, W, 1);
  if (!wsb) { fprintf(stderr, "uwm error: %s\n", uwm_last_error()); return 3; }
  HCHK(hipMalloc((void**)&x, hx.size() * sizeof(float))); HCHK(hipMalloc((void**)&t, npix));
  HCHK(hipMalloc((void**)&logits, npix * CP * sizeof(float))); HCHK(hipMalloc((void**)&dlogits, npix * CP * sizeof(float)));
  HCHK(hipMalloc((void**)&loss3, 3 * sizeof(float))); HCHK(hipMalloc(&scr, 64)); HCHK(hipMalloc(&ws, wsb));
  HCHK(hipMemcpy(x, hx.data(), hx.size() * sizeof(float), hipMemcpyHostToDevice));
  HCHK(hipMemcpy(t, ht.data(), npix, hipMemcpyHostToDevice));
  hipStream_t st; HCHK(hipStreamCreate(&st));

  printf("arch %d encoder %d: %lld parameters, workspace %.1f MB\n", arch, enc, uwm_param_count(h), wsb / 1048576.0);
  for (int step = 1; step <= steps; ++step) {
    UCHK(uwm_forward(h, x, logits, ws, wsb, N, H, W, 1, st));
    UCHK(uwm_loss(logits, CP, t, 2 /* uint8 */, (long long)npix, 1.f, 0.f, 1e-5f, 1e-7f, scr, loss3, dlogits, CP, 1.f, st));
    UCHK(uwm_backward(h, dlogits, ws, 0, uwm_num_stages(h), st));
    UCHK(uwm_adam(params, grads, m1, m2, np, 2e-3f, 0.9f, 0.999f, 1e-8f, 0.f, step, 1.f, st));
    float l[3]; HCHK(hipMemcpyAsync(l, loss3, sizeof(l), hipMemcpyDeviceToHost, st)); HCHK(hipStreamSynchronize(st));
    printf("step %d dice_loss %.6f\n", step, l[1]);
  }
  uwm_destroy(h);
  return 0;
}
