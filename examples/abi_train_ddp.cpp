// Data-parallel training on the C ABI alone (include/uwm.h + the HIP runtime + RCCL; no Python, no torch): one process per
// GPU, every rank a replica, per-rank batches, the gradient exchange of SURVEY.md 8(e) through uwm_allreduce_grads — one
// bucket per backward stage on a communication stream, overlapped with the rest of the backward — and Adam with
// grad_scale = 1/world.  Rank 0 creates the ncclUniqueId and hands it over through a file.
//
//   usage: abi_train_ddp <rank> <world> <id-file> [arch 0|1] [encoder 18|34|50] [N] [H] [W] [steps]
//
// Prints one line per step (per-rank Dice loss) and, at the end, a checksum of the parameter arena: replicas must print
// the SAME checksum.  Built by __graft_entry__.build() and run by tests/test_abi_example_gpu.py (1 rank on a 1-GPU box,
// 2 ranks where two devices are visible).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "uwm.h"

#define HCHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define UCHK(e) do { if ((e) != 0) { fprintf(stderr, "uwm error: %s (line %d)\n", uwm_last_error(), __LINE__); return 3; } } while (0)
#define NCHK(e) do { ncclResult_t e_ = (e); if (e_ != ncclSuccess) { fprintf(stderr, "RCCL error %s at line %d\n", ncclGetErrorString(e_), __LINE__); return 4; } } while (0)

struct Rng {
  unsigned long long s;
  float u() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xFFFFFF) / 16777216.f; }
  float n() { const float a = u() + 1e-7f, b = u(); return sqrtf(-2.f * logf(a)) * cosf(6.2831853f * b); }
};

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: abi_train_ddp <rank> <world> <id-file> [arch enc N H W steps]\n"); return 1; }
  const int rank = atoi(argv[1]), world = atoi(argv[2]); const std::string idfile = argv[3];
  const int arch = argc > 4 ? atoi(argv[4]) : 0, enc = argc > 5 ? atoi(argv[5]) : 18;
  const int N = argc > 6 ? atoi(argv[6]) : 2, H = argc > 7 ? atoi(argv[7]) : 96, W = argc > 8 ? atoi(argv[8]) : 96;
  const int steps = argc > 9 ? atoi(argv[9]) : 4;
  int ndev = 0; HCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1 || world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "bad rank/world/devices\n"); return 1; }
  HCHK(hipSetDevice(rank % ndev));

  // ---- communicator: rank 0 publishes the unique id through a file (write to a temporary name, then rename)
  ncclUniqueId id;
  if (rank == 0) {
    NCHK(ncclGetUniqueId(&id));
    const std::string tmp = idfile + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb"); if (!f) { perror("id file"); return 1; }
    fwrite(&id, sizeof(id), 1, f); fclose(f);
    if (rename(tmp.c_str(), idfile.c_str()) != 0) { perror("rename"); return 1; }
  } else {
    FILE* f = nullptr;
    for (int i = 0; i < 600 && !(f = fopen(idfile.c_str(), "rb")); ++i) usleep(100000);
    if (!f || fread(&id, sizeof(id), 1, f) != 1) { fprintf(stderr, "rank %d: no unique id at %s\n", rank, idfile.c_str()); return 1; }
    fclose(f);
  }
  ncclComm_t comm; NCHK(ncclCommInitRank(&comm, world, id, rank));

  uwm_unet_desc d; memset(&d, 0, sizeof(d));
  d.encoder = enc; d.in_channels = 3; d.classes = 1; d.bn_eps = 1e-5f; d.bn_momentum = 0.1f; d.arch = arch;
  const int dc[5] = {256, 128, 64, 32, 16};
  for (int i = 0; i < 5; ++i) d.decoder_channels[i] = dc[i];
  uwm_handle h = nullptr;
  UCHK(uwm_create(&d, &h));
  const long long np = uwm_param_arena_floats(h), nb = uwm_buffer_arena_floats(h);
  const int CP = uwm_logits_channels(h), nst = uwm_num_stages(h);

  // identical initial replicas: the same seed on every rank (a real host would broadcast rank 0's arena)
  Rng wr{0x9E3779B97F4A7C15ull};
  std::vector<float> hp((size_t)np, 0.f), hb((size_t)nb, 0.f);
  for (int i = 0; i < uwm_num_tensors(h); ++i) {
    uwm_tensor_info t; UCHK(uwm_tensor_info_get(h, i, &t));
    float* base = (t.arena == UWM_ARENA_PARAM ? hp.data() : hb.data()) + t.offset;
    if (t.kind == UWM_KIND_CONV_W) {
      const float sd = sqrtf(2.f / (float)(t.shape[1] * t.shape[2] * t.shape[3]));
      for (long long o = 0; o < t.shape[0]; ++o) for (long long c = 0; c < t.shape[1]; ++c)
        for (long long r = 0; r < t.shape[2]; ++r) for (long long s = 0; s < t.shape[3]; ++s)
          base[o * t.stride[0] + c * t.stride[1] + r * t.stride[2] + s * t.stride[3]] = sd * wr.n();
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
  if (uwm_grad_arena(h) != grads) { fprintf(stderr, "uwm_grad_arena mismatch\n"); return 3; }

  // rank-distinct synthetic batch
  Rng dr{0xD1B54A32D192ED03ull + 7919ull * (unsigned long long)rank};
  const size_t npix = (size_t)N * H * W;
  std::vector<float> hx(npix * 3); std::vector<unsigned char> ht(npix, 0);
  for (int n = 0; n < N; ++n) {
    const int y0 = H / 4 + 3 * n + 2 * rank, x0 = W / 5 + 5 * n + rank, hh = H / 3, ww = W / 2;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
      const bool in = y >= y0 && y < y0 + hh && x >= x0 && x < x0 + ww;
      ht[((size_t)n * H + y) * W + x] = in ? 1 : 0;
      for (int c = 0; c < 3; ++c) hx[(((size_t)n * 3 + c) * H + y) * W + x] = 0.5f * dr.n() + (in ? 1.5f : 0.f);
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
  hipStream_t st, cst; HCHK(hipStreamCreate(&st)); HCHK(hipStreamCreateWithFlags(&cst, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev((size_t)nst); hipEvent_t evc;
  for (auto& e : ev) HCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HCHK(hipEventCreateWithFlags(&evc, hipEventDisableTiming));
  // each stage's weight gradients (library side stream) gate that stage's all-reduce on the communication stream
  UCHK(uwm_set_join_stream(h, cst));

  printf("rank %d/%d arch %d encoder %d: %lld parameters, %d gradient buckets\n", rank, world, arch, enc, uwm_param_count(h), nst);
  for (int step = 1; step <= steps; ++step) {
    UCHK(uwm_forward(h, x, logits, ws, wsb, N, H, W, 1, st));
    UCHK(uwm_loss(logits, CP, t, 2 /* uint8 */, (long long)npix, 1.f, 0.f, 1e-5f, 1e-7f, scr, loss3, dlogits, CP, 1.f, st));
    for (int k = 0; k < nst; ++k) {
      UCHK(uwm_backward(h, dlogits, ws, k, k + 1, st));
      HCHK(hipEventRecord(ev[(size_t)k], st));                     // bucket k's dgrad chain has been enqueued ...
      HCHK(hipStreamWaitEvent(cst, ev[(size_t)k], 0));             // ... its all-reduce runs behind it on the comm stream
      UCHK(uwm_allreduce_grads(h, comm, k, k + 1, cst));
    }
    HCHK(hipEventRecord(evc, cst)); HCHK(hipStreamWaitEvent(st, evc, 0));
    UCHK(uwm_adam(params, grads, m1, m2, np, 2e-3f, 0.9f, 0.999f, 1e-3f, 0.f, step, 1.f / (float)world, st));
    float l[3]; HCHK(hipMemcpyAsync(l, loss3, sizeof(l), hipMemcpyDeviceToHost, st)); HCHK(hipStreamSynchronize(st));
    printf("rank %d step %d dice_loss %.6f\n", rank, step, l[1]);
  }
  HCHK(hipMemcpy(hp.data(), params, np * sizeof(float), hipMemcpyDeviceToHost));
  double cs = 0.0, ca = 0.0; for (long long i = 0; i < np; ++i) { cs += hp[(size_t)i] * (double)((i % 251) + 1); ca += fabs(hp[(size_t)i]); }
  printf("rank %d param_checksum %.10e %.10e\n", rank, cs, ca);
  HCHK(hipStreamSynchronize(cst));
  uwm_destroy(h);
  NCHK(ncclCommDestroy(comm));
  return 0;
}
