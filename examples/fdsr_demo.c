/*
 * fdsr_demo.c -- the C ABI of libfdsr_hip.so driven from plain C (no Python, no torch):
 * what a host written in any language binds.  Reads a bundle directory written by
 * tools/export_bundle.py (hyper-parameters, a reference-format checkpoint flattened to one file,
 * the schedule scalars, a conditioning batch and optionally the noise planes), runs
 * GaussianDiffusion.p_sample_loop (diffusion.py:192-221) through fdsr_sample and writes out.bin.
 *
 *   fdsr_demo <bundle_dir> [precision 0|1|2] [graph 0|1]
 *
 * tests/test_gpu_c_abi.py checks out.bin bit-for-bit against the Python facade on the same inputs.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fdsr.h"

#define HIP_OK(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)
#define FDSR_CHECK(h, x)                                                                 \
  do {                                                                                   \
    int rc_ = (x);                                                                       \
    if (rc_ != FDSR_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, fdsr_last_error(h)); return 3; } \
  } while (0)

static void* slurp(const char* dir, const char* name, size_t* bytes) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%s", dir, name);
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(n > 0 ? (size_t)n : 1);
  if (p && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
  fclose(f);
  if (bytes) *bytes = (size_t)n;
  return p;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <bundle_dir> [precision] [graph]\n", argv[0]); return 1; }
  const char* dir = argv[1];
  const int precision = argc > 2 ? atoi(argv[2]) : FDSR_PREC_F16X3;
  const int graph = argc > 3 ? atoi(argv[3]) : 0;

  /* config.bin: int32 in,out,inner,groups,n_mults,mults[8],res_blocks,variant,image_size,n_attn,attn[8],B,H,W,seed ; float dropout */
  size_t nb = 0;
  int32_t* ci = (int32_t*)slurp(dir, "config.bin", &nb);
  if (!ci || nb < 29 * 4) { fprintf(stderr, "bad config.bin\n"); return 1; }
  fdsr_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.in_channel = ci[0]; cfg.out_channel = ci[1]; cfg.inner_channel = ci[2]; cfg.norm_groups = ci[3];
  cfg.n_mults = ci[4];
  for (int i = 0; i < FDSR_MAX_MULTS; ++i) cfg.channel_mults[i] = ci[5 + i];
  cfg.res_blocks = ci[13]; cfg.variant = ci[14]; cfg.image_size = ci[15]; cfg.n_attn_res = ci[16];
  for (int i = 0; i < FDSR_MAX_MULTS; ++i) cfg.attn_res[i] = ci[17 + i];
  const int B = ci[25], H = ci[26], W = ci[27];
  const uint64_t seed = (uint32_t)ci[28];
  cfg.dropout = 0.0f;

  fdsr_handle h = NULL;
  if (fdsr_create(&cfg, &h) != FDSR_OK) { fprintf(stderr, "fdsr_create: %s\n", fdsr_last_error(NULL)); return 3; }
  printf("%s\n", fdsr_version());

  /* weights.bin: records [int32 keylen][key][int32 ndim][int64 shape[ndim]][fp32 data], reference layout */
  size_t wb = 0;
  unsigned char* w = (unsigned char*)slurp(dir, "weights.bin", &wb);
  if (!w) { fprintf(stderr, "no weights.bin\n"); return 1; }
  size_t off = 0;
  int n_loaded = 0;
  while (off < wb) {
    int32_t klen, ndim;
    char key[256];
    int64_t shape[4] = {1, 1, 1, 1};
    memcpy(&klen, w + off, 4); off += 4;
    memcpy(key, w + off, (size_t)klen); key[klen] = 0; off += (size_t)klen;
    memcpy(&ndim, w + off, 4); off += 4;
    memcpy(shape, w + off, 8 * (size_t)ndim); off += 8 * (size_t)ndim;
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    float* data = (float*)malloc(numel * 4);          /* records are not 4-byte aligned in the file */
    memcpy(data, w + off, numel * 4); off += numel * 4;
    FDSR_CHECK(h, fdsr_load_weight(h, key, data, shape, ndim));
    free(data);
    ++n_loaded;
  }
  free(w);
  if (!fdsr_weights_complete(h)) { fprintf(stderr, "checkpoint incomplete after %d tensors\n", n_loaded); return 3; }

  /* schedule.bin: int32 T, then noise_level, sqrt_recip, sqrt_recipm1, coef1, coef2, sigma (T floats each) */
  unsigned char* sb = (unsigned char*)slurp(dir, "schedule.bin", &nb);
  if (!sb) { fprintf(stderr, "no schedule.bin\n"); return 1; }
  int32_t T;
  memcpy(&T, sb, 4);
  const float* tab = (const float*)(sb + 4);
  fdsr_schedule s = {T, tab, tab + T, tab + 2 * T, tab + 3 * T, tab + 4 * T, tab + 5 * T};
  FDSR_CHECK(h, fdsr_set_schedule(h, &s));
  FDSR_CHECK(h, fdsr_set_precision(h, precision));

  const size_t img = (size_t)B * 3 * H * W;
  size_t cb = 0, nzb = 0;
  float* cond_h = (float*)slurp(dir, "cond.bin", &cb);
  float* noise_h = (float*)slurp(dir, "noise.bin", &nzb);   /* optional: absent => the engine draws (seed) */
  if (!cond_h || cb != img * 4) { fprintf(stderr, "bad cond.bin\n"); return 1; }
  if (noise_h && nzb != img * 4 * (size_t)T) { fprintf(stderr, "bad noise.bin\n"); return 1; }

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  float *cond_d = NULL, *noise_d = NULL, *out_d = NULL;
  void* ws = NULL;
  size_t ws_bytes = 0;
  FDSR_CHECK(h, fdsr_workspace_bytes(h, B, H, W, &ws_bytes));
  HIP_OK(hipMalloc((void**)&cond_d, img * 4));
  HIP_OK(hipMalloc((void**)&out_d, img * 4));
  HIP_OK(hipMalloc(&ws, ws_bytes));
  HIP_OK(hipMemcpy(cond_d, cond_h, img * 4, hipMemcpyHostToDevice));
  if (noise_h) {
    HIP_OK(hipMalloc((void**)&noise_d, img * 4 * (size_t)T));
    HIP_OK(hipMemcpy(noise_d, noise_h, img * 4 * (size_t)T, hipMemcpyHostToDevice));
  } else {
    FDSR_CHECK(h, fdsr_set_seed(h, seed));
  }
  FDSR_CHECK(h, fdsr_sample(h, cond_d, noise_d, out_d, NULL, B, H, W, ws, ws_bytes, st, graph ? FDSR_SAMPLE_GRAPH : 0));
  HIP_OK(hipStreamSynchronize(st));

  float* out_h = (float*)malloc(img * 4);
  HIP_OK(hipMemcpy(out_h, out_d, img * 4, hipMemcpyDeviceToHost));
  char path[1024];
  snprintf(path, sizeof path, "%s/out.bin", dir);
  FILE* f = fopen(path, "wb");
  if (!f || fwrite(out_h, 4, img, f) != img) { fprintf(stderr, "cannot write %s\n", path); return 1; }
  fclose(f);
  printf("sampled B=%d %dx%d T=%d precision=%d graph=%d workspace=%.1f MB -> %s\n", B, H, W, T, precision, graph,
         ws_bytes / 1e6, path);

  /* error behaviour at the boundary: codes, never exceptions */
  if (fdsr_sample(h, cond_d, noise_d, out_d, NULL, B, H, W, ws, 1024, st, 0) != FDSR_E_WORKSPACE) return 4;
  if (fdsr_sample(h, cond_d, noise_d, out_d, NULL, B, H + 1, W, ws, ws_bytes, st, 0) != FDSR_E_INVALID) return 4;

  fdsr_destroy(h);
  hipFree(cond_d); hipFree(noise_d); hipFree(out_d); hipFree(ws);
  free(cond_h); free(noise_h); free(out_h); free(ci); free(sb);
  return 0;
}
