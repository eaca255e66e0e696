// What does a last-arriver ("ticket") fold of a tiny dependent kernel cost on gfx950, against launching that kernel?
// Round-3 verdict item 2 asks to fold gn_finalize (and splitk_reduce) into their producers at batch 1.  The knock-out bound says the 82
// launches cost 12.3 of 35.3 ms per image; this probe prices what replaces them.  Chain of NREP steps inside one hipGraph, two forms:
//   (A) producer kernel (G workgroups: each writes its 256-B partial) -> finalize kernel (32 workgroups sum the partials, write 1 KB)
//   (B) producer kernel with a ticket: partial stored write-through (sc1), s_waitcnt vmcnt(0), barrier, one agent-scope atomic add;
//       the workgroup that draws the last ticket acquires, sums ALL partials in the same fixed order and writes the 1 KB; counter reset
// Both forms also do ~W microseconds of dummy work per workgroup before publishing (a stand-in for the conv) so that the tail behaviour,
// not the launch of an empty kernel, is what differs.   hipcc --offload-arch=gfx950 -O3 ticket_probe.hip -o ticket_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void busy(int iters, float* sink) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.678f) *sink = a;
}

__global__ void __launch_bounds__(256) producer(float* part, int iters, float* sink) {
  busy(iters, sink);
  if (threadIdx.x < 64) part[blockIdx.x * 64 + threadIdx.x] = (float)(blockIdx.x + threadIdx.x);
}

__global__ void __launch_bounds__(256) finalize(const float* part, int G, float* out) {
  // workgroup b sums column block b of the partials over all G producers, in order (fp64), like gn_finalize_kernel
  const int c = blockIdx.x * 2 + (threadIdx.x >> 7);      // 64 columns over 32 workgroups: 2 per workgroup
  __shared__ double sh[256];
  double s = 0.0;
  for (int g = threadIdx.x & 127; g < G; g += 128) s += part[g * 64 + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  if ((threadIdx.x & 127) == 0) {
    double t = 0.0;
    for (int i = 0; i < 128; ++i) t += sh[threadIdx.x + i];
    out[c] = (float)t;
  }
}

__global__ void __launch_bounds__(256) producer_ticket(float* part, int iters, float* sink, unsigned* counter, int G, float* out) {
  busy(iters, sink);
  if (threadIdx.x < 64)
    __hip_atomic_store(part + blockIdx.x * 64 + threadIdx.x, (float)(blockIdx.x + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ unsigned last;
  if (threadIdx.x == 0) last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(G - 1);
  __syncthreads();
  if (!last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ double sh[256];
  {   // all 64 columns at once: thread = (column, quarter of the producers); quarters folded in order
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    double s = 0.0;
    for (int g = q; g < G; g += 4) s += part[g * 64 + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (q == 0) out[c] = (float)(((sh[c] + sh[64 + c]) + sh[128 + c]) + sh[192 + c]);
  }
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char** argv) {
  const int NREP = 200;
  float *part, *out, *sink; unsigned* counter;
  CK(hipMalloc(&part, 4096 * 64 * 4)); CK(hipMalloc(&out, 64 * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&counter, 64));
  CK(hipMemset(counter, 0, 64));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int G : {16, 64, 256}) for (int iters : {0, 4000, 16000}) {
    float ms[2] = {0, 0};
    for (int form = 0; form < 2; ++form) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int r = 0; r < NREP; ++r) {
        if (form == 0) {
          hipLaunchKernelGGL(producer, dim3(G), dim3(256), 0, st, part, iters, sink);
          hipLaunchKernelGGL(finalize, dim3(32), dim3(256), 0, st, part, G, out);
        } else {
          hipLaunchKernelGGL(producer_ticket, dim3(G), dim3(256), 0, st, part, iters, sink, counter, G, out);
        }
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < 5; ++k) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms[form], e0, e1));
      std::vector<float> h(64);
      CK(hipMemcpy(h.data(), out, 256, hipMemcpyDeviceToHost));
      double want = 0; for (int q = 0; q < G; ++q) want += q + 5;          // column 5
      if (h[5] != (float)want) printf("  form %d WRONG: %g vs %g\n", form, h[5], want);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    printf("G=%4d workgroups, busy iters %6d: producer+finalize %.2f us/step, ticket form %.2f us/step  (%+.2f us)\n", G, iters,
           1e3 * ms[0] / (5 * NREP), 1e3 * ms[1] / (5 * NREP), 1e3 * (ms[1] - ms[0]) / (5 * NREP));
  }
  return 0;
}
