// Micro-benchmark (not part of the product): latency of an all-gather between M workgroups through L2 with flags,
// as a multi-CU scan would need per GEMV phase.  Each group of M WGs repeats: write 64 floats, release-increment own
// flag, spin until the partners' flags reach the sequence number, read the partners' floats.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(1024) void exch(float* buf, unsigned* flags, int M, int iters, float* out, long long* cycles) {
  const int grp = blockIdx.x / M, m = blockIdx.x % M;
  float* gb = buf + (size_t)grp * M * 64 * 2;
  unsigned* gf = flags + (size_t)grp * M * 32;  // one flag per 128 B
  __shared__ float vec[256];
  float acc = 0.f;
  const long long t0 = clock64();
  for (int it = 1; it <= iters; ++it) {
    float* slot = gb + (size_t)(it & 1) * M * 64;
    if (threadIdx.x < 64) slot[m * 64 + threadIdx.x] = (float)(it + m) + acc * 1e-9f;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&gf[m * 32], (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < M && threadIdx.x != m) {
      while (__hip_atomic_load(&gf[threadIdx.x * 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {}
    }
    __syncthreads();
    if (threadIdx.x < M * 64) vec[threadIdx.x] = __builtin_nontemporal_load(&slot[threadIdx.x]);
    __syncthreads();
    acc += vec[(threadIdx.x + it) % (M * 64)];
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) { out[blockIdx.x] = acc; cycles[blockIdx.x] = t1 - t0; }
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4, groups = argc > 2 ? atoi(argv[2]) : 64, iters = 2000;
  float *buf, *out; unsigned* flags; long long* cyc;
  hipMalloc(&buf, (size_t)groups * M * 64 * 2 * 4); hipMalloc(&flags, (size_t)groups * M * 32 * 4);
  hipMalloc(&out, groups * M * 4); hipMalloc(&cyc, groups * M * 8);
  hipMemset(flags, 0, (size_t)groups * M * 32 * 4); hipMemset(buf, 0, (size_t)groups * M * 64 * 2 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(exch, dim3(groups * M), dim3(1024), 0, 0, buf, flags, M, iters, out, cyc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("M=%d groups=%d: %.3f us per exchange (%s)\n", M, groups, ms * 1e3 / iters, hipGetErrorString(hipGetLastError()));
  return 0;
}
