// Micro-benchmark (not part of the product): rate of a single dependent chain of v_mfma_f32_32x32x16_bf16, one wave per SIMD,
// with the A operand in AGPRs or VGPRs and the accumulator in AGPRs or VGPRs.
// build: hipcc --offload-arch=gfx950 -O3 mfma_operands.hip -o mfma_operands     run: ./mfma_operands
#include <hip/hip_runtime.h>
#include <cstdio>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>  // bit 0: A in AGPR, bit 1: accumulator in AGPR
__global__ __launch_bounds__(256, 1) void chain(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
      if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
      if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(float* out, const char* name) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chain<MODE>, dim3(256), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = 16.0 * iters;
  printf("%-40s %.2f ns per MFMA  (%.1f TFLOP/s chip-wide)\n", name, ms * 1e6 / n, 256.0 * 4 * n * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * sizeof(float));
  run<0>(out, "A in VGPR, accumulator in VGPR");
  run<1>(out, "A in AGPR, accumulator in VGPR");
  run<2>(out, "A in VGPR, accumulator in AGPR");
  run<3>(out, "A in AGPR, accumulator in AGPR");
  return 0;
}
