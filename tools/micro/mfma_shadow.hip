// Micro-benchmark (not part of the product): how much VALU / LDS work fits in the shadow of back-to-back independent
// v_mfma_f32_32x32x16_bf16 on one wave per SIMD (and two), gfx950.  Every variant runs ITER rounds of 9 MFMAs into 9
// accumulator tiles; between consecutive MFMAs it issues NV dependent-free v_fma_f32 and / or one ds_read_b128.
// Accumulators are VGPR operands of the asm: with "+a" the compiler copies every tile to AGPRs and back around each
// statement (32 v_accvgpr moves per MFMA), which measures the copies.
// Measured (profiles/round2_notes.md): 32.0 cycles per MFMA alone; + 7 VALU per slot 40.9; 7 VALU alone 36.0 (5.1 cycles per
// instruction from one wave; two waves per SIMD issue twice as many in the same time); ds_read_b128 alone 16 per read.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shadow.hip -o tools/_dbg/mfma_shadow
#include <hip/hip_runtime.h>

#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NV, bool LDS, bool MFMA, bool AGPR>
__global__ __launch_bounds__(512) void shadow_kernel(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 1.f + lane * 0.001f + k;
  const float m = 0.999f, c = 0.001f;
  const unsigned addr = (unsigned)(size_t)lds + (unsigned)((threadIdx.x & 255) * 144 % 32000 & ~15u);
  u32x4 ld = a;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (MFMA) {
        if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
      }
      if (LDS) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(addr));
#pragma unroll
      for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(m), "v"(c));
    }
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) s += acc[t][0] + acc[t][15];
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k];
  s += __builtin_bit_cast(float, ld.x);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, bool LDS, bool MFMA, bool AGPR>
static int run(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
  const int iters = 2000;
  hipLaunchKernelGGL((shadow_kernel<NV, LDS, MFMA, AGPR>), dim3(256), dim3(threads), 0, nullptr, out, cyc, iters);
  hipLaunchKernelGGL((shadow_kernel<NV, LDS, MFMA, AGPR>), dim3(256), dim3(threads), 0, nullptr, out, cyc, iters);
  CK(hipDeviceSynchronize());
  unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-44s waves/SIMD %d: %7.1f cycles per round of 9 MFMA slots (%.1f per slot)\n", name, threads / 256, (double)h / iters, (double)h / iters / 9);
  (void)hipFree(out); (void)hipFree(cyc);
  return 0;
}

int main() {
  for (int threads : {256, 512}) {
    if (run<0, false, true, false>("MFMA only", threads)) return 1;
    if (run<4, false, true, false>("MFMA + 4 VALU per slot", threads)) return 1;
    if (run<7, false, true, false>("MFMA + 7 VALU per slot", threads)) return 1;
    if (run<12, false, true, false>("MFMA + 12 VALU per slot", threads)) return 1;
    if (run<7, false, false, false>("7 VALU per slot, no MFMA", threads)) return 1;
    if (run<12, false, false, false>("12 VALU per slot, no MFMA", threads)) return 1;
    if (run<0, true, true, false>("MFMA + 1 ds_read_b128 per slot", threads)) return 1;
    if (run<4, true, true, false>("MFMA + 1 ds_read_b128 + 4 VALU per slot", threads)) return 1;
    if (run<0, true, false, false>("1 ds_read_b128 per slot, no MFMA", threads)) return 1;
  }
  return 0;
}
