// Development harness (not part of the product): conv3x3_wgrad_resident_kernel alone (no library, 20 s to build), timed
// by HIP events and by its own phase stamps -- the place to try a change of its loop before rebuilding conv.hip.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/wgrad_ablate.hip -o tools/_dbg/wgrad_ablate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../multimodal_mtrssm_amd/csrc/scan_common.h"

namespace mtrssm {
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
template <int SPLIT>
__device__ __forceinline__ void split_bf16(float x, unsigned short (&p)[SPLIT]) {
  float r = x;
#pragma unroll
  for (int s = 0; s < SPLIT; ++s) {
    const __bf16 h = (__bf16)r;
    p[s] = __builtin_bit_cast(unsigned short, h);
    r -= (float)h;
  }
}
__device__ unsigned long long* g_res_prof = nullptr;
}  // namespace mtrssm
#include "../../multimodal_mtrssm_amd/csrc/conv_wgrad_resident.h"

using namespace mtrssm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int C, int W>
static int run(int cout, int n) {
  MtrssmConvGeom g{};
  g.N = n; g.C = C; g.Hs = 64 / W; g.Ws = W; g.Cpad = C; g.KH = g.KW = 3; g.SS = g.TS = 1; g.OFFY = g.OFFX = -1;
  g.Hq = 64 / W; g.Wq = W; g.OS = 1; g.Ho = 64 / W; g.Wo = W; g.Cout = cout; g.CoutPad = cout; g.pre_act = 1; g.act = MTRSSM_ACT_ELU; g.mfma_split = 2;
  const size_t na = (size_t)n * cout * 64, nx = (size_t)n * C * 64, m = (size_t)cout * 9 * C;
  std::vector<float> ha(na), hx(nx);
  for (size_t i = 0; i < na; ++i) ha[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  for (size_t i = 0; i < nx; ++i) hx[i] = (float)((i * 40503u) % 2001) / 1000.f - 1.f;
  float *a, *x, *dwp, *part, *dbias;
  unsigned long long* prof;
  const int cogroups = cout / 64;
  int wgs = (C == 64 ? 256 : 512) / cogroups;
  if (wgs > n) wgs = n;
  const int per = (n + wgs - 1) / wgs;
  const dim3 grid((n + per - 1) / per, cogroups);
  CK(hipMalloc(&a, na * 4)); CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&dwp, m * 4)); CK(hipMalloc(&part, (size_t)grid.x * cogroups * wgres_set_floats(C) * 4)); CK(hipMalloc(&dbias, cout * 4));
  CK(hipMalloc(&prof, 64 * 8));
  CK(hipMemcpy(a, ha.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dwp, 0, m * 4)); CK(hipMemset(dbias, 0, cout * 4)); CK(hipMemset(prof, 0, 64 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_res_prof), &prof, sizeof(prof)));
  constexpr int lds_b = wgres_lds_bytes<2, C, W>();
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_resident_kernel<2, C, W>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((conv3x3_wgrad_resident_kernel<2, C, W>), grid, dim3(64 * 2 * (C / 32)), lds_b, nullptr, g, a, x, dwp, part, dbias, per);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned long long st[64];
  CK(hipMemcpy(st, prof, sizeof(st), hipMemcpyDeviceToHost));
  const double fr = per;
  printf("v=%d C=%d W=%d Cout=%d frames/wg=%d: kernel %.1f us | wg0 prologue %llu loop %llu (%.0f / frame) epilogue %llu | mid-wg loop %llu epilogue %llu cycles\n",
         0, C, W, cout, per, best * 1e3, st[33] - st[32], st[34] - st[33], (double)(st[34] - st[33]) / fr, st[35] - st[34], st[42] - st[41], st[43] - st[42]);
  (void)hipFree(a); (void)hipFree(x); (void)hipFree(dwp); (void)hipFree(part); (void)hipFree(dbias); (void)hipFree(prof);
  return 0;
}

template <int C>
static int run1x1(int cout, int n) {
  MtrssmConvGeom g{};
  g.N = n; g.C = C; g.Hs = 8; g.Ws = 8; g.Cpad = C; g.KH = g.KW = 1; g.SS = g.TS = 1;
  g.Hq = 8; g.Wq = 8; g.OS = 1; g.Ho = 8; g.Wo = 8; g.Cout = cout; g.CoutPad = cout; g.pre_act = 1; g.act = MTRSSM_ACT_ELU; g.mfma_split = 2;
  const size_t na = (size_t)n * cout * 64, nx = (size_t)n * C * 64, m = (size_t)cout * C;
  float *a, *x, *dwp, *dbias, *part;
  unsigned long long* prof;
  const int cogroups = cout / 64;
  int wgs = 256 / cogroups;
  if (wgs > n) wgs = n;
  const int per = (n + wgs - 1) / wgs;
  const dim3 grid((n + per - 1) / per, cogroups);
  CK(hipMalloc(&a, na * 4)); CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&dwp, m * 4)); CK(hipMalloc(&dbias, cout * 4)); CK(hipMalloc(&prof, 64 * 8));
  CK(hipMalloc(&part, (size_t)grid.x * cogroups * kWg1x1SetFloats * 4));
  CK(hipMemset(a, 0, na * 4)); CK(hipMemset(x, 0, nx * 4)); CK(hipMemset(dwp, 0, m * 4)); CK(hipMemset(dbias, 0, cout * 4)); CK(hipMemset(prof, 0, 64 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_res_prof), &prof, sizeof(prof)));
  constexpr int lds_b = wg1x1_lds_bytes<2, C>();
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_wgrad_staged_kernel<2, C>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((conv1x1_wgrad_staged_kernel<2, C>), grid, dim3(512), lds_b, nullptr, g, a, x, dwp, part, dbias, per);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned long long st[64];
  CK(hipMemcpy(st, prof, sizeof(st), hipMemcpyDeviceToHost));
  printf("1x1 C=%d Cout=%d frames/wg=%d: kernel %.1f us | wg0 prologue %llu loop %llu (%.0f / frame) epilogue %llu | mid-wg loop %llu epilogue %llu cycles\n",
         C, cout, per, best * 1e3, st[33] - st[32], st[34] - st[33], (double)(st[34] - st[33]) / per, st[35] - st[34], st[42] - st[41], st[43] - st[42]);
  (void)hipFree(a); (void)hipFree(x); (void)hipFree(dwp); (void)hipFree(dbias); (void)hipFree(prof);
  return 0;
}

int main() {
  if (run1x1<64>(64, 3200)) return 1;
  if (run1x1<128>(64, 3200)) return 1;
  if (run1x1<64>(128, 3200)) return 1;
  if (run<64, 8>(64, 3200)) return 1;
  if (run<64, 4>(64, 3200)) return 1;
  if (run<64, 8>(128, 3200)) return 1;
  if (run<32, 8>(64, 3200)) return 1;
  return 0;
}
