// Dense fp32 GEMMs of the rollout path on the fp32 MFMA (v_mfma_f32_32x32x2_f32, gfx950): the Linear layers of the
// encoder / decoder stacks (cnn.Encoder.head, cnn.Decoder.stem), init_proj and the prior head of the initial state, the
// three recurrence-independent projections hoisted out of the scan (networks.py:165-166, 79-82) and every weight gradient
// of the scan ([out, B*T] x [B*T, in]).  Bitwise a k-ordered fp32 fma chain per output element
// (MI355X_MICROARCH.md, "FP32-input MFMA"), i.e. the numerics of the reference's fp32 nn.Linear, with the elementwise
// neighbours of each GEMM fused in: activation of an operand while it is staged, bias, output activation, the
// multiplication by act'(z) of the data gradient, bias gradients as column sums of the staged operand, and accumulation
// straight into the flat gradient buffer (optim.FlatParameters) instead of a temporary + add.
//
// One kernel, three operand layouts (which index of each operand is contiguous in memory):
//   C[i][j] (+)= epi( sum_r A'(i, r) * B'(j, r) ),  i < M, j < N, r < R
//   A' : a_rmajor ? A[r * lda + i] : A[i * lda + r]        B' : b_rmajor ? B[r * ldb + j] : B[j * ldb + r]
//     forward  Y = X W^T        : A = X [M][K] (i-major rows),  B = W [N][K]                     (neither r-major)
//     data     dX = dY W        : A = dY [M][N],                B = W [N][K] read as B'(k, n)    (b_rmajor)
//     weight   dW = dY^T X      : A = dY read as A'(n, m),      B = X read as B'(k, m)           (both r-major)
// Workgroup tile 64 x 64, four waves of one 32 x 32 accumulator each, reduction in steps of 32 through double-buffered LDS
// (operand rows padded to 33 floats, r-major images to 96: conflict-free ds_read_b32 for the two k halves of a wave).
#include <cstdlib>

#include "scan_common.h"

namespace mtrssm {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);
int device_cu_count();

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kTile = 64;   // rows of A' / B' per workgroup tile
constexpr int kStep = 32;   // reduction extent per LDS stage
constexpr int kLdRow = kStep + 1;   // [row][r] image: 33 floats per row
constexpr int kLdT = kTile + 32;    // [r][row] image: 96 floats per r
constexpr int kImage = kTile * kLdRow > kStep * kLdT ? kTile * kLdRow : kStep * kLdT;  // floats per operand image

struct GemmArgs {
  const float* A; const float* B; float* C;
  const float* bias;      // [N], added to every row (forward) or null
  const float* zgrad;     // [M][ldz] pre-activation tensor: C *= act'(z) (data gradient) or null
  float* colsum;          // [M] (+)= sum_r A'(i, r) (bias gradient of the weight form; atomics) or null
  int M, N, R, lda, ldb, ldc, ldz;
  int a_rmajor, b_rmajor;
  int act_a, act_b;       // activation applied to A / B elements while staging (0 = none)
  int act_out;            // activation of the output
  int act_z;              // activation whose derivative multiplies the output (with zgrad)
  int accumulate;         // C += instead of C =
  int splits;             // reduction split over gridDim.z (atomic accumulation when > 1)
  int* tickets;           // per output tile: arrivals of the reduction slices (only for a split reduction WITH an epilogue)
};

// Branch-free activations for the staging path (uniform scalar branch on `act`, no libm call per element: an inlined
// expm1f / tanhf per staged element cost more VALU time than the MFMAs it feeds).  ELU: v_exp_f32 - 1 on the negative side
// (absolute error < 1.2e-7, as in the conv kernels); Tanh: 1 - 2 / (exp(2x) + 1) (absolute error ~1e-7, saturates cleanly).
__device__ __forceinline__ float4 act4(float4 q, int act) {
  if (act == MTRSSM_ACT_ELU) {
    q.x = q.x > 0.f ? q.x : __expf(q.x) - 1.f; q.y = q.y > 0.f ? q.y : __expf(q.y) - 1.f;
    q.z = q.z > 0.f ? q.z : __expf(q.z) - 1.f; q.w = q.w > 0.f ? q.w : __expf(q.w) - 1.f;
  } else if (act == MTRSSM_ACT_RELU) {
    q.x = fmaxf(q.x, 0.f); q.y = fmaxf(q.y, 0.f); q.z = fmaxf(q.z, 0.f); q.w = fmaxf(q.w, 0.f);
  } else if (act == MTRSSM_ACT_TANH) {
    q.x = 1.f - 2.f / (__expf(2.f * q.x) + 1.f); q.y = 1.f - 2.f / (__expf(2.f * q.y) + 1.f);
    q.z = 1.f - 2.f / (__expf(2.f * q.z) + 1.f); q.w = 1.f - 2.f / (__expf(2.f * q.w) + 1.f);
  }
  return q;
}
__device__ __forceinline__ float act_grad_from_in(float z, int act) {
  if (act == MTRSSM_ACT_ELU) return z > 0.f ? 1.f : __expf(z);
  if (act == MTRSSM_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == MTRSSM_ACT_TANH) { const float t = 1.f - 2.f / (__expf(2.f * z) + 1.f); return 1.f - t * t; }
  return 1.f;
}

// one operand tile (64 rows x 32 r) from global memory into registers: 2 x float4 per thread.  Loads are branch-free per
// lane -- clamped addresses, masked afterwards: a load inside a divergent branch is waited for at the join, which serialises
// the prefetch (the first version of this kernel ran at 13-20 TFLOP/s for that reason).
struct Stage { float4 v[2]; };

// y: index along the strided axis (limit ylim), x: first of 4 consecutive indices along the contiguous axis (limit xlim).
// Returns the RAW quad from clamped addresses; mask_quad() zeroes the out-of-range lanes later, when the tile goes to LDS
// (a select right behind the load would make the compiler wait for the load on the spot).
__device__ __forceinline__ float4 load_quad(const float* __restrict__ P, int ld, int y, int ylim, int x, int xlim, bool vec) {
  const int yc = y < ylim ? y : ylim - 1;
  const float* row = P + (size_t)yc * ld;
  float4 q;
  if (vec) {
    // uniform branch: rows are 16-byte aligned and ld % 4 == 0.  Clamp to the LAST quad that still overlaps [0, xlim): its
    // tail may pass xlim by up to 3 elements, which stay inside the (16-byte granular) row of the parent tensor; quads past
    // it are fully masked later.  (Clamping against ld instead walks off the end of a column-offset view's last row.)
    const int xq = (xlim - 1) & ~3;
    const int xc = x < xq ? x : xq;
    q = *reinterpret_cast<const float4*>(row + xc);
  } else {
    const int last = xlim - 1;
    q.x = row[x < last ? x : last];
    q.y = row[x + 1 < last ? x + 1 : last];
    q.z = row[x + 2 < last ? x + 2 : last];
    q.w = row[x + 3 < last ? x + 3 : last];
  }
  return q;
}
__device__ __forceinline__ float4 mask_quad(float4 q, int y, int ylim, int x, int xlim) {
  const bool oky = y < ylim;
  q.x = (oky && x < xlim) ? q.x : 0.f;
  q.y = (oky && x + 1 < xlim) ? q.y : 0.f;
  q.z = (oky && x + 2 < xlim) ? q.z : 0.f;
  q.w = (oky && x + 3 < xlim) ? q.w : 0.f;
  return q;
}

template <bool RMAJOR>
__device__ __forceinline__ void load_tile(Stage& st, const float* __restrict__ P, int ld, int rows, int row0, int r0, int r_end, bool vec) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int f = tid + 256 * j;
    if (RMAJOR) st.v[j] = load_quad(P, ld, r0 + (f >> 4), r_end, row0 + 4 * (f & 15), rows, vec);  // memory [r][row]
    else st.v[j] = load_quad(P, ld, row0 + (f >> 3), rows, r0 + 4 * (f & 7), r_end, vec);          // memory [row][r]
  }
}

template <bool RMAJOR>
__device__ __forceinline__ void store_tile(const Stage& st, float* img, int act, int rows, int row0, int r0, int r_end) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int f = tid + 256 * j;
    float4 q = RMAJOR ? mask_quad(st.v[j], r0 + (f >> 4), r_end, row0 + 4 * (f & 15), rows)
                      : mask_quad(st.v[j], row0 + (f >> 3), rows, r0 + 4 * (f & 7), r_end);
    q = act4(q, act);  // act(0) = 0 for every supported activation
    if (RMAJOR) {
      *reinterpret_cast<float4*>(img + (f >> 4) * kLdT + 4 * (f & 15)) = q;
    } else {
      float* d = img + (f >> 3) * kLdRow + 4 * (f & 7);
      d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
    }
  }
}

// full tiles (FAST instantiation): nothing to mask
template <bool RMAJOR>
__device__ __forceinline__ void store_tile_full(const Stage& st, float* img, int act) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int f = tid + 256 * j;
    const float4 q = act4(st.v[j], act);
    if (RMAJOR) {
      *reinterpret_cast<float4*>(img + (f >> 4) * kLdT + 4 * (f & 15)) = q;
    } else {
      float* d = img + (f >> 3) * kLdRow + 4 * (f & 7);
      d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
    }
  }
}

template <bool RMAJOR>
__device__ __forceinline__ float image_at(const float* img, int row, int r) {
  return RMAJOR ? img[r * kLdT + row] : img[row * kLdRow + r];
}

// FAST: every tile of the launch is full (M, N multiples of 64, every reduction slice a whole number of 32-steps) and both
// operands take 16-byte loads: per-thread pointers advance by one k-step per request and nothing is clamped or masked.  Per
// k-step the general form spent ~3000 cycles of address arithmetic, clamps, masks and their waits around 1100 cycles of
// MFMAs (stamped, one workgroup per CU) -- the MFMA pipe idled 3/4 of the time on the 3200 x 4096 x 256 head GEMMs.
template <bool AR, bool BR, bool FAST>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs& g, const int bx, const int by, const int bz, const int tiles_x) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][kImage];  // [buffer][operand][image]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i0 = by * kTile, j0 = bx * kTile;
  // this workgroup's share of the reduction (multiples of kStep)
  const int steps = (g.R + kStep - 1) / kStep;
  const int per = (steps + g.splits - 1) / g.splits;
  const int s_lo = bz * per, s_hi = min(steps, s_lo + per);
  if (s_lo >= s_hi) return;
  const int r_end = min(g.R, s_hi * kStep);
  const bool vec_a = ((g.lda & 3) == 0) && (((uintptr_t)g.A & 15) == 0);
  const bool vec_b = ((g.ldb & 3) == 0) && (((uintptr_t)g.B & 15) == 0);
  const bool want_colsum = g.colsum && bx == 0;
  float csum = 0.f;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // Software pipeline: the operand tiles of steps s+1 and s+2 are in flight in registers (two stages) while step s is
  // multiplied out of LDS, so a global load has two whole steps to land; with one stage the ~1.5 us load latency, not the
  // 0.43 us of MFMAs, set the step time on grids of < 1 workgroup per SIMD (measured: 13-16 TFLOP/s).
  Stage sa[2], sb[2];
  // FAST: this thread's two quads of each operand at the NEXT step to request (requests come in step order)
  const float4* pa[2];
  const float4* pb[2];
  size_t bump_a = 0, bump_b = 0;  // float4s per k-step
  if (FAST) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f = tid + 256 * j;
      pa[j] = reinterpret_cast<const float4*>(AR ? g.A + (size_t)(s_lo * kStep + (f >> 4)) * g.lda + i0 + 4 * (f & 15)
                                                 : g.A + (size_t)(i0 + (f >> 3)) * g.lda + s_lo * kStep + 4 * (f & 7));
      pb[j] = reinterpret_cast<const float4*>(BR ? g.B + (size_t)(s_lo * kStep + (f >> 4)) * g.ldb + j0 + 4 * (f & 15)
                                                 : g.B + (size_t)(j0 + (f >> 3)) * g.ldb + s_lo * kStep + 4 * (f & 7));
    }
    bump_a = AR ? (size_t)kStep * g.lda / 4 : kStep / 4;
    bump_b = BR ? (size_t)kStep * g.ldb / 4 : kStep / 4;
  }
  auto issue = [&](int which, int s) {
    if (s < s_hi) {
      if (FAST) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          sa[which].v[j] = *pa[j];
          sb[which].v[j] = *pb[j];
          pa[j] += bump_a;
          pb[j] += bump_b;
        }
      } else {
        load_tile<AR>(sa[which], g.A, g.lda, g.M, i0, s * kStep, r_end, vec_a);
        load_tile<BR>(sb[which], g.B, g.ldb, g.N, j0, s * kStep, r_end, vec_b);
      }
    }
  };
  auto to_lds = [&](int which, int s, int buf) {
    if (FAST) {
      store_tile_full<AR>(sa[which], lds[buf][0], g.act_a);
      store_tile_full<BR>(sb[which], lds[buf][1], g.act_b);
    } else {
      store_tile<AR>(sa[which], lds[buf][0], g.act_a, g.M, i0, s * kStep, r_end);
      store_tile<BR>(sb[which], lds[buf][1], g.act_b, g.N, j0, s * kStep, r_end);
    }
  };
  const int ar = wr * 32 + (lane & 31), bc = wc * 32 + (lane & 31), kh = lane >> 5;
  auto step = [&](int which, int s, int cur) {  // `which`: the register stage that holds step s + 1
    if (s + 1 < s_hi) to_lds(which, s + 1, cur ^ 1);
    issue(which, s + 3);
    const float* ia = lds[cur][0];
    const float* ib = lds[cur][1];
    float av[kStep / 2], bv[kStep / 2];  // all operand values of the step first: the LDS reads pipeline under the MFMAs
#pragma unroll
    for (int k = 0; k < kStep / 2; ++k) {
      av[k] = image_at<AR>(ia, ar, 2 * k + kh);
      bv[k] = image_at<BR>(ib, bc, 2 * k + kh);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMA chain (the scheduler re-sinks them pair by pair otherwise)
#pragma unroll
    for (int k = 0; k < kStep / 2; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv[k], acc, 0, 0, 0);
    if (want_colsum && tid < kTile) {
#pragma unroll 8
      for (int k = 0; k < kStep; ++k) csum += image_at<AR>(ia, tid, k);
    }
    lds_barrier();  // LDS only: a __syncthreads also drains vmcnt, i.e. waits for the prefetched tiles at every k-step
  };
  issue(0, s_lo);
  to_lds(0, s_lo, 0);
  issue(0, s_lo + 1);
  issue(1, s_lo + 2);
  lds_barrier();  // LDS only: a __syncthreads also drains vmcnt, i.e. waits for the prefetched tiles at every k-step
  for (int s = s_lo; s < s_hi; s += 2) {
    step(0, s, 0);
    if (s + 1 < s_hi) step(1, s + 1, 1);
  }

  // epilogue: C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
  // All loads first (clamped, branch-free), then the arithmetic, then the stores.
  const int j = j0 + wc * 32 + (lane & 31);
  const bool okj = j < g.N;
  const int jc = okj ? j : g.N - 1;
  const bool atomic = g.splits > 1;
  const int ibase = i0 + wr * 32 + 4 * (lane >> 5);
  // A split reduction whose output needs an epilogue (bias / act' of the data gradient): the slices add their raw sums
  // atomically into the zeroed C, take a ticket per tile, and the LAST slice to arrive reads the finished sums back
  // (sc1 loads: the atomics were performed memory-side), applies the epilogue and stores.  Without an epilogue the atomics
  // are the result (bias by slice 0).
  const bool finalize = atomic && g.tickets != nullptr;
  if (atomic) {
    const float bias0 = (g.bias && !finalize && bz == 0) ? g.bias[jc] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = ibase + (reg & 3) + 8 * (reg >> 2);
      if (i < g.M && okj) atomicAdd(g.C + (size_t)i * g.ldc + j, acc[reg] + bias0);
    }
    if (!finalize) {
      if (want_colsum && tid < kTile && i0 + tid < g.M) atomicAdd(g.colsum + i0 + tid, csum);
      return;
    }
    __shared__ int last_flag;
    __threadfence();
    __syncthreads();
    if (tid == 0) last_flag = atomicAdd(g.tickets + by * tiles_x + bx, 1) == g.splits - 1;
    __syncthreads();
    if (!last_flag) {
      if (want_colsum && tid < kTile && i0 + tid < g.M) atomicAdd(g.colsum + i0 + tid, csum);
      return;
    }
    __threadfence();
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = ibase + (reg & 3) + 8 * (reg >> 2);
      acc[reg] = __hip_atomic_load(g.C + (size_t)(i < g.M ? i : g.M - 1) * g.ldc + jc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const float bias = g.bias ? g.bias[jc] : 0.f;
  float zv[16];
  if (g.zgrad) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = ibase + (reg & 3) + 8 * (reg >> 2);
      zv[reg] = g.zgrad[(size_t)(i < g.M ? i : g.M - 1) * g.ldz + jc];
    }
  }
  float cv[16];
  if (g.accumulate && !atomic) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = ibase + (reg & 3) + 8 * (reg >> 2);
      cv[reg] = g.C[(size_t)(i < g.M ? i : g.M - 1) * g.ldc + jc];
    }
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int i = ibase + (reg & 3) + 8 * (reg >> 2);
    float v = acc[reg] + bias;
    if (g.act_out) v = act_fwd(v, g.act_out);
    if (g.zgrad) v *= act_grad_from_in(zv[reg], g.act_z);
    if (g.accumulate && !atomic) v += cv[reg];
    if (i < g.M && okj) g.C[(size_t)i * g.ldc + j] = v;
  }
  if (want_colsum && tid < kTile && i0 + tid < g.M) atomicAdd(g.colsum + i0 + tid, csum);
}

template <bool AR, bool BR, bool FAST>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  gemm_f32_body<AR, BR, FAST>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// Several INDEPENDENT problems of one layout in one launch (mtrssm_gemm_group): the scan's ten to fifteen weight-gradient GEMMs
// ([out, B*T] x [B*T, in], 0.02-0.8 GFLOP each) are latency-bound launches of 15-35 us one by one -- a few k-steps behind a
// load round trip each, on a fraction of the chip -- and independent of each other: side by side they cost one such latency.
// A 1-D grid; first[p] = first workgroup of problem p, whose grid is (tiles_x, tiles_y, splits) linearised x-fastest.
constexpr int kGroupMax = 24;
struct GemmGroup {
  int count;
  int first[kGroupMax + 1];
  GemmArgs g[kGroupMax];
};
template <bool AR, bool BR, bool FAST>
__global__ __launch_bounds__(256) void gemm_f32_group_kernel(const GemmGroup grp) {
  int p = 0;
  while (p + 1 < grp.count && (int)blockIdx.x >= grp.first[p + 1]) ++p;
  const GemmArgs& g = grp.g[p];
  const int local = blockIdx.x - grp.first[p];
  const int tx = (g.N + kTile - 1) / kTile, ty = (g.M + kTile - 1) / kTile;
  gemm_f32_body<AR, BR, FAST>(g, local % tx, (local / tx) % ty, local / (tx * ty), tx);
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 variant (MtrssmGemm.mfma_split = 2) for the large GEMMs inside the conv stacks (encoder head, decoder stem:
// 3200 x 4096 x 256 and its gradients).  Same contract and epilogues; every fp32 operand value becomes two bf16 pieces while
// it is staged (after the fused activation) and a k-block of 16 is three v_mfma_f32_32x32x16_bf16 (hi*lo + lo*hi + hi*hi,
// fp32 accumulation): the arithmetic of the conv kernels' default mode (conv_split.h), 16 significant bits per operand.
// Both operand layouts are staged as "8 consecutive r of one row" per thread (two float4 loads of an r-contiguous operand,
// eight 4-byte loads -- coalesced across the rows of neighbouring lanes -- of an r-major one), so the LDS images are always
// [row][32 r] bf16 with an 80-byte pitch (an odd number of 16-byte slots: conflict-free ds_read_b128 fragments) and a
// k-step costs a wave 12 ds_read_b128 + 12 MFMAs instead of 32 ds_read_b32 + 16 fp32 MFMAs of 64 cycles.
// Workgroup tile 128 x 64, four waves of 64 x 32 (two accumulators).
// ------------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
constexpr int kSM = 128, kSN = 64;
constexpr int kSPitch = kStep * 2 + 16;
constexpr int kSImgA = kSM * kSPitch, kSImgB = kSN * kSPitch;  // bytes of one piece
constexpr int kSBuf = 2 * kSImgA + 2 * kSImgB;                 // one buffer: A hi, A lo, B hi, B lo

struct SItem { float v[8]; };

// item idx of an operand tile with TROWS rows: 8 consecutive r of one row, raw (clamped addresses; masked in store_item)
template <bool RMAJOR, int TROWS>
__device__ __forceinline__ void load_item(SItem& it, const float* __restrict__ P, int ld, int rows, int row0, int r0, int r_end, int idx,
                                          bool vec) {
  if (RMAJOR) {  // memory [r][row]
    const int row = idx % TROWS, kg = idx / TROWS;
    const int x = row0 + row < rows ? row0 + row : rows - 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int y = r0 + kg * 8 + u;
      it.v[u] = P[(size_t)(y < r_end ? y : r_end - 1) * ld + x];
    }
  } else {  // memory [row][r]
    const int row = idx >> 2, kg = idx & 3;
    const float4 q0 = load_quad(P, ld, row0 + row, rows, r0 + kg * 8, r_end, vec);
    const float4 q1 = load_quad(P, ld, row0 + row, rows, r0 + kg * 8 + 4, r_end, vec);
    it.v[0] = q0.x; it.v[1] = q0.y; it.v[2] = q0.z; it.v[3] = q0.w;
    it.v[4] = q1.x; it.v[5] = q1.y; it.v[6] = q1.z; it.v[7] = q1.w;
  }
}
template <bool RMAJOR, int TROWS>
__device__ __forceinline__ void store_item(const SItem& it, unsigned char* img, int img_bytes, int act, int rows, int row0, int r0, int r_end,
                                           int idx) {
  const int row = RMAJOR ? idx % TROWS : idx >> 2, kg = RMAJOR ? idx / TROWS : idx & 3;
  const bool okrow = row0 + row < rows;
  float x[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = (okrow && r0 + kg * 8 + u < r_end) ? it.v[u] : 0.f;
  // ONE uniform branch per item (per element it became a branch chain per value: 580 branches in the step, 2x slower than
  // the fp32 kernel); act(0) = 0 for every supported activation
  if (act == MTRSSM_ACT_ELU) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = x[u] > 0.f ? x[u] : __expf(x[u]) - 1.f;
  } else if (act == MTRSSM_ACT_RELU) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = fmaxf(x[u], 0.f);
  } else if (act == MTRSSM_ACT_TANH) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = 1.f - 2.f / (__expf(2.f * x[u]) + 1.f);
  }
  u16x8 hi, lo;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const __bf16 h = (__bf16)x[u];
    const __bf16 l = (__bf16)(x[u] - (float)h);
    hi[u] = __builtin_bit_cast(unsigned short, h);
    lo[u] = __builtin_bit_cast(unsigned short, l);
  }
  unsigned char* d = img + row * kSPitch + kg * 16;
  *reinterpret_cast<u16x8*>(d) = hi;
  *reinterpret_cast<u16x8*>(d + img_bytes) = lo;
}

template <bool AR, bool BR>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kSBuf];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, il = lane & 31, kl = lane >> 5;
  const int i0 = blockIdx.y * kSM, j0 = blockIdx.x * kSN;
  const int steps = (g.R + kStep - 1) / kStep;
  const int per = (steps + g.splits - 1) / g.splits;
  const int s_lo = blockIdx.z * per, s_hi = min(steps, s_lo + per);
  if (s_lo >= s_hi) return;
  const int r_end = min(g.R, s_hi * kStep);
  const bool vec_a = ((g.lda & 3) == 0) && (((uintptr_t)g.A & 15) == 0);
  const bool vec_b = ((g.ldb & 3) == 0) && (((uintptr_t)g.B & 15) == 0);
  const bool want_colsum = g.colsum && blockIdx.x == 0;
  float csum = 0.f;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // two register stages: the tiles of steps s + 1 and s + 2 are in flight while step s is multiplied out of LDS
  SItem sa[2][2], sb[2];
  auto issue = [&](int which, int s) __attribute__((always_inline)) {
    if (s < s_hi) {
      load_item<AR, kSM>(sa[which][0], g.A, g.lda, g.M, i0, s * kStep, r_end, tid, vec_a);
      load_item<AR, kSM>(sa[which][1], g.A, g.lda, g.M, i0, s * kStep, r_end, tid + 256, vec_a);
      load_item<BR, kSN>(sb[which], g.B, g.ldb, g.N, j0, s * kStep, r_end, tid, vec_b);
    }
  };
  auto to_lds = [&](int which, int s, int buf) __attribute__((always_inline)) {
    unsigned char* base = lds + buf * kSBuf;
    store_item<AR, kSM>(sa[which][0], base, kSImgA, g.act_a, g.M, i0, s * kStep, r_end, tid);
    store_item<AR, kSM>(sa[which][1], base, kSImgA, g.act_a, g.M, i0, s * kStep, r_end, tid + 256);
    store_item<BR, kSN>(sb[which], base + 2 * kSImgA, kSImgB, g.act_b, g.N, j0, s * kStep, r_end, tid);
  };
  auto step = [&](int which, int s, int cur) __attribute__((always_inline)) {  // `which`: the register stage that holds step s + 1
    if (s + 1 < s_hi) to_lds(which, s + 1, cur ^ 1);
    issue(which, s + 3);
    const unsigned char* ia = lds + cur * kSBuf + (wr * 64 + il) * kSPitch + kl * 16;
    const unsigned char* ib = lds + cur * kSBuf + 2 * kSImgA + (wc * 32 + il) * kSPitch + kl * 16;
    bf16x8 fa[2][2][2], fb[2][2];  // [k-block][row tile][piece], [k-block][piece]: all reads first, they pipeline under the MFMAs
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      fb[kb][0] = *reinterpret_cast<const bf16x8*>(ib + kb * 32);
      fb[kb][1] = *reinterpret_cast<const bf16x8*>(ib + kb * 32 + kSImgB);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[kb][t][0] = *reinterpret_cast<const bf16x8*>(ia + t * 32 * kSPitch + kb * 32);
        fa[kb][t][1] = *reinterpret_cast<const bf16x8*>(ia + t * 32 * kSPitch + kb * 32 + kSImgA);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][t][0], fb[kb][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][t][1], fb[kb][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][t][0], fb[kb][0], acc[t], 0, 0, 0);
      }
    if (want_colsum && tid < kSM) {  // bias gradient: the staged A' row (hi + lo = the fp32 value to 2^-17)
      const unsigned char* row = lds + cur * kSBuf + tid * kSPitch;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u16x8 h = *reinterpret_cast<const u16x8*>(row + q * 16), l = *reinterpret_cast<const u16x8*>(row + q * 16 + kSImgA);
#pragma unroll
        for (int u = 0; u < 8; ++u) csum += __uint_as_float((unsigned)h[u] << 16) + __uint_as_float((unsigned)l[u] << 16);
      }
    }
    lds_barrier();  // LDS only: a __syncthreads also drains vmcnt, i.e. waits for the prefetched tiles at every k-step
  };
  issue(0, s_lo);
  to_lds(0, s_lo, 0);
  issue(0, s_lo + 1);
  issue(1, s_lo + 2);
  lds_barrier();  // LDS only: a __syncthreads also drains vmcnt, i.e. waits for the prefetched tiles at every k-step
  for (int s = s_lo; s < s_hi; s += 2) {
    step(0, s, 0);
    if (s + 1 < s_hi) step(1, s + 1, 1);
  }

  // epilogue: as gemm_f32_kernel, once per 32 x 32 accumulator of the wave
  const int j = j0 + wc * 32 + il;
  const bool okj = j < g.N;
  const int jc = okj ? j : g.N - 1;
  const bool atomic = g.splits > 1;
  const bool finalize = atomic && g.tickets != nullptr;
  const bool colsum_now = want_colsum && tid < kSM && i0 + tid < g.M;
  if (atomic) {
    const float bias0 = (g.bias && !finalize && blockIdx.z == 0) ? g.bias[jc] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ibase = i0 + wr * 64 + t * 32 + 4 * kl;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = ibase + (reg & 3) + 8 * (reg >> 2);
        if (i < g.M && okj) atomicAdd(g.C + (size_t)i * g.ldc + j, acc[t][reg] + bias0);
      }
    }
    if (!finalize) {
      if (colsum_now) atomicAdd(g.colsum + i0 + tid, csum);
      return;
    }
    __shared__ int last_flag;
    __threadfence();
    __syncthreads();
    if (tid == 0) last_flag = atomicAdd(g.tickets + blockIdx.y * gridDim.x + blockIdx.x, 1) == g.splits - 1;
    __syncthreads();
    if (!last_flag) {
      if (colsum_now) atomicAdd(g.colsum + i0 + tid, csum);
      return;
    }
    __threadfence();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ibase = i0 + wr * 64 + t * 32 + 4 * kl;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = ibase + (reg & 3) + 8 * (reg >> 2);
        acc[t][reg] = __hip_atomic_load(g.C + (size_t)(i < g.M ? i : g.M - 1) * g.ldc + jc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  const float bias = g.bias ? g.bias[jc] : 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ibase = i0 + wr * 64 + t * 32 + 4 * kl;
    float zv[16], cv[16];
    if (g.zgrad) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = ibase + (reg & 3) + 8 * (reg >> 2);
        zv[reg] = g.zgrad[(size_t)(i < g.M ? i : g.M - 1) * g.ldz + jc];
      }
    }
    if (g.accumulate && !atomic) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = ibase + (reg & 3) + 8 * (reg >> 2);
        cv[reg] = g.C[(size_t)(i < g.M ? i : g.M - 1) * g.ldc + jc];
      }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = ibase + (reg & 3) + 8 * (reg >> 2);
      float v = acc[t][reg] + bias;
      if (g.act_out) v = act_fwd(v, g.act_out);
      if (g.zgrad) v *= act_grad_from_in(zv[reg], g.act_z);
      if (g.accumulate && !atomic) v += cv[reg];
      if (i < g.M && okj) g.C[(size_t)i * g.ldc + j] = v;
    }
  }
  if (colsum_now) atomicAdd(g.colsum + i0 + tid, csum);
}

}  // namespace mtrssm
#include "gemm_tile.h"
namespace mtrssm {

// 0 = the tile kernel (gemm_tile.h) does not take this problem; else its column tile (128 or 64)
static int tile_kernel_columns(const MtrssmGemm* p) {
  if (p->mfma_split != 2 && p->mfma_split != 3) return 0;
  if (p->M % kGM || p->R % kGK || p->R < kGK) return 0;
  if ((!p->a_rmajor && ((p->lda & 3) || ((uintptr_t)p->A & 15))) || (!p->b_rmajor && ((p->ldb & 3) || ((uintptr_t)p->B & 15)))) return 0;
  // a short reduction (<= 16 k-steps) is prologue + epilogue: 64-column tiles put two workgroups on a CU, one's request
  // latency and output stores under the other's steps (3200 x 4096 x 256: 101 -> 54 us; x 64: 48 -> 27)
  // two pieces and a wide output: 64-column tiles too (61 KB of LDS: two workgroups per CU; 3200 x 4096 x 1024: 204 -> 142 us,
  // 3200 x 1024 x 1024: 47 -> 39); narrow outputs (N <= 512) keep the 128-column tile's operand reuse
  if (p->N % 128 == 0 && p->R > 512 && !(p->mfma_split == 2 && p->N >= 1024)) return 128;
  if (p->N % 64 == 0) return 64;
  return 0;
}

// Validation, split choice and the clears a problem needs before its kernel; fills `g`, the grid and whether the FAST
// (full-tile) instantiation applies.
static int gemm_prepare(const MtrssmGemm* p, GemmArgs& g, dim3& grid, bool& fast, hipStream_t stream, int fill_target = 768) {
  if (!p || !p->A || !p->B || !p->C || p->M <= 0 || p->N <= 0 || p->R <= 0) {
    set_error("gemm: null operand or non-positive extent");
    return MTRSSM_EINVAL;
  }
  const int min_lda = p->a_rmajor ? p->M : p->R, min_ldb = p->b_rmajor ? p->N : p->R;
  if (p->lda < min_lda || p->ldb < min_ldb || p->ldc < p->N || (p->zgrad && p->ldz < p->N)) {
    set_error("gemm: a leading dimension is smaller than its row (lda %d ldb %d ldc %d ldz %d for M %d N %d R %d)", p->lda, p->ldb, p->ldc,
              p->ldz, p->M, p->N, p->R);
    return MTRSSM_EINVAL;
  }
  for (int a : {p->act_a, p->act_b, p->act_out, p->act_z})
    if (a < MTRSSM_ACT_IDENTITY || a > MTRSSM_ACT_TANH) {
      set_error("gemm: unknown activation id %d", a);
      return MTRSSM_EINVAL;
    }
  g.A = p->A; g.B = p->B; g.C = p->C; g.bias = p->bias; g.zgrad = p->zgrad; g.colsum = p->colsum;
  g.M = p->M; g.N = p->N; g.R = p->R; g.lda = p->lda; g.ldb = p->ldb; g.ldc = p->ldc; g.ldz = p->ldz;
  g.a_rmajor = p->a_rmajor; g.b_rmajor = p->b_rmajor; g.act_a = p->act_a; g.act_b = p->act_b; g.act_out = p->act_out;
  g.act_z = p->act_z; g.accumulate = p->accumulate;
  if (p->mfma_split != 0 && p->mfma_split != 2 && p->mfma_split != 3) {
    set_error("gemm: mfma_split must be 0 (fp32 MFMA), 2 or 3 (bf16 pieces per operand), got %d", p->mfma_split);
    return MTRSSM_EINVAL;
  }
  const int tile_cols = tile_kernel_columns(p);
  const bool split_mfma = p->mfma_split == 2 && !tile_cols;   // the older 128 x 64 split kernel: ragged shapes at two pieces
  const int tile_m = tile_cols ? kGM : (split_mfma ? kSM : kTile), tile_n = tile_cols ? tile_cols : (split_mfma ? kSN : kTile);
  const int ti = (p->M + tile_m - 1) / tile_m, tj = (p->N + tile_n - 1) / tile_n;
  int splits = p->split_r;
  const int steps = (p->R + kStep - 1) / kStep;
  const bool plain_epilogue = !p->act_out && !p->zgrad;
  const bool dense_c = p->ldc == p->N;
  const bool can_finalize = p->tickets && p->n_tickets >= ti * tj && !p->accumulate && dense_c;
  if (splits <= 0) {
    // Automatic: a workgroup's time is (its reduction steps) x (a load round trip), so a grid that does not fill the 256 CUs
    // three deep is cut along the reduction, down to 4 steps per slice.  Slices meet by fp32 atomics: into the running
    // target when accumulating; otherwise into C zeroed here first (dense C only); an output epilogue (act' of a data
    // gradient) then needs the ticket words for its last-arriver pass.
    splits = 1;
    if ((plain_epilogue && (p->accumulate || dense_c)) || (!plain_epilogue && can_finalize))
      // (64 x 64-tile kernels: a grid of >= 128 tiles is not cut -- its workgroups already overlap their load round trips, and a
      //  cut costs a clear launch of C plus the atomics: 3200 x 200 x 256 19.8 -> 18.7 us per launch, one clear less)
      while (ti * tj * splits < (tile_cols ? 200 : fill_target) && steps / (splits * 2) >= 4 && (tile_cols || ti * tj < 128)) splits *= 2;
  }
  while (splits > 1 && (splits - 1) * ((steps + splits - 1) / splits) >= steps) --splits;   // no empty slice: each one takes a ticket
  bool finalize = false;
  if (splits > 1 && !plain_epilogue) {
    if (!can_finalize) {
      set_error("gemm: a split reduction with an output epilogue needs ticket words (>= one per 64x64 tile), a dense C and no accumulate");
      return MTRSSM_EINVAL;
    }
    finalize = true;
  }
  if (splits > 1 && !p->accumulate) {
    if (!dense_c) {
      set_error("gemm: a split reduction into a strided C needs accumulate (C is zeroed by the caller)");
      return MTRSSM_EINVAL;
    }
    if (int rc = clear_async(p->C, (size_t)p->M * p->N * sizeof(float), stream)) return rc;
    if (finalize)
      if (int rc = clear_async(p->tickets, (size_t)ti * tj * sizeof(int), stream)) return rc;
  }
  g.tickets = finalize ? p->tickets : nullptr;
  g.splits = splits;
  grid = dim3(tj, ti, splits);
  const int per_slice = (steps + splits - 1) / splits;
  fast = !split_mfma && !tile_cols && p->M % kTile == 0 && p->N % kTile == 0 && p->R % kStep == 0 && steps % splits == 0 && per_slice >= 1 &&
         (p->lda & 3) == 0 && (p->ldb & 3) == 0 && ((uintptr_t)p->A & 15) == 0 && ((uintptr_t)p->B & 15) == 0;
  return MTRSSM_OK;
}

int gemm_launch(const MtrssmGemm* p, hipStream_t stream) {
  GemmArgs g;
  dim3 grid;
  bool fast = false;
  if (int rc = gemm_prepare(p, g, grid, fast, stream)) return rc;
  if (const int tn = tile_kernel_columns(p)) {
    const size_t lds_bytes = 2 * (size_t)p->mfma_split * (kGM + tn) * kGPitch;   // two buffers of P pieces of both operand images
#define MTRSSM_GEMM_TILE(AR_, BR_, P_, TN_)                                                                            \
  {                                                                                                                   \
    auto kern = gemm_tile_kernel<AR_, BR_, P_, TN_>;                                                                   \
    if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
    set_last_kernel("mtrssm::gemm_tile_kernel<" #AR_ ", " #BR_ ", " #P_ ", " #TN_ ">");                                \
    hipLaunchKernelGGL(kern, dim3(grid.x * grid.y * grid.z), dim3(512), lds_bytes, stream, g);                        \
  }
#define MTRSSM_GEMM_TILE_PT(AR_, BR_)                                                                                 \
  {                                                                                                                   \
    if (p->mfma_split == 3) { if (tn == 128) MTRSSM_GEMM_TILE(AR_, BR_, 3, 128) else MTRSSM_GEMM_TILE(AR_, BR_, 3, 64) } \
    else { if (tn == 128) MTRSSM_GEMM_TILE(AR_, BR_, 2, 128) else MTRSSM_GEMM_TILE(AR_, BR_, 2, 64) }                 \
  }
    if (p->a_rmajor && p->b_rmajor) MTRSSM_GEMM_TILE_PT(true, true)
    else if (!p->a_rmajor && p->b_rmajor) MTRSSM_GEMM_TILE_PT(false, true)
    else if (!p->a_rmajor && !p->b_rmajor) MTRSSM_GEMM_TILE_PT(false, false)
    else MTRSSM_GEMM_TILE_PT(true, false)
#undef MTRSSM_GEMM_TILE_PT
#undef MTRSSM_GEMM_TILE
  } else if (p->mfma_split == 2) {
    if (p->a_rmajor && p->b_rmajor) {
      set_last_kernel("mtrssm::gemm_split_kernel<true, true>");
      hipLaunchKernelGGL((gemm_split_kernel<true, true>), grid, dim3(256), 0, stream, g);
    } else if (!p->a_rmajor && p->b_rmajor) {
      set_last_kernel("mtrssm::gemm_split_kernel<false, true>");
      hipLaunchKernelGGL((gemm_split_kernel<false, true>), grid, dim3(256), 0, stream, g);
    } else if (!p->a_rmajor && !p->b_rmajor) {
      set_last_kernel("mtrssm::gemm_split_kernel<false, false>");
      hipLaunchKernelGGL((gemm_split_kernel<false, false>), grid, dim3(256), 0, stream, g);
    } else {
      set_last_kernel("mtrssm::gemm_split_kernel<true, false>");
      hipLaunchKernelGGL((gemm_split_kernel<true, false>), grid, dim3(256), 0, stream, g);
    }
  } else {
#define MTRSSM_GEMM_LAUNCH(AR_, BR_)                                                                                  \
  {                                                                                                                   \
    set_last_kernel("mtrssm::gemm_f32_kernel<" #AR_ ", " #BR_ ">");                                                    \
    if (fast) hipLaunchKernelGGL((gemm_f32_kernel<AR_, BR_, true>), grid, dim3(256), 0, stream, g);                   \
    else hipLaunchKernelGGL((gemm_f32_kernel<AR_, BR_, false>), grid, dim3(256), 0, stream, g);                       \
  }
    if (p->a_rmajor && p->b_rmajor) MTRSSM_GEMM_LAUNCH(true, true)
    else if (!p->a_rmajor && p->b_rmajor) MTRSSM_GEMM_LAUNCH(false, true)
    else if (!p->a_rmajor && !p->b_rmajor) MTRSSM_GEMM_LAUNCH(false, false)
    else MTRSSM_GEMM_LAUNCH(true, false)
#undef MTRSSM_GEMM_LAUNCH
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("gemm launch failed: %s", hipGetErrorString(e));
    return MTRSSM_ELAUNCH;
  }
  return MTRSSM_OK;
}

// `count` independent problems (no output of one is read or written by another; fp32 MFMA form only).  Problems of one operand
// layout share launches, kGroupMax at a time; the FAST instantiation is used when every problem of a launch qualifies.
int gemm_group_launch(const MtrssmGemm* ps, int count, hipStream_t stream) {
  if (!ps || count <= 0) { set_error("gemm_group: no problems"); return MTRSSM_EINVAL; }
  for (int layout = 0; layout < 4; ++layout) {
    const int ar = layout >> 1, br = layout & 1;
    GemmGroup grp;
    grp.count = 0;
    grp.first[0] = 0;
    bool all_fast = true;
    auto flush = [&]() -> int {
      if (grp.count == 0) return MTRSSM_OK;
      const dim3 grid((unsigned)grp.first[grp.count]);
#define MTRSSM_GEMM_GROUP_LAUNCH(AR_, BR_)                                                                               \
  {                                                                                                                      \
    set_last_kernel("mtrssm::gemm_f32_group_kernel<" #AR_ ", " #BR_ ">");                                                 \
    if (all_fast) hipLaunchKernelGGL((gemm_f32_group_kernel<AR_, BR_, true>), grid, dim3(256), 0, stream, grp);          \
    else hipLaunchKernelGGL((gemm_f32_group_kernel<AR_, BR_, false>), grid, dim3(256), 0, stream, grp);                  \
  }
      if (ar && br) MTRSSM_GEMM_GROUP_LAUNCH(true, true)
      else if (!ar && br) MTRSSM_GEMM_GROUP_LAUNCH(false, true)
      else if (!ar && !br) MTRSSM_GEMM_GROUP_LAUNCH(false, false)
      else MTRSSM_GEMM_GROUP_LAUNCH(true, false)
#undef MTRSSM_GEMM_GROUP_LAUNCH
      grp.count = 0;
      all_fast = true;
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { set_error("gemm_group launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
      return MTRSSM_OK;
    };
    // the problems of a launch share the chip: each one's reduction is cut to its share of ~1500 workgroups (alone it would be
    // cut to fill 768 by itself, and ten of those queue up thirty deep per CU)
    int members = 0;
    for (int i = 0; i < count; ++i) members += ((ps[i].a_rmajor != 0) == (ar != 0) && (ps[i].b_rmajor != 0) == (br != 0)) ? 1 : 0;
    const int fill = members > 0 ? (1536 / members < 64 ? 64 : 1536 / members) : 768;
    for (int i = 0; i < count; ++i) {
      const MtrssmGemm* p = ps + i;
      if ((p->a_rmajor != 0) != (ar != 0) || (p->b_rmajor != 0) != (br != 0)) continue;
      if (p->mfma_split != 0 && !(p->mfma_split == 3 && !tile_kernel_columns(p))) {
        set_error("gemm_group: problems of the fp32 MFMA kernel only (mfma_split = 0, or 3 on a shape the tile kernel does not take)");
        return MTRSSM_EINVAL;
      }
      GemmArgs g;
      dim3 grid;
      bool fast = false;
      if (int rc = gemm_prepare(p, g, grid, fast, stream, fill < 768 ? fill : 768)) return rc;
      grp.g[grp.count] = g;
      grp.first[grp.count + 1] = grp.first[grp.count] + (int)(grid.x * grid.y * grid.z);
      all_fast = all_fast && fast;
      if (++grp.count == kGroupMax)
        if (int rc = flush()) return rc;
    }
    if (int rc = flush()) return rc;
  }
  return MTRSSM_OK;
}

}  // namespace mtrssm
