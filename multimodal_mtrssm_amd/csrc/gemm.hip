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
#include "scan_common.h"

namespace mtrssm {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);

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
};

__device__ __forceinline__ float act_grad_from_in(float z, int act) {
  switch (act) {
    case MTRSSM_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case MTRSSM_ACT_ELU: return z > 0.f ? 1.f : expf(z);
    case MTRSSM_ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
    default: return 1.f;
  }
}

// one operand tile (64 rows x 32 r) from global memory into registers: 2 x float4 per thread
struct Stage { float4 v[2]; };

template <bool RMAJOR>
__device__ __forceinline__ void load_tile(Stage& st, const float* __restrict__ P, int ld, int rows, int R, int row0, int r0,
                                          int r_end, bool vec) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int f = tid + 256 * j;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RMAJOR) {  // memory [r][row]: 16 float4 per r line of 64 rows
      const int r = r0 + (f >> 4), row = row0 + 4 * (f & 15);
      if (r < r_end) {
        const float* p = P + (size_t)r * ld + row;
        if (vec && row + 3 < rows) q = *reinterpret_cast<const float4*>(p);
        else {
          if (row < rows) q.x = p[0];
          if (row + 1 < rows) q.y = p[1];
          if (row + 2 < rows) q.z = p[2];
          if (row + 3 < rows) q.w = p[3];
        }
      }
    } else {  // memory [row][r]: 8 float4 per row of 32 r
      const int row = row0 + (f >> 3), r = r0 + 4 * (f & 7);
      if (row < rows) {
        const float* p = P + (size_t)row * ld + r;
        if (vec && r + 3 < r_end) q = *reinterpret_cast<const float4*>(p);
        else {
          if (r < r_end) q.x = p[0];
          if (r + 1 < r_end) q.y = p[1];
          if (r + 2 < r_end) q.z = p[2];
          if (r + 3 < r_end) q.w = p[3];
        }
      }
    }
    st.v[j] = q;
  }
  (void)R;
}

template <bool RMAJOR>
__device__ __forceinline__ void store_tile(const Stage& st, float* img, int act) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int f = tid + 256 * j;
    float4 q = st.v[j];
    if (act) { q.x = act_fwd(q.x, act); q.y = act_fwd(q.y, act); q.z = act_fwd(q.z, act); q.w = act_fwd(q.w, act); }
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

template <bool AR, bool BR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][kImage];  // [buffer][operand][image]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i0 = blockIdx.y * kTile, j0 = blockIdx.x * kTile;
  // this workgroup's share of the reduction (multiples of kStep)
  const int steps = (g.R + kStep - 1) / kStep;
  const int per = (steps + g.splits - 1) / g.splits;
  const int s_lo = blockIdx.z * per, s_hi = min(steps, s_lo + per);
  if (s_lo >= s_hi) return;
  const int r_end = min(g.R, s_hi * kStep);
  const bool vec_a = ((g.lda & 3) == 0) && (((uintptr_t)g.A & 15) == 0);
  const bool vec_b = ((g.ldb & 3) == 0) && (((uintptr_t)g.B & 15) == 0);
  const bool want_colsum = g.colsum && blockIdx.x == 0;
  float csum = 0.f;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // Software pipeline: the operand tiles of steps s+1 and s+2 are in flight in registers (two stages) while step s is
  // multiplied out of LDS, so a global load has two whole steps to land; with one stage the ~1.5 us load latency, not the
  // 0.43 us of MFMAs, set the step time on grids of < 1 workgroup per SIMD (measured: 13-16 TFLOP/s).
  Stage sa[2], sb[2];
  auto issue = [&](int which, int s) {
    if (s < s_hi) {
      load_tile<AR>(sa[which], g.A, g.lda, g.M, g.R, i0, s * kStep, r_end, vec_a);
      load_tile<BR>(sb[which], g.B, g.ldb, g.N, g.R, j0, s * kStep, r_end, vec_b);
    }
  };
  auto step = [&](int which, int s, int cur) {  // `which`: the register stage that holds step s + 1
    if (s + 1 < s_hi) {
      store_tile<AR>(sa[which], lds[cur ^ 1][0], g.act_a);
      store_tile<BR>(sb[which], lds[cur ^ 1][1], g.act_b);
    }
    issue(which, s + 3);
    const float* ia = lds[cur][0];
    const float* ib = lds[cur][1];
    const int ar = wr * 32 + (lane & 31), bc = wc * 32 + (lane & 31), kh = lane >> 5;
#pragma unroll
    for (int k = 0; k < kStep; k += 2) {
      const float a = image_at<AR>(ia, ar, k + kh);
      const float b = image_at<BR>(ib, bc, k + kh);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (want_colsum && tid < kTile) {
#pragma unroll 8
      for (int k = 0; k < kStep; ++k) csum += image_at<AR>(ia, tid, k);
    }
    __syncthreads();
  };
  issue(0, s_lo);
  store_tile<AR>(sa[0], lds[0][0], g.act_a);
  store_tile<BR>(sb[0], lds[0][1], g.act_b);
  issue(0, s_lo + 1);
  issue(1, s_lo + 2);
  __syncthreads();
  for (int s = s_lo; s < s_hi; s += 2) {
    step(0, s, 0);
    if (s + 1 < s_hi) step(1, s + 1, 1);
  }

  // epilogue: C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int j = j0 + wc * 32 + (lane & 31);
  const float bias = (g.bias && j < g.N) ? g.bias[j] : 0.f;
  const bool atomic = g.splits > 1;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int i = i0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    if (i < g.M && j < g.N) {
      float v = acc[reg];
      if (!atomic || blockIdx.z == 0) v += bias;
      if (g.act_out) v = act_fwd(v, g.act_out);
      if (g.zgrad) v *= act_grad_from_in(g.zgrad[(size_t)i * g.ldz + j], g.act_z);
      float* c = g.C + (size_t)i * g.ldc + j;
      if (atomic) atomicAdd(c, v);
      else if (g.accumulate) *c += v;
      else *c = v;
    }
  }
  if (want_colsum && tid < kTile && i0 + tid < g.M) atomicAdd(g.colsum + i0 + tid, csum);
}

int gemm_launch(const MtrssmGemm* p, hipStream_t stream) {
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
  GemmArgs g;
  g.A = p->A; g.B = p->B; g.C = p->C; g.bias = p->bias; g.zgrad = p->zgrad; g.colsum = p->colsum;
  g.M = p->M; g.N = p->N; g.R = p->R; g.lda = p->lda; g.ldb = p->ldb; g.ldc = p->ldc; g.ldz = p->ldz;
  g.a_rmajor = p->a_rmajor; g.b_rmajor = p->b_rmajor; g.act_a = p->act_a; g.act_b = p->act_b; g.act_out = p->act_out;
  g.act_z = p->act_z; g.accumulate = p->accumulate;
  const int ti = (p->M + kTile - 1) / kTile, tj = (p->N + kTile - 1) / kTile;
  int splits = p->split_r;
  const int steps = (p->R + kStep - 1) / kStep;
  const bool plain_epilogue = !p->act_out && !p->zgrad;
  const bool dense_c = p->ldc == p->N;
  if (splits <= 0) {
    // Automatic: a workgroup's time is (its reduction steps) x (a load round trip), so a grid that does not fill the 256 CUs
    // three deep is cut along the reduction, down to 4 steps per slice.  Slices meet by fp32 atomics: into the running
    // target when accumulating; otherwise into C zeroed here first (dense C only).
    splits = 1;
    if (plain_epilogue && (p->accumulate || dense_c))
      while (ti * tj * splits < 768 && steps / (splits * 2) >= 4) splits *= 2;
  }
  if (splits > 1 && !plain_epilogue) {
    set_error("gemm: an output activation / act' epilogue cannot be combined with a split reduction");
    return MTRSSM_EINVAL;
  }
  if (splits > 1 && !p->accumulate) {
    if (!dense_c) {
      set_error("gemm: a split reduction into a strided C needs accumulate (C is zeroed by the caller)");
      return MTRSSM_EINVAL;
    }
    hipError_t e = hipMemsetAsync(p->C, 0, (size_t)p->M * p->N * sizeof(float), stream);
    if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  }
  g.splits = splits;
  const dim3 grid(tj, ti, splits);
  if (p->a_rmajor && p->b_rmajor) {
    set_last_kernel("mtrssm::gemm_f32_kernel<true, true>");
    hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, stream, g);
  } else if (!p->a_rmajor && p->b_rmajor) {
    set_last_kernel("mtrssm::gemm_f32_kernel<false, true>");
    hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, stream, g);
  } else if (!p->a_rmajor && !p->b_rmajor) {
    set_last_kernel("mtrssm::gemm_f32_kernel<false, false>");
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, stream, g);
  } else {
    set_last_kernel("mtrssm::gemm_f32_kernel<true, false>");
    hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, stream, g);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("gemm launch failed: %s", hipGetErrorString(e));
    return MTRSSM_ELAUNCH;
  }
  return MTRSSM_OK;
}

}  // namespace mtrssm
