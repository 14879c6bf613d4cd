// Device helpers shared by the MRSSM / MMTRSSM scan kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mtrssm.h"

namespace mtrssm {

constexpr int kWave = 64;

// Zero `bytes` (a multiple of 4, 4-byte aligned) of device memory on `stream` with a plain kernel (train_ops.hip).  The library's
// launch functions are captured into hipGraphs (graph.CapturedTrainStep); a captured hipMemsetAsync node of a few megabytes
// went together with foreign bytes at the head of the buffer on replay (ROCm 7.2: the kernel arguments of an unrelated eager
// launch showed up there), so no memset node is ever recorded.
int clear_async(void* p, size_t bytes, hipStream_t stream);
constexpr float kLogThird = -1.0986122886681098f;  // log(1/3), mrssm/mopoe_mrssm/core.py:141-142

__device__ __forceinline__ float act_fwd(float z, int act) {
  switch (act) {
    case MTRSSM_ACT_RELU: return z > 0.f ? z : 0.f;
    case MTRSSM_ACT_ELU: return z > 0.f ? z : expm1f(z);
    case MTRSSM_ACT_TANH: return tanhf(z);
    default: return z;
  }
}

// derivative of the activation expressed through its OUTPUT h = act(z)
__device__ __forceinline__ float act_grad_from_out(float h, int act) {
  switch (act) {
    case MTRSSM_ACT_RELU: return h > 0.f ? 1.f : 0.f;
    case MTRSSM_ACT_ELU: return h > 0.f ? 1.f : h + 1.f;
    case MTRSSM_ACT_TANH: return 1.f - h * h;
    default: return 1.f;
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// Workgroup barrier for kernels whose waves talk to each other through LDS only.  __syncthreads() is a workgroup-scope
// release/acquire fence + s_barrier, and the fence drains the vector-memory counter too (s_waitcnt vmcnt(0)): every barrier
// that follows a global STORE -- the scan kernels write their per-step outputs and saved activations between almost any two
// barriers -- then waits for the HBM write acknowledgement, a full memory round trip on the timestep's critical path
// (measured in the cluster scan: ~1,000 cycles per barrier, 9 barriers per step).  Here only LDS traffic is drained.
// Safe because no wave of these kernels reads global memory that another wave of the same launch wrote with a plain store
// (outputs are read by later launches; the cluster exchanges are polled sc1 granules).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Wave-wide reductions on the DPP data path (every lane gets the result; all 64 lanes must be active).  __shfl_xor goes
// through the LDS crossbar (ds_bpermute_b32: ~60 cycles per step of a dependent chain of six); a DPP step is one VALU
// instruction.  Steps: quad_perm swaps, row_half_mirror, row_mirror (16-lane rows complete), row_bcast15 into rows 1 and 3,
// row_bcast31 into rows 2 and 3, lane 63 holds the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1, 0xF>(0.f, v);   // quad_perm [1, 0, 3, 2]
  v += dpp_move<0x4E, 0xF>(0.f, v);   // quad_perm [2, 3, 0, 1]
  v += dpp_move<0x141, 0xF>(0.f, v);  // row_half_mirror
  v += dpp_move<0x140, 0xF>(0.f, v);  // row_mirror
  v += dpp_move<0x142, 0xA>(0.f, v);  // row_bcast15 -> rows 1, 3
  v += dpp_move<0x143, 0xC>(0.f, v);  // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x140, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x142, 0xA>(v, v));  // rows 0, 2 keep their own value (old = v)
  v = fmaxf(v, dpp_move<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---------------------------------------------------------------------------------------
// Row-tile GEMV, thread-per-output ("wide O"):
//   acc[rb] = init(rb, o) + sum_r M[r*ld + o] * vin[rb][r];   fin(rb, o, acc[rb])
// M is row-major [R][ld] in global memory (streamed from L2); consecutive threads read consecutive
// o -> coalesced 256-B wave loads.  vin lives in LDS (broadcast reads).  The sum is a sequential
// fp32 fma chain over r (deterministic).
// ---------------------------------------------------------------------------------------
template <int RB, typename Init, typename Fin>
__device__ __forceinline__ void gemv_t(const float* __restrict__ M, int ld, int R, int O,
                                       const float* vin, int vin_stride, Init init, Fin fin) {
  for (int o = threadIdx.x; o < O; o += blockDim.x) {
    float acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = init(rb, o);
    const float* m = M + o;
    int r = 0;
    for (; r + 8 <= R; r += 8) {
      float w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = m[(size_t)(r + i) * ld];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = fmaf(w[i], vin[rb * vin_stride + r + i], acc[rb]);
      }
    }
    for (; r < R; ++r) {
      const float w = m[(size_t)r * ld];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = fmaf(w, vin[rb * vin_stride + r], acc[rb]);
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) fin(rb, o, acc[rb]);
  }
}

// ---------------------------------------------------------------------------------------
// Row-tile GEMV, split-K ("wide O", the scan's workhorse):
//   out[rb][o] = init(rb, o) + sum_r M[r*ld + o] * vin[rb][r];   fin(rb, o, out)
// The step is latency-bound (one workgroup streams ~1.7 MB of weights from L2 per timestep), so the point
// is to have as many independent loads in flight as the CU takes: thread = (column group of W outputs,
// K slice); a K slice walks rows r = ks, ks+KS, ... with 16-byte loads when VEC (W = 4), the KS partial
// sums meet in LDS (`red`, >= blockDim.x * W * RB floats) and the first O threads finish.
// Contains two barriers; callers need no barrier between the GEMV and a consumer of fin()'s LDS writes
// other than their usual one.
// ---------------------------------------------------------------------------------------
template <int RB, bool VEC, typename Init, typename Fin>
__device__ __forceinline__ void gemv_sk(const float* __restrict__ M, int ld, int R, int O, const float* vin,
                                        int vin_stride, float* red, Init init, Fin fin) {
  constexpr int W = VEC ? 4 : 1;
  const int nthr = blockDim.x;
  const int CG = (O + W - 1) / W;              // column groups of W outputs
  const int CGP = CG < nthr ? CG : nthr;       // column groups per pass
  int KS = nthr / CGP;                         // K slices
  if (KS > R) KS = R;
  const int OP = CG * W;                       // padded output extent (row length of `red`)
  const int t = threadIdx.x;
  const int lcg = t % CGP, ks = t / CGP;
  for (int base = 0; base < CG; base += CGP) {  // a single pass unless O/W exceeds the block
    const int cg = base + lcg;
    if (cg < CG && ks < KS) {
      float acc[RB][W];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int j = 0; j < W; ++j) acc[rb][j] = 0.f;
      const float* m = M + (size_t)cg * W;
#pragma unroll 4
      for (int r = ks; r < R; r += KS) {
        float w[W];
        if constexpr (VEC) {
          const float4 q = *reinterpret_cast<const float4*>(m + (size_t)r * ld);
          w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
        } else {
          w[0] = m[(size_t)r * ld];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const float v = vin[rb * vin_stride + r];
#pragma unroll
          for (int j = 0; j < W; ++j) acc[rb][j] = fmaf(w[j], v, acc[rb][j]);
        }
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int j = 0; j < W; ++j) red[((size_t)ks * RB + rb) * OP + cg * W + j] = acc[rb][j];
    }
    lds_barrier();
    const int o_lo = base * W;
    const int o_hi = (base + CGP) * W < O ? (base + CGP) * W : O;
    for (int o = o_lo + t; o < o_hi; o += nthr) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        float sum = init(rb, o);
        for (int k = 0; k < KS; ++k) sum += red[((size_t)k * RB + rb) * OP + o];
        fin(rb, o, sum);
      }
    }
    lds_barrier();
  }
}

// Wave-per-output dot products ("narrow O"): acc[rb] = sum_r wrow[r] * vin[rb][r], all lanes return the sum.
// wrow is one contiguous weight row in global memory; lanes stride r (coalesced), butterfly reduce.
template <int RB>
__device__ __forceinline__ void wave_dot(const float* __restrict__ wrow, int R, const float* vin,
                                         int vin_stride, int lane, float (&acc)[RB]) {
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = 0.f;
  for (int r = lane; r < R; r += kWave) {
    const float w = wrow[r];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = fmaf(w, vin[rb * vin_stride + r], acc[rb]);
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = wave_sum(acc[rb]);
}

// ---------------------------------------------------------------------------------------
// Categorical helpers, executed by ONE wave for one row.  Logits are flat [K*C] in LDS.
// ---------------------------------------------------------------------------------------

// exp / log of the categorical helpers.  FAST = the hardware transcendentals (v_exp_f32 / v_log_f32, ~1e-6 relative): the
// cluster scan's categorical block sits on every timestep's critical path, where the libm versions cost ~5 us per step.
template <bool FAST> __device__ __forceinline__ float cexp(float x) { return FAST ? __expf(x) : expf(x); }
template <bool FAST> __device__ __forceinline__ float clog(float x) { return FAST ? __logf(x) : logf(x); }

// flat log-sum-exp pieces over S entries: returns (max, log(sum exp(x - max))), torch.log_softmax form
template <bool FAST = false>
__device__ __forceinline__ void wave_flat_lse(const float* x, int S, int lane, float& mx, float& lsum) {
  float m = -INFINITY;
  for (int s = lane; s < S; s += kWave) m = fmaxf(m, x[s]);
  m = wave_max(m);
  float acc = 0.f;
  for (int s = lane; s < S; s += kWave) acc += cexp<FAST>(x[s] - m);
  acc = wave_sum(acc);
  mx = m;
  lsum = clog<FAST>(acc);
}

// MoPoE mix of two experts' flat logits (core.py:241-243 + 112-163) -> mixed[S] (LDS)
template <bool FAST = false>
__device__ __forceinline__ void wave_mopoe_mix(const float* la, const float* lv, float* mixed, int S, int lane) {
  float ma, lsa, mv, lsv;
  wave_flat_lse<FAST>(la, S, lane, ma, lsa);
  wave_flat_lse<FAST>(lv, S, lane, mv, lsv);
  for (int s = lane; s < S; s += kWave) {
    const float a = (la[s] - ma) - lsa;
    const float v = (lv[s] - mv) - lsv;
    const float f = a + v;
    const float x1 = kLogThird + a, x2 = kLogThird + v, x3 = kLogThird + f;
    const float m = fmaxf(x1, fmaxf(x2, x3));
    mixed[s] = clog<FAST>(cexp<FAST>(x1 - m) + cexp<FAST>(x2 - m) + cexp<FAST>(x3 - m)) + m;
  }
}

// Per-categorical softmax statistics for category k: max and log-sum
template <bool FAST = false>
__device__ __forceinline__ void cat_stats(const float* x, int C, float& mx, float& sum) {
  float m = x[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += cexp<FAST>(x[c] - m);
  mx = m;
  sum = s;
}

// inverse-CDF index: number of c in [0, C-2] whose inclusive cumulative probability is <= u
template <bool FAST = false>
__device__ __forceinline__ int cat_sample(const float* x, int C, float mx, float sum, float u) {
  float acc = 0.f;
  int idx = 0;
  for (int c = 0; c + 1 < C; ++c) {
    acc += cexp<FAST>(x[c] - mx) / sum;
    idx += (acc <= u) ? 1 : 0;
  }
  return idx;
}


// ---------------------------------------------------------------------------------------
// Forward categorical block for one row, executed by one wave (lane k owns categorical k).
//   POST: q = softmax_k(q_logits), p = softmax_k(p_logits); returns this lane's partial of
//         sum_k KL(q_k || p_k); posterior sample -> s_lds (+ post_stoch_g); optional prior sample.
//   !POST: prior sample -> s_lds (+ prior_stoch_g); returns 0.
// Follows state.py:17 (sample on construction) and core.py:212-216 (KL over independent(1)).
// ---------------------------------------------------------------------------------------
template <bool POST, bool FAST = false>
__device__ __forceinline__ float cat_block_fwd(const float* q_logits, const float* p_logits, int K, int C, int lane,
                                               const float* u_post, const float* u_prior, float* s_lds,
                                               float* post_stoch_g, float* prior_stoch_g, bool ok) {
  float kl = 0.f;
  for (int k = lane; k < K; k += kWave) {
    const float* pl = p_logits + k * C;
    float pm, ps;
    cat_stats<FAST>(pl, C, pm, ps);
    if (POST) {
      const float* ql = q_logits + k * C;
      float qm, qs;
      cat_stats<FAST>(ql, C, qm, qs);
      const float lqs = clog<FAST>(qs), lps = clog<FAST>(ps);
      float klk = 0.f;
      for (int c = 0; c < C; ++c) {
        const float qc = cexp<FAST>(ql[c] - qm) / qs;
        klk += qc * (((ql[c] - qm) - lqs) - ((pl[c] - pm) - lps));
      }
      kl += klk;
      const int idx = cat_sample<FAST>(ql, C, qm, qs, u_post[k]);
      for (int c = 0; c < C; ++c) {
        const float v = c == idx ? 1.f : 0.f;
        s_lds[k * C + c] = v;
        if (ok) post_stoch_g[k * C + c] = v;
      }
      if (u_prior && prior_stoch_g && ok) {
        const int pidx = cat_sample<FAST>(pl, C, pm, ps, u_prior[k]);
        for (int c = 0; c < C; ++c) prior_stoch_g[k * C + c] = c == pidx ? 1.f : 0.f;
      }
    } else {
      const int pidx = cat_sample<FAST>(pl, C, pm, ps, u_prior[k]);
      for (int c = 0; c < C; ++c) {
        const float v = c == pidx ? 1.f : 0.f;
        s_lds[k * C + c] = v;
        if (ok) prior_stoch_g[k * C + c] = v;
      }
    }
  }
  return kl;
}

// The same block for the cluster scan's critical path (FAST arithmetic, POST, C <= 8): every exponential is evaluated once
// (the generic form recomputes them for the statistics, the KL and the inverse CDF) and the divisions by the partition sums
// are multiplications by one reciprocal (an IEEE division is ~10 instructions; 25 of them per categorical).
__device__ __forceinline__ float cat_block_fwd_fast8(const float* q_logits, const float* p_logits, int K, int C, int lane,
                                                     const float* u_post, const float* u_prior, float* s_lds, float* post_stoch_g,
                                                     float* prior_stoch_g, bool ok) {
  float kl = 0.f;
  for (int k = lane; k < K; k += kWave) {
    const float* ql = q_logits + k * C;
    const float* pl = p_logits + k * C;
    float q[8], p[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      q[c] = c < C ? ql[c] : -INFINITY;
      p[c] = c < C ? pl[c] : -INFINITY;
    }
    float qm = q[0], pm = p[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) { qm = fmaxf(qm, q[c]); pm = fmaxf(pm, p[c]); }
    float eq[8], ep[8], qs = 0.f, ps = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      eq[c] = c < C ? __expf(q[c] - qm) : 0.f;
      ep[c] = c < C ? __expf(p[c] - pm) : 0.f;
      qs += eq[c];
      ps += ep[c];
    }
    const float rq = __frcp_rn(qs), rp = __frcp_rn(ps);
    const float lqs = __logf(qs), lps = __logf(ps);
    float klk = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < C) klk += eq[c] * rq * (((q[c] - qm) - lqs) - ((p[c] - pm) - lps));
    kl += klk;
    const float u = u_post[k];
    float acc = 0.f;
    int idx = 0;
#pragma unroll
    for (int c = 0; c < 7; ++c)
      if (c + 1 < C) { acc += eq[c] * rq; idx += (acc <= u) ? 1 : 0; }
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < C) {
        const float v = c == idx ? 1.f : 0.f;
        s_lds[k * C + c] = v;
        if (ok) post_stoch_g[k * C + c] = v;
      }
    if (u_prior && prior_stoch_g && ok) {
      const float up = u_prior[k];
      float pacc = 0.f;
      int pidx = 0;
#pragma unroll
      for (int c = 0; c < 7; ++c)
        if (c + 1 < C) { pacc += ep[c] * rp; pidx += (pacc <= up) ? 1 : 0; }
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < C) prior_stoch_g[k * C + c] = c == pidx ? 1.f : 0.f;
    }
  }
  return kl;
}

// ---------------------------------------------------------------------------------------
// Backward categorical block for one row (one wave, lane k owns categorical k):
//   dq[s] = q (gq - <q,gq>)                      straight-through sample, gq = g_post_stoch + carry
//         + gk * w_post * q ((lq - lp) - KL_k)   KL wrt posterior logits
//         + g_post_logits[s]
//   dp[s] = gk * w_prior * (p - q) + g_prior_logits[s] + p (gps - <p,gps>)
// (kl balancing: w_post = 1-alpha, w_prior = alpha; distribution_extension.kl_divergence restated
//  in oracle/ref_dists.py).  Global gradient row pointers may be null (= zero).
// ---------------------------------------------------------------------------------------
template <bool FAST = false>
__device__ __forceinline__ void cat_block_bwd(const float* q_logits, const float* p_logits, int K, int C, int lane,
                                              const float* g_post_stoch, const float* carry_s,
                                              const float* g_prior_stoch, const float* g_post_logits,
                                              const float* g_prior_logits, float gk, float w_post, float w_prior,
                                              float* dq, float* dp) {
  for (int k = lane; k < K; k += kWave) {
    const float* ql = q_logits + k * C;
    const float* pl = p_logits + k * C;
    float qm, qs, pm, ps;
    cat_stats<FAST>(ql, C, qm, qs);
    cat_stats<FAST>(pl, C, pm, ps);
    const float lqs = clog<FAST>(qs), lps = clog<FAST>(ps);
    const float rq = 1.f / qs, rp = 1.f / ps;
    float dot = 0.f, klk = 0.f, pdot = 0.f;
    for (int c = 0; c < C; ++c) {
      const int s = k * C + c;
      const float qc = FAST ? cexp<FAST>(ql[c] - qm) * rq : expf(ql[c] - qm) / qs;
      const float gq = (g_post_stoch ? g_post_stoch[s] : 0.f) + carry_s[s];
      dot += qc * gq;
      klk += qc * (((ql[c] - qm) - lqs) - ((pl[c] - pm) - lps));
      if (g_prior_stoch) pdot += (FAST ? cexp<FAST>(pl[c] - pm) * rp : expf(pl[c] - pm) / ps) * g_prior_stoch[s];
    }
    for (int c = 0; c < C; ++c) {
      const int s = k * C + c;
      const float qc = FAST ? cexp<FAST>(ql[c] - qm) * rq : expf(ql[c] - qm) / qs;
      const float pc = FAST ? cexp<FAST>(pl[c] - pm) * rp : expf(pl[c] - pm) / ps;
      const float gq = (g_post_stoch ? g_post_stoch[s] : 0.f) + carry_s[s];
      const float diff = ((ql[c] - qm) - lqs) - ((pl[c] - pm) - lps);
      float a = qc * (gq - dot) + gk * w_post * qc * (diff - klk);
      if (g_post_logits) a += g_post_logits[s];
      float b = gk * w_prior * (pc - qc);
      if (g_prior_logits) b += g_prior_logits[s];
      if (g_prior_stoch) b += pc * (g_prior_stoch[s] - pdot);
      dq[s] = a;
      dp[s] = b;
    }
  }
}

// The same block for the cooperative kernels' critical path (C <= 8, hardware exp / log): a categorical's logits and incoming
// gradients are read ONCE into registers, every exponential is evaluated once, the loops are unrolled (the generic form walks
// the classes twice with a dependent LDS read chain: 6.3 us per step at K = 16, C = 8 on one wave).
__device__ __forceinline__ void cat_block_bwd_fast8(const float* q_logits, const float* p_logits, int K, int C, int lane,
                                                    const float* g_post_stoch, const float* carry_s, const float* g_prior_stoch,
                                                    const float* g_post_logits, const float* g_prior_logits, float gk, float w_post,
                                                    float w_prior, float* dq, float* dp) {
  for (int k = lane; k < K; k += kWave) {
    const int s0 = k * C;
    float q[8], p[8], gq[8], gps[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const bool in = c < C;
      q[c] = in ? q_logits[s0 + c] : -INFINITY;
      p[c] = in ? p_logits[s0 + c] : -INFINITY;
      gq[c] = in ? (g_post_stoch ? g_post_stoch[s0 + c] : 0.f) + carry_s[s0 + c] : 0.f;
      gps[c] = (in && g_prior_stoch) ? g_prior_stoch[s0 + c] : 0.f;
    }
    float qm = q[0], pm = p[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) { qm = fmaxf(qm, q[c]); pm = fmaxf(pm, p[c]); }
    float eq[8], ep[8], qs = 0.f, ps = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      eq[c] = c < C ? __expf(q[c] - qm) : 0.f;
      ep[c] = c < C ? __expf(p[c] - pm) : 0.f;
      qs += eq[c];
      ps += ep[c];
    }
    const float rq = __frcp_rn(qs), rp = __frcp_rn(ps), lqs = __logf(qs), lps = __logf(ps);
    float diff[8], dot = 0.f, klk = 0.f, pdot = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      eq[c] *= rq;   // q_c
      ep[c] *= rp;   // p_c
      diff[c] = c < C ? ((q[c] - qm) - lqs) - ((p[c] - pm) - lps) : 0.f;
      dot += eq[c] * gq[c];
      klk += eq[c] * diff[c];
      pdot += ep[c] * gps[c];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < C) {
        float a = eq[c] * (gq[c] - dot) + gk * w_post * eq[c] * (diff[c] - klk);
        if (g_post_logits) a += g_post_logits[s0 + c];
        float b = gk * w_prior * (ep[c] - eq[c]);
        if (g_prior_logits) b += g_prior_logits[s0 + c];
        if (g_prior_stoch) b += ep[c] * (gps[c] - pdot);
        dq[s0 + c] = a;
        dp[s0 + c] = b;
      }
    }
  }
}

// Backward of wave_mopoe_mix: given d mixed (dmx, LDS) and the saved expert logits, writes d la / d lv (LDS).
template <bool FAST = false>
__device__ __forceinline__ void wave_mopoe_mix_bwd(const float* la, const float* lv, const float* mixed, const float* dmx,
                                                   float* dla, float* dlv, int S, int lane) {
  float ma, lsa, mv, lsv;
  wave_flat_lse<FAST>(la, S, lane, ma, lsa);
  wave_flat_lse<FAST>(lv, S, lane, mv, lsv);
  float suma = 0.f, sumv = 0.f;
  for (int s = lane; s < S; s += kWave) {
    const float a = (la[s] - ma) - lsa;
    const float v = (lv[s] - mv) - lsv;
    const float mx = mixed[s];
    const float wa = cexp<FAST>(kLogThird + a - mx), wv = cexp<FAST>(kLogThird + v - mx), wf = cexp<FAST>(kLogThird + a + v - mx);
    const float g = dmx[s];
    const float da = g * (wa + wf), dv = g * (wv + wf);
    dla[s] = da;
    dlv[s] = dv;
    suma += da;
    sumv += dv;
  }
  suma = wave_sum(suma);
  sumv = wave_sum(sumv);
  for (int s = lane; s < S; s += kWave) {
    dla[s] -= cexp<FAST>((la[s] - ma) - lsa) * suma;
    dlv[s] -= cexp<FAST>((lv[s] - mv) - lsv) * sumv;
  }
}

}  // namespace mtrssm
