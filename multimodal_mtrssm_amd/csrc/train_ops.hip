// Streaming train-step ops for gfx950: fused Gaussian NLL (objective.py:7-23) and a flat fused
// AdamW with global-norm clipping (default.yaml:103-107,119).  All are HBM-bound: 16 B per lane
// coalesced accesses, grid-stride over <= 2048 workgroups, wave64 shuffle reductions, one atomic
// per workgroup.
#include "scan_common.h"

namespace mtrssm {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);

constexpr int kThreads = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x < kWave) {
    t = threadIdx.x < blockDim.x / kWave ? red[threadIdx.x] : 0.f;
    t = wave_sum(t);
  }
  return t;  // valid in wave 0
}

// tanh on the hardware exponential: 1 - 2 / (exp(2x) + 1), absolute error ~1e-7 (as the GEMM / conv epilogues), saturates
// cleanly.  libm's tanhf made the two NLL kernels VALU-bound (41 / 25 us for a 105 MB stream).
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f / (__expf(2.f * x) + 1.f); }

// out += scale * sum 0.5 (t - p)^2   (+ the constant on block 0)
template <bool TANH>
__global__ void nll_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n,
                               float scale, float constant, float* __restrict__ out) {
  __shared__ float red[kThreads / kWave];
  const int64_t n4 = n / 4;
  const float4* p4 = reinterpret_cast<const float4*>(pred);
  const float4* t4 = reinterpret_cast<const float4*>(target);
  float acc = 0.f;
  auto term = [](float4 p, const float4 t) {
    if (TANH) { p.x = tanh_fast(p.x); p.y = tanh_fast(p.y); p.z = tanh_fast(p.z); p.w = tanh_fast(p.w); }
    const float a = t.x - p.x, b = t.y - p.y, c = t.z - p.z, d = t.w - p.w;
    return 0.5f * (a * a) + 0.5f * (b * b) + 0.5f * (c * c) + 0.5f * (d * d);
  };
  // four quads per thread and iteration in flight; few workgroups (the launch picks 512): every workgroup ends in ONE atomic
  // on the same word, ~25 ns each when they queue up (2048 of them were 40 us of a 41 us kernel)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 pa = p4[i], pb = p4[i + stride], pc = p4[i + 2 * stride], pd = p4[i + 3 * stride];
    const float4 ta = t4[i], tb = t4[i + stride], tc = t4[i + 2 * stride], td = t4[i + 3 * stride];
    acc += (term(pa, ta) + term(pb, tb)) + (term(pc, tc) + term(pd, td));
  }
  for (; i < n4; i += stride) acc += term(p4[i], t4[i]);
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float a = target[i] - (TANH ? tanh_fast(pred[i]) : pred[i]);
      acc += 0.5f * a * a;
    }
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, tot * scale + (blockIdx.x == 0 ? constant : 0.f));
}

// d/dz of 0.5 (t - act(z))^2 = (act(z) - t) act'(z); Tanh: act' = 1 - act^2
template <bool TANH>
__global__ void nll_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                               const float* __restrict__ g_out, int64_t n, float inv_frames, float* __restrict__ g_pred) {
  const float g = g_out[0] * inv_frames;
  const int64_t n4 = n / 4;
  const float4* p4 = reinterpret_cast<const float4*>(pred);
  const float4* t4 = reinterpret_cast<const float4*>(target);
  float4* o4 = reinterpret_cast<float4*>(g_pred);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 p = p4[i];
    const float4 t = t4[i];
    float4 d = make_float4(1.f, 1.f, 1.f, 1.f);
    if (TANH) {
      p.x = tanh_fast(p.x); p.y = tanh_fast(p.y); p.z = tanh_fast(p.z); p.w = tanh_fast(p.w);
      d = make_float4(1.f - p.x * p.x, 1.f - p.y * p.y, 1.f - p.z * p.z, 1.f - p.w * p.w);
    }
    o4[i] = make_float4(g * (p.x - t.x) * d.x, g * (p.y - t.y) * d.y, g * (p.z - t.z) * d.z, g * (p.w - t.w) * d.w);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float p = TANH ? tanh_fast(pred[i]) : pred[i];
      g_pred[i] = g * (p - target[i]) * (TANH ? 1.f - p * p : 1.f);
    }
}

__global__ void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[kThreads / kWave];
  const int64_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) acc += x[i] * x[i];
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, tot);
}

// torch.optim.AdamW semantics (decoupled decay, bias correction, eps outside the sqrt's bias term):
//   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// preceded by clip_grad_norm_: g *= min(1, clip / (||g|| + 1e-6)), and by grad_scale (e.g. 1/world).
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, const float* __restrict__ sumsq, float clip, float gscale, float lr, float b1,
                             float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  float coef = gscale;
  if (clip > 0.f && sumsq) {
    const float norm = sqrtf(sumsq[0]) * gscale;
    coef *= fminf(1.f, clip / (norm + 1e-6f));
  }
  const float step = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi;
  }
}

// Graph-safe variant: the learning rate and the step count live in device memory (`state`: [0] lr, [1] steps taken,
// [2] 1 - b1^step, [3] sqrt(1 - b2^step)), so a captured train step replays with the right bias correction and sees a
// scheduler's new lr without re-capture.  `active` (one byte per element, may be null) marks the parameters that have ever
// received a gradient: torch.optim.AdamW skips parameters whose .grad is None (no decay, no moments) -- MMTRSSM's
// l_posterior and dummy transition (mmtrssm/mopoe_mmtrssm/core.py:143-151,188).
__global__ void sumsq_tick_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out, float* __restrict__ state,
                                  const int* __restrict__ status, float b1, float b2) {
  __shared__ float red[kThreads / kWave];
  if (blockIdx.x == 0 && threadIdx.x == 0 && state && !(status && *status)) {  // one optimizer step = one launch of this kernel
    const float step = state[1] + 1.f;
    state[1] = step;
    state[2] = 1.f - powf(b1, step);
    state[3] = sqrtf(1.f - powf(b2, step));
  }
  const int64_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) acc += x[i] * x[i];
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, tot);
}

__global__ void adamw_masked_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                    const unsigned char* __restrict__ active, int64_t n, const float* __restrict__ sumsq,
                                    const float* __restrict__ state, const int* __restrict__ status, float clip, float gscale, float b1,
                                    float b2, float eps, float wd) {
  // a cooperative scan kernel of this step gave up on an exchange (sticky status word of its workspace): its outputs, hence
  // these gradients, are invalid -- leave the parameters and the moments alone; the host raises at its next status poll
  if (status && *status) return;
  const float lr = state[0], bc1 = state[2], bc2_sqrt = state[3];
  float coef = gscale;
  if (clip > 0.f && sumsq) {
    const float norm = sqrtf(sumsq[0]) * gscale;
    coef *= fminf(1.f, clip / (norm + 1e-6f));
  }
  const float step = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (active && !active[i]) continue;
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi;
  }
}

__global__ void clear_words_kernel(unsigned* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

int clear_async(void* p, size_t bytes, hipStream_t stream) {
  if (!p || (bytes & 3) || ((uintptr_t)p & 3)) { set_error("clear_async: needs a 4-byte aligned buffer of a multiple of 4 bytes"); return MTRSSM_EINVAL; }
  const size_t n = bytes / 4;
  if (n == 0) return MTRSSM_OK;
  const size_t blocks = (n + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(clear_words_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kThreads), 0, stream, static_cast<unsigned*>(p), n);
  return hipGetLastError() == hipSuccess ? MTRSSM_OK : MTRSSM_ELAUNCH;
}

static int grid_for(int64_t n) {
  int64_t g = (n + kThreads - 1) / kThreads;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s launch failed: %s", what, hipGetErrorString(e));
    return MTRSSM_ELAUNCH;
  }
  return MTRSSM_OK;
}

// ------------------------------------------------------------------------------------------------
// Categorical head of the INITIAL state (core.py:121-135; mmtrssm core.py:321-362): flat logits [rows][K * C] + one uniform per
// categorical -> log-probabilities, probabilities and the inverse-CDF one-hot sample, one thread per (row, categorical), the
// classes walked in order (the cumulative sum is a left fold, as torch's on the host).  Replaces a dozen eager launches per
// level (softmax, log_softmax, cumsum, compare, sum, one_hot, the straight-through add / sub).
// ------------------------------------------------------------------------------------------------
__global__ void categorical_sample_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ u, int64_t n, int C,
                                              float* __restrict__ logp, float* __restrict__ probs, float* __restrict__ onehot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (row, categorical)
  if (i >= n) return;
  const float* x = logits + i * C;
  float m = x[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += expf(x[c] - m);
  const float ls = logf(sum), uu = u[i];
  float acc = 0.f;
  int idx = 0;
  for (int c = 0; c < C; ++c) {
    const float p = expf(x[c] - m) / sum;
    probs[i * C + c] = p;
    logp[i * C + c] = (x[c] - m) - ls;
    acc += p;
    if (c + 1 < C && acc <= uu) ++idx;
  }
  for (int c = 0; c < C; ++c) onehot[i * C + c] = c == idx ? 1.f : 0.f;
}

// d logits of the same head: through the probabilities (g_p: the straight-through sample's gradient plus any gradient of the
// probabilities themselves; may be null) and through the log-probabilities (g_l, may be null):
//   d x_c = p_c (g_p[c] - sum_j p_j g_p[j]) + g_l[c] - p_c sum_j g_l[j]
__global__ void categorical_sample_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ g_p, const float* __restrict__ g_l,
                                              int64_t n, int C, float* __restrict__ d_logits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float dot = 0.f, sl = 0.f;
  for (int c = 0; c < C; ++c) {
    const float p = probs[i * C + c];
    if (g_p) dot += p * g_p[i * C + c];
    if (g_l) sl += g_l[i * C + c];
  }
  for (int c = 0; c < C; ++c) {
    const float p = probs[i * C + c];
    float d = 0.f;
    if (g_p) d += p * (g_p[i * C + c] - dot);
    if (g_l) d += g_l[i * C + c] - p * sl;
    d_logits[i * C + c] = d;
  }
}

int categorical_sample_fwd_launch(const float* logits, const float* u, int64_t rows, int K, int C, float* logp, float* probs, float* onehot,
                                  hipStream_t s) {
  if (!logits || !u || !logp || !probs || !onehot || rows <= 0 || K <= 0 || C <= 0) { set_error("categorical_sample_fwd: bad argument"); return MTRSSM_EINVAL; }
  const int64_t n = rows * K;
  set_last_kernel("mtrssm::categorical_sample_fwd_kernel");
  hipLaunchKernelGGL(categorical_sample_fwd_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, logits, u, n, C, logp, probs, onehot);
  return check_launch("categorical_sample_fwd");
}

int categorical_sample_bwd_launch(const float* probs, const float* g_p, const float* g_l, int64_t rows, int K, int C, float* d_logits,
                                  hipStream_t s) {
  if (!probs || !d_logits || rows <= 0 || K <= 0 || C <= 0) { set_error("categorical_sample_bwd: bad argument"); return MTRSSM_EINVAL; }
  const int64_t n = rows * K;
  set_last_kernel("mtrssm::categorical_sample_bwd_kernel");
  hipLaunchKernelGGL(categorical_sample_bwd_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, probs, g_p, g_l, n, C, d_logits);
  return check_launch("categorical_sample_bwd");
}

// ------------------------------------------------------------------------------------------------
// The scalar end of shared_step (core.py:187-221; mmtrssm core.py:563-606) in one launch each way:
//   recon = nll_a + nll_v;  kl_j = coeff_j * mean(kl_bt_j)  (j < nkl <= 2);  loss = recon + sum_j kl_j
// Four scalar outputs (o_k1 may be null).  Backward: g = { g_recon, g_kl_0, g_kl_1, g_loss } (any may be null = 0) ->
// g_nll_a = g_nll_v = g_recon + g_loss;  g_kl_bt_j[i] = (g_kl_j + g_loss) * coeff_j / n.
// ------------------------------------------------------------------------------------------------
__global__ void elbo_combine_fwd_kernel(const float* __restrict__ nll_a, const float* __restrict__ nll_v, const float* __restrict__ kl0,
                                        const float* __restrict__ kl1, int64_t n, float c0, float c1, float* __restrict__ o_recon, float* __restrict__ o_k0,
                                        float* __restrict__ o_k1, float* __restrict__ o_loss) {
  __shared__ float red[kThreads / kWave];
  float a0 = 0.f, a1 = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    a0 += kl0[i];
    if (kl1) a1 += kl1[i];
  }
  const float s0 = block_sum(a0, red);
  __syncthreads();
  const float s1 = block_sum(a1, red);
  if (threadIdx.x == 0) {
    const float recon = nll_a[0] + nll_v[0];
    const float k0 = s0 / (float)n * c0, k1 = kl1 ? s1 / (float)n * c1 : 0.f;
    o_recon[0] = recon; o_k0[0] = k0; if (o_k1) o_k1[0] = k1; o_loss[0] = recon + k0 + k1;
  }
}
__global__ void elbo_combine_bwd_kernel(const float* __restrict__ g_recon, const float* __restrict__ g_k0, const float* __restrict__ g_k1,
                                        const float* __restrict__ g_loss, int64_t n, float c0, float c1, float* __restrict__ g_nll_a,
                                        float* __restrict__ g_nll_v, float* __restrict__ g_kl0, float* __restrict__ g_kl1) {
  const float gl = g_loss ? g_loss[0] : 0.f;
  const float gn = (g_recon ? g_recon[0] : 0.f) + gl;
  const float v0 = ((g_k0 ? g_k0[0] : 0.f) + gl) * c0 / (float)n, v1 = ((g_k1 ? g_k1[0] : 0.f) + gl) * c1 / (float)n;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { g_nll_a[0] = gn; g_nll_v[0] = gn; }
  if (i < n) {
    g_kl0[i] = v0;
    if (g_kl1) g_kl1[i] = v1;
  }
}
int elbo_combine_fwd_launch(const float* nll_a, const float* nll_v, const float* kl0, const float* kl1, int64_t n, float c0, float c1,
                            float* o_recon, float* o_k0, float* o_k1, float* o_loss, hipStream_t s) {
  if (!nll_a || !nll_v || !kl0 || !o_recon || !o_k0 || !o_loss || n <= 0) { set_error("elbo_combine_fwd: bad argument"); return MTRSSM_EINVAL; }
  set_last_kernel("mtrssm::elbo_combine_fwd_kernel");
  hipLaunchKernelGGL(elbo_combine_fwd_kernel, dim3(1), dim3(kThreads), 0, s, nll_a, nll_v, kl0, kl1, n, c0, c1, o_recon, o_k0, o_k1, o_loss);
  return check_launch("elbo_combine_fwd");
}
int elbo_combine_bwd_launch(const float* g_recon, const float* g_k0, const float* g_k1, const float* g_loss, int64_t n, float c0, float c1,
                            float* g_nll_a, float* g_nll_v, float* g_kl0, float* g_kl1, hipStream_t s) {
  if (!g_nll_a || !g_nll_v || !g_kl0 || n <= 0) { set_error("elbo_combine_bwd: bad argument"); return MTRSSM_EINVAL; }
  set_last_kernel("mtrssm::elbo_combine_bwd_kernel");
  hipLaunchKernelGGL(elbo_combine_bwd_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, g_recon, g_k0, g_k1, g_loss, n,
                     c0, c1, g_nll_a, g_nll_v, g_kl0, g_kl1);
  return check_launch("elbo_combine_bwd");
}

int nll_fwd_launch(const float* pred, const float* target, int64_t frames, int64_t event, int act, float* out, hipStream_t s) {
  if (!pred || !target || !out || frames <= 0 || event <= 0) { set_error("gaussian_nll_fwd: bad argument"); return MTRSSM_EINVAL; }
  if (act != MTRSSM_ACT_IDENTITY && act != MTRSSM_ACT_TANH) { set_error("gaussian_nll: the fused output activation is Identity or Tanh (got %d)", act); return MTRSSM_EINVAL; }
  if (((uintptr_t)pred | (uintptr_t)target) & 15) { set_error("gaussian_nll_fwd: pred/target must be 16-byte aligned"); return MTRSSM_EINVAL; }
  if (int rc = clear_async(out, sizeof(float), s)) return rc;
  const int64_t n = frames * event;
  const float constant = 0.5f * 1.8378770664093453f * (float)event;  // 0.5 log(2 pi) per element
  const int sum_grid = grid_for(n / 4) < 512 ? grid_for(n / 4) : 512;
  set_last_kernel("mtrssm::nll_fwd_kernel");
  if (act == MTRSSM_ACT_TANH)
    hipLaunchKernelGGL(nll_fwd_kernel<true>, dim3(sum_grid), dim3(kThreads), 0, s, pred, target, n, 1.f / (float)frames, constant, out);
  else
    hipLaunchKernelGGL(nll_fwd_kernel<false>, dim3(sum_grid), dim3(kThreads), 0, s, pred, target, n, 1.f / (float)frames, constant, out);
  return check_launch("gaussian_nll_fwd");
}

int nll_bwd_launch(const float* pred, const float* target, const float* g_out, int64_t frames, int64_t event, int act, float* g_pred, hipStream_t s) {
  if (!pred || !target || !g_out || !g_pred || frames <= 0 || event <= 0) { set_error("gaussian_nll_bwd: bad argument"); return MTRSSM_EINVAL; }
  if (act != MTRSSM_ACT_IDENTITY && act != MTRSSM_ACT_TANH) { set_error("gaussian_nll: the fused output activation is Identity or Tanh (got %d)", act); return MTRSSM_EINVAL; }
  if (((uintptr_t)pred | (uintptr_t)target | (uintptr_t)g_pred) & 15) { set_error("gaussian_nll_bwd: buffers must be 16-byte aligned"); return MTRSSM_EINVAL; }
  const int64_t n = frames * event;
  set_last_kernel("mtrssm::nll_bwd_kernel");
  if (act == MTRSSM_ACT_TANH)
    hipLaunchKernelGGL(nll_bwd_kernel<true>, dim3(grid_for(n / 4)), dim3(kThreads), 0, s, pred, target, g_out, n, 1.f / (float)frames, g_pred);
  else
    hipLaunchKernelGGL(nll_bwd_kernel<false>, dim3(grid_for(n / 4)), dim3(kThreads), 0, s, pred, target, g_out, n, 1.f / (float)frames, g_pred);
  return check_launch("gaussian_nll_bwd");
}

int sumsq_launch(const float* x, int64_t n, float* out, hipStream_t s) {
  if (!x || !out || n <= 0) { set_error("sumsq: bad argument"); return MTRSSM_EINVAL; }
  if ((uintptr_t)x & 15) { set_error("sumsq: x must be 16-byte aligned"); return MTRSSM_EINVAL; }
  if (int rc = clear_async(out, sizeof(float), s)) return rc;
  set_last_kernel("mtrssm::sumsq_kernel");
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4) < 512 ? grid_for(n / 4) : 512), dim3(kThreads), 0, s, x, n, out);  // one same-address atomic per workgroup
  return check_launch("sumsq");
}

int adamw_launch(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float clip, float gscale,
                 float lr, float b1, float b2, float eps, float wd, int step, hipStream_t s) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) { set_error("adamw_step: bad argument"); return MTRSSM_EINVAL; }
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(b2, (float)step));
  set_last_kernel("mtrssm::adamw_kernel");
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, p, g, m, v, n, sumsq, clip, gscale, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
  return check_launch("adamw_step");
}

int adamw_prepare_launch(const float* g, int64_t n, float* sumsq, float* state, const int* status, float b1, float b2, hipStream_t s) {
  if (!g || !sumsq || !state || n <= 0) { set_error("adamw_prepare: bad argument"); return MTRSSM_EINVAL; }
  if ((uintptr_t)g & 15) { set_error("adamw_prepare: grad must be 16-byte aligned"); return MTRSSM_EINVAL; }
  if (int rc = clear_async(sumsq, sizeof(float), s)) return rc;
  set_last_kernel("mtrssm::sumsq_tick_kernel");
  hipLaunchKernelGGL(sumsq_tick_kernel, dim3(grid_for(n / 4) < 512 ? grid_for(n / 4) : 512), dim3(kThreads), 0, s, g, n, sumsq, state, status, b1, b2);  // (as sumsq)
  return check_launch("adamw_prepare");
}

int adamw_apply_launch(float* p, const float* g, float* m, float* v, const unsigned char* active, int64_t n, const float* sumsq,
                       const float* state, const int* status, float clip, float gscale, float b1, float b2, float eps, float wd, hipStream_t s) {
  if (!p || !g || !m || !v || !state || n <= 0) { set_error("adamw_apply: bad argument"); return MTRSSM_EINVAL; }
  set_last_kernel("mtrssm::adamw_masked_kernel");
  hipLaunchKernelGGL(adamw_masked_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, p, g, m, v, active, n, sumsq, state, status, clip, gscale,
                     b1, b2, eps, wd);
  return check_launch("adamw_apply");
}

// ------------------------------------------------------------------------------------------------
// Conv weight gradients, end of backward: every conv weight gradient of the step was accumulated (fp32 atomics, 128-byte
// coalesced) in the kernels' packed layout [OPad][taps][IPad]; ONE launch adds them all into the flat gradient buffer in
// the parameter layout [O][I][KH][KW] and clears the packed buffers for the next step.  Replaces one strided torch add
// (AccumulateGrad) and one zero fill per conv weight (~80 of each per MoPoE-MRSSM train step).
// table: `count` rows of 8 int64 = { packed*, grad*, O, I, taps, IPad, 0, 0 }.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void unpack_conv_grads_kernel(const long* __restrict__ table) {
  const long* row = table + (long)blockIdx.y * 8;
  float* packed = reinterpret_cast<float*>(row[0]);
  float* grad = reinterpret_cast<float*>(row[1]);
  const long O = row[2], I = row[3], taps = row[4], ipad = row[5];
  const long total = O * taps * ipad;  // rows o >= O of the padded buffer are never written by the kernels
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long i = e % ipad, ot = e / ipad, tap = ot % taps, o = ot / taps;
    const float v = packed[e];
    if (v != 0.f) packed[e] = 0.f;
    if (i < I) grad[(o * I + i) * taps + tap] += v;
  }
}

int unpack_conv_grads_launch(const int64_t* table, int count, int blocks_per_entry, hipStream_t s) {
  if (!table || count <= 0 || blocks_per_entry <= 0) { set_error("unpack_conv_grads: bad argument"); return MTRSSM_EINVAL; }
  static_assert(sizeof(long) == sizeof(int64_t), "descriptor words are 64-bit");
  set_last_kernel("mtrssm::unpack_conv_grads_kernel");
  hipLaunchKernelGGL(unpack_conv_grads_kernel, dim3(blocks_per_entry, count), dim3(kThreads), 0, s, reinterpret_cast<const long*>(table));
  return check_launch("unpack_conv_grads");
}

// ------------------------------------------------------------------------------------------------
// Episode feed: one [B, T, E] input / target pair from the HBM-resident episode store.
//   target[b, t, :] = store[idx[b], t, :]            (TakeFirstN: t < T <= Tfull, transform.py:31-52)
//   input [b, t, :] = target + noise[b, t, :] * std  (GaussianNoise, transform.py:55-72: two roundings, mul then add)
// Replaces EpisodeDataset.__getitem__ + default collate of the 6-tuple StackDataset (dataset.py:84-112,
// mrssm/dataset.py:155-183).  One float4 per thread; rows are E floats, E % 4 == 0.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void episode_gather_kernel(
    const float* __restrict__ store, const long* __restrict__ idx, const float* __restrict__ noise, long B, long T, long Tfull,
    long E4, float std_, float* __restrict__ input, float* __restrict__ target) {
#pragma clang fp contract(off)  // mul then add, each rounded (torch's `data + randn * std`): hipcc would contract to v_pk_fma_f32
  const long per_b = T * E4;
  const long total = B * per_b;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / per_b, r = i - b * per_b;  // r = t * E4 + e4
    const float4 x = reinterpret_cast<const float4*>(store)[idx[b] * Tfull * E4 + r];
    if (target) reinterpret_cast<float4*>(target)[i] = x;
    if (input) {
      float4 y = x;
      if (noise) {
        const float4 n = reinterpret_cast<const float4*>(noise)[i];
        const float px = n.x * std_, py = n.y * std_, pz = n.z * std_, pw = n.w * std_;  // plain expressions: the pragma
        y.x = x.x + px;                                                                  // above does not reach into the
        y.y = x.y + py;                                                                  // headers' __fmul_rn / __fadd_rn
        y.z = x.z + pz;
        y.w = x.w + pw;
      }
      reinterpret_cast<float4*>(input)[i] = y;
    }
  }
}

int episode_gather_launch(const float* store, const int64_t* idx, const float* noise, int64_t n_episodes, int64_t B, int64_t T,
                          int64_t Tfull, int64_t E, float std_, float* input, float* target, hipStream_t s) {
  if (!store || !idx || (!input && !target) || n_episodes <= 0 || B <= 0 || T <= 0 || Tfull < T || E <= 0) {
    set_error("episode_gather: bad argument (need 0 < T <= Tfull, B, E > 0, an output)");
    return MTRSSM_EINVAL;
  }
  if (E % 4) { set_error("episode_gather: the event size %ld must be a multiple of 4 floats", (long)E); return MTRSSM_EINVAL; }
  if (((uintptr_t)store | (uintptr_t)noise | (uintptr_t)input | (uintptr_t)target) & 15) {
    set_error("episode_gather: buffers must be 16-byte aligned");
    return MTRSSM_EINVAL;
  }
  set_last_kernel("mtrssm::episode_gather_kernel");
  hipLaunchKernelGGL(episode_gather_kernel, dim3(grid_for(B * T * E / 4)), dim3(kThreads), 0, s, store, reinterpret_cast<const long*>(idx),
                     noise, (long)B, (long)T, (long)Tfull, (long)(E / 4), std_, input, target);
  return check_launch("episode_gather");
}

}  // namespace mtrssm
