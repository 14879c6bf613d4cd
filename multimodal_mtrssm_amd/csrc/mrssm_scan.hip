// MoPoE-MRSSM scan kernels for gfx950 (MI355X).
//
// One workgroup owns RB batch rows for the WHOLE sequence: rows never interact
// (mrssm/mopoe_mrssm/core.py:221-256 has no cross-row op), so there is no grid-wide
// synchronisation.  The recurrent state (deter, stoch, every intermediate vector of the step) lives
// in LDS; weights are re-streamed from L2 every step (1.7 MB fp32 at D=H=200); the only HBM traffic
// is the per-step xa/pa/pv/u reads and the state / saved-activation writes.  The kernels are
// latency-bound by the T-step dependency chain, not by a roofline.
#include <initializer_list>

#include "scan_common.h"

namespace mtrssm {

struct MrssmLds {
  int s, dp, dn, h1, h2, gi, gh, hd, lp, la, lv, mx, stride;
  __host__ __device__ MrssmLds(int D, int H, int S) {
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    s = take(S); dp = take(D); dn = take(D); h1 = take(H); h2 = take(H);
    gi = take(3 * D); gh = take(3 * D); hd = take(3 * H);
    lp = take(S); la = take(S); lv = take(S); mx = take(S);
    stride = o;
  }
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int RB, bool POST, bool VEC>
__global__ __launch_bounds__(1024) void mrssm_fwd_kernel(const MtrssmMrssmDims dm, const MtrssmMrssmFwdWeights w, const MtrssmMrssmFwdIO io) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = dm.D, H = dm.H, K = dm.K, C = dm.C, S = K * C, T = dm.T, act = dm.act;
  const int NH = POST ? 3 : 1;
  const MrssmLds L(D, H, S);
  float* red = lds + RB * L.stride;  // split-K partial sums
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave, nwave = blockDim.x / kWave;
  const int row0 = blockIdx.x * RB;
  int brow[RB];
  bool valid[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    valid[rb] = row0 + rb < dm.B;
    brow[rb] = valid[rb] ? row0 + rb : dm.B - 1;
  }
  int cur = L.dp, nxt = L.dn;  // toggled every step

  for (int rb = 0; rb < RB; ++rb) {
    float* r = lds + rb * L.stride;
    for (int i = tid; i < D; i += blockDim.x) r[cur + i] = io.deter0[(size_t)brow[rb] * D + i];
    for (int i = tid; i < S; i += blockDim.x) r[L.s + i] = io.stoch0[(size_t)brow[rb] * S + i];
  }
  lds_barrier();

  for (int t = 0; t < T; ++t) {
    size_t bt[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) bt[rb] = (size_t)brow[rb] * T + t;

    // (1) h1 = act(xa + W1s s)                                   networks.py:165-166 (first Linear + act)
    gemv_sk<RB, VEC>(w.w1s_t, H, S, H, lds + L.s, L.stride, red,
               [&](int rb, int o) { return io.xa[bt[rb] * H + o]; },
               [&](int rb, int o, float a) {
                 const float h = act_fwd(a, act);
                 lds[rb * L.stride + L.h1 + o] = h;
                 if (io.sv_h1 && valid[rb]) io.sv_h1[bt[rb] * H + o] = h;
               });
    // (no barrier: gemv_sk ends with one)
    // (2) h2 = W2 h1 + b2                                         networks.py:166 (second Linear)
    gemv_sk<RB, VEC>(w.w2_t, H, H, H, lds + L.h1, L.stride, red,
               [&](int, int o) { return w.b2[o]; },
               [&](int rb, int o, float a) {
                 lds[rb * L.stride + L.h2 + o] = a;
                 if (io.sv_h2 && valid[rb]) io.sv_h2[bt[rb] * H + o] = a;
               });
    // (no barrier: gemv_sk ends with one)
    // (3) gi = W_ih h2 + b_ih ; gh = W_hh d_prev + b_hh           networks.py:170 (nn.GRUCell)
    gemv_sk<RB, VEC>(w.wih_t, 3 * D, H, 3 * D, lds + L.h2, L.stride, red,
               [&](int, int o) { return w.bih[o]; },
               [&](int rb, int o, float a) { lds[rb * L.stride + L.gi + o] = a; });
    gemv_sk<RB, VEC>(w.whh_t, 3 * D, D, 3 * D, lds + cur, L.stride, red,
               [&](int, int o) { return w.bhh[o]; },
               [&](int rb, int o, float a) { lds[rb * L.stride + L.gh + o] = a; });
    // (no barrier: gemv_sk ends with one)
    // (4) gates: r, z, n ; d = (d_prev - n) z + n
    for (int rb = 0; rb < RB; ++rb) {
      float* r_ = lds + rb * L.stride;
      for (int i = tid; i < D; i += blockDim.x) {
        const float rg = sigmoidf_(r_[L.gh + i] + r_[L.gi + i]);
        const float zg = sigmoidf_(r_[L.gh + D + i] + r_[L.gi + D + i]);
        const float ghn = r_[L.gh + 2 * D + i];
        const float ng = tanhf(r_[L.gi + 2 * D + i] + ghn * rg);
        const float dnew = (r_[cur + i] - ng) * zg + ng;
        r_[nxt + i] = dnew;
        if (valid[rb]) {
          io.deter[bt[rb] * D + i] = dnew;
          if (io.sv_gates) {
            float* g = io.sv_gates + bt[rb] * 4 * D;
            g[i] = rg; g[D + i] = zg; g[2 * D + i] = ng; g[3 * D + i] = ghn;
          }
        }
      }
    }
    lds_barrier();
    // (5) head layer 0 on the new deter: prior | audio | vision   networks.py:171, 82 ; core.py:82
    gemv_sk<RB, VEC>(w.wh1_t, NH * H, D, NH * H, lds + nxt, L.stride, red,
               [&](int rb, int o) {
                 if (o < H) return w.b3[o];
                 if (o < 2 * H) return io.pa[bt[rb] * H + (o - H)];
                 return io.pv[bt[rb] * H + (o - 2 * H)];
               },
               [&](int rb, int o, float a) {
                 const float h = act_fwd(a, act);
                 lds[rb * L.stride + L.hd + o] = h;
                 if (io.sv_heads && valid[rb]) io.sv_heads[bt[rb] * 3 * H + o] = h;
               });
    // (no barrier: gemv_sk ends with one)
    // (6) head layer 1: three narrow [S x H] products, one wave per output
    for (int o = wave; o < NH * S; o += nwave) {
      const int which = o / S, s = o - which * S;
      const float* W = which == 0 ? w.w4 : (which == 1 ? w.wa2 : w.wv2);
      const float* bias = which == 0 ? w.b4 : (which == 1 ? w.ba2 : w.bv2);
      float acc[RB];
      wave_dot<RB>(W + (size_t)s * H, H, lds + L.hd + which * H, L.stride, lane, acc);
      if (lane == 0) {
        const float b = bias[s];
        const int dst = which == 0 ? L.lp : (which == 1 ? L.la : L.lv);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) lds[rb * L.stride + dst + s] = acc[rb] + b;
      }
    }
    lds_barrier();
    // (7) fusion, per-categorical softmax, KL, sampling: wave rb handles row rb
    for (int rb = wave; rb < RB; rb += nwave) {
      float* r_ = lds + rb * L.stride;
      const size_t q = bt[rb];
      const bool ok = valid[rb];
      if (POST) wave_mopoe_mix(r_ + L.la, r_ + L.lv, r_ + L.mx, S, lane);
      for (int s = lane; s < S; s += kWave) {
        if (ok) {
          io.prior_logits[q * S + s] = r_[L.lp + s];
          if (POST) {
            io.post_logits[q * S + s] = r_[L.mx + s];
            if (io.sv_la) { io.sv_la[q * S + s] = r_[L.la + s]; io.sv_lv[q * S + s] = r_[L.lv + s]; }
          }
        }
      }
      float kl = cat_block_fwd<POST>(r_ + L.mx, r_ + L.lp, K, C, lane, POST ? io.u_post + q * K : nullptr,
                                     io.u_prior ? io.u_prior + q * K : nullptr, r_ + L.s,
                                     POST ? io.post_stoch + q * S : nullptr,
                                     io.prior_stoch ? io.prior_stoch + q * S : nullptr, ok);
      if (POST && io.kl) {
        kl = wave_sum(kl);
        if (lane == 0 && ok) io.kl[q] = kl;
      }
    }
    lds_barrier();
    const int tmp = cur; cur = nxt; nxt = tmp;
  }
}

// ------------------------------------------------------------------------------------------------
// backward (reverse-time scan)
// ------------------------------------------------------------------------------------------------
struct MrssmBwdLds {
  // carried: cd (grad wrt deter_t from step t+1), cs (grad wrt stoch_t from step t+1)
  int cd, cs, dprev, la, lv, mx, lp, dmx, dlp, dla, dlv, hd, dzh, dd, gate, dgi, dgh, dh2, h1, dz1, stride;
  __host__ __device__ MrssmBwdLds(int D, int H, int S) {
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    cd = take(D); cs = take(S); dprev = take(D);
    la = take(S); lv = take(S); mx = take(S); lp = take(S);
    dmx = take(S); dlp = take(S); dla = take(S); dlv = take(S);
    hd = take(3 * H); dzh = take(3 * H); dd = take(D); gate = take(4 * D);
    dgi = take(3 * D); dgh = take(3 * D); dh2 = take(H); h1 = take(H); dz1 = take(H);
    stride = o;
  }
};

template <int RB, bool VEC>
__global__ __launch_bounds__(1024) void mrssm_bwd_kernel(const MtrssmMrssmDims dm, const MtrssmMrssmBwdWeights w, const MtrssmMrssmBwdIO io) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = dm.D, H = dm.H, K = dm.K, C = dm.C, S = K * C, T = dm.T, act = dm.act;
  const MrssmBwdLds L(D, H, S);
  float* red = lds + RB * L.stride;  // split-K partial sums
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave, nwave = blockDim.x / kWave;
  const int row0 = blockIdx.x * RB;
  int brow[RB];
  bool valid[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    valid[rb] = row0 + rb < dm.B;
    brow[rb] = valid[rb] ? row0 + rb : dm.B - 1;
  }
  for (int rb = 0; rb < RB; ++rb) {
    float* r = lds + rb * L.stride;
    for (int i = tid; i < D; i += blockDim.x) r[L.cd + i] = 0.f;
    for (int i = tid; i < S; i += blockDim.x) r[L.cs + i] = 0.f;
  }
  lds_barrier();

  for (int t = T - 1; t >= 0; --t) {
    size_t bt[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) bt[rb] = (size_t)brow[rb] * T + t;

    // (a) stage this step's saved vectors
    for (int rb = 0; rb < RB; ++rb) {
      float* r = lds + rb * L.stride;
      const size_t q = bt[rb];
      for (int s = tid; s < S; s += blockDim.x) {
        r[L.la + s] = io.sv_la[q * S + s];
        r[L.lv + s] = io.sv_lv[q * S + s];
        r[L.mx + s] = io.post_logits[q * S + s];
        r[L.lp + s] = io.prior_logits[q * S + s];
      }
      for (int i = tid; i < 3 * H; i += blockDim.x) r[L.hd + i] = io.sv_heads[q * 3 * H + i];
      for (int i = tid; i < 4 * D; i += blockDim.x) r[L.gate + i] = io.sv_gates[q * 4 * D + i];
      for (int i = tid; i < H; i += blockDim.x) r[L.h1 + i] = io.sv_h1[q * H + i];
      const float* dsrc = t > 0 ? io.deter + (q - 1) * D : io.deter0 + (size_t)brow[rb] * D;
      for (int i = tid; i < D; i += blockDim.x) r[L.dprev + i] = dsrc[i];
    }
    lds_barrier();

    // (b) categorical block: straight-through sample, KL, per-categorical softmax, MoE/PoE, flat log-softmax
    for (int rb = wave; rb < RB; rb += nwave) {
      float* r = lds + rb * L.stride;
      const size_t q = bt[rb];
      const float gk = io.g_kl ? io.g_kl[q] : 0.f;
      cat_block_bwd(r + L.mx, r + L.lp, K, C, lane, io.g_post_stoch ? io.g_post_stoch + q * S : nullptr, r + L.cs,
                    io.g_prior_stoch ? io.g_prior_stoch + q * S : nullptr,
                    io.g_post_logits ? io.g_post_logits + q * S : nullptr,
                    io.g_prior_logits ? io.g_prior_logits + q * S : nullptr, gk, dm.kl_w_post, dm.kl_w_prior,
                    r + L.dmx, r + L.dlp);
      // back through logsumexp over {A, V, A+V} and the two flat log-softmaxes
      wave_mopoe_mix_bwd(r + L.la, r + L.lv, r + L.mx, r + L.dmx, r + L.dla, r + L.dlv, S, lane);
      if (valid[rb]) {
        for (int s = lane; s < S; s += kWave) {
          io.d_la[q * S + s] = r[L.dla + s];
          io.d_lv[q * S + s] = r[L.dlv + s];
          io.d_lp[q * S + s] = r[L.dlp + s];
        }
      }
    }
    lds_barrier();

    // (c) head layer 1 transposed: dzh[which][j] = act'(hd) * sum_s W[s][j] dl[s]
    gemv_sk<RB, VEC>(w.w4, H, S, H, lds + L.dlp, L.stride, red, [](int, int) { return 0.f; },
               [&](int rb, int o, float a) {
                 const float g = a * act_grad_from_out(lds[rb * L.stride + L.hd + o], act);
                 lds[rb * L.stride + L.dzh + o] = g;
                 if (valid[rb]) io.d_zh[bt[rb] * 3 * H + o] = g;
               });
    gemv_sk<RB, VEC>(w.wa2, H, S, H, lds + L.dla, L.stride, red, [](int, int) { return 0.f; },
               [&](int rb, int o, float a) {
                 const float g = a * act_grad_from_out(lds[rb * L.stride + L.hd + H + o], act);
                 lds[rb * L.stride + L.dzh + H + o] = g;
                 if (valid[rb]) io.d_zh[bt[rb] * 3 * H + H + o] = g;
               });
    gemv_sk<RB, VEC>(w.wv2, H, S, H, lds + L.dlv, L.stride, red, [](int, int) { return 0.f; },
               [&](int rb, int o, float a) {
                 const float g = a * act_grad_from_out(lds[rb * L.stride + L.hd + 2 * H + o], act);
                 lds[rb * L.stride + L.dzh + 2 * H + o] = g;
                 if (valid[rb]) io.d_zh[bt[rb] * 3 * H + 2 * H + o] = g;
               });
    // (no barrier: gemv_sk ends with one)

    // (d) dd = g_deter + carry + Wh1^T dzh ; then the GRU gate gradients
    gemv_sk<RB, VEC>(w.wh1, D, 3 * H, D, lds + L.dzh, L.stride, red,
               [&](int rb, int o) {
                 return (io.g_deter ? io.g_deter[bt[rb] * D + o] : 0.f) + lds[rb * L.stride + L.cd + o];
               },
               [&](int rb, int o, float dd) {
                 float* r = lds + rb * L.stride;
                 const float rg = r[L.gate + o], zg = r[L.gate + D + o], ng = r[L.gate + 2 * D + o], ghn = r[L.gate + 3 * D + o];
                 const float dprev = r[L.dprev + o];
                 const float dn = dd * (1.f - zg);
                 const float dz = dd * (dprev - ng);
                 const float dn_pre = dn * (1.f - ng * ng);
                 const float dr = dn_pre * ghn;
                 const float dr_pre = dr * rg * (1.f - rg);
                 const float dz_pre = dz * zg * (1.f - zg);
                 r[L.dd + o] = dd * zg;  // direct path into d_prev
                 r[L.dgi + o] = dr_pre; r[L.dgi + D + o] = dz_pre; r[L.dgi + 2 * D + o] = dn_pre;
                 r[L.dgh + o] = dr_pre; r[L.dgh + D + o] = dz_pre; r[L.dgh + 2 * D + o] = dn_pre * rg;
                 if (valid[rb]) {
                   float* gi = io.d_gi + bt[rb] * 3 * D;
                   float* gh = io.d_gh + bt[rb] * 3 * D;
                   gi[o] = dr_pre; gi[D + o] = dz_pre; gi[2 * D + o] = dn_pre;
                   gh[o] = dr_pre; gh[D + o] = dz_pre; gh[2 * D + o] = dn_pre * rg;
                 }
               });
    // (no barrier: gemv_sk ends with one)

    // (e) carry_d = dd z + W_hh^T dgh ; dh2 = W_ih^T dgi
    gemv_sk<RB, VEC>(w.whh, D, 3 * D, D, lds + L.dgh, L.stride, red,
               [&](int rb, int o) { return lds[rb * L.stride + L.dd + o]; },
               [&](int rb, int o, float a) { lds[rb * L.stride + L.cd + o] = a; });
    gemv_sk<RB, VEC>(w.wih, H, 3 * D, H, lds + L.dgi, L.stride, red, [](int, int) { return 0.f; },
               [&](int rb, int o, float a) {
                 lds[rb * L.stride + L.dh2 + o] = a;
                 if (valid[rb]) io.d_h2[bt[rb] * H + o] = a;
               });
    // (no barrier: gemv_sk ends with one)
    // (f) dz1 = act'(h1) * W2^T dh2
    gemv_sk<RB, VEC>(w.w2, H, H, H, lds + L.dh2, L.stride, red, [](int, int) { return 0.f; },
               [&](int rb, int o, float a) {
                 const float g = a * act_grad_from_out(lds[rb * L.stride + L.h1 + o], act);
                 lds[rb * L.stride + L.dz1 + o] = g;
                 if (valid[rb]) io.d_z1[bt[rb] * H + o] = g;
               });
    // (no barrier: gemv_sk ends with one)
    // (g) carry_s[s] = sum_j W1s[j][s] dz1[j]  (narrow output: one wave per s, rows of W1s^T)
    for (int s = wave; s < S; s += nwave) {
      float acc[RB];
      wave_dot<RB>(w.w1s_t + (size_t)s * H, H, lds + L.dz1, L.stride, lane, acc);
      if (lane == 0) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) lds[rb * L.stride + L.cs + s] = acc[rb];
      }
    }
    lds_barrier();
  }

  for (int rb = 0; rb < RB; ++rb) {
    if (!valid[rb]) continue;
    const float* r = lds + rb * L.stride;
    for (int i = tid; i < D; i += blockDim.x) io.g_deter0[(size_t)brow[rb] * D + i] = r[L.cd + i];
    for (int i = tid; i < S; i += blockDim.x) io.g_stoch0[(size_t)brow[rb] * S + i] = r[L.cs + i];
  }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);

static int pick_rows(int B, int requested, size_t bytes_per_row, size_t extra_bytes) {
  if (requested > 0) return requested;
  int rb = 1;
  // keep the grid near the CU count; grow the row tile only for big batches
  while (rb < 4 && (B + rb - 1) / rb > 1024) rb *= 2;
  while (rb > 1 && (size_t)rb * (bytes_per_row + extra_bytes) > 160 * 1024) rb /= 2;
  return rb;
}

static bool aligned16(std::initializer_list<const void*> ptrs) {
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

template <typename Kern, typename... Args>
static int launch(const char* name, Kern kern, int grid, int threads, size_t lds_bytes, hipStream_t stream, Args... args) {
  if (lds_bytes > 160 * 1024) {
    set_error("scan kernel needs %zu bytes of LDS per workgroup (> 160 KiB): dims too large for the row-parallel regime", lds_bytes);
    return MTRSSM_ELDS;
  }
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds_bytes, hipGetErrorString(e));
      return MTRSSM_ELAUNCH;
    }
  }
  set_last_kernel(name);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, stream, args...);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("kernel launch failed: %s", hipGetErrorString(e));
    return MTRSSM_ELAUNCH;
  }
  return MTRSSM_OK;
}

static int check_dims(const MtrssmMrssmDims* d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->D <= 0 || d->H <= 0 || d->K <= 0 || d->C <= 0) {
    set_error("mrssm: B,T,D,H,K,C must all be positive");
    return MTRSSM_EINVAL;
  }
  if (d->act < MTRSSM_ACT_IDENTITY || d->act > MTRSSM_ACT_TANH) {
    set_error("mrssm: unknown activation id %d", d->act);
    return MTRSSM_EINVAL;
  }
  if (d->threads < 0 || d->threads > 1024 || d->threads % kWave) {
    set_error("mrssm: threads must be a multiple of 64 in [64, 1024] (0 = default)");
    return MTRSSM_EINVAL;
  }
  return MTRSSM_OK;
}

// split-K scratch: KS * RB * padded O floats <= RB * max(4 * threads, widest output + 3)
static size_t red_floats(int threads, int widest) { return (size_t)(4 * threads > widest + 3 ? 4 * threads : widest + 3); }

int mrssm_fwd_launch(const MtrssmMrssmDims* d, const MtrssmMrssmFwdWeights* w, const MtrssmMrssmFwdIO* io, hipStream_t stream) {
  if (int rc = check_dims(d)) return rc;
  if (!w || !io || !io->xa || !io->deter0 || !io->stoch0 || !io->deter || !io->prior_logits) {
    set_error("mrssm_rollout_fwd: null required pointer");
    return MTRSSM_EINVAL;
  }
  if (d->post && (!io->pa || !io->pv || !io->u_post || !io->post_logits || !io->post_stoch)) {
    set_error("mrssm_rollout_fwd: posterior rollout needs pa, pv, u_post, post_logits, post_stoch");
    return MTRSSM_EINVAL;
  }
  if (!d->post && (!io->u_prior || !io->prior_stoch)) {
    set_error("mrssm_rollout_fwd: prior-only rollout needs u_prior and prior_stoch");
    return MTRSSM_EINVAL;
  }
  const MrssmLds L(d->D, d->H, d->K * d->C);
  const int threads = d->threads > 0 ? d->threads : 1024;
  const size_t red = red_floats(threads, 3 * (d->D > d->H ? d->D : d->H)) * sizeof(float);
  const int rb = pick_rows(d->B, d->rows_per_block, L.stride * sizeof(float), red);
  const int grid = (d->B + rb - 1) / rb;
  const size_t lds = (size_t)rb * (L.stride * sizeof(float) + red);
  if (rb > threads / kWave) {
    set_error("rows_per_block %d exceeds waves per block %d", rb, threads / kWave);
    return MTRSSM_EINVAL;
  }
  const bool vec = d->D % 4 == 0 && d->H % 4 == 0 && aligned16({w->w1s_t, w->w2_t, w->wih_t, w->whh_t, w->wh1_t});
#define MTRSSM_FWD_VARIANT(R, P, V) \
  launch("mtrssm::mrssm_fwd_kernel<" #R ", " #P ", " #V ">", mrssm_fwd_kernel<R, P, V>, grid, threads, lds, stream, *d, *w, *io)
#define MTRSSM_FWD_CASE(R)                                                                       \
  case R:                                                                                         \
    if (d->post) return vec ? MTRSSM_FWD_VARIANT(R, true, true) : MTRSSM_FWD_VARIANT(R, true, false); \
    return vec ? MTRSSM_FWD_VARIANT(R, false, true) : MTRSSM_FWD_VARIANT(R, false, false);
  switch (rb) {
    MTRSSM_FWD_CASE(1)
    MTRSSM_FWD_CASE(2)
    MTRSSM_FWD_CASE(4)
    default:
      set_error("rows_per_block must be 1, 2 or 4 (got %d)", rb);
      return MTRSSM_EINVAL;
  }
#undef MTRSSM_FWD_CASE
#undef MTRSSM_FWD_VARIANT
}

int mrssm_bwd_launch(const MtrssmMrssmDims* d, const MtrssmMrssmBwdWeights* w, const MtrssmMrssmBwdIO* io, hipStream_t stream) {
  if (int rc = check_dims(d)) return rc;
  if (!w || !io || !io->deter0 || !io->deter || !io->prior_logits || !io->post_logits || !io->sv_h1 || !io->sv_gates ||
      !io->sv_heads || !io->sv_la || !io->sv_lv || !io->g_deter0 || !io->g_stoch0 || !io->d_z1 || !io->d_h2 || !io->d_gi ||
      !io->d_gh || !io->d_zh || !io->d_lp || !io->d_la || !io->d_lv) {
    set_error("mrssm_rollout_bwd: null required pointer");
    return MTRSSM_EINVAL;
  }
  const MrssmBwdLds L(d->D, d->H, d->K * d->C);
  const int threads = d->threads > 0 ? d->threads : 1024;
  const size_t red = red_floats(threads, d->D > d->H ? d->D : d->H) * sizeof(float);
  const int rb = pick_rows(d->B, d->rows_per_block, L.stride * sizeof(float), red);
  const int grid = (d->B + rb - 1) / rb;
  const size_t lds = (size_t)rb * (L.stride * sizeof(float) + red);
  if (rb > threads / kWave) {
    set_error("rows_per_block %d exceeds waves per block %d", rb, threads / kWave);
    return MTRSSM_EINVAL;
  }
  const bool vec = d->D % 4 == 0 && d->H % 4 == 0 && aligned16({w->w2, w->wih, w->whh, w->wh1, w->w4, w->wa2, w->wv2});
#define MTRSSM_BWD_CASE(R)                                                                                              \
  case R:                                                                                                                \
    return vec ? launch("mtrssm::mrssm_bwd_kernel<" #R ", true>", mrssm_bwd_kernel<R, true>, grid, threads, lds, stream, *d, *w, *io) \
               : launch("mtrssm::mrssm_bwd_kernel<" #R ", false>", mrssm_bwd_kernel<R, false>, grid, threads, lds, stream, *d, *w, *io);
  switch (rb) {
    MTRSSM_BWD_CASE(1)
    MTRSSM_BWD_CASE(2)
    MTRSSM_BWD_CASE(4)
    default:
      set_error("rows_per_block must be 1, 2 or 4 (got %d)", rb);
      return MTRSSM_EINVAL;
  }
#undef MTRSSM_BWD_CASE
}

}  // namespace mtrssm
