// MoPoE-MMTRSSM (two-timescale, MTState) scan kernels for gfx950 (MI355X).
// Same regime as mrssm_scan.hip: one workgroup owns RB rows for the whole sequence, state in LDS,
// weights streamed from L2 every step, no grid-wide synchronisation.
// Reference: mmtrssm/mopoe_mmtrssm/core.py:405-490 (posterior rollout), :496-544 (prior-only).
#include <initializer_list>

#include "scan_common.h"

namespace mtrssm {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);

struct MmtLds {
  // slh = [stoch_l ; stoch_h] contiguous (the MTRNN input vector without the action part)
  int slh, dl0, dl1, hl, dh0, dh1, hh, tmp, l1, h1, hq, lpl, la, lv, mx, lph, lqh, stride;
  __host__ __device__ MmtLds(int LD, int HD, int H, int LS, int HS) {
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    slh = take(LS + HS); dl0 = take(LD); dl1 = take(LD); hl = take(LD);
    dh0 = take(HD); dh1 = take(HD); hh = take(HD); tmp = take(LD > HD ? LD : HD);
    l1 = take(4 * H); h1 = take(2 * H); hq = take(H);
    lpl = take(LS); la = take(LS); lv = take(LS); mx = take(LS); lph = take(HS); lqh = take(HS);
    stride = o;
  }
};

template <int RB, bool POST, bool VEC>
__global__ __launch_bounds__(1024) void mmtrssm_fwd_kernel(const MtrssmMmtrssmDims dm, const MtrssmMmtrssmFwdWeights w, const MtrssmMmtrssmFwdIO io) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int LD = dm.LD, HD = dm.HD, H = dm.H, KL = dm.KL, CL = dm.CL, KH = dm.KH, CH = dm.CH;
  const int LS = KL * CL, HS = KH * CH, T = dm.T, act = dm.act;
  const int NL = POST ? 4 : 1, NHh = POST ? 2 : 1;
  const MmtLds L(LD, HD, H, LS, HS);
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
  int lcur = L.dl0, lnxt = L.dl1, hcur = L.dh0, hnxt = L.dh1;

  for (int rb = 0; rb < RB; ++rb) {
    float* r = lds + rb * L.stride;
    const size_t b = brow[rb];
    for (int i = tid; i < LD; i += blockDim.x) { r[lcur + i] = io.deter_l0[b * LD + i]; r[L.hl + i] = io.hidden_l0[b * LD + i]; }
    for (int i = tid; i < HD; i += blockDim.x) { r[hcur + i] = io.deter_h0[b * HD + i]; r[L.hh + i] = io.hidden_h0[b * HD + i]; }
    for (int i = tid; i < LS; i += blockDim.x) r[L.slh + i] = io.stoch_l0[b * LS + i];
    for (int i = tid; i < HS; i += blockDim.x) r[L.slh + LS + i] = io.stoch_h0[b * HS + i];
  }
  lds_barrier();

  for (int t = 0; t < T; ++t) {
    size_t bt[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) bt[rb] = (size_t)brow[rb] * T + t;

    // (1) both MTRNN cells (core.py:59-60): hidden = keep*hidden + (W_d d_prev + W_x x + b)/tau ; d = tanh(hidden)
    gemv_sk<RB, VEC>(w.wxl_s_t, LD, LS + HS, LD, lds + L.slh, L.stride, red,
               [&](int rb, int o) { return io.xl[bt[rb] * LD + o]; },
               [&](int rb, int o, float a) { lds[rb * L.stride + L.tmp + o] = a; });
    gemv_sk<RB, VEC>(w.wdl_t, LD, LD, LD, lds + lcur, L.stride, red,
               [&](int rb, int o) { return lds[rb * L.stride + L.tmp + o]; },
               [&](int rb, int o, float u) {
                 float* r = lds + rb * L.stride;
                 const float hid = dm.keep_l * r[L.hl + o] + u / dm.tau_l;
                 const float d = tanhf(hid);
                 r[L.hl + o] = hid;
                 r[lnxt + o] = d;
                 if (valid[rb]) { io.deter_l[bt[rb] * LD + o] = d; io.hidden_l[bt[rb] * LD + o] = hid; }
               });
    // (no barrier: gemv_sk ends with one)  // tmp is reused by the higher cell
    gemv_sk<RB, VEC>(w.wxh_t, HD, HS, HD, lds + L.slh + LS, L.stride, red,
               [&](int, int o) { return w.bh[o]; },
               [&](int rb, int o, float a) { lds[rb * L.stride + L.tmp + o] = a; });
    gemv_sk<RB, VEC>(w.wdh_t, HD, HD, HD, lds + hcur, L.stride, red,
               [&](int rb, int o) { return lds[rb * L.stride + L.tmp + o]; },
               [&](int rb, int o, float u) {
                 float* r = lds + rb * L.stride;
                 const float hid = dm.keep_h * r[L.hh + o] + u / dm.tau_h;
                 const float d = tanhf(hid);
                 r[L.hh + o] = hid;
                 r[hnxt + o] = d;
                 if (valid[rb]) { io.deter_h[bt[rb] * HD + o] = d; io.hidden_h[bt[rb] * HD + o] = hid; }
               });
    // (no barrier: gemv_sk ends with one)

    // (2a) layer 0 of every head: on d_l -> [l_prior | audio | vision | h_posterior(l part, raw)], on d_h -> [h_prior | h_posterior(h part, raw)]
    gemv_sk<RB, VEC>(w.wl1_t, NL * H, LD, NL * H, lds + lnxt, L.stride, red,
               [&](int rb, int o) {
                 if (o < H) return w.bl1[o];
                 if (o < 2 * H) return io.pa[bt[rb] * H + (o - H)];
                 if (o < 3 * H) return io.pv[bt[rb] * H + (o - 2 * H)];
                 return 0.f;
               },
               [&](int rb, int o, float a) {
                 const float h = o < 3 * H ? act_fwd(a, act) : a;
                 lds[rb * L.stride + L.l1 + o] = h;
                 if (io.sv_l1 && valid[rb] && o < 3 * H) io.sv_l1[bt[rb] * 4 * H + o] = h;
               });
    gemv_sk<RB, VEC>(w.wh1_t, NHh * H, HD, NHh * H, lds + hnxt, L.stride, red,
               [&](int, int o) { return w.bh1[o]; },
               [&](int rb, int o, float a) {
                 const float h = o < H ? act_fwd(a, act) : a;
                 lds[rb * L.stride + L.h1 + o] = h;
                 if (io.sv_h1 && valid[rb] && o < H) io.sv_h1[bt[rb] * H + o] = h;
               });
    // (no barrier: gemv_sk ends with one)
    // (2b) h_posterior layer 0 = act(l part + h part)   (core.py:315-316: cat([l_deter, h_deter]))
    if (POST) {
      for (int rb = 0; rb < RB; ++rb) {
        float* r = lds + rb * L.stride;
        for (int o = tid; o < H; o += blockDim.x) {
          const float h = act_fwd(r[L.l1 + 3 * H + o] + r[L.h1 + H + o], act);
          r[L.hq + o] = h;
          if (io.sv_l1 && valid[rb]) io.sv_l1[bt[rb] * 4 * H + 3 * H + o] = h;
        }
      }
      lds_barrier();
    }
    // (3) layer 1 of every head (narrow outputs): lpl | la | lv | lph | lqh
    {
      const int n_out = POST ? 3 * LS + 2 * HS : LS + HS;
      for (int o = wave; o < n_out; o += nwave) {
        const float* W; const float* bias; const float* vin; int dst, s;
        if (POST) {
          if (o < LS) { s = o; W = w.wlp2; bias = w.blp2; vin = lds + L.l1; dst = L.lpl; }
          else if (o < 2 * LS) { s = o - LS; W = w.wa2; bias = w.ba2; vin = lds + L.l1 + H; dst = L.la; }
          else if (o < 3 * LS) { s = o - 2 * LS; W = w.wv2; bias = w.bv2; vin = lds + L.l1 + 2 * H; dst = L.lv; }
          else if (o < 3 * LS + HS) { s = o - 3 * LS; W = w.whp2; bias = w.bhp2; vin = lds + L.h1; dst = L.lph; }
          else { s = o - 3 * LS - HS; W = w.whq2; bias = w.bhq2; vin = lds + L.hq; dst = L.lqh; }
        } else {
          if (o < LS) { s = o; W = w.wlp2; bias = w.blp2; vin = lds + L.l1; dst = L.lpl; }
          else { s = o - LS; W = w.whp2; bias = w.bhp2; vin = lds + L.h1; dst = L.lph; }
        }
        float acc[RB];
        wave_dot<RB>(W + (size_t)s * H, H, vin, L.stride, lane, acc);
        if (lane == 0) {
          const float b = bias[s];
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) lds[rb * L.stride + dst + s] = acc[rb] + b;
        }
      }
    }
    lds_barrier();
    // (4) MoPoE on the lower level, categorical blocks on both levels
    for (int rb = wave; rb < RB; rb += nwave) {
      float* r = lds + rb * L.stride;
      const size_t q = bt[rb];
      const bool ok = valid[rb];
      if (POST) wave_mopoe_mix(r + L.la, r + L.lv, r + L.mx, LS, lane);
      if (ok) {
        for (int s = lane; s < LS; s += kWave) {
          io.prior_logits_l[q * LS + s] = r[L.lpl + s];
          if (POST) {
            io.post_logits_l[q * LS + s] = r[L.mx + s];
            if (io.sv_la) { io.sv_la[q * LS + s] = r[L.la + s]; io.sv_lv[q * LS + s] = r[L.lv + s]; }
          }
        }
        for (int s = lane; s < HS; s += kWave) {
          io.prior_logits_h[q * HS + s] = r[L.lph + s];
          if (POST) io.post_logits_h[q * HS + s] = r[L.lqh + s];
        }
      }
      float kll = cat_block_fwd<POST>(r + L.mx, r + L.lpl, KL, CL, lane, POST ? io.u_post_l + q * KL : nullptr,
                                      io.u_prior_l ? io.u_prior_l + q * KL : nullptr, r + L.slh,
                                      POST ? io.post_stoch_l + q * LS : nullptr,
                                      io.prior_stoch_l ? io.prior_stoch_l + q * LS : nullptr, ok);
      float klh = cat_block_fwd<POST>(r + L.lqh, r + L.lph, KH, CH, lane, POST ? io.u_post_h + q * KH : nullptr,
                                      io.u_prior_h ? io.u_prior_h + q * KH : nullptr, r + L.slh + LS,
                                      POST ? io.post_stoch_h + q * HS : nullptr,
                                      io.prior_stoch_h ? io.prior_stoch_h + q * HS : nullptr, ok);
      if (POST) {
        kll = wave_sum(kll);
        klh = wave_sum(klh);
        if (lane == 0 && ok) {
          if (io.kl_l) io.kl_l[q] = kll;
          if (io.kl_h) io.kl_h[q] = klh;
        }
      }
    }
    lds_barrier();
    int tswap = lcur; lcur = lnxt; lnxt = tswap;
    tswap = hcur; hcur = hnxt; hnxt = tswap;
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct MmtBwdLds {
  int c_dl, c_hl, c_dh, c_hh, c_s, dl, dh, dlp, dhp, l1, h1, la, lv, mx, lpl, lqh, lph;
  int dmx, dlpl, dla, dlv, dlqh, dlph, dzl, dzh, dul, duh, stride;
  __host__ __device__ MmtBwdLds(int LD, int HD, int H, int LS, int HS) {
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    c_dl = take(LD); c_hl = take(LD); c_dh = take(HD); c_hh = take(HD); c_s = take(LS + HS);
    dl = take(LD); dh = take(HD); dlp = take(LD); dhp = take(HD);
    l1 = take(4 * H); h1 = take(H);
    la = take(LS); lv = take(LS); mx = take(LS); lpl = take(LS); lqh = take(HS); lph = take(HS);
    dmx = take(LS); dlpl = take(LS); dla = take(LS); dlv = take(LS); dlqh = take(HS); dlph = take(HS);
    dzl = take(4 * H); dzh = take(2 * H); dul = take(LD); duh = take(HD);
    stride = o;
  }
};

template <int RB, bool VEC>
__global__ __launch_bounds__(1024) void mmtrssm_bwd_kernel(const MtrssmMmtrssmDims dm, const MtrssmMmtrssmBwdWeights w, const MtrssmMmtrssmBwdIO io) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int LD = dm.LD, HD = dm.HD, H = dm.H, KL = dm.KL, CL = dm.CL, KH = dm.KH, CH = dm.CH;
  const int LS = KL * CL, HS = KH * CH, T = dm.T, act = dm.act;
  const MmtBwdLds L(LD, HD, H, LS, HS);
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
    for (int i = tid; i < LD; i += blockDim.x) { r[L.c_dl + i] = 0.f; r[L.c_hl + i] = 0.f; }
    for (int i = tid; i < HD; i += blockDim.x) { r[L.c_dh + i] = 0.f; r[L.c_hh + i] = 0.f; }
    for (int i = tid; i < LS + HS; i += blockDim.x) r[L.c_s + i] = 0.f;
  }
  lds_barrier();

  for (int t = T - 1; t >= 0; --t) {
    size_t bt[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) bt[rb] = (size_t)brow[rb] * T + t;

    // (a) stage saved vectors
    for (int rb = 0; rb < RB; ++rb) {
      float* r = lds + rb * L.stride;
      const size_t q = bt[rb];
      for (int s = tid; s < LS; s += blockDim.x) {
        r[L.la + s] = io.sv_la[q * LS + s];
        r[L.lv + s] = io.sv_lv[q * LS + s];
        r[L.mx + s] = io.post_logits_l[q * LS + s];
        r[L.lpl + s] = io.prior_logits_l[q * LS + s];
      }
      for (int s = tid; s < HS; s += blockDim.x) {
        r[L.lqh + s] = io.post_logits_h[q * HS + s];
        r[L.lph + s] = io.prior_logits_h[q * HS + s];
      }
      for (int i = tid; i < 4 * H; i += blockDim.x) r[L.l1 + i] = io.sv_l1[q * 4 * H + i];
      for (int i = tid; i < H; i += blockDim.x) r[L.h1 + i] = io.sv_h1[q * H + i];
      const float* dlsrc = t > 0 ? io.deter_l + (q - 1) * LD : io.deter_l0 + (size_t)brow[rb] * LD;
      const float* dhsrc = t > 0 ? io.deter_h + (q - 1) * HD : io.deter_h0 + (size_t)brow[rb] * HD;
      for (int i = tid; i < LD; i += blockDim.x) { r[L.dl + i] = io.deter_l[q * LD + i]; r[L.dlp + i] = dlsrc[i]; }
      for (int i = tid; i < HD; i += blockDim.x) { r[L.dh + i] = io.deter_h[q * HD + i]; r[L.dhp + i] = dhsrc[i]; }
    }
    lds_barrier();

    // (b) categorical blocks (lower: mixed posterior; higher: plain posterior) + MoPoE backward
    for (int rb = wave; rb < RB; rb += nwave) {
      float* r = lds + rb * L.stride;
      const size_t q = bt[rb];
      cat_block_bwd(r + L.mx, r + L.lpl, KL, CL, lane, io.g_post_stoch_l ? io.g_post_stoch_l + q * LS : nullptr, r + L.c_s,
                    io.g_prior_stoch_l ? io.g_prior_stoch_l + q * LS : nullptr,
                    io.g_post_logits_l ? io.g_post_logits_l + q * LS : nullptr,
                    io.g_prior_logits_l ? io.g_prior_logits_l + q * LS : nullptr, io.g_kl_l ? io.g_kl_l[q] : 0.f,
                    dm.kl_w_post, dm.kl_w_prior, r + L.dmx, r + L.dlpl);
      cat_block_bwd(r + L.lqh, r + L.lph, KH, CH, lane, io.g_post_stoch_h ? io.g_post_stoch_h + q * HS : nullptr, r + L.c_s + LS,
                    io.g_prior_stoch_h ? io.g_prior_stoch_h + q * HS : nullptr,
                    io.g_post_logits_h ? io.g_post_logits_h + q * HS : nullptr,
                    io.g_prior_logits_h ? io.g_prior_logits_h + q * HS : nullptr, io.g_kl_h ? io.g_kl_h[q] : 0.f,
                    dm.kl_w_post, dm.kl_w_prior, r + L.dlqh, r + L.dlph);
      wave_mopoe_mix_bwd(r + L.la, r + L.lv, r + L.mx, r + L.dmx, r + L.dla, r + L.dlv, LS, lane);
      if (valid[rb]) {
        for (int s = lane; s < LS; s += kWave) {
          io.d_lpl[q * LS + s] = r[L.dlpl + s];
          io.d_la[q * LS + s] = r[L.dla + s];
          io.d_lv[q * LS + s] = r[L.dlv + s];
        }
        for (int s = lane; s < HS; s += kWave) {
          io.d_lph[q * HS + s] = r[L.dlph + s];
          io.d_lqh[q * HS + s] = r[L.dlqh + s];
        }
      }
    }
    lds_barrier();

    // (c) layer 1 transposed -> pre-activation grads of layer 0
    auto head_bwd = [&](const float* W, int R, int vin_off, int act_off, int dst_off, int g_off, bool to_l, int dup_off) {
      gemv_sk<RB, VEC>(W, H, R, H, lds + vin_off, L.stride, red, [](int, int) { return 0.f; },
                 [&](int rb, int o, float a) {
                   float* r = lds + rb * L.stride;
                   const float g = a * act_grad_from_out(r[act_off + o], act);
                   r[dst_off + o] = g;
                   if (dup_off >= 0) r[dup_off + o] = g;
                   if (valid[rb]) {
                     if (to_l) io.d_zl1[bt[rb] * 4 * H + g_off + o] = g;
                     else io.d_zh1[bt[rb] * H + g_off + o] = g;
                   }
                 });
    };
    head_bwd(w.wlp2, LS, L.dlpl, L.l1, L.dzl, 0, true, -1);
    head_bwd(w.wa2, LS, L.dla, L.l1 + H, L.dzl + H, H, true, -1);
    head_bwd(w.wv2, LS, L.dlv, L.l1 + 2 * H, L.dzl + 2 * H, 2 * H, true, -1);
    head_bwd(w.whq2, HS, L.dlqh, L.l1 + 3 * H, L.dzl + 3 * H, 3 * H, true, L.dzh + H);
    head_bwd(w.whp2, HS, L.dlph, L.h1, L.dzh, 0, false, -1);
    lds_barrier();

    // (d) grads at d_l / d_h, through tanh into the leaky integrators
    gemv_sk<RB, VEC>(w.wl1, LD, 4 * H, LD, lds + L.dzl, L.stride, red,
               [&](int rb, int o) { return (io.g_deter_l ? io.g_deter_l[bt[rb] * LD + o] : 0.f) + lds[rb * L.stride + L.c_dl + o]; },
               [&](int rb, int o, float dd) {
                 float* r = lds + rb * L.stride;
                 const float d = r[L.dl + o];
                 const float dhid = dd * (1.f - d * d) + (io.g_hidden_l ? io.g_hidden_l[bt[rb] * LD + o] : 0.f) + r[L.c_hl + o];
                 const float du = dhid / dm.tau_l;
                 r[L.dul + o] = du;
                 r[L.c_hl + o] = dhid * dm.keep_l;
                 if (valid[rb]) io.d_ul[bt[rb] * LD + o] = du;
               });
    gemv_sk<RB, VEC>(w.wh1, HD, 2 * H, HD, lds + L.dzh, L.stride, red,
               [&](int rb, int o) { return (io.g_deter_h ? io.g_deter_h[bt[rb] * HD + o] : 0.f) + lds[rb * L.stride + L.c_dh + o]; },
               [&](int rb, int o, float dd) {
                 float* r = lds + rb * L.stride;
                 const float d = r[L.dh + o];
                 const float dhid = dd * (1.f - d * d) + (io.g_hidden_h ? io.g_hidden_h[bt[rb] * HD + o] : 0.f) + r[L.c_hh + o];
                 const float du = dhid / dm.tau_h;
                 r[L.duh + o] = du;
                 r[L.c_hh + o] = dhid * dm.keep_h;
                 if (valid[rb]) io.d_uh[bt[rb] * HD + o] = du;
               });
    // (no barrier: gemv_sk ends with one)

    // (e) carries into step t-1: d_prev via W_d^T ; [stoch_l ; stoch_h] via W_x^T (narrow outputs)
    gemv_sk<RB, VEC>(w.wdl, LD, LD, LD, lds + L.dul, L.stride, red, [](int, int) { return 0.f; },
               [&](int rb, int o, float a) { lds[rb * L.stride + L.c_dl + o] = a; });
    gemv_sk<RB, VEC>(w.wdh, HD, HD, HD, lds + L.duh, L.stride, red, [](int, int) { return 0.f; },
               [&](int rb, int o, float a) { lds[rb * L.stride + L.c_dh + o] = a; });
    for (int s = wave; s < LS + HS; s += nwave) {
      float acc[RB], acc2[RB];
      wave_dot<RB>(w.wxl_s_t + (size_t)s * LD, LD, lds + L.dul, L.stride, lane, acc);
      if (s >= LS) {
        wave_dot<RB>(w.wxh_t + (size_t)(s - LS) * HD, HD, lds + L.duh, L.stride, lane, acc2);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] += acc2[rb];
      }
      if (lane == 0) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) lds[rb * L.stride + L.c_s + s] = acc[rb];
      }
    }
    lds_barrier();
  }

  for (int rb = 0; rb < RB; ++rb) {
    if (!valid[rb]) continue;
    const float* r = lds + rb * L.stride;
    const size_t b = brow[rb];
    for (int i = tid; i < LD; i += blockDim.x) { io.g_deter_l0[b * LD + i] = r[L.c_dl + i]; io.g_hidden_l0[b * LD + i] = r[L.c_hl + i]; }
    for (int i = tid; i < HD; i += blockDim.x) { io.g_deter_h0[b * HD + i] = r[L.c_dh + i]; io.g_hidden_h0[b * HD + i] = r[L.c_hh + i]; }
    for (int i = tid; i < LS; i += blockDim.x) io.g_stoch_l0[b * LS + i] = r[L.c_s + i];
    for (int i = tid; i < HS; i += blockDim.x) io.g_stoch_h0[b * HS + i] = r[L.c_s + LS + i];
  }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
template <typename Kern, typename... Args>
static int launch_mmt(const char* name, Kern kern, int grid, int threads, size_t lds_bytes, hipStream_t stream, Args... args) {
  if (lds_bytes > 160 * 1024) {
    set_error("mmtrssm scan needs %zu bytes of LDS per workgroup (> 160 KiB): dims too large for the row-parallel regime", lds_bytes);
    return MTRSSM_ELDS;
  }
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds_bytes, hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  }
  set_last_kernel(name);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, stream, args...);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("kernel launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  return MTRSSM_OK;
}

static int check_mmt_dims(const MtrssmMmtrssmDims* d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->LD <= 0 || d->HD <= 0 || d->H <= 0 || d->KL <= 0 || d->CL <= 0 || d->KH <= 0 || d->CH <= 0) {
    set_error("mmtrssm: B,T,LD,HD,H,KL,CL,KH,CH must all be positive");
    return MTRSSM_EINVAL;
  }
  if (!(d->tau_l > 1.f) || !(d->tau_h > 1.f)) {
    set_error("mmtrssm: tau must be greater than 1.0");  // core.py:34
    return MTRSSM_EINVAL;
  }
  if (d->act < MTRSSM_ACT_IDENTITY || d->act > MTRSSM_ACT_TANH) {
    set_error("mmtrssm: unknown activation id %d", d->act);
    return MTRSSM_EINVAL;
  }
  if (d->threads < 0 || d->threads > 1024 || d->threads % kWave) {
    set_error("mmtrssm: threads must be a multiple of 64 in [64, 1024] (0 = default)");
    return MTRSSM_EINVAL;
  }
  return MTRSSM_OK;
}

static size_t mmt_red_floats(int threads, int widest) { return (size_t)(4 * threads > widest + 3 ? 4 * threads : widest + 3); }

static int mmt_rows(const MtrssmMmtrssmDims* d, size_t row_bytes, size_t red_bytes) {
  if (d->rows_per_block > 0) return d->rows_per_block;
  int rb = 1;
  while (rb < 4 && (d->B + rb - 1) / rb > 1024) rb *= 2;
  while (rb > 1 && (size_t)rb * (row_bytes + red_bytes) > 160 * 1024) rb /= 2;
  return rb;
}

static bool mmt_aligned16(std::initializer_list<const void*> ptrs) {
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

int mmtrssm_fwd_launch(const MtrssmMmtrssmDims* d, const MtrssmMmtrssmFwdWeights* w, const MtrssmMmtrssmFwdIO* io, hipStream_t stream) {
  if (int rc = check_mmt_dims(d)) return rc;
  if (!w || !io || !io->xl || !io->deter_l0 || !io->deter_h0 || !io->hidden_l0 || !io->hidden_h0 || !io->stoch_l0 || !io->stoch_h0 ||
      !io->deter_l || !io->deter_h || !io->hidden_l || !io->hidden_h || !io->prior_logits_l || !io->prior_logits_h) {
    set_error("mmtrssm_rollout_fwd: null required pointer");
    return MTRSSM_EINVAL;
  }
  if (d->post && (!io->pa || !io->pv || !io->u_post_l || !io->u_post_h || !io->post_logits_l || !io->post_logits_h ||
                  !io->post_stoch_l || !io->post_stoch_h)) {
    set_error("mmtrssm_rollout_fwd: posterior rollout needs pa, pv, u_post_*, post_logits_*, post_stoch_*");
    return MTRSSM_EINVAL;
  }
  if (!d->post && (!io->u_prior_l || !io->u_prior_h || !io->prior_stoch_l || !io->prior_stoch_h)) {
    set_error("mmtrssm_rollout_fwd: prior-only rollout needs u_prior_* and prior_stoch_*");
    return MTRSSM_EINVAL;
  }
  const MmtLds L(d->LD, d->HD, d->H, d->KL * d->CL, d->KH * d->CH);
  const int threads = d->threads > 0 ? d->threads : 1024;
  const int widest = 4 * d->H > d->LD ? (4 * d->H > d->HD ? 4 * d->H : d->HD) : (d->LD > d->HD ? d->LD : d->HD);
  const size_t red = mmt_red_floats(threads, widest) * sizeof(float);
  const int rb = mmt_rows(d, L.stride * sizeof(float), red);
  const int grid = (d->B + rb - 1) / rb;
  const size_t lds = (size_t)rb * (L.stride * sizeof(float) + red);
  if (rb > threads / kWave) { set_error("rows_per_block %d exceeds waves per block %d", rb, threads / kWave); return MTRSSM_EINVAL; }
  const bool vec = d->LD % 4 == 0 && d->HD % 4 == 0 && d->H % 4 == 0 &&
                   mmt_aligned16({w->wxl_s_t, w->wdl_t, w->wxh_t, w->wdh_t, w->wl1_t, w->wh1_t});
#define MTRSSM_VARIANT(R, P, V) \
  launch_mmt("mtrssm::mmtrssm_fwd_kernel<" #R ", " #P ", " #V ">", mmtrssm_fwd_kernel<R, P, V>, grid, threads, lds, stream, *d, *w, *io)
#define MTRSSM_CASE(R)                                                                      \
  case R:                                                                                    \
    if (d->post) return vec ? MTRSSM_VARIANT(R, true, true) : MTRSSM_VARIANT(R, true, false); \
    return vec ? MTRSSM_VARIANT(R, false, true) : MTRSSM_VARIANT(R, false, false);
  switch (rb) {
    MTRSSM_CASE(1)
    MTRSSM_CASE(2)
    MTRSSM_CASE(4)
    default: set_error("rows_per_block must be 1, 2 or 4 (got %d)", rb); return MTRSSM_EINVAL;
  }
#undef MTRSSM_CASE
#undef MTRSSM_VARIANT
}

int mmtrssm_bwd_launch(const MtrssmMmtrssmDims* d, const MtrssmMmtrssmBwdWeights* w, const MtrssmMmtrssmBwdIO* io, hipStream_t stream) {
  if (int rc = check_mmt_dims(d)) return rc;
  if (!w || !io || !io->deter_l0 || !io->deter_h0 || !io->deter_l || !io->deter_h || !io->prior_logits_l || !io->prior_logits_h ||
      !io->post_logits_l || !io->post_logits_h || !io->sv_l1 || !io->sv_h1 || !io->sv_la || !io->sv_lv || !io->g_deter_l0 ||
      !io->g_deter_h0 || !io->g_hidden_l0 || !io->g_hidden_h0 || !io->g_stoch_l0 || !io->g_stoch_h0 || !io->d_ul || !io->d_uh ||
      !io->d_zl1 || !io->d_zh1 || !io->d_lpl || !io->d_la || !io->d_lv || !io->d_lph || !io->d_lqh) {
    set_error("mmtrssm_rollout_bwd: null required pointer");
    return MTRSSM_EINVAL;
  }
  const MmtBwdLds L(d->LD, d->HD, d->H, d->KL * d->CL, d->KH * d->CH);
  const int threads = d->threads > 0 ? d->threads : 1024;
  const int widest = d->H > d->LD ? (d->H > d->HD ? d->H : d->HD) : (d->LD > d->HD ? d->LD : d->HD);
  const size_t red = mmt_red_floats(threads, widest) * sizeof(float);
  const int rb = mmt_rows(d, L.stride * sizeof(float), red);
  const int grid = (d->B + rb - 1) / rb;
  const size_t lds = (size_t)rb * (L.stride * sizeof(float) + red);
  if (rb > threads / kWave) { set_error("rows_per_block %d exceeds waves per block %d", rb, threads / kWave); return MTRSSM_EINVAL; }
  const bool vec = d->LD % 4 == 0 && d->HD % 4 == 0 && d->H % 4 == 0 &&
                   mmt_aligned16({w->wdl, w->wdh, w->wl1, w->wh1, w->wlp2, w->wa2, w->wv2, w->whp2, w->whq2});
#define MTRSSM_CASE(R)                                                                                                         \
  case R:                                                                                                                       \
    return vec ? launch_mmt("mtrssm::mmtrssm_bwd_kernel<" #R ", true>", mmtrssm_bwd_kernel<R, true>, grid, threads, lds, stream, *d, *w, *io) \
               : launch_mmt("mtrssm::mmtrssm_bwd_kernel<" #R ", false>", mmtrssm_bwd_kernel<R, false>, grid, threads, lds, stream, *d, *w, *io);
  switch (rb) {
    MTRSSM_CASE(1)
    MTRSSM_CASE(2)
    MTRSSM_CASE(4)
    default: set_error("rows_per_block must be 1, 2 or 4 (got %d)", rb); return MTRSSM_EINVAL;
  }
#undef MTRSSM_CASE
}

}  // namespace mtrssm
