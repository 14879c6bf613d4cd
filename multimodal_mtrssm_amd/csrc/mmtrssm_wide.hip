// MoPoE-MMTRSSM (two-timescale, MTState) scan on ALL compute units: the wide form of mmtrssm_scan.hip (wide_common.h has the
// regime, the packed layouts, the exchange vectors and the two-level grid barrier).  One tile of up to 64 batch rows (four
// 16-row MFMA tiles) per pass; every matrix product of a timestep is cut into 16-column output tiles, one workgroup each.
// Replaces the loop body of mmtrssm/mopoe_mmtrssm/core.py:405-490 and its BPTT where the one-CU-per-row form takes 38 / 46 us
// per timestep at ld = hd = H = 200 (BASELINE configs[2]).
//
// State exchange vector XS (double-buffered): k = [ d_l (LD, padded to 32) | d_h (HD, padded) | s_l ; s_h (LS + HS, padded) ].
// Forward, per timestep (four grid barriers):
//   F0 [one workgroup per batch row]  finish step t-1: logits -> MoPoE mix (lower level) -> both categorical blocks (KL, samples);
//                                     the samples go into XS's s segment
//   F1 [one per 16 units of d_l / d_h] both MTRNN cells: u = [W_d | W_x] . XS (+ xl / bias), hidden = keep hidden + u / tau,
//                                     d = tanh(hidden)  (the workgroup keeps its hidden units in registers across steps)
//   F2 [one per 16 head units]        layer 0 of the five heads on [d_l | d_h]: l_prior, audio, vision, h_posterior (both
//                                     halves in one product), h_prior
//   F3 [one per 16 logits]            layer 1 of the five heads
// Backward, per timestep (four barriers): R0 categorical blocks + mix backward per row; R1 pre-activation gradients of the
// five layer-0 heads; R2 gradients at d_l / d_h through tanh into the leaky integrators (du); R3 the carries into step t-1:
// W_d^T du per level (kept by the workgroup of R2) and W_x^T du -> [s_l ; s_h].
#include <cstdlib>

#include "wide_common.h"

namespace mtrssm {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);
int device_cu_count();
WidePackJob wide_make_job(const float* src, long sn, long sk, int N, int K, uint4* dst);
WidePackJob wide_make_block(const float* src, long sn, long sk, int N, int K, int nskip, int kskip, int nt0, int ks0, int KST, uint4* dst);
int wide_launch_pack(const WidePackJobs& jobs, int pieces, hipStream_t stream);

// Development aid (tools/wide_probe.py mmt-fwd | mmt-bwd): as g_wide_prof of mrssm_wide.hip.
__device__ unsigned long long* g_mmt_prof = nullptr;
#define MTRSSM_MMT_STAMP(i)                                                                                               \
  do {                                                                                                                    \
    if (prof && tstamp >= 8 && tstamp < 12) prof[((size_t)blockIdx.x * 4 + (tstamp - 8)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)

constexpr int kMRT = 4;             // row tiles per pass
constexpr int kMRows = 16 * kMRT;   // 64 batch rows
constexpr int kMNS = 3;             // register stages of the operand ring

struct MmtWideGeom {
  int LD, HD, H, LS, HS;
  int NTL, NTHd, HP, NTHP;   // tiles of d_l, d_h; head width padded to 16 and its tiles
  int KLD, KHD, KSS;         // k extents (multiples of 32) of the d_l, d_h and s segments of XS
  int KT;                    // KLD + KHD + KSS
  int HK;                    // head width padded to 32 (k segment of a head in the backward exchange vector)
  int LSp, HSp;              // stochastic sizes padded to 16
  __host__ __device__ MmtWideGeom(const MtrssmMmtrssmDims& d) {
    LD = d.LD; HD = d.HD; H = d.H; LS = d.KL * d.CL; HS = d.KH * d.CH;
    NTL = (LD + 15) / 16; NTHd = (HD + 15) / 16; HP = (H + 15) / 16 * 16; NTHP = HP / 16;
    KLD = (LD + 31) / 32 * 32; KHD = (HD + 31) / 32 * 32; KSS = (LS + HS + 31) / 32 * 32;
    KT = KLD + KHD + KSS;
    HK = (H + 31) / 32 * 32;
    LSp = (LS + 15) / 16 * 16; HSp = (HS + 15) / 16 * 16;
  }
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
struct MmtWideFwdArgs {
  MtrssmMmtrssmDims dm;
  MtrssmMmtrssmFwdWeights w;
  MtrssmMmtrssmFwdIO io;
  const uint4 *pk_rnn, *pk_l0, *pk_l1[5];   // packed weights: both cells; layer 0 of the five heads; layer 1 (lpl, la, lv, lqh, lph)
  uint4 *xs[2], *xh[5];                      // exchange vectors: state (double-buffered); layer-0 activations of the five heads
  float* lg;                                 // [64][3 LSp + 2 HSp] raw logits of the step (without bias): lpl | la | lv | lqh | lph
  void* ctl;
  int* status;
  int nblk;
  int acquire;
};

template <int P>
__global__ __launch_bounds__(kWT) void mmtrssm_wide_fwd_kernel(const MmtWideFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MtrssmMmtrssmFwdIO& io = a.io;
  const MtrssmMmtrssmFwdWeights& w = a.w;
  const MmtWideGeom G(a.dm);
  const int B = a.dm.B, T = a.dm.T, LD = G.LD, HD = G.HD, H = G.H, LS = G.LS, HS = G.HS, act = a.dm.act;
  const int KL = a.dm.KL, CL = a.dm.CL, KH = a.dm.KH, CH = a.dm.CH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = a.nblk, blk = blockIdx.x;
  const int KST = G.KT / 32, KS2 = (G.KLD + G.KHD) / 32, KSH = G.HK / 32;
  const int LGW = 3 * G.LSp + 2 * G.HSp;

  wf32x4* red = reinterpret_cast<wf32x4*>(lds);   // 4 waves x 1 tile x 4 row tiles x 64 lanes x 16 B = 16 KiB
  float* rowv = lds + kWW * kMRT * kWave * 4;
  float* Llpl = rowv, *Lla = Llpl + G.LSp, *Llv = Lla + G.LSp, *Lmx = Llv + G.LSp, *Llqh = Lmx + G.LSp, *Llph = Llqh + G.HSp;
  float* Ls = Llph + G.HSp;            // [LS + HS] one-hot samples
  float* Lu = Ls + G.LSp + G.HSp;      // [4][64] uniforms: post_l, prior_l, post_h, prior_h
  int* abort_flag = reinterpret_cast<int*>(Lu + 256);
  if (tid == 0) *abort_flag = 0;
  __syncthreads();
  WideBarrier bar;
  bar.init(a.ctl, a.status, abort_flag, nblk, blk, a.acquire != 0);

  auto krange = [&](int KS, int& k0, int& k1) { k0 = KS * wave / kWW; k1 = KS * (wave + 1) / kWW; };
  const size_t tile_rnn = (size_t)KST * P * 64, tile_l0 = (size_t)KS2 * P * 64, tile_l1 = (size_t)KSH * P * 64;
  const int e_rt = tid >> 6, e_slot = lane, e_row = 16 * e_rt + (e_slot & 15), e_cq = 4 * (e_slot >> 4);

  for (int rb = 0; rb < B; rb += kMRows) {
    const int nrows = B - rb < kMRows ? B - rb : kMRows;
    const bool e_valid = e_row < nrows;
    const size_t e_b = (size_t)(rb + (e_valid ? e_row : 0));
    // ---- set-up: XS[0] <- [deter_l0 | deter_h0 | (s: written by F0)]; the launch zeroed every exchange vector
    for (int i = blk * kWT + tid; i < nrows * ((LD + HD) / 4); i += nblk * kWT) {
      const int row = i / ((LD + HD) / 4), q = i - row * ((LD + HD) / 4);
      const bool lower = q < LD / 4;
      const int c = lower ? q * 4 : (q - LD / 4) * 4;
      const float4 v4 = *reinterpret_cast<const float4*>(lower ? io.deter_l0 + (size_t)(rb + row) * LD + c : io.deter_h0 + (size_t)(rb + row) * HD + c);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
      wide_x_store4<P, kMRows>(a.xs[0], KST, row, lower ? c : G.KLD + c, v);
    }
    // hidden state of the units this workgroup owns in F1: registers across the steps
    float hid[4] = {0.f, 0.f, 0.f, 0.f};
    if (blk < G.NTL + G.NTHd && e_valid) {
      const bool lower = blk < G.NTL;
      const int c = (lower ? blk : blk - G.NTL) * 16 + e_cq;
      if (c < (lower ? LD : HD)) {
        const float4 v4 = *reinterpret_cast<const float4*>(lower ? io.hidden_l0 + e_b * LD + c : io.hidden_h0 + e_b * HD + c);
        hid[0] = v4.x; hid[1] = v4.y; hid[2] = v4.z; hid[3] = v4.w;
      }
    }
    int cur = 0;
    unsigned long long* const prof = (tid == 0 && rb == 0) ? g_mmt_prof : nullptr;

    for (int t = 0; t <= T; ++t) {
      const int tstamp = t;
      MTRSSM_MMT_STAMP(0);
      // ============ F0: one workgroup per batch row ============
      for (int r = nblk - 1 - blk; r < nrows; r += nblk) {
        const size_t b = (size_t)(rb + r);
        if (t > 0) {
          const size_t q = b * T + (t - 1);
          if (wave == 0) {
            if (lane < KL) { Lu[lane] = io.u_post_l[q * KL + lane]; Lu[64 + lane] = io.u_prior_l ? io.u_prior_l[q * KL + lane] : 0.f; }
            if (lane < KH) { Lu[128 + lane] = io.u_post_h[q * KH + lane]; Lu[192 + lane] = io.u_prior_h ? io.u_prior_h[q * KH + lane] : 0.f; }
          }
          for (int i = tid; i < 3 * LS + 2 * HS; i += kWT) {
            const float* lgr = a.lg + (size_t)r * LGW;
            if (i < 3 * LS) {
              const int which = i / LS, s2 = i - which * LS;
              const float bias = (which == 0 ? w.blp2 : (which == 1 ? w.ba2 : w.bv2))[s2];
              (which == 0 ? Llpl : (which == 1 ? Lla : Llv))[s2] = wide_load_f(lgr + which * G.LSp + s2) + bias;
            } else {
              const int j = i - 3 * LS, which = j / HS, s2 = j - which * HS;
              const float bias = (which == 0 ? w.bhq2 : w.bhp2)[s2];
              (which == 0 ? Llqh : Llph)[s2] = wide_load_f(lgr + 3 * G.LSp + which * G.HSp + s2) + bias;
            }
          }
          lds_barrier();
          MTRSSM_MMT_STAMP(10);
          if (wave == 0) {   // lower level: MoPoE mix, categorical block
            wave_mopoe_mix<true>(Lla, Llv, Lmx, LS, lane);
            for (int s2 = lane; s2 < LS; s2 += kWave) {
              io.prior_logits_l[q * LS + s2] = Llpl[s2];
              io.post_logits_l[q * LS + s2] = Lmx[s2];
              if (io.sv_la) { io.sv_la[q * LS + s2] = Lla[s2]; io.sv_lv[q * LS + s2] = Llv[s2]; }
            }
            float kll = CL <= 8 ? cat_block_fwd_fast8(Lmx, Llpl, KL, CL, lane, Lu, io.u_prior_l ? Lu + 64 : nullptr, Ls, io.post_stoch_l + q * LS,
                                                     io.prior_stoch_l ? io.prior_stoch_l + q * LS : nullptr, true)
                                : cat_block_fwd<true, true>(Lmx, Llpl, KL, CL, lane, Lu, io.u_prior_l ? Lu + 64 : nullptr, Ls, io.post_stoch_l + q * LS,
                                                            io.prior_stoch_l ? io.prior_stoch_l + q * LS : nullptr, true);
            kll = wave_sum(kll);
            if (lane == 0 && io.kl_l) io.kl_l[q] = kll;
            MTRSSM_MMT_STAMP(11);
          } else if (wave == 1) {   // higher level, beside it on another SIMD
            for (int s2 = lane; s2 < HS; s2 += kWave) {
              io.prior_logits_h[q * HS + s2] = Llph[s2];
              io.post_logits_h[q * HS + s2] = Llqh[s2];
            }
            float klh = CH <= 8 ? cat_block_fwd_fast8(Llqh, Llph, KH, CH, lane, Lu + 128, io.u_prior_h ? Lu + 192 : nullptr, Ls + LS,
                                                     io.post_stoch_h + q * HS, io.prior_stoch_h ? io.prior_stoch_h + q * HS : nullptr, true)
                                : cat_block_fwd<true, true>(Llqh, Llph, KH, CH, lane, Lu + 128, io.u_prior_h ? Lu + 192 : nullptr, Ls + LS,
                                                            io.post_stoch_h + q * HS, io.prior_stoch_h ? io.prior_stoch_h + q * HS : nullptr, true);
            klh = wave_sum(klh);
            if (lane == 0 && io.kl_h) io.kl_h[q] = klh;
          }
          lds_barrier();
        } else {
          for (int i = tid; i < LS; i += kWT) Ls[i] = io.stoch_l0[b * LS + i];
          for (int i = tid; i < HS; i += kWT) Ls[LS + i] = io.stoch_h0[b * HS + i];
          lds_barrier();
        }
        if (t < T) {  // [s_l ; s_h] into the s segment of the state vector F1 reads
          for (int i = tid; i < G.KSS / 4; i += kWT) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 4 * i + j < LS + HS ? Ls[4 * i + j] : 0.f;
            wide_x_store4<P, kMRows>(a.xs[cur], KST, r, G.KLD + G.KHD + 4 * i, v);
          }
        }
        lds_barrier();
      }
      if (t == T) break;
      MTRSSM_MMT_STAMP(1);
      if (!bar.sync(1 + 4 * t)) return;
      MTRSSM_MMT_STAMP(2);

      // ============ F1: both MTRNN cells, one workgroup per 16 deter units ============
      for (int u = blk; u < G.NTL + G.NTHd; u += nblk) {
        const bool lower = u < G.NTL;
        const int c = (lower ? u : u - G.NTL) * 16 + e_cq, width = lower ? LD : HD;
        const bool c_ok = e_valid && c < width;
        float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_ok) add = *reinterpret_cast<const float4*>(lower ? io.xl + (e_b * T + t) * LD + c : w.bh + c);
        int k0, k1;
        krange(KST, k0, k1);
        wf32x4 acc[1][kMRT];
#pragma unroll
        for (int rt = 0; rt < kMRT; ++rt) acc[0][rt] = wf32x4{0.f, 0.f, 0.f, 0.f};
        const uint4* const wt[1] = {a.pk_rnn + (size_t)u * tile_rnn};
        wide_mfma_stream<1, P, kMNS, kMRT>(acc, wt, a.xs[cur], KST, k0, k1, lane);
        wide_red_store<1, kMRT>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        {
          const wf32x4 sm = wide_red_sum<1, kMRT>(red, 0, e_rt, e_slot);
          const float keep = lower ? a.dm.keep_l : a.dm.keep_h, tau = lower ? a.dm.tau_l : a.dm.tau_h;
          const float uv[4] = {sm[0] + add.x, sm[1] + add.y, sm[2] + add.z, sm[3] + add.w};
          float d[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            hid[j] = keep * hid[j] + uv[j] / tau;
            d[j] = tanhf(hid[j]);
          }
          if (c_ok) {
            const size_t q = e_b * T + t;
            *reinterpret_cast<float4*>((lower ? io.deter_l + q * LD : io.deter_h + q * HD) + c) = make_float4(d[0], d[1], d[2], d[3]);
            *reinterpret_cast<float4*>((lower ? io.hidden_l + q * LD : io.hidden_h + q * HD) + c) = make_float4(hid[0], hid[1], hid[2], hid[3]);
            wide_x_store4<P, kMRows>(a.xs[cur ^ 1], KST, e_row, (lower ? 0 : G.KLD) + c, d);
          }
        }
        lds_barrier();
      }
      cur ^= 1;
      MTRSSM_MMT_STAMP(3);
      if (!bar.sync(2 + 4 * t)) return;
      MTRSSM_MMT_STAMP(4);

      // ============ F2: layer 0 of the five heads on [d_l | d_h] ============
      for (int u = blk; u < 5 * G.NTHP; u += nblk) {
        const int q5 = u / G.NTHP, c = (u - q5 * G.NTHP) * 16 + e_cq;
        const bool c_ok = e_valid && c < H;
        float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_ok) {
          const size_t q = e_b * T + t;
          if (q5 == 0) add = *reinterpret_cast<const float4*>(w.bl1 + c);
          else if (q5 == 1) add = *reinterpret_cast<const float4*>(io.pa + q * H + c);
          else if (q5 == 2) add = *reinterpret_cast<const float4*>(io.pv + q * H + c);
          else if (q5 == 3) add = *reinterpret_cast<const float4*>(w.bh1 + H + c);
          else add = *reinterpret_cast<const float4*>(w.bh1 + c);
        }
        int k0, k1;
        krange(KS2, k0, k1);
        wf32x4 acc[1][kMRT];
#pragma unroll
        for (int rt = 0; rt < kMRT; ++rt) acc[0][rt] = wf32x4{0.f, 0.f, 0.f, 0.f};
        const uint4* const wt[1] = {a.pk_l0 + (size_t)u * tile_l0};
        wide_mfma_stream<1, P, kMNS, kMRT>(acc, wt, a.xs[cur], KST, k0, k1, lane);   // XS has KST k-blocks per piece; only the d part is read
        wide_red_store<1, kMRT>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        if (c_ok) {
          const wf32x4 sm = wide_red_sum<1, kMRT>(red, 0, e_rt, e_slot);
          const float h[4] = {act_fwd(sm[0] + add.x, act), act_fwd(sm[1] + add.y, act), act_fwd(sm[2] + add.z, act), act_fwd(sm[3] + add.w, act)};
          const size_t q = e_b * T + t;
          if (q5 < 4) {
            if (io.sv_l1) *reinterpret_cast<float4*>(io.sv_l1 + q * 4 * H + q5 * H + c) = make_float4(h[0], h[1], h[2], h[3]);
          } else if (io.sv_h1) {
            *reinterpret_cast<float4*>(io.sv_h1 + q * H + c) = make_float4(h[0], h[1], h[2], h[3]);
          }
          wide_x_store4<P, kMRows>(a.xh[q5], KSH, e_row, c, h);
        }
        lds_barrier();
      }
      MTRSSM_MMT_STAMP(5);
      if (!bar.sync(3 + 4 * t)) return;
      MTRSSM_MMT_STAMP(6);

      // ============ F3: layer 1 of the five heads: lpl | la | lv | lqh | lph ============
      {
        const int nl = G.LSp / 16, nh = G.HSp / 16;
        for (int u = blk; u < 3 * nl + 2 * nh; u += nblk) {
          const int q5 = u < 3 * nl ? u / nl : 3 + (u - 3 * nl) / nh;
          const int st = u < 3 * nl ? u - q5 * nl : (u - 3 * nl) - (q5 - 3) * nh;
          int k0, k1;
          krange(KSH, k0, k1);
          wf32x4 acc[1][kMRT];
#pragma unroll
          for (int rt = 0; rt < kMRT; ++rt) acc[0][rt] = wf32x4{0.f, 0.f, 0.f, 0.f};
          const uint4* const wt[1] = {a.pk_l1[q5] + (size_t)st * tile_l1};
          wide_mfma_stream<1, P, kMNS, kMRT>(acc, wt, a.xh[q5], KSH, k0, k1, lane);
          wide_red_store<1, kMRT>(red, wave, 0, lane, acc[0]);
          lds_barrier();
          if (e_valid) {
            const wf32x4 sm = wide_red_sum<1, kMRT>(red, 0, e_rt, e_slot);
            float* dst = a.lg + (size_t)e_row * LGW + (q5 < 3 ? q5 * G.LSp : 3 * G.LSp + (q5 - 3) * G.HSp) + st * 16 + e_cq;
            wide_store_f2(dst, sm[0], sm[1]);
            wide_store_f2(dst + 2, sm[2], sm[3]);
          }
          lds_barrier();
        }
      }
      MTRSSM_MMT_STAMP(7);
      if (!bar.sync(4 + 4 * t)) return;
      MTRSSM_MMT_STAMP(8);
    }
    if (!bar.sync(0x40000000)) return;   // the next tile's set-up overwrites the exchange vectors
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct MmtWideBwdArgs {
  MtrssmMmtrssmDims dm;
  MtrssmMmtrssmBwdIO io;
  const uint4 *pk_l1t[5];   // layer 1 transposed (N = H, K = LS | HS): lpl, la, lv, lqh, lph
  const uint4 *pk_l0t;      // N = [d_l tiles | d_h tiles], K = 5 head segments of HK: l_prior, audio, vision, h_posterior, h_prior
  const uint4 *pk_rnnt;     // N = [d_l tiles | d_h tiles | s tiles], K = [du_l (KLD) | du_h (KHD)]
  uint4 *x_dl[5], *x_dz, *x_du;
  float* cs;                // [64][LSHp] carry into [s_l ; s_h]
  void* ctl;
  int* status;
  int nblk;
  int acquire;
};

template <int P>
__global__ __launch_bounds__(kWT) void mmtrssm_wide_bwd_kernel(const MmtWideBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MtrssmMmtrssmBwdIO& io = a.io;
  const MmtWideGeom G(a.dm);
  const int B = a.dm.B, T = a.dm.T, LD = G.LD, HD = G.HD, H = G.H, LS = G.LS, HS = G.HS, act = a.dm.act;
  const int KL = a.dm.KL, CL = a.dm.CL, KH = a.dm.KH, CH = a.dm.CH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = a.nblk, blk = blockIdx.x;
  const int KSL = (LS + 31) / 32, KSHs = (HS + 31) / 32, KSZ = 5 * G.HK / 32, KSU = (G.KLD + G.KHD) / 32;
  const int LSHp = (LS + HS + 15) / 16 * 16, NTS = LSHp / 16;

  wf32x4* red = reinterpret_cast<wf32x4*>(lds);
  float* rowv = lds + kWW * kMRT * kWave * 4;
  float* Lla = rowv, *Llv = Lla + G.LSp, *Lmx = Llv + G.LSp, *Llpl = Lmx + G.LSp, *Llqh = Llpl + G.LSp, *Llph = Llqh + G.HSp;
  float* Ldmx = Llph + G.HSp, *Ldlpl = Ldmx + G.LSp, *Ldla = Ldlpl + G.LSp, *Ldlv = Ldla + G.LSp, *Ldlqh = Ldlv + G.LSp, *Ldlph = Ldlqh + G.HSp;
  float* Lcs = Ldlph + G.HSp;          // [LS + HS]
  float* Lgps = Lcs + LSHp;            // [LS + HS] g_post_stoch_l | g_post_stoch_h
  int* abort_flag = reinterpret_cast<int*>(Lgps + LSHp);
  if (tid == 0) *abort_flag = 0;
  __syncthreads();
  WideBarrier bar;
  bar.init(a.ctl, a.status, abort_flag, nblk, blk, a.acquire != 0);

  auto krange = [&](int KS, int& k0, int& k1) { k0 = KS * wave / kWW; k1 = KS * (wave + 1) / kWW; };
  const size_t tile_z = (size_t)KSZ * P * 64, tile_u = (size_t)KSU * P * 64;
  const int e_rt = tid >> 6, e_slot = lane, e_row = 16 * e_rt + (e_slot & 15), e_cq = 4 * (e_slot >> 4);

  for (int rb = 0; rb < B; rb += kMRows) {
    const int nrows = B - rb < kMRows ? B - rb : kMRows;
    const bool e_valid = e_row < nrows;
    const size_t e_b = (size_t)(rb + (e_valid ? e_row : 0));
    float c_d[4] = {0.f, 0.f, 0.f, 0.f}, c_hid[4] = {0.f, 0.f, 0.f, 0.f};   // carries of the deter units this workgroup owns (R2, R3)
    unsigned long long* const prof = (tid == 0 && rb == 0) ? g_mmt_prof : nullptr;

    for (int t = T - 1; t >= 0; --t) {
      const int tstamp = T - 1 - t;
      MTRSSM_MMT_STAMP(0);
      // ============ R0: categorical blocks + MoPoE mix backward, one workgroup per batch row ============
      for (int r = nblk - 1 - blk; r < nrows; r += nblk) {
        const size_t q = (size_t)(rb + r) * T + t;
        for (int s2 = tid; s2 < LS; s2 += kWT) {
          Lla[s2] = io.sv_la[q * LS + s2];
          Llv[s2] = io.sv_lv[q * LS + s2];
          Lmx[s2] = io.post_logits_l[q * LS + s2];
          Llpl[s2] = io.prior_logits_l[q * LS + s2];
          Lgps[s2] = io.g_post_stoch_l ? io.g_post_stoch_l[q * LS + s2] : 0.f;
        }
        for (int s2 = tid; s2 < HS; s2 += kWT) {
          Llqh[s2] = io.post_logits_h[q * HS + s2];
          Llph[s2] = io.prior_logits_h[q * HS + s2];
          Lgps[LS + s2] = io.g_post_stoch_h ? io.g_post_stoch_h[q * HS + s2] : 0.f;
        }
        for (int s2 = tid; s2 < LS + HS; s2 += kWT) Lcs[s2] = t == T - 1 ? 0.f : wide_load_f(a.cs + (size_t)r * LSHp + s2);
        lds_barrier();
        MTRSSM_MMT_STAMP(10);
        if (wave == 0) {   // lower level: categorical block, then the MoPoE mix backward
          const float gkl = io.g_kl_l ? io.g_kl_l[q] : 0.f;
          const float* gpsl = io.g_prior_stoch_l ? io.g_prior_stoch_l + q * LS : nullptr;
          const float* gpll = io.g_post_logits_l ? io.g_post_logits_l + q * LS : nullptr;
          const float* gprl = io.g_prior_logits_l ? io.g_prior_logits_l + q * LS : nullptr;
          if (CL <= 8) cat_block_bwd_fast8(Lmx, Llpl, KL, CL, lane, Lgps, Lcs, gpsl, gpll, gprl, gkl, a.dm.kl_w_post, a.dm.kl_w_prior, Ldmx, Ldlpl);
          else cat_block_bwd<true>(Lmx, Llpl, KL, CL, lane, Lgps, Lcs, gpsl, gpll, gprl, gkl, a.dm.kl_w_post, a.dm.kl_w_prior, Ldmx, Ldlpl);
          wave_mopoe_mix_bwd<true>(Lla, Llv, Lmx, Ldmx, Ldla, Ldlv, LS, lane);
          MTRSSM_MMT_STAMP(11);
        } else if (wave == 1) {   // higher level, beside it on another SIMD
          const float gkh = io.g_kl_h ? io.g_kl_h[q] : 0.f;
          const float* gpsh = io.g_prior_stoch_h ? io.g_prior_stoch_h + q * HS : nullptr;
          const float* gplh = io.g_post_logits_h ? io.g_post_logits_h + q * HS : nullptr;
          const float* gprh = io.g_prior_logits_h ? io.g_prior_logits_h + q * HS : nullptr;
          if (CH <= 8) cat_block_bwd_fast8(Llqh, Llph, KH, CH, lane, Lgps + LS, Lcs + LS, gpsh, gplh, gprh, gkh, a.dm.kl_w_post, a.dm.kl_w_prior, Ldlqh, Ldlph);
          else cat_block_bwd<true>(Llqh, Llph, KH, CH, lane, Lgps + LS, Lcs + LS, gpsh, gplh, gprh, gkh, a.dm.kl_w_post, a.dm.kl_w_prior, Ldlqh, Ldlph);
        }
        lds_barrier();
        // outputs and exchange vectors: lpl, la, lv (K = LS), lqh, lph (K = HS)
        for (int i = tid; i < 3 * (G.LSp / 4) + 2 * (G.HSp / 4); i += kWT) {
          const bool low = i < 3 * (G.LSp / 4);
          const int which = low ? i / (G.LSp / 4) : 3 + (i - 3 * (G.LSp / 4)) / (G.HSp / 4);
          const int s4 = (low ? i - which * (G.LSp / 4) : (i - 3 * (G.LSp / 4)) - (which - 3) * (G.HSp / 4)) * 4;
          const int S2 = low ? LS : HS;
          const float* src = which == 0 ? Ldlpl : (which == 1 ? Ldla : (which == 2 ? Ldlv : (which == 3 ? Ldlqh : Ldlph)));
          float* dst = (which == 0 ? io.d_lpl : (which == 1 ? io.d_la : (which == 2 ? io.d_lv : (which == 3 ? io.d_lqh : io.d_lph)))) + q * S2;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = s4 + j < S2 ? src[s4 + j] : 0.f;
            if (s4 + j < S2) dst[s4 + j] = v[j];
          }
          wide_x_store4<P, kMRows>(a.x_dl[which], low ? KSL : KSHs, r, s4, v);
        }
        lds_barrier();
      }
      MTRSSM_MMT_STAMP(1);
      if (!bar.sync(1 + 4 * t)) return;
      MTRSSM_MMT_STAMP(2);

      // ============ R1: pre-activation gradients of layer 0: dz = act'(h) * (W2nd^T dl), five heads ============
      for (int u = blk; u < 5 * G.NTHP; u += nblk) {
        const int q5 = u / G.NTHP, c = (u - q5 * G.NTHP) * 16 + e_cq;   // l_prior, audio, vision, h_posterior, h_prior
        const bool c_ok = e_valid && c < H;
        float4 hs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_ok) {
          const size_t q = e_b * T + t;
          hs = *reinterpret_cast<const float4*>(q5 < 4 ? io.sv_l1 + q * 4 * H + q5 * H + c : io.sv_h1 + q * H + c);
        }
        const int KS = q5 < 3 ? KSL : KSHs;
        int k0, k1;
        krange(KS, k0, k1);
        wf32x4 acc[1][kMRT];
#pragma unroll
        for (int rt = 0; rt < kMRT; ++rt) acc[0][rt] = wf32x4{0.f, 0.f, 0.f, 0.f};
        const uint4* const wt[1] = {a.pk_l1t[q5] + (size_t)(u - q5 * G.NTHP) * ((size_t)KS * P * 64)};
        wide_mfma_stream<1, P, kMNS, kMRT>(acc, wt, a.x_dl[q5], KS, k0, k1, lane);
        wide_red_store<1, kMRT>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        if (c_ok) {
          const wf32x4 sm = wide_red_sum<1, kMRT>(red, 0, e_rt, e_slot);
          const float g[4] = {sm[0] * act_grad_from_out(hs.x, act), sm[1] * act_grad_from_out(hs.y, act), sm[2] * act_grad_from_out(hs.z, act),
                              sm[3] * act_grad_from_out(hs.w, act)};
          const size_t q = e_b * T + t;
          if (q5 < 4) *reinterpret_cast<float4*>(io.d_zl1 + q * 4 * H + q5 * H + c) = make_float4(g[0], g[1], g[2], g[3]);
          else *reinterpret_cast<float4*>(io.d_zh1 + q * H + c) = make_float4(g[0], g[1], g[2], g[3]);
          wide_x_store4<P, kMRows>(a.x_dz, KSZ, e_row, q5 * G.HK + c, g);
        }
        lds_barrier();
      }
      MTRSSM_MMT_STAMP(3);
      if (!bar.sync(2 + 4 * t)) return;
      MTRSSM_MMT_STAMP(4);

      // ============ R2: gradients at d_l / d_h, through tanh into the leaky integrators ============
      for (int u = blk; u < G.NTL + G.NTHd; u += nblk) {
        const bool lower = u < G.NTL;
        const int c = (lower ? u : u - G.NTL) * 16 + e_cq, width = lower ? LD : HD;
        const bool c_ok = e_valid && c < width;
        float4 gd = make_float4(0.f, 0.f, 0.f, 0.f), gh = gd, dv = gd;
        if (c_ok) {
          const size_t q = e_b * T + t;
          const float* gdp = lower ? io.g_deter_l : io.g_deter_h;
          const float* ghp = lower ? io.g_hidden_l : io.g_hidden_h;
          if (gdp) gd = *reinterpret_cast<const float4*>(gdp + q * width + c);
          if (ghp) gh = *reinterpret_cast<const float4*>(ghp + q * width + c);
          dv = *reinterpret_cast<const float4*>((lower ? io.deter_l : io.deter_h) + q * width + c);
        }
        int k0, k1;
        krange(KSZ, k0, k1);
        wf32x4 acc[1][kMRT];
#pragma unroll
        for (int rt = 0; rt < kMRT; ++rt) acc[0][rt] = wf32x4{0.f, 0.f, 0.f, 0.f};
        const uint4* const wt[1] = {a.pk_l0t + (size_t)u * tile_z};
        wide_mfma_stream<1, P, kMNS, kMRT>(acc, wt, a.x_dz, KSZ, k0, k1, lane);
        wide_red_store<1, kMRT>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        {
          const wf32x4 sm = wide_red_sum<1, kMRT>(red, 0, e_rt, e_slot);
          const float keep = lower ? a.dm.keep_l : a.dm.keep_h, tau = lower ? a.dm.tau_l : a.dm.tau_h;
          const float gdv[4] = {gd.x, gd.y, gd.z, gd.w}, ghv[4] = {gh.x, gh.y, gh.z, gh.w}, dd4[4] = {dv.x, dv.y, dv.z, dv.w};
          float du[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float dd = sm[j] + gdv[j] + c_d[j];
            const float dhid = dd * (1.f - dd4[j] * dd4[j]) + ghv[j] + c_hid[j];
            du[j] = dhid / tau;
            c_hid[j] = dhid * keep;
          }
          if (c_ok) {
            const size_t q = e_b * T + t;
            *reinterpret_cast<float4*>((lower ? io.d_ul + q * LD : io.d_uh + q * HD) + c) = make_float4(du[0], du[1], du[2], du[3]);
            wide_x_store4<P, kMRows>(a.x_du, KSU, e_row, (lower ? 0 : G.KLD) + c, du);
          }
        }
        lds_barrier();
      }
      MTRSSM_MMT_STAMP(5);
      if (!bar.sync(3 + 4 * t)) return;
      MTRSSM_MMT_STAMP(6);

      // ============ R3: carries into step t-1: W_d^T du (kept by the workgroups of R2) | W_x^T du -> [s_l ; s_h] ============
      for (int u = blk; u < G.NTL + G.NTHd + NTS; u += nblk) {
        const bool is_s = u >= G.NTL + G.NTHd, lower = u < G.NTL;
        const int c = (is_s ? u - G.NTL - G.NTHd : (lower ? u : u - G.NTL)) * 16 + e_cq;
        int k0, k1;
        krange(KSU, k0, k1);
        wf32x4 acc[1][kMRT];
#pragma unroll
        for (int rt = 0; rt < kMRT; ++rt) acc[0][rt] = wf32x4{0.f, 0.f, 0.f, 0.f};
        const uint4* const wt[1] = {a.pk_rnnt + (size_t)u * tile_u};
        wide_mfma_stream<1, P, kMNS, kMRT>(acc, wt, a.x_du, KSU, k0, k1, lane);
        wide_red_store<1, kMRT>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        {
          const wf32x4 sm = wide_red_sum<1, kMRT>(red, 0, e_rt, e_slot);
          if (is_s) {
            if (e_valid) {
              float* dst = a.cs + (size_t)e_row * LSHp + c;
              wide_store_f2(dst, sm[0], sm[1]);
              wide_store_f2(dst + 2, sm[2], sm[3]);
              if (t == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const int s2 = c + j;
                  if (s2 < LS) io.g_stoch_l0[e_b * LS + s2] = sm[j];
                  else if (s2 < LS + HS) io.g_stoch_h0[e_b * HS + (s2 - LS)] = sm[j];
                }
              }
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) c_d[j] = sm[j];
            const int width = lower ? LD : HD;
            if (t == 0 && e_valid && c < width) {
              *reinterpret_cast<float4*>((lower ? io.g_deter_l0 + e_b * LD : io.g_deter_h0 + e_b * HD) + c) = make_float4(c_d[0], c_d[1], c_d[2], c_d[3]);
              *reinterpret_cast<float4*>((lower ? io.g_hidden_l0 + e_b * LD : io.g_hidden_h0 + e_b * HD) + c) =
                  make_float4(c_hid[0], c_hid[1], c_hid[2], c_hid[3]);
            }
          }
        }
        lds_barrier();
      }
      MTRSSM_MMT_STAMP(7);
      if (!bar.sync(4 + 4 * t)) return;
      MTRSSM_MMT_STAMP(8);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int debug_set_mmt_profile(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_mmt_prof), &p, sizeof(p)) == hipSuccess ? MTRSSM_OK : MTRSSM_ELAUNCH;
}

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static bool mmt_wide_dims_ok(const MtrssmMmtrssmDims* d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->LD <= 0 || d->HD <= 0 || d->H <= 0 || d->KL <= 0 || d->CL <= 0 || d->KH <= 0 || d->CH <= 0 || !d->post)
    return false;
  if (d->LD % 4 || d->HD % 4 || d->H % 4) return false;          // 16-byte quads of the row-major tensors
  if (d->KL > 64 || d->KH > 64) return false;                    // one lane per categorical
  if (!(d->tau_l > 1.f) || !(d->tau_h > 1.f)) return false;
  if (d->act < MTRSSM_ACT_IDENTITY || d->act > MTRSSM_ACT_TANH) return false;
  return true;
}

int mmtrssm_wide_supported(const MtrssmMmtrssmDims* d, int pieces) {
  if (!mmt_wide_dims_ok(d) || (pieces != 2 && pieces != 3)) return 0;
  if (d->LD < 128 && d->HD < 128) return 0;   // below that the one-CU form's timestep is shorter than four grid barriers
  const MmtWideGeom G(*d);
  const int cus = device_cu_count();
  if (cus < 64) return 0;
  if (G.NTL + G.NTHd + (G.LS + G.HS + 15) / 16 > cus) return 0;   // the stateful phases: one column tile per workgroup
  return 1;
}

struct MmtFwdLayout { size_t xs[2], xh[5], lg, exch_end, rnn, l0, l1[5], total; };
static MmtFwdLayout mmt_fwd_layout(const MtrssmMmtrssmDims* d, int P) {
  const MmtWideGeom G(*d);
  MmtFwdLayout L;
  size_t o = kWideCtl;
  auto take = [&](size_t bytes) { const size_t r = o; o += al256(bytes); return r; };
  for (int i = 0; i < 2; ++i) L.xs[i] = take(wide_x_uint4(G.KT, P, kMRows) * 16);
  for (int i = 0; i < 5; ++i) L.xh[i] = take(wide_x_uint4(G.HK, P, kMRows) * 16);
  L.lg = take((size_t)kMRows * (3 * G.LSp + 2 * G.HSp) * sizeof(float));
  L.exch_end = o;
  L.rnn = take(wide_pack_uint4(16 * (G.NTL + G.NTHd), G.KT, P) * 16);
  L.l0 = take(wide_pack_uint4(5 * G.HP, G.KLD + G.KHD, P) * 16);
  for (int i = 0; i < 5; ++i) L.l1[i] = take(wide_pack_uint4(i < 3 ? G.LS : G.HS, G.HK, P) * 16);
  L.total = o;
  return L;
}
size_t mmtrssm_wide_workspace_bytes(const MtrssmMmtrssmDims* d, int pieces) {
  if (!mmt_wide_dims_ok(d) || (pieces != 2 && pieces != 3)) return 0;
  return mmt_fwd_layout(d, pieces).total;
}

struct MmtBwdLayout { size_t x_dl[5], x_dz, x_du, cs, exch_end, l1t[5], l0t, rnnt, total; };
static MmtBwdLayout mmt_bwd_layout(const MtrssmMmtrssmDims* d, int P) {
  const MmtWideGeom G(*d);
  MmtBwdLayout L;
  size_t o = kWideCtl;
  auto take = [&](size_t bytes) { const size_t r = o; o += al256(bytes); return r; };
  for (int i = 0; i < 5; ++i) L.x_dl[i] = take(wide_x_uint4(i < 3 ? G.LS : G.HS, P, kMRows) * 16);
  L.x_dz = take(wide_x_uint4(5 * G.HK, P, kMRows) * 16);
  L.x_du = take(wide_x_uint4(G.KLD + G.KHD, P, kMRows) * 16);
  L.cs = take((size_t)kMRows * ((G.LS + G.HS + 15) / 16 * 16) * sizeof(float));
  L.exch_end = o;
  for (int i = 0; i < 5; ++i) L.l1t[i] = take(wide_pack_uint4(G.HP, i < 3 ? G.LS : G.HS, P) * 16);
  L.l0t = take(wide_pack_uint4(16 * (G.NTL + G.NTHd), 5 * G.HK, P) * 16);
  L.rnnt = take(wide_pack_uint4(16 * (G.NTL + G.NTHd) + (G.LS + G.HS + 15) / 16 * 16, G.KLD + G.KHD, P) * 16);
  L.total = o;
  return L;
}
size_t mmtrssm_wide_bwd_workspace_bytes(const MtrssmMmtrssmDims* d, int pieces) {
  if (!mmt_wide_dims_ok(d) || (pieces != 2 && pieces != 3)) return 0;
  return mmt_bwd_layout(d, pieces).total;
}

static int mmt_acquire_fence() {
  static const int on = [] { const char* e = getenv("MTRSSM_WIDE_ACQUIRE"); return (e && e[0] == '1') ? 1 : 0; }();
  return on;
}

static size_t mmt_fwd_lds(const MmtWideGeom& G) {
  const size_t need = (size_t)kWW * kMRT * kWave * 16 + ((size_t)5 * G.LSp + 3 * G.HSp + 256 + 8) * sizeof(float);
  return need < 84 * 1024 ? 84 * 1024 : need;   // > 80 KiB: never two workgroups on one CU
}
static size_t mmt_bwd_lds(const MmtWideGeom& G) {
  const size_t LSHp = (G.LS + G.HS + 15) / 16 * 16;
  const size_t need = (size_t)kWW * kMRT * kWave * 16 + ((size_t)8 * G.LSp + 4 * G.HSp + 2 * LSHp + 8) * sizeof(float);
  return need < 84 * 1024 ? 84 * 1024 : need;
}

int mmtrssm_wide_fwd_launch(const MtrssmMmtrssmDims* d, const MtrssmMmtrssmFwdWeights* w, const MtrssmMmtrssmFwdIO* io, int pieces,
                            void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!mmtrssm_wide_supported(d, pieces)) {
    set_error("mmtrssm_rollout_fwd_wide: dims / device outside the wide kernel's regime (ask mtrssm_mmtrssm_wide_supported first)");
    return MTRSSM_EINVAL;
  }
  if (!w || !io || !workspace || !w->wxl_s_t || !w->wdl_t || !w->wxh_t || !w->wdh_t || !w->bh || !w->wl1_t || !w->bl1 || !w->wh1_t || !w->bh1 ||
      !w->wlp2 || !w->blp2 || !w->wa2 || !w->ba2 || !w->wv2 || !w->bv2 || !w->whp2 || !w->bhp2 || !w->whq2 || !w->bhq2 || !io->xl || !io->pa ||
      !io->pv || !io->deter_l0 || !io->deter_h0 || !io->hidden_l0 || !io->hidden_h0 || !io->stoch_l0 || !io->stoch_h0 || !io->u_post_l ||
      !io->u_post_h || !io->deter_l || !io->deter_h || !io->hidden_l || !io->hidden_h || !io->prior_logits_l || !io->prior_logits_h ||
      !io->post_logits_l || !io->post_logits_h || !io->post_stoch_l || !io->post_stoch_h) {
    set_error("mmtrssm_rollout_fwd_wide: null required pointer");
    return MTRSSM_EINVAL;
  }
  const MmtFwdLayout L = mmt_fwd_layout(d, pieces);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) {
    set_error("mmtrssm_rollout_fwd_wide: workspace too small (%zu < %zu) or not 256-byte aligned", workspace_bytes, L.total);
    return MTRSSM_EINVAL;
  }
  char* ws = static_cast<char*>(workspace);
  const MmtWideGeom G(*d);
  const int LD = G.LD, HD = G.HD, H = G.H, LS = G.LS, HS = G.HS, P = pieces;
  // barrier words, exchange vectors (padding and rows beyond the batch stay zero) and the packed matrices (blocks the jobs do not
  // fill are zero): everything behind the sticky status word
  if (int rc = clear_async(ws + 16, L.total - 16, stream)) return rc;
  const int KST = G.KT / 32, KS2 = (G.KLD + G.KHD) / 32;
  uint4* rnn = reinterpret_cast<uint4*>(ws + L.rnn);
  uint4* l0 = reinterpret_cast<uint4*>(ws + L.l0);
  WidePackJobs jobs;
  jobs.count = 4;
  // u_l rows: [W_d_l | 0 | W_x_l over (s_l ; s_h)];  u_h rows: [0 | W_d_h | W_x_h over s_h]
  jobs.j[0] = wide_make_block(w->wdl_t, 1, LD, LD, LD, 0, 0, 0, 0, KST, rnn);
  jobs.j[1] = wide_make_block(w->wxl_s_t, 1, LD, LD, LS + HS, 0, 0, 0, KS2, KST, rnn);
  jobs.j[2] = wide_make_block(w->wdh_t, 1, HD, HD, HD, 0, 0, G.NTL, G.KLD / 32, KST, rnn);
  jobs.j[3] = wide_make_block(w->wxh_t, 1, HD, HD, LS + HS, 0, LS, G.NTL, KS2, KST, rnn);
  if (int rc = wide_launch_pack(jobs, P, stream)) return rc;
  jobs.count = 6;
  // layer 0: heads l_prior, audio, vision, h_posterior (l part), then h_posterior (h part), h_prior
  for (int q = 0; q < 4; ++q) jobs.j[q] = wide_make_block(w->wl1_t + (size_t)q * H, 1, 4 * H, H, LD, 0, 0, q * G.NTHP, 0, KS2, l0);
  jobs.j[4] = wide_make_block(w->wh1_t + H, 1, 2 * H, H, HD, 0, 0, 3 * G.NTHP, G.KLD / 32, KS2, l0);
  jobs.j[5] = wide_make_block(w->wh1_t, 1, 2 * H, H, HD, 0, 0, 4 * G.NTHP, G.KLD / 32, KS2, l0);
  if (int rc = wide_launch_pack(jobs, P, stream)) return rc;
  const float* l1src[5] = {w->wlp2, w->wa2, w->wv2, w->whq2, w->whp2};
  jobs.count = 5;
  for (int q = 0; q < 5; ++q) jobs.j[q] = wide_make_job(l1src[q], H, 1, q < 3 ? LS : HS, H, reinterpret_cast<uint4*>(ws + L.l1[q]));
  if (int rc = wide_launch_pack(jobs, P, stream)) return rc;

  MmtWideFwdArgs a;
  a.dm = *d; a.w = *w; a.io = *io;
  a.pk_rnn = rnn; a.pk_l0 = l0;
  for (int q = 0; q < 5; ++q) {
    a.pk_l1[q] = reinterpret_cast<const uint4*>(ws + L.l1[q]);
    a.xh[q] = reinterpret_cast<uint4*>(ws + L.xh[q]);
  }
  a.xs[0] = reinterpret_cast<uint4*>(ws + L.xs[0]);
  a.xs[1] = reinterpret_cast<uint4*>(ws + L.xs[1]);
  a.lg = reinterpret_cast<float*>(ws + L.lg);
  a.ctl = ws; a.status = reinterpret_cast<int*>(ws);
  a.nblk = device_cu_count();
  a.acquire = mmt_acquire_fence();
  const size_t lds = mmt_fwd_lds(G);
  if (lds > 160 * 1024) { set_error("mmtrssm_rollout_fwd_wide: %zu bytes of LDS", lds); return MTRSSM_ELDS; }
  hipError_t e;
#define MTRSSM_MMT_WIDE_FWD(PV)                                                                                                        \
  {                                                                                                                                   \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mmtrssm_wide_fwd_kernel<PV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e)); return MTRSSM_ELAUNCH; }     \
    set_last_kernel("mtrssm::mmtrssm_wide_fwd_kernel<" #PV ">");                                                                      \
    hipLaunchKernelGGL(mmtrssm_wide_fwd_kernel<PV>, dim3(a.nblk), dim3(kWT), lds, stream, a);                                        \
  }
  if (pieces == 3) MTRSSM_MMT_WIDE_FWD(3) else MTRSSM_MMT_WIDE_FWD(2)
#undef MTRSSM_MMT_WIDE_FWD
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("wide MMTRSSM forward scan launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  return MTRSSM_OK;
}

int mmtrssm_wide_bwd_launch(const MtrssmMmtrssmDims* d, const MtrssmMmtrssmBwdWeights* w, const MtrssmMmtrssmBwdIO* io, int pieces,
                            void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!mmtrssm_wide_supported(d, pieces)) {
    set_error("mmtrssm_rollout_bwd_wide: dims / device outside the wide kernel's regime (ask mtrssm_mmtrssm_wide_supported first)");
    return MTRSSM_EINVAL;
  }
  if (!w || !io || !workspace || !w->wxl_s_t || !w->wdl || !w->wxh_t || !w->wdh || !w->wl1 || !w->wh1 || !w->wlp2 || !w->wa2 || !w->wv2 ||
      !w->whp2 || !w->whq2 || !io->deter_l0 || !io->deter_h0 || !io->deter_l || !io->deter_h || !io->prior_logits_l || !io->prior_logits_h ||
      !io->post_logits_l || !io->post_logits_h || !io->sv_l1 || !io->sv_h1 || !io->sv_la || !io->sv_lv || !io->g_deter_l0 || !io->g_deter_h0 ||
      !io->g_hidden_l0 || !io->g_hidden_h0 || !io->g_stoch_l0 || !io->g_stoch_h0 || !io->d_ul || !io->d_uh || !io->d_zl1 || !io->d_zh1 ||
      !io->d_lpl || !io->d_la || !io->d_lv || !io->d_lph || !io->d_lqh) {
    set_error("mmtrssm_rollout_bwd_wide: null required pointer");
    return MTRSSM_EINVAL;
  }
  const MmtBwdLayout L = mmt_bwd_layout(d, pieces);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) {
    set_error("mmtrssm_rollout_bwd_wide: workspace too small (%zu < %zu) or not 256-byte aligned", workspace_bytes, L.total);
    return MTRSSM_EINVAL;
  }
  char* ws = static_cast<char*>(workspace);
  const MmtWideGeom G(*d);
  const int LD = G.LD, HD = G.HD, H = G.H, LS = G.LS, HS = G.HS, P = pieces;
  if (int rc = clear_async(ws + 16, L.total - 16, stream)) return rc;
  const int KSZ = 5 * G.HK / 32, KSU = (G.KLD + G.KHD) / 32, NTD = G.NTL + G.NTHd;
  uint4* l0t = reinterpret_cast<uint4*>(ws + L.l0t);
  uint4* rnnt = reinterpret_cast<uint4*>(ws + L.rnnt);
  WidePackJobs jobs;
  // layer 1 transposed: dz[j] = sum_s W[s][j] dl[s]  -> (n = j, k = s) = W[s H + j]
  const float* l1src[5] = {w->wlp2, w->wa2, w->wv2, w->whq2, w->whp2};
  jobs.count = 5;
  for (int q = 0; q < 5; ++q) jobs.j[q] = wide_make_job(l1src[q], 1, H, H, q < 3 ? LS : HS, reinterpret_cast<uint4*>(ws + L.l1t[q]));
  if (int rc = wide_launch_pack(jobs, P, stream)) return rc;
  // layer 0 transposed: dd_l[i] = sum over the four lower heads of wl1[q H + j][i] dz_q[j];  dd_h[i] = wh1[H + j][i] dz_hpost[j] + wh1[j][i] dz_hprior[j]
  jobs.count = 6;
  for (int q = 0; q < 4; ++q) jobs.j[q] = wide_make_block(w->wl1 + (size_t)q * H * LD, 1, LD, LD, H, 0, 0, 0, q * G.HK / 32, KSZ, l0t);
  jobs.j[4] = wide_make_block(w->wh1 + (size_t)H * HD, 1, HD, HD, H, 0, 0, G.NTL, 3 * G.HK / 32, KSZ, l0t);
  jobs.j[5] = wide_make_block(w->wh1, 1, HD, HD, H, 0, 0, G.NTL, 4 * G.HK / 32, KSZ, l0t);
  if (int rc = wide_launch_pack(jobs, P, stream)) return rc;
  // the cells transposed: c_dl[i] = sum_j wdl[j][i] du_l[j];  c_dh likewise;  c_s[s] = sum_j wxl_s_t[s][j] du_l[j] (+ wxh_t[s - LS][j] du_h[j])
  jobs.count = 4;
  jobs.j[0] = wide_make_block(w->wdl, 1, LD, LD, LD, 0, 0, 0, 0, KSU, rnnt);
  jobs.j[1] = wide_make_block(w->wdh, 1, HD, HD, HD, 0, 0, G.NTL, G.KLD / 32, KSU, rnnt);
  jobs.j[2] = wide_make_block(w->wxl_s_t, LD, 1, LS + HS, LD, 0, 0, NTD, 0, KSU, rnnt);
  jobs.j[3] = wide_make_block(w->wxh_t, HD, 1, LS + HS, HD, LS, 0, NTD, G.KLD / 32, KSU, rnnt);
  if (int rc = wide_launch_pack(jobs, P, stream)) return rc;

  MmtWideBwdArgs a;
  a.dm = *d; a.io = *io;
  for (int q = 0; q < 5; ++q) {
    a.pk_l1t[q] = reinterpret_cast<const uint4*>(ws + L.l1t[q]);
    a.x_dl[q] = reinterpret_cast<uint4*>(ws + L.x_dl[q]);
  }
  a.pk_l0t = l0t; a.pk_rnnt = rnnt;
  a.x_dz = reinterpret_cast<uint4*>(ws + L.x_dz);
  a.x_du = reinterpret_cast<uint4*>(ws + L.x_du);
  a.cs = reinterpret_cast<float*>(ws + L.cs);
  a.ctl = ws; a.status = reinterpret_cast<int*>(ws);
  a.nblk = device_cu_count();
  a.acquire = mmt_acquire_fence();
  const size_t lds = mmt_bwd_lds(G);
  if (lds > 160 * 1024) { set_error("mmtrssm_rollout_bwd_wide: %zu bytes of LDS", lds); return MTRSSM_ELDS; }
  hipError_t e;
#define MTRSSM_MMT_WIDE_BWD(PV)                                                                                                        \
  {                                                                                                                                   \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mmtrssm_wide_bwd_kernel<PV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e)); return MTRSSM_ELAUNCH; }     \
    set_last_kernel("mtrssm::mmtrssm_wide_bwd_kernel<" #PV ">");                                                                      \
    hipLaunchKernelGGL(mmtrssm_wide_bwd_kernel<PV>, dim3(a.nblk), dim3(kWT), lds, stream, a);                                        \
  }
  if (pieces == 3) MTRSSM_MMT_WIDE_BWD(3) else MTRSSM_MMT_WIDE_BWD(2)
#undef MTRSSM_MMT_WIDE_BWD
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("wide MMTRSSM backward scan launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  return MTRSSM_OK;
}

}  // namespace mtrssm
