// MoPoE-MRSSM scan for LARGE deterministic / hidden sizes: all CUs of the chip on one tile of 32 batch rows, MFMA products.
// (wide_common.h has the regime, the packed layouts and the barrier.)  Replaces the loop body of
// mrssm/mopoe_mrssm/core.py:221-256 and its BPTT where mrssm_scan.hip's one-CU-per-row form re-streams 42 MB per row-step.
//
// Forward, per timestep t (four grid barriers):
//   A  [one workgroup per batch row]  finish step t-1: logits + bias -> MoPoE mix -> KL, samples (one wave); then
//                                     h1 = act(xa_t + sum of the sampled columns of W1s)  (fp32, a K-term gather)
//   B  [one workgroup per 16 deter columns]  gi = (W_ih W2) h1, gh = W_hh d: three gate tiles x two streams on the MFMA,
//                                     GRU gates, d_t (the workgroup keeps its d columns in registers across steps)
//   C  [one per 16 head-layer-0 columns]     hd = act(Wh1 d_t + {b3 | pa_t | pv_t})
//   D  [one per 16 logits]                   raw logits of the prior / audio / vision heads
// Backward, per timestep (five barriers): R0 categorical block + mix backward per row; R1 dzh = act'(hd) W2nd^T dl;
// R2 dd = g + carry + Wh1^T dzh and the gate gradients; R3 carry_d += W_hh^T dgh | dz1 = act'(h1) (W_ih W2)^T dgi;
// R4 carry_s = W1s^T dz1.  As in the four-CU cluster kernels the GRU input path is fused (wf_t = (W_ih W2)^T, sv_h2 / d_h2
// are not produced: the caller forms them as batched GEMMs).
#include <cstdlib>

#include "wide_common.h"

namespace mtrssm {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);
int device_cu_count();

// ------------------------------------------------------------------------------------------------
// weight packing (once per launch: the weights change every optimizer step)
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void wide_pack_kernel(const WidePackJobs jobs) {
  const WidePackJob jb = jobs.j[blockIdx.y];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const long nfrag = (long)jb.NT * jb.KS;
  for (long f = (long)blockIdx.x * 4 + sub; f < nfrag; f += (long)gridDim.x * 4) {
    const int nt = (int)(f / jb.KS), ks = (int)(f - (long)nt * jb.KS);
    const int n = nt * 16 + (lane & 15), k0 = ks * 32 + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = (n < jb.N && n >= jb.nskip && k0 + j < jb.K && k0 + j >= jb.kskip) ? jb.src[(long)(n - jb.nskip) * jb.sn + (long)(k0 + j - jb.kskip) * jb.sk] : 0.f;
    uint2 lo[P], hi[P];
    const float v0[4] = {v[0], v[1], v[2], v[3]}, v1[4] = {v[4], v[5], v[6], v[7]};
    wide_split4<P>(v0, lo);
    wide_split4<P>(v1, hi);
    const size_t frag = (size_t)(jb.nt0 + nt) * jb.KST + (jb.ks0 + ks);
#pragma unroll
    for (int p = 0; p < P; ++p) jb.dst[(frag * P + p) * 64 + lane] = make_uint4(lo[p].x, lo[p].y, hi[p].x, hi[p].y);
  }
}

WidePackJob wide_make_job(const float* src, long sn, long sk, int N, int K, uint4* dst) {
  WidePackJob j;
  j.src = src; j.sn = sn; j.sk = sk; j.N = N; j.K = K; j.nskip = 0; j.kskip = 0; j.NT = (N + 15) / 16; j.KS = (K + 31) / 32;
  j.nt0 = 0; j.ks0 = 0; j.KST = j.KS; j.dst = dst;
  return j;
}
// a block of a larger packed matrix (KST k-blocks per tile) at tile nt0 / k-block ks0
WidePackJob wide_make_block(const float* src, long sn, long sk, int N, int K, int nskip, int kskip, int nt0, int ks0, int KST, uint4* dst) {
  WidePackJob j = wide_make_job(src, sn, sk, N, K, dst);
  j.nskip = nskip; j.kskip = kskip; j.nt0 = nt0; j.ks0 = ks0; j.KST = KST;
  return j;
}
static WidePackJob make_job(const float* src, long sn, long sk, int N, int K, uint4* dst) { return wide_make_job(src, sn, sk, N, K, dst); }

int wide_launch_pack(const WidePackJobs& jobs, int pieces, hipStream_t stream);
static int launch_pack(const WidePackJobs& jobs, int pieces, hipStream_t stream) { return wide_launch_pack(jobs, pieces, stream); }
int wide_launch_pack(const WidePackJobs& jobs, int pieces, hipStream_t stream) {
  dim3 grid(256, jobs.count);
  if (pieces == 3) hipLaunchKernelGGL(wide_pack_kernel<3>, grid, dim3(256), 0, stream, jobs);
  else hipLaunchKernelGGL(wide_pack_kernel<2>, grid, dim3(256), 0, stream, jobs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("wide scan: weight pack launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  return MTRSSM_OK;
}

// Development aid (tools/wide_probe.py): when set, lane 0 of EVERY workgroup stamps s_memrealtime (100 MHz) at the phase
// boundaries of timesteps 8..11 of the first row tile into this buffer: [workgroup][step - 8][16 stamps].  Null in normal use.
__device__ unsigned long long* g_wide_prof = nullptr;
#define MTRSSM_WIDE_STAMP(i)                                                                                              \
  do {                                                                                                                    \
    if (prof && tstamp >= 8 && tstamp < 12) prof[((size_t)blockIdx.x * 4 + (tstamp - 8)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
constexpr int kNS1 = 4, kNS3 = 2;   // register stages of the operand ring: one-tile units / the GRU's three-tile units
constexpr int kAQ = 4;   // float4 columns of h1 per thread in the row phase: H <= 16 kWT

struct WideFwdArgs {
  MtrssmMrssmDims dm;
  MtrssmMrssmClusterWeights w;
  MtrssmMrssmFwdIO io;
  const uint4 *pk_wf, *pk_whh, *pk_wh1, *pk_h2[3];   // packed weights (pk_h2: prior / audio / vision second layers)
  uint4 *x_h1, *x_d[2], *x_hd[3];                     // exchange vectors
  float* lg;                                          // [32][3][Sp] raw logits of the step (without bias)
  void* ctl;                                           // control block (barrier words)
  int* status;
  int nblk;
  int acquire;                                         // 1: an agent acquire fence after every barrier (A/B switch)
};

template <int P>
__global__ __launch_bounds__(kWT) void mrssm_wide_fwd_kernel(const WideFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MtrssmMrssmFwdIO& io = a.io;
  const MtrssmMrssmClusterWeights& w = a.w;
  const int B = a.dm.B, T = a.dm.T, D = a.dm.D, H = a.dm.H, K = a.dm.K, C = a.dm.C, S = K * C, act = a.dm.act;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KSH = (H + 31) / 32, KSD = (D + 31) / 32;
  const int NTD = D / 16, NTH = H / 16, NTS = (S + 15) / 16, Sp = NTS * 16;
  const int nblk = a.nblk, blk = blockIdx.x;

  // LDS: reduction scratch (4 waves x 4 tiles x 2 row tiles x 64 lanes x 16 B = 32 KiB), then the row phase's vectors
  wf32x4* red = reinterpret_cast<wf32x4*>(lds);
  float* rowv = lds + kWW * 4 * 2 * kWave * 4;
  float* Llp = rowv, *Lla = Llp + Sp, *Llv = Lla + Sp, *Lmx = Llv + Sp, *Ls = Lmx + Sp, *Lu = Ls + Sp;   // Lu: [2][64] uniforms
  int* abort_flag = reinterpret_cast<int*>(Lu + 128);
  int* Lidx = abort_flag + 4;   // [64] flat index of the sampled class of each categorical
  if (tid == 0) *abort_flag = 0;
  __syncthreads();
  WideBarrier bar;
  bar.init(a.ctl, a.status, abort_flag, nblk, blk, a.acquire != 0);

  const int ks0h = KSH * wave / kWW, ks1h = KSH * (wave + 1) / kWW;
  const int ks0d = KSD * wave / kWW, ks1d = KSD * (wave + 1) / kWW;
  const size_t tileH = (size_t)KSH * P * 64, tileD = (size_t)KSD * P * 64;   // uint4 per packed n-tile (K = H / K = D)

  // epilogue item of this thread (threads 0..127): row tile, lane slot -> batch row of the tile, four consecutive columns
  const int e_rt = tid >> 6, e_slot = lane, e_row = 16 * e_rt + (e_slot & 15), e_cq = 4 * (e_slot >> 4);
  const bool e_thread = tid < 2 * kWave;

  for (int rb = 0; rb < B; rb += kWRows) {
    const int nrows = B - rb < kWRows ? B - rb : kWRows;
    const bool e_valid = e_thread && e_row < nrows;
    const size_t e_b = (size_t)(rb + (e_row < nrows ? e_row : 0));
    // ---- set-up of the tile: x_d[0] <- deter0 (every workgroup converts a slice), zero rows beyond the batch
    for (int i = blk * kWT + tid; i < kWRows * (D / 4); i += nblk * kWT) {
      const int row = i / (D / 4), k = (i - row * (D / 4)) * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (row < nrows) {
        const float4 q = *reinterpret_cast<const float4*>(io.deter0 + (size_t)(rb + row) * D + k);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      }
      wide_x_store4<P>(a.x_d[0], KSD, row, k, v);
      if (row >= nrows) {  // rows beyond the batch stay zero in every exchange vector
        wide_x_store4<P>(a.x_d[1], KSD, row, k, v);
      }
    }
    for (int i = blk * kWT + tid; i < kWRows * (H / 4); i += nblk * kWT) {
      const int row = i / (H / 4), k = (i - row * (H / 4)) * 4;
      if (row >= nrows) {
        const float v[4] = {0.f, 0.f, 0.f, 0.f};
        wide_x_store4<P>(a.x_h1, KSH, row, k, v);
#pragma unroll
        for (int q = 0; q < 3; ++q) wide_x_store4<P>(a.x_hd[q], KSH, row, k, v);
      }
    }
    // the deter columns this workgroup owns in phase B live in registers across the steps
    float dprev[4] = {0.f, 0.f, 0.f, 0.f};
    if (blk < NTD && e_valid) {
      const float4 q = *reinterpret_cast<const float4*>(io.deter0 + e_b * D + blk * 16 + e_cq);
      dprev[0] = q.x; dprev[1] = q.y; dprev[2] = q.z; dprev[3] = q.w;
    }
    int cur = 0;
    unsigned long long* const prof = (tid == 0 && rb == 0) ? g_wide_prof : nullptr;

    for (int t = 0; t <= T; ++t) {
      const int tstamp = t;
      MTRSSM_WIDE_STAMP(0);
      // ============ phase A: one workgroup per batch row (the last workgroups of the grid) ============
      for (int r = nblk - 1 - blk; r < nrows; r += nblk) {
        const size_t b = (size_t)(rb + r);
        // streamed inputs first: their HBM latency hides behind the categorical block
        float4 xa4[kAQ];
        const int hq = H / 4;
        if (t < T) {
#pragma unroll
          for (int i = 0; i < kAQ; ++i)
            if (tid + i * kWT < hq) xa4[i] = *reinterpret_cast<const float4*>(io.xa + (b * T + t) * H + (size_t)(tid + i * kWT) * 4);
        }
        if (t > 0) {
          const size_t q = b * T + (t - 1);
          if (wave == 0 && lane < K) {
            Lu[lane] = io.u_post[q * K + lane];
            Lu[64 + lane] = io.u_prior ? io.u_prior[q * K + lane] : 0.f;
          }
          for (int i = tid; i < 3 * S; i += kWT) {
            const int which = i / S, s2 = i - which * S;
            const float bias = (which == 0 ? w.b4 : (which == 1 ? w.ba2 : w.bv2))[s2];
            (which == 0 ? Llp : (which == 1 ? Lla : Llv))[s2] = wide_load_f(a.lg + ((size_t)r * 3 + which) * Sp + s2) + bias;
          }
          lds_barrier();
          MTRSSM_WIDE_STAMP(10);
          if (wave == 0) {
            wave_mopoe_mix<true>(Lla, Llv, Lmx, S, lane);
            MTRSSM_WIDE_STAMP(11);
            for (int s2 = lane; s2 < S; s2 += kWave) {
              io.prior_logits[q * S + s2] = Llp[s2];
              io.post_logits[q * S + s2] = Lmx[s2];
              if (io.sv_la) { io.sv_la[q * S + s2] = Lla[s2]; io.sv_lv[q * S + s2] = Llv[s2]; }
            }
            float kl = C <= 8 ? cat_block_fwd_fast8(Lmx, Llp, K, C, lane, Lu, io.u_prior ? Lu + 64 : nullptr, Ls, io.post_stoch + q * S,
                                                    io.prior_stoch ? io.prior_stoch + q * S : nullptr, true)
                              : cat_block_fwd<true, true>(Lmx, Llp, K, C, lane, Lu, io.u_prior ? Lu + 64 : nullptr, Ls, io.post_stoch + q * S,
                                                          io.prior_stoch ? io.prior_stoch + q * S : nullptr, true);
            if (io.kl) {
              kl = wave_sum(kl);
              if (lane == 0) io.kl[q] = kl;
            }
            MTRSSM_WIDE_STAMP(12);
          }
          lds_barrier();
        } else {
          for (int i = tid; i < S; i += kWT) Ls[i] = io.stoch0[b * S + i];
          lds_barrier();
        }
        if (t < T) {
          // h1 = act(xa + W1s s): fp32.  t > 0: s is one-hot per categorical -> the sum of K rows of w1s_t (row indices from the
          // sample, the K loads of a thread issued eight at a time); t = 0: stoch0 is whatever the caller passed: dense.
          if (t > 0) {
            if (tid < K) {
              int idx = 0;
              for (int c = 1; c < C; ++c) idx = Ls[tid * C + c] != 0.f ? c : idx;
              Lidx[tid] = tid * C + idx;
            }
            lds_barrier();
          }
#pragma unroll
          for (int i = 0; i < kAQ; ++i) {
            const int c4 = tid + i * kWT;
            if (c4 < hq) {
              float acc[4] = {xa4[i].x, xa4[i].y, xa4[i].z, xa4[i].w};
              const float* wc = w.w1s_t + (size_t)c4 * 4;
              if (t > 0) {
                for (int k0 = 0; k0 < K; k0 += 8) {
                  float4 q4[8];
#pragma unroll
                  for (int j = 0; j < 8; ++j) q4[j] = *reinterpret_cast<const float4*>(wc + (size_t)Lidx[k0 + j < K ? k0 + j : K - 1] * H);
#pragma unroll
                  for (int j = 0; j < 8; ++j)
                    if (k0 + j < K) { acc[0] += q4[j].x; acc[1] += q4[j].y; acc[2] += q4[j].z; acc[3] += q4[j].w; }
                }
              } else {
                for (int s2 = 0; s2 < S; ++s2) {
                  const float sv = Ls[s2];
                  const float4 q4 = *reinterpret_cast<const float4*>(wc + (size_t)s2 * H);
                  acc[0] = fmaf(q4.x, sv, acc[0]); acc[1] = fmaf(q4.y, sv, acc[1]); acc[2] = fmaf(q4.z, sv, acc[2]); acc[3] = fmaf(q4.w, sv, acc[3]);
                }
              }
              float h[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) h[j] = act_fwd(acc[j], act);
              if (io.sv_h1) *reinterpret_cast<float4*>(io.sv_h1 + (b * T + t) * H + (size_t)c4 * 4) = make_float4(h[0], h[1], h[2], h[3]);
              wide_x_store4<P>(a.x_h1, KSH, r, c4 * 4, h);
            }
          }
        }
        lds_barrier();
      }
      if (t == T) break;
      MTRSSM_WIDE_STAMP(1);
      if (!bar.sync(1 + 4 * t)) return;
      MTRSSM_WIDE_STAMP(2);

      // ============ phase B: GRU, one workgroup per 16 deter columns ============
      for (int u = blk; u < NTD; u += nblk) {
        wf32x4 accA[3][2], accB[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) { accA[g][rt] = wf32x4{0.f, 0.f, 0.f, 0.f}; accB[g][rt] = wf32x4{0.f, 0.f, 0.f, 0.f}; }
        const uint4* const wtA[3] = {a.pk_wf + (size_t)u * tileH, a.pk_wf + (size_t)(NTD + u) * tileH, a.pk_wf + (size_t)(2 * NTD + u) * tileH};
        const uint4* const wtB[3] = {a.pk_whh + (size_t)u * tileD, a.pk_whh + (size_t)(NTD + u) * tileD, a.pk_whh + (size_t)(2 * NTD + u) * tileD};
        wide_mfma_stream<3, P, kNS3>(accB, wtB, a.x_d[cur], KSD, ks0d, ks1d, lane);   // gh first: its operand has been there longest
        wide_mfma_stream<3, P, kNS3>(accA, wtA, a.x_h1, KSH, ks0h, ks1h, lane);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) { accA[0][rt] += accB[0][rt]; accA[1][rt] += accB[1][rt]; }
        wide_red_store<4>(red, wave, 0, lane, accA[0]);
        wide_red_store<4>(red, wave, 1, lane, accA[1]);
        wide_red_store<4>(red, wave, 2, lane, accA[2]);
        wide_red_store<4>(red, wave, 3, lane, accB[2]);
        lds_barrier();
        if (e_thread) {
          const wf32x4 sr = wide_red_sum<4>(red, 0, e_rt, e_slot), sz = wide_red_sum<4>(red, 1, e_rt, e_slot);
          const wf32x4 sni = wide_red_sum<4>(red, 2, e_rt, e_slot), snh = wide_red_sum<4>(red, 3, e_rt, e_slot);
          const int c = u * 16 + e_cq;
          float rg[4], zg[4], ng[4], gn[4], dn[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float pr = sr[j] + w.bf[c + j] + w.bhh[c + j];
            const float pz = sz[j] + w.bf[D + c + j] + w.bhh[D + c + j];
            gn[j] = snh[j] + w.bhh[2 * D + c + j];
            rg[j] = sigmoidf_(pr);
            zg[j] = sigmoidf_(pz);
            ng[j] = tanhf(sni[j] + w.bf[2 * D + c + j] + gn[j] * rg[j]);
            dn[j] = (dprev[j] - ng[j]) * zg[j] + ng[j];
            dprev[j] = dn[j];
          }
          if (e_valid) {
            const size_t q = e_b * T + t;
            *reinterpret_cast<float4*>(io.deter + q * D + c) = make_float4(dn[0], dn[1], dn[2], dn[3]);
            if (io.sv_gates) {
              float* gs = io.sv_gates + q * 4 * D + c;
              *reinterpret_cast<float4*>(gs) = make_float4(rg[0], rg[1], rg[2], rg[3]);
              *reinterpret_cast<float4*>(gs + D) = make_float4(zg[0], zg[1], zg[2], zg[3]);
              *reinterpret_cast<float4*>(gs + 2 * D) = make_float4(ng[0], ng[1], ng[2], ng[3]);
              *reinterpret_cast<float4*>(gs + 3 * D) = make_float4(gn[0], gn[1], gn[2], gn[3]);
            }
            wide_x_store4<P>(a.x_d[cur ^ 1], KSD, e_row, c, dn);
          }
        }
        lds_barrier();
      }
      cur ^= 1;
      MTRSSM_WIDE_STAMP(3);
      if (!bar.sync(2 + 4 * t)) return;
      MTRSSM_WIDE_STAMP(4);

      // ============ phase C: head layer 0 on the new deter, one workgroup per 16 of the 3H columns ============
      for (int u = blk; u < 3 * NTH; u += nblk) {
        const int which = u / NTH, c = (u - which * NTH) * 16 + e_cq;
        float4 pin = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e_valid) {
          if (which == 0) pin = *reinterpret_cast<const float4*>(w.b3 + c);
          else pin = *reinterpret_cast<const float4*>((which == 1 ? io.pa : io.pv) + (e_b * T + t) * H + c);
        }
        wf32x4 acc[1][2] = {{wf32x4{0.f, 0.f, 0.f, 0.f}, wf32x4{0.f, 0.f, 0.f, 0.f}}};
        const uint4* const wt[1] = {a.pk_wh1 + (size_t)u * tileD};
        wide_mfma_stream<1, P, kNS1>(acc, wt, a.x_d[cur], KSD, ks0d, ks1d, lane);
        wide_red_store<1>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        if (e_valid) {
          const wf32x4 sm = wide_red_sum<1>(red, 0, e_rt, e_slot);
          float h[4] = {act_fwd(sm[0] + pin.x, act), act_fwd(sm[1] + pin.y, act), act_fwd(sm[2] + pin.z, act), act_fwd(sm[3] + pin.w, act)};
          if (io.sv_heads) *reinterpret_cast<float4*>(io.sv_heads + (e_b * T + t) * 3 * H + which * H + c) = make_float4(h[0], h[1], h[2], h[3]);
          wide_x_store4<P>(a.x_hd[which], KSH, e_row, c, h);
        }
        lds_barrier();
      }
      MTRSSM_WIDE_STAMP(5);
      if (!bar.sync(3 + 4 * t)) return;
      MTRSSM_WIDE_STAMP(6);

      // ============ phase D: head layer 1, one workgroup per 16 logits ============
      for (int u = blk; u < 3 * NTS; u += nblk) {
        const int which = u / NTS, st = u - which * NTS;
        wf32x4 acc[1][2] = {{wf32x4{0.f, 0.f, 0.f, 0.f}, wf32x4{0.f, 0.f, 0.f, 0.f}}};
        const uint4* const wt[1] = {a.pk_h2[which] + (size_t)st * tileH};
        wide_mfma_stream<1, P, kNS1>(acc, wt, a.x_hd[which], KSH, ks0h, ks1h, lane);
        wide_red_store<1>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        if (e_valid) {
          const wf32x4 sm = wide_red_sum<1>(red, 0, e_rt, e_slot);
          float* dst = a.lg + ((size_t)e_row * 3 + which) * Sp + st * 16 + e_cq;
          wide_store_f2(dst, sm[0], sm[1]);
          wide_store_f2(dst + 2, sm[2], sm[3]);
        }
        lds_barrier();
      }
      MTRSSM_WIDE_STAMP(7);
      if (!bar.sync(4 + 4 * t)) return;
      MTRSSM_WIDE_STAMP(8);
    }
    if (!bar.sync(0x40000000)) return;   // the next tile's set-up overwrites the exchange vectors
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct WideBwdArgs {
  MtrssmMrssmDims dm;
  MtrssmMrssmBwdIO io;
  const uint4 *pk_h2t[3];     // second head layers transposed: N = H, K = S
  const uint4 *pk_wh1t;       // N = D, K = 3H
  const uint4 *pk_whht;       // N = D, K = 3D
  const uint4 *pk_wft;        // N = H, K = 3D
  const uint4 *pk_w1s;        // N = S, K = H
  uint4 *x_dl[3], *x_dzh, *x_dgi, *x_dgh, *x_dz1;
  float* cs;                  // [32][Sp] carry_s of the step
  void* ctl;                                           // control block (barrier words)
  int* status;
  int nblk;
  int acquire;                                         // 1: an agent acquire fence after every barrier (A/B switch)
};

template <int P>
__global__ __launch_bounds__(kWT) void mrssm_wide_bwd_kernel(const WideBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MtrssmMrssmBwdIO& io = a.io;
  const int B = a.dm.B, T = a.dm.T, D = a.dm.D, H = a.dm.H, K = a.dm.K, C = a.dm.C, S = K * C, act = a.dm.act;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KSH = (H + 31) / 32, KSS = (S + 31) / 32, KS3H = (3 * H + 31) / 32, KS3D = (3 * D + 31) / 32;
  const int NTD = D / 16, NTH = H / 16, NTS = (S + 15) / 16, Sp = NTS * 16;
  const int nblk = a.nblk, blk = blockIdx.x;

  wf32x4* red = reinterpret_cast<wf32x4*>(lds);   // 4 waves x 1 tile x 2 row tiles x 64 lanes x 16 B = 8 KiB
  float* rowv = lds + kWW * 1 * 2 * kWave * 4;
  float* Lla = rowv, *Llv = Lla + Sp, *Lmx = Llv + Sp, *Llp = Lmx + Sp, *Ldmx = Llp + Sp, *Ldlp = Ldmx + Sp, *Ldla = Ldlp + Sp, *Ldlv = Ldla + Sp;
  float* Lcs = Ldlv + Sp, *Lgps = Lcs + Sp;
  int* abort_flag = reinterpret_cast<int*>(Lgps + Sp);
  if (tid == 0) *abort_flag = 0;
  __syncthreads();
  WideBarrier bar;
  bar.init(a.ctl, a.status, abort_flag, nblk, blk, a.acquire != 0);

  auto krange = [&](int KS, int& k0, int& k1) { k0 = KS * wave / kWW; k1 = KS * (wave + 1) / kWW; };
  const size_t tileS = (size_t)KSS * P * 64, tile3H = (size_t)KS3H * P * 64, tile3D = (size_t)KS3D * P * 64, tileH = (size_t)KSH * P * 64;
  const int e_rt = tid >> 6, e_slot = lane, e_row = 16 * e_rt + (e_slot & 15), e_cq = 4 * (e_slot >> 4);
  const bool e_thread = tid < 2 * kWave;

  for (int rb = 0; rb < B; rb += kWRows) {
    const int nrows = B - rb < kWRows ? B - rb : kWRows;
    const bool e_valid = e_thread && e_row < nrows;
    const size_t e_b = (size_t)(rb + (e_row < nrows ? e_row : 0));
    // rows beyond the batch: zero in every exchange vector (each workgroup clears a slice)
    {
      const float z4[4] = {0.f, 0.f, 0.f, 0.f};
      for (int i = blk * kWT + tid; i < kWRows * (3 * D / 4); i += nblk * kWT) {
        const int row = i / (3 * D / 4), k = (i - row * (3 * D / 4)) * 4;
        if (row >= nrows) { wide_x_store4<P>(a.x_dgi, KS3D, row, k, z4); wide_x_store4<P>(a.x_dgh, KS3D, row, k, z4); }
      }
      for (int i = blk * kWT + tid; i < kWRows * (3 * H / 4); i += nblk * kWT) {
        const int row = i / (3 * H / 4), k = (i - row * (3 * H / 4)) * 4;
        if (row >= nrows) wide_x_store4<P>(a.x_dzh, KS3H, row, k, z4);
      }
      for (int i = blk * kWT + tid; i < kWRows * (H / 4); i += nblk * kWT) {
        const int row = i / (H / 4), k = (i - row * (H / 4)) * 4;
        if (row >= nrows) wide_x_store4<P>(a.x_dz1, KSH, row, k, z4);
      }
      for (int i = blk * kWT + tid; i < kWRows * (Sp / 4); i += nblk * kWT) {
        const int row = i / (Sp / 4), k = (i - row * (Sp / 4)) * 4;
        if (row >= nrows)
#pragma unroll
          for (int q = 0; q < 3; ++q) wide_x_store4<P>(a.x_dl[q], KSS, row, k, z4);
      }
    }
    float cd[4] = {0.f, 0.f, 0.f, 0.f};    // carry into d_{t-1}: owned by the workgroup of these 16 deter columns (phases R2, R3)
    float ddz[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned long long* const prof = (tid == 0 && rb == 0) ? g_wide_prof : nullptr;

    for (int t = T - 1; t >= 0; --t) {
      const int tstamp = T - 1 - t;
      MTRSSM_WIDE_STAMP(0);
      // ============ R0: categorical block + MoPoE mix backward, one workgroup per batch row ============
      for (int r = nblk - 1 - blk; r < nrows; r += nblk) {
        const size_t q = (size_t)(rb + r) * T + t;
        for (int s2 = tid; s2 < S; s2 += kWT) {
          Lla[s2] = io.sv_la[q * S + s2];
          Llv[s2] = io.sv_lv[q * S + s2];
          Lmx[s2] = io.post_logits[q * S + s2];
          Llp[s2] = io.prior_logits[q * S + s2];
          Lcs[s2] = t == T - 1 ? 0.f : wide_load_f(a.cs + (size_t)r * Sp + s2);
          Lgps[s2] = io.g_post_stoch ? io.g_post_stoch[q * S + s2] : 0.f;   // (read twice per class by one lane: from LDS, not L2)
        }
        lds_barrier();
        MTRSSM_WIDE_STAMP(11);
        if (wave == 0) {
          const float gk = io.g_kl ? io.g_kl[q] : 0.f;
          if (C <= 8)
            cat_block_bwd_fast8(Lmx, Llp, K, C, lane, Lgps, Lcs, io.g_prior_stoch ? io.g_prior_stoch + q * S : nullptr,
                                io.g_post_logits ? io.g_post_logits + q * S : nullptr, io.g_prior_logits ? io.g_prior_logits + q * S : nullptr,
                                gk, a.dm.kl_w_post, a.dm.kl_w_prior, Ldmx, Ldlp);
          else
            cat_block_bwd<true>(Lmx, Llp, K, C, lane, Lgps, Lcs, io.g_prior_stoch ? io.g_prior_stoch + q * S : nullptr,
                                io.g_post_logits ? io.g_post_logits + q * S : nullptr, io.g_prior_logits ? io.g_prior_logits + q * S : nullptr,
                                gk, a.dm.kl_w_post, a.dm.kl_w_prior, Ldmx, Ldlp);
          MTRSSM_WIDE_STAMP(12);
          wave_mopoe_mix_bwd<true>(Lla, Llv, Lmx, Ldmx, Ldla, Ldlv, S, lane);
          MTRSSM_WIDE_STAMP(13);
        }
        lds_barrier();
        for (int i = tid; i < 3 * (Sp / 4); i += kWT) {
          const int which = i / (Sp / 4), s4 = (i - which * (Sp / 4)) * 4;
          const float* src = which == 0 ? Ldlp : (which == 1 ? Ldla : Ldlv);
          float* dst = (which == 0 ? io.d_lp : (which == 1 ? io.d_la : io.d_lv)) + q * S;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = s4 + j < S ? src[s4 + j] : 0.f;
            if (s4 + j < S) dst[s4 + j] = v[j];
          }
          wide_x_store4<P>(a.x_dl[which], KSS, r, s4, v);
        }
        lds_barrier();
      }
      MTRSSM_WIDE_STAMP(1);
      if (!bar.sync(1 + 5 * t)) return;
      MTRSSM_WIDE_STAMP(2);

      // ============ R1: dzh = act'(hd) * (W2nd^T dl), one workgroup per 16 of the 3H head units ============
      for (int u = blk; u < 3 * NTH; u += nblk) {
        const int which = u / NTH, c = (u - which * NTH) * 16 + e_cq;
        float4 hd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e_valid) hd = *reinterpret_cast<const float4*>(io.sv_heads + (e_b * T + t) * 3 * H + which * H + c);
        int k0, k1;
        krange(KSS, k0, k1);
        wf32x4 acc[1][2] = {{wf32x4{0.f, 0.f, 0.f, 0.f}, wf32x4{0.f, 0.f, 0.f, 0.f}}};
        const uint4* const wt[1] = {a.pk_h2t[which] + (size_t)(u - which * NTH) * tileS};
        wide_mfma_stream<1, P, kNS1>(acc, wt, a.x_dl[which], KSS, k0, k1, lane);
        wide_red_store<1>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        if (e_valid) {
          const wf32x4 sm = wide_red_sum<1>(red, 0, e_rt, e_slot);
          float g[4] = {sm[0] * act_grad_from_out(hd.x, act), sm[1] * act_grad_from_out(hd.y, act), sm[2] * act_grad_from_out(hd.z, act),
                        sm[3] * act_grad_from_out(hd.w, act)};
          *reinterpret_cast<float4*>(io.d_zh + (e_b * T + t) * 3 * H + which * H + c) = make_float4(g[0], g[1], g[2], g[3]);
          wide_x_store4<P>(a.x_dzh, KS3H, e_row, which * H + c, g);
        }
        lds_barrier();
      }
      MTRSSM_WIDE_STAMP(3);
      if (!bar.sync(2 + 5 * t)) return;
      MTRSSM_WIDE_STAMP(4);

      // ============ R2: dd = g_deter + carry + Wh1^T dzh, GRU gate gradients; one workgroup per 16 deter columns ============
      for (int u = blk; u < NTD; u += nblk) {
        const int c = u * 16 + e_cq;
        float4 gd = make_float4(0.f, 0.f, 0.f, 0.f), rg4 = gd, zg4 = gd, ng4 = gd, gn4 = gd, dp4 = gd;
        if (e_valid) {
          const size_t q = e_b * T + t;
          if (io.g_deter) gd = *reinterpret_cast<const float4*>(io.g_deter + q * D + c);
          const float* gs = io.sv_gates + q * 4 * D + c;
          rg4 = *reinterpret_cast<const float4*>(gs);
          zg4 = *reinterpret_cast<const float4*>(gs + D);
          ng4 = *reinterpret_cast<const float4*>(gs + 2 * D);
          gn4 = *reinterpret_cast<const float4*>(gs + 3 * D);
          dp4 = *reinterpret_cast<const float4*>(t > 0 ? io.deter + (q - 1) * D + c : io.deter0 + e_b * D + c);
        }
        int k0, k1;
        krange(KS3H, k0, k1);
        wf32x4 acc[1][2] = {{wf32x4{0.f, 0.f, 0.f, 0.f}, wf32x4{0.f, 0.f, 0.f, 0.f}}};
        const uint4* const wt[1] = {a.pk_wh1t + (size_t)u * tile3H};
        wide_mfma_stream<1, P, kNS1>(acc, wt, a.x_dzh, KS3H, k0, k1, lane);
        wide_red_store<1>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        if (e_valid) {
          const wf32x4 sm = wide_red_sum<1>(red, 0, e_rt, e_slot);
          const float rg[4] = {rg4.x, rg4.y, rg4.z, rg4.w}, zg[4] = {zg4.x, zg4.y, zg4.z, zg4.w}, ng[4] = {ng4.x, ng4.y, ng4.z, ng4.w};
          const float gn[4] = {gn4.x, gn4.y, gn4.z, gn4.w}, dp[4] = {dp4.x, dp4.y, dp4.z, dp4.w}, gdv[4] = {gd.x, gd.y, gd.z, gd.w};
          float gi[3][4], gh[3][4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float dd = sm[j] + gdv[j] + cd[j];
            const float dn = dd * (1.f - zg[j]);
            const float dz = dd * (dp[j] - ng[j]);
            const float dn_pre = dn * (1.f - ng[j] * ng[j]);
            const float dr = dn_pre * gn[j];
            const float dr_pre = dr * rg[j] * (1.f - rg[j]);
            const float dz_pre = dz * zg[j] * (1.f - zg[j]);
            ddz[j] = dd * zg[j];
            gi[0][j] = dr_pre; gi[1][j] = dz_pre; gi[2][j] = dn_pre;
            gh[0][j] = dr_pre; gh[1][j] = dz_pre; gh[2][j] = dn_pre * rg[j];
          }
          const size_t q = e_b * T + t;
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            *reinterpret_cast<float4*>(io.d_gi + q * 3 * D + g * D + c) = make_float4(gi[g][0], gi[g][1], gi[g][2], gi[g][3]);
            *reinterpret_cast<float4*>(io.d_gh + q * 3 * D + g * D + c) = make_float4(gh[g][0], gh[g][1], gh[g][2], gh[g][3]);
            wide_x_store4<P>(a.x_dgi, KS3D, e_row, g * D + c, gi[g]);
            wide_x_store4<P>(a.x_dgh, KS3D, e_row, g * D + c, gh[g]);
          }
        }
        lds_barrier();
      }
      MTRSSM_WIDE_STAMP(5);
      if (!bar.sync(3 + 5 * t)) return;
      MTRSSM_WIDE_STAMP(6);

      // ============ R3: carry_d = dd z + W_hh^T dgh (the workgroups of R2)  |  dz1 = act'(h1) (W_ih W2)^T dgi ============
      for (int u = blk; u < NTD + NTH; u += nblk) {
        const bool is_cd = u < NTD;
        const int c = (is_cd ? u : u - NTD) * 16 + e_cq;
        float4 h1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!is_cd && e_valid) h1 = *reinterpret_cast<const float4*>(io.sv_h1 + (e_b * T + t) * H + c);
        int k0, k1;
        krange(KS3D, k0, k1);
        wf32x4 acc[1][2] = {{wf32x4{0.f, 0.f, 0.f, 0.f}, wf32x4{0.f, 0.f, 0.f, 0.f}}};
        const uint4* const wt[1] = {is_cd ? a.pk_whht + (size_t)u * tile3D : a.pk_wft + (size_t)(u - NTD) * tile3D};
        wide_mfma_stream<1, P, kNS1>(acc, wt, is_cd ? a.x_dgh : a.x_dgi, KS3D, k0, k1, lane);
        wide_red_store<1>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        if (e_valid) {
          const wf32x4 sm = wide_red_sum<1>(red, 0, e_rt, e_slot);
          if (is_cd) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cd[j] = ddz[j] + sm[j];
            if (t == 0) *reinterpret_cast<float4*>(io.g_deter0 + e_b * D + c) = make_float4(cd[0], cd[1], cd[2], cd[3]);
          } else {
            float g[4] = {sm[0] * act_grad_from_out(h1.x, act), sm[1] * act_grad_from_out(h1.y, act), sm[2] * act_grad_from_out(h1.z, act),
                          sm[3] * act_grad_from_out(h1.w, act)};
            *reinterpret_cast<float4*>(io.d_z1 + (e_b * T + t) * H + c) = make_float4(g[0], g[1], g[2], g[3]);
            wide_x_store4<P>(a.x_dz1, KSH, e_row, c, g);
          }
        }
        lds_barrier();
      }
      MTRSSM_WIDE_STAMP(7);
      if (!bar.sync(4 + 5 * t)) return;
      MTRSSM_WIDE_STAMP(8);

      // ============ R4: carry_s = W1s^T dz1, one workgroup per 16 stochastic units ============
      for (int u = blk; u < NTS; u += nblk) {
        int k0, k1;
        krange(KSH, k0, k1);
        wf32x4 acc[1][2] = {{wf32x4{0.f, 0.f, 0.f, 0.f}, wf32x4{0.f, 0.f, 0.f, 0.f}}};
        const uint4* const wt[1] = {a.pk_w1s + (size_t)u * tileH};
        wide_mfma_stream<1, P, kNS1>(acc, wt, a.x_dz1, KSH, k0, k1, lane);
        wide_red_store<1>(red, wave, 0, lane, acc[0]);
        lds_barrier();
        if (e_valid) {
          const wf32x4 sm = wide_red_sum<1>(red, 0, e_rt, e_slot);
          const int s0 = u * 16 + e_cq;
          float* dst = a.cs + (size_t)e_row * Sp + s0;
          wide_store_f2(dst, sm[0], sm[1]);
          wide_store_f2(dst + 2, sm[2], sm[3]);
          if (t == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (s0 + j < S) io.g_stoch0[e_b * S + s0 + j] = sm[j];
          }
        }
        lds_barrier();
      }
      MTRSSM_WIDE_STAMP(9);
      if (!bar.sync(5 + 5 * t)) return;
      MTRSSM_WIDE_STAMP(10);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int debug_set_wide_profile(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wide_prof), &p, sizeof(p)) == hipSuccess ? MTRSSM_OK : MTRSSM_ELAUNCH;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static bool wide_dims_ok(const MtrssmMrssmDims* d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->D <= 0 || d->H <= 0 || d->K <= 0 || d->C <= 0 || !d->post) return false;
  if (d->D % 16 || d->H % 16) return false;                 // whole 16-column MFMA tiles, 16-byte aligned quads
  if (d->K > 64 || d->H > 4 * kAQ * kWT) return false;      // one lane per categorical; kAQ float4 per thread in the row phase
  if (d->act < MTRSSM_ACT_IDENTITY || d->act > MTRSSM_ACT_TANH) return false;
  return true;
}

// 1 when the wide kernels take these dims on the current device: tile state lives in registers of ONE workgroup per column
// tile, so the grid (= CU count, all resident) must cover the largest stateful phase.
int mrssm_wide_supported(const MtrssmMrssmDims* d, int pieces) {
  if (!wide_dims_ok(d) || (pieces != 2 && pieces != 3)) return 0;
  if (d->D < 256 && d->H < 256) return 0;                   // below that the one-CU / four-CU kernels are the faster forms
  const int cus = device_cu_count();
  if (cus < 64) return 0;
  if (d->D / 16 + d->H / 16 > cus) return 0;
  return 1;
}

struct WideFwdLayout {
  size_t wf, whh, wh1, h2[3], x_h1, x_d[2], x_hd[3], lg, total;
};
static WideFwdLayout wide_fwd_layout(const MtrssmMrssmDims* d, int P) {
  const int D = d->D, H = d->H, S = d->K * d->C, Sp = (S + 15) / 16 * 16;
  WideFwdLayout L;
  size_t o = kWideCtl;
  auto take = [&](size_t bytes) { const size_t r = o; o += align256(bytes); return r; };
  L.x_h1 = take(wide_x_uint4(H, P) * 16);
  L.x_d[0] = take(wide_x_uint4(D, P) * 16);
  L.x_d[1] = take(wide_x_uint4(D, P) * 16);
  for (int q = 0; q < 3; ++q) L.x_hd[q] = take(wide_x_uint4(H, P) * 16);
  L.lg = take((size_t)kWRows * 3 * Sp * sizeof(float));
  L.wf = take(wide_pack_uint4(3 * D, H, P) * 16);
  L.whh = take(wide_pack_uint4(3 * D, D, P) * 16);
  L.wh1 = take(wide_pack_uint4(3 * H, D, P) * 16);
  for (int q = 0; q < 3; ++q) L.h2[q] = take(wide_pack_uint4(S, H, P) * 16);
  L.total = o;
  return L;
}
size_t mrssm_wide_workspace_bytes(const MtrssmMrssmDims* d, int pieces) {
  if (!wide_dims_ok(d) || (pieces != 2 && pieces != 3)) return 0;
  return wide_fwd_layout(d, pieces).total;
}

struct WideBwdLayout {
  size_t h2t[3], wh1t, whht, wft, w1s, x_dl[3], x_dzh, x_dgi, x_dgh, x_dz1, cs, total;
};
static WideBwdLayout wide_bwd_layout(const MtrssmMrssmDims* d, int P) {
  const int D = d->D, H = d->H, S = d->K * d->C, Sp = (S + 15) / 16 * 16;
  WideBwdLayout L;
  size_t o = kWideCtl;
  auto take = [&](size_t bytes) { const size_t r = o; o += align256(bytes); return r; };
  for (int q = 0; q < 3; ++q) L.x_dl[q] = take(wide_x_uint4(Sp, P) * 16);
  L.x_dzh = take(wide_x_uint4(3 * H, P) * 16);
  L.x_dgi = take(wide_x_uint4(3 * D, P) * 16);
  L.x_dgh = take(wide_x_uint4(3 * D, P) * 16);
  L.x_dz1 = take(wide_x_uint4(H, P) * 16);
  L.cs = take((size_t)kWRows * Sp * sizeof(float));
  for (int q = 0; q < 3; ++q) L.h2t[q] = take(wide_pack_uint4(H, S, P) * 16);
  L.wh1t = take(wide_pack_uint4(D, 3 * H, P) * 16);
  L.whht = take(wide_pack_uint4(D, 3 * D, P) * 16);
  L.wft = take(wide_pack_uint4(H, 3 * D, P) * 16);
  L.w1s = take(wide_pack_uint4(S, H, P) * 16);
  L.total = o;
  return L;
}
size_t mrssm_wide_bwd_workspace_bytes(const MtrssmMrssmDims* d, int pieces) {
  if (!wide_dims_ok(d) || (pieces != 2 && pieces != 3)) return 0;
  return wide_bwd_layout(d, pieces).total;
}

// MTRSSM_WIDE_ACQUIRE=1: an agent-scope acquire fence after every grid barrier on top of the sc1 loads (A/B runs)
static int wide_acquire_fence() {
  static const int on = [] { const char* e = getenv("MTRSSM_WIDE_ACQUIRE"); return (e && e[0] == '1') ? 1 : 0; }();
  return on;
}

static int wide_grid(const MtrssmMrssmDims* d) {
  (void)d;
  return device_cu_count();   // one workgroup per CU (each asks for more than half a CU's LDS)
}

static size_t wide_fwd_lds(const MtrssmMrssmDims* d) {
  const int Sp = (d->K * d->C + 15) / 16 * 16;
  size_t need = (size_t)kWW * 4 * 2 * kWave * 16 + ((size_t)5 * Sp + 128 + 4 + 64) * sizeof(float);
  return need < 84 * 1024 ? 84 * 1024 : need;    // > 80 KiB: never two workgroups on one CU
}
static size_t wide_bwd_lds(const MtrssmMrssmDims* d) {
  const int Sp = (d->K * d->C + 15) / 16 * 16;
  size_t need = (size_t)kWW * 1 * 2 * kWave * 16 + ((size_t)10 * Sp + 4) * sizeof(float);
  return need < 84 * 1024 ? 84 * 1024 : need;
}

int mrssm_wide_fwd_launch(const MtrssmMrssmDims* d, const MtrssmMrssmClusterWeights* w, const MtrssmMrssmFwdIO* io, int pieces,
                          void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!mrssm_wide_supported(d, pieces)) {
    set_error("mrssm_rollout_fwd_wide: dims / device outside the wide kernel's regime (ask mtrssm_mrssm_wide_supported first)");
    return MTRSSM_EINVAL;
  }
  if (!w || !io || !workspace || !w->w1s_t || !w->wf_t || !w->bf || !w->whh_t || !w->bhh || !w->wh1_t || !w->b3 || !w->w4 || !w->b4 ||
      !w->wa2 || !w->ba2 || !w->wv2 || !w->bv2 || !io->xa || !io->pa || !io->pv || !io->deter0 || !io->stoch0 || !io->u_post ||
      !io->deter || !io->prior_logits || !io->post_logits || !io->post_stoch) {
    set_error("mrssm_rollout_fwd_wide: null required pointer");
    return MTRSSM_EINVAL;
  }
  const WideFwdLayout L = wide_fwd_layout(d, pieces);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) {
    set_error("mrssm_rollout_fwd_wide: workspace too small (%zu < %zu) or not 256-byte aligned", workspace_bytes, L.total);
    return MTRSSM_EINVAL;
  }
  char* ws = static_cast<char*>(workspace);
  const int D = d->D, H = d->H, S = d->K * d->C;
  // control words after the sticky status word + the logit exchange (read before it is first written? no: zero anyway)
  if (int rc = clear_async(ws + 16, kWideCtl - 16, stream)) return rc;
  WidePackJobs jobs;
  jobs.count = 6;
  jobs.j[0] = make_job(w->wf_t, 1, 3 * D, 3 * D, H, reinterpret_cast<uint4*>(ws + L.wf));
  jobs.j[1] = make_job(w->whh_t, 1, 3 * D, 3 * D, D, reinterpret_cast<uint4*>(ws + L.whh));
  jobs.j[2] = make_job(w->wh1_t, 1, 3 * H, 3 * H, D, reinterpret_cast<uint4*>(ws + L.wh1));
  jobs.j[3] = make_job(w->w4, H, 1, S, H, reinterpret_cast<uint4*>(ws + L.h2[0]));
  jobs.j[4] = make_job(w->wa2, H, 1, S, H, reinterpret_cast<uint4*>(ws + L.h2[1]));
  jobs.j[5] = make_job(w->wv2, H, 1, S, H, reinterpret_cast<uint4*>(ws + L.h2[2]));
  if (int rc = launch_pack(jobs, pieces, stream)) return rc;

  WideFwdArgs a;
  a.dm = *d; a.w = *w; a.io = *io;
  a.pk_wf = reinterpret_cast<const uint4*>(ws + L.wf);
  a.pk_whh = reinterpret_cast<const uint4*>(ws + L.whh);
  a.pk_wh1 = reinterpret_cast<const uint4*>(ws + L.wh1);
  for (int q = 0; q < 3; ++q) {
    a.pk_h2[q] = reinterpret_cast<const uint4*>(ws + L.h2[q]);
    a.x_hd[q] = reinterpret_cast<uint4*>(ws + L.x_hd[q]);
  }
  a.x_h1 = reinterpret_cast<uint4*>(ws + L.x_h1);
  a.x_d[0] = reinterpret_cast<uint4*>(ws + L.x_d[0]);
  a.x_d[1] = reinterpret_cast<uint4*>(ws + L.x_d[1]);
  a.lg = reinterpret_cast<float*>(ws + L.lg);
  a.status = reinterpret_cast<int*>(ws);
  a.ctl = ws;
  a.nblk = wide_grid(d);
  a.acquire = wide_acquire_fence();
  const size_t lds = wide_fwd_lds(d);
  if (lds > 160 * 1024) { set_error("mrssm_rollout_fwd_wide: %zu bytes of LDS", lds); return MTRSSM_ELDS; }
  hipError_t e;
#define MTRSSM_WIDE_FWD(PV)                                                                                                          \
  {                                                                                                                                 \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mrssm_wide_fwd_kernel<PV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e)); return MTRSSM_ELAUNCH; }   \
    set_last_kernel("mtrssm::mrssm_wide_fwd_kernel<" #PV ">");                                                                      \
    hipLaunchKernelGGL(mrssm_wide_fwd_kernel<PV>, dim3(a.nblk), dim3(kWT), lds, stream, a);                                        \
  }
  if (pieces == 3) MTRSSM_WIDE_FWD(3) else MTRSSM_WIDE_FWD(2)
#undef MTRSSM_WIDE_FWD
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("wide forward scan launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  return MTRSSM_OK;
}

int mrssm_wide_bwd_launch(const MtrssmMrssmDims* d, const MtrssmMrssmClusterWeights* w, const MtrssmMrssmBwdIO* io, int pieces,
                          void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!mrssm_wide_supported(d, pieces)) {
    set_error("mrssm_rollout_bwd_wide: dims / device outside the wide kernel's regime (ask mtrssm_mrssm_wide_supported first)");
    return MTRSSM_EINVAL;
  }
  if (!w || !io || !workspace || !w->w1s_t || !w->wf_t || !w->whh_t || !w->wh1_t || !w->w4 || !w->wa2 || !w->wv2 || !io->deter0 ||
      !io->deter || !io->prior_logits || !io->post_logits || !io->sv_h1 || !io->sv_gates || !io->sv_heads || !io->sv_la || !io->sv_lv ||
      !io->g_deter0 || !io->g_stoch0 || !io->d_z1 || !io->d_gi || !io->d_gh || !io->d_zh || !io->d_lp || !io->d_la || !io->d_lv) {
    set_error("mrssm_rollout_bwd_wide: null required pointer");
    return MTRSSM_EINVAL;
  }
  const WideBwdLayout L = wide_bwd_layout(d, pieces);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) {
    set_error("mrssm_rollout_bwd_wide: workspace too small (%zu < %zu) or not 256-byte aligned", workspace_bytes, L.total);
    return MTRSSM_EINVAL;
  }
  char* ws = static_cast<char*>(workspace);
  const int D = d->D, H = d->H, S = d->K * d->C;
  if (int rc = clear_async(ws + 16, kWideCtl - 16, stream)) return rc;
  WidePackJobs jobs;
  jobs.count = 7;
  jobs.j[0] = make_job(w->w4, 1, H, H, S, reinterpret_cast<uint4*>(ws + L.h2t[0]));       // (n = head unit j, k = s): w4[s][j]
  jobs.j[1] = make_job(w->wa2, 1, H, H, S, reinterpret_cast<uint4*>(ws + L.h2t[1]));
  jobs.j[2] = make_job(w->wv2, 1, H, H, S, reinterpret_cast<uint4*>(ws + L.h2t[2]));
  jobs.j[3] = make_job(w->wh1_t, 3 * H, 1, D, 3 * H, reinterpret_cast<uint4*>(ws + L.wh1t));  // (n = i, k = j): wh1_t[i][j]
  jobs.j[4] = make_job(w->whh_t, 3 * D, 1, D, 3 * D, reinterpret_cast<uint4*>(ws + L.whht));
  jobs.j[5] = make_job(w->wf_t, 3 * D, 1, H, 3 * D, reinterpret_cast<uint4*>(ws + L.wft));
  jobs.j[6] = make_job(w->w1s_t, H, 1, S, H, reinterpret_cast<uint4*>(ws + L.w1s));          // (n = s, k = j): w1s_t[s][j]
  if (int rc = launch_pack(jobs, pieces, stream)) return rc;

  WideBwdArgs a;
  a.dm = *d; a.io = *io;
  for (int q = 0; q < 3; ++q) {
    a.pk_h2t[q] = reinterpret_cast<const uint4*>(ws + L.h2t[q]);
    a.x_dl[q] = reinterpret_cast<uint4*>(ws + L.x_dl[q]);
  }
  a.pk_wh1t = reinterpret_cast<const uint4*>(ws + L.wh1t);
  a.pk_whht = reinterpret_cast<const uint4*>(ws + L.whht);
  a.pk_wft = reinterpret_cast<const uint4*>(ws + L.wft);
  a.pk_w1s = reinterpret_cast<const uint4*>(ws + L.w1s);
  a.x_dzh = reinterpret_cast<uint4*>(ws + L.x_dzh);
  a.x_dgi = reinterpret_cast<uint4*>(ws + L.x_dgi);
  a.x_dgh = reinterpret_cast<uint4*>(ws + L.x_dgh);
  a.x_dz1 = reinterpret_cast<uint4*>(ws + L.x_dz1);
  a.cs = reinterpret_cast<float*>(ws + L.cs);
  a.status = reinterpret_cast<int*>(ws);
  a.ctl = ws;
  a.nblk = wide_grid(d);
  a.acquire = wide_acquire_fence();
  const size_t lds = wide_bwd_lds(d);
  if (lds > 160 * 1024) { set_error("mrssm_rollout_bwd_wide: %zu bytes of LDS", lds); return MTRSSM_ELDS; }
  hipError_t e;
#define MTRSSM_WIDE_BWD(PV)                                                                                                          \
  {                                                                                                                                 \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mrssm_wide_bwd_kernel<PV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e)); return MTRSSM_ELAUNCH; }   \
    set_last_kernel("mtrssm::mrssm_wide_bwd_kernel<" #PV ">");                                                                      \
    hipLaunchKernelGGL(mrssm_wide_bwd_kernel<PV>, dim3(a.nblk), dim3(kWT), lds, stream, a);                                        \
  }
  if (pieces == 3) MTRSSM_WIDE_BWD(3) else MTRSSM_WIDE_BWD(2)
#undef MTRSSM_WIDE_BWD
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("wide backward scan launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  return MTRSSM_OK;
}

}  // namespace mtrssm
