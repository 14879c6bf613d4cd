// MoPoE-MRSSM forward scan, one batch row on a CLUSTER of four compute units (gfx950).
//
// The single-CU scan (mrssm_scan.hip) streams the step's 1.5 MB of fp32 weights from L2 through one CU's 64 B/clk load
// path every timestep: 31 us per step against 0.4 us of arithmetic.  Here a row's step is split over kClu = 4 workgroups
// (= 4 CUs: each workgroup takes a whole CU's LDS), each of which keeps ITS QUARTER of the weights resident for all T
// steps -- its columns of the fused GRU input matrix and of W_hh in registers, cut into pieces spread evenly over all 256
// threads (240 floats per thread at D = H = 200: a 256-thread workgroup is one wave per SIMD, so each thread may use all
// 512 registers), the head matrix and W1s in LDS (144 KB) -- so nothing is streamed and the step costs its arithmetic plus
// two exchanges:
//
//   every member   h1 = act(xa + W1s s)                                    (redundant: H x S, W1s in LDS)
//   member m       gi | gh for its D/4 deter units, gates, d_new[own]      -> publish D/4 values, gather D      (exchange 1)
//   member m       head layer 0 for its H/4 units of the prior / audio / vision heads, then its share of the 3 S logit
//                  dot products (partial sums over its units)              -> publish 3 S partials, gather 12 S (exchange 2)
//   every member   logits = bias + the four partials in member order, MoPoE mix, KL, sample   (redundant, identical bits)
//
// Exchanges are 8-byte {epoch, value} granules written by ONE sc1 (write-through) store and polled by ONE wave with sc1 loads:
// the data is the flag, no fence, no ordering requirement (cdna_hip_programming.md Guideline 16, form R2); granule words are
// zeroed by a memset node in the launch function, epochs count from 1 inside the launch.  Every spin is bounded: a wave that
// gives up writes a code to the launch's status word and the whole workgroup leaves (the host reads the word after the
// launch; results are then invalid).  The grid is 4 x min(B, 64) workgroups with ~150 KB of LDS each: one per CU, all
// resident on the 256 CUs (a cluster loops over rows c, c + clusters, ...), which the spins rely on.
//
// The GRU's input path is fused: gi = W_ih (W2 h1 + b2) + b_ih = (W_ih W2) h1 + (W_ih b2 + b_ih); the caller hands in the
// product (one small GEMM per launch).  Rounding differs from the two-step form by ~1e-7 relative; h2 itself (needed by
// dW_ih) is recomputed after the scan as one batched GEMM.
#include "scan_common.h"

namespace mtrssm {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);
int device_cu_count();

constexpr int kClu = 4;            // workgroups (CUs) per row
constexpr int kCluThreads = 256;   // one wave per SIMD: 512 registers per thread
constexpr unsigned kSpinLimit = 1u << 22;

// Development aid (tools/cluster_probe.py): when set, workgroup 0 stamps s_memtime at the phase boundaries of timesteps
// 8..11 of its first row into this buffer (16 stamps per step).  Null in normal operation.
__device__ unsigned long long* g_cluster_prof = nullptr;
#define MTRSSM_CLU_STAMP(i)                                                                                  \
  do {                                                                                                       \
    if (prof && t >= 8 && t < 12) prof[(t - 8) * 16 + (i)] = __builtin_readcyclecounter();                   \
  } while (0)

__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned epoch, float v) {
  __hip_atomic_store(p, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// One wave gathers `total` CONSECUTIVE granules g[0 .. total) into LDS dst[0 .. total); returns false on time-out.
// (The exchange arrays are laid out so that what a member needs -- all four members' slots, its own included -- is one
// contiguous run in destination order: no index arithmetic in the poll loop.)
template <int NJ>
__device__ __forceinline__ bool wave_gather(const unsigned long long* g, float* dst, int total, unsigned epoch, int lane) {
  unsigned done = 0;
  const unsigned full = (1u << NJ) - 1u;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (lane + 64 * j >= total) done |= 1u << j;
  for (unsigned spins = 0;; ++spins) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!(done & (1u << j))) {
        const unsigned long long x = __hip_atomic_load(g + lane + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == epoch) {
          dst[lane + 64 * j] = __uint_as_float((unsigned)x);
          done |= 1u << j;
        }
      }
    }
    if (__all(done == full)) return true;
    if (spins > kSpinLimit) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// DH: D = H (compile time: every stride an immediate).  The 2 * 3 * DH / 4 gate columns are cut into NP pieces of KP = DH / NP
// reduction terms; thread t holds pieces t * PPT .. t * PPT + PPT - 1 (column = piece / NP, part = piece % NP).
template <int DH, int NP, int PPT>
__global__ __launch_bounds__(kCluThreads) void mrssm_fwd_cluster_kernel(const MtrssmMrssmDims dm, const MtrssmMrssmClusterWeights w,
                                                                        const MtrssmMrssmFwdIO io, unsigned long long* __restrict__ gran,
                                                                        int* __restrict__ status, int nclusters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int D = DH, H = DH;
  constexpr int UD = D / kClu, UH = H / kClu;
  constexpr int NG = 3 * UD, NH = 3 * UH;  // own gate outputs / own head units
  constexpr int KP = DH / NP;              // reduction terms per piece
  constexpr int NPIECE = 2 * NG * NP;      // gi columns first, then gh columns
  static_assert(KP % 4 == 0 && DH % NP == 0, "pieces are whole float4 runs");
  static_assert(PPT * kCluThreads >= NPIECE, "every piece has a thread");
  const int K = dm.K, C = dm.C, S = K * C, T = dm.T, act = dm.act;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  // blocks b, b + 8, b + 16, b + 24 share an XCD under round-robin placement (speed only)
  const int blk = blockIdx.x;
  const int member = (blk >> 3) & (kClu - 1);
  const int cluster = (blk & 7) + 8 * (blk >> 5);
  if (cluster >= nclusters) return;  // whole clusters drop out together

  // ---- LDS carve-up (floats)
  int o = 0;
  auto take = [&](int n) { const int r = o; o += (n + 3) & ~3; return r; };
  const int Ls = take(S), Lh1 = take(DH), Ld0 = take(DH), Ld1 = take(DH), Lhd = take(3 * 64);  // head units [which][64], zero padded
  const int Llp = take(S), Lla = take(S), Llv = take(S), Lmx = take(S);
  const int Lpart = take(kClu * 3 * S);   // logit partial sums of the four members
  const int Lred = take(NPIECE > NH * NP ? NPIECE : NH * NP);   // per-piece partial sums
  const int Lw1 = take(S * H);            // W1s^T [S][H]
  const int Lwh = take(D * NH);           // [k][o]: this member's columns of the head layer 0
  const int Lbl = take(3 * S);            // biases of the 3 S logits
  const int Lu = take(2 * 64);            // this step's uniforms: posterior [K], prior [K]
  const int Lflag = take(4);
  (void)o;
  int* abort_flag = reinterpret_cast<int*>(lds + Lflag);

  // ---- resident weights: PPT pieces of KP floats per thread
  float wreg[PPT][KP];
  int xsel[PPT];   // per piece: 0 = multiplies h1 (gi column), 1 = multiplies d_prev (gh column); part offset in the high bits
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int pc = tid * PPT + i;
    const int pcc = pc < NPIECE ? pc : NPIECE - 1;          // threads past the last piece hold a copy (their sums are unused)
    const int col = pcc / NP, part = pcc - col * NP;
    const bool is_h = col >= NG;
    const int og = is_h ? col - NG : col;                   // own gate output og = g * UD + u
    const int gcol = (og / UD) * D + member * UD + (og % UD);   // <-> column g * D + member * UD + u of the [.][3D] matrices
    const float* src = (is_h ? w.whh_t : w.wf_t) + (size_t)(part * KP) * 3 * D + gcol;
    xsel[i] = (is_h ? 1 : 0) | (part * KP) << 1;
#pragma unroll
    for (int k = 0; k < KP; ++k) wreg[i][k] = src[(size_t)k * 3 * D];
  }
  // the logit rows (second-layer slices, UH <= 64 terms): thread r < 3 S holds row r
  float wrow[64];
  {
    const bool roleL = tid < 3 * S;
    const int which = roleL ? tid / S : 0, srow = roleL ? tid - which * S : 0;
    const float* rowsrc = (which == 0 ? w.w4 : (which == 1 ? w.wa2 : w.wv2)) + (size_t)srow * H + member * UH;
#pragma unroll
    for (int k = 0; k < 64; ++k) wrow[k] = k < UH ? rowsrc[k] : 0.f;
  }
  for (int idx = tid; idx < S * H; idx += kCluThreads) lds[Lw1 + idx] = w.w1s_t[idx];
  // head layer 0: own head unit oh = which * UH + u  <->  column which * H + member * UH + u of wh1_t [D][3H]
  for (int idx = tid; idx < D * NH; idx += kCluThreads) {
    const int k = idx / NH, oh = idx - k * NH;
    lds[Lwh + idx] = w.wh1_t[(size_t)k * 3 * H + (oh / UH) * H + member * UH + (oh % UH)];
  }
  for (int i = tid; i < 3 * 64; i += kCluThreads) lds[Lhd + i] = 0.f;
  for (int i = tid; i < 3 * S; i += kCluThreads) {
    const int which = i / S, s2 = i - which * S;
    lds[Lbl + i] = (which == 0 ? w.b4 : (which == 1 ? w.ba2 : w.bv2))[s2];
  }
  // per-thread constants of the finishing threads
  float bias_g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // thread u < UD: bf / bhh of its three gates
  if (tid < UD) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      bias_g[g] = w.bf[g * D + member * UD + tid];
      bias_g[3 + g] = w.bhh[g * D + member * UD + tid];
    }
  }
  const float bias_h = (tid < UH) ? w.b3[member * UH + tid] : 0.f;  // prior head units only (audio / vision take pa / pv)
  if (tid == 0) *abort_flag = 0;

  // granule slots of this cluster: [parity][ D-exchange: member x UD | logit exchange: member x 3S ]
  const int slotsD = UD, slotsL = 3 * S;
  const int per_parity = kClu * (slotsD + slotsL);
  unsigned long long* gbase = gran + (size_t)cluster * 2 * per_parity;
  unsigned epoch = 0;
  unsigned long long* prof = (blockIdx.x == 0 && tid == 0) ? g_cluster_prof : nullptr;

  for (int row = cluster; row < dm.B; row += nclusters) {
    int cur = Ld0, nxt = Ld1;
    lds_barrier();
    for (int i = tid; i < D; i += kCluThreads) lds[cur + i] = io.deter0[(size_t)row * D + i];
    for (int i = tid; i < S; i += kCluThreads) lds[Ls + i] = io.stoch0[(size_t)row * S + i];
    lds_barrier();

    // The streamed inputs of a step (xa, pa / pv of the own head units, the uniforms) are loaded one step AHEAD into registers:
    // loaded at the top of the step that consumes them, their HBM latency (~1 us) sat on the step's critical path.
    auto load_inputs = [&](size_t q, float& xa_r, float& pa_r, float& up_r, float& ur_r) {
      xa_r = tid < H ? io.xa[q * H + tid] : 0.f;
      pa_r = 0.f;
      if (tid >= UH && tid < NH) {
        const int which = tid / UH, u = tid - which * UH;
        pa_r = (which == 1 ? io.pa : io.pv)[q * H + member * UH + u];
      }
      up_r = (wave == 3 && lane < K) ? io.u_post[q * K + lane] : 0.f;
      ur_r = (wave == 3 && lane < K && io.u_prior) ? io.u_prior[q * K + lane] : 0.f;
    };
    float xa_n, pav_n, up_n, ur_n;
    load_inputs((size_t)row * T, xa_n, pav_n, up_n, ur_n);

    for (int t = 0; t < T; ++t) {
      const size_t bt = (size_t)row * T + t;
      unsigned long long* gpar = gbase + (size_t)(t & 1) * per_parity;
      const float xa_v = xa_n, pav = pav_n;
      if (wave == 3 && lane < K) { lds[Lu + lane] = up_n; lds[Lu + 64 + lane] = ur_n; }
      if (t + 1 < T) load_inputs(bt + 1, xa_n, pav_n, up_n, ur_n);

      MTRSSM_CLU_STAMP(0);
      // (1) h1 = act(xa + W1s s): every member, all H outputs                    networks.py:165-166
      if (tid < H) {
        // S <= 42 terms in four independent chains, reads batched eight deep: a rolled, one-chain loop pays the LDS latency
        // (~100 cycles) per term on the step's critical path
        const float* wc = lds + Lw1 + tid;
        float a0 = xa_v, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = 0;
        for (; k + 8 <= S; k += 8) {
          float wv[8], sv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) { wv[i] = wc[(size_t)(k + i) * H]; sv[i] = lds[Ls + k + i]; }
          a0 = fmaf(wv[0], sv[0], a0); a1 = fmaf(wv[1], sv[1], a1); a2 = fmaf(wv[2], sv[2], a2); a3 = fmaf(wv[3], sv[3], a3);
          a0 = fmaf(wv[4], sv[4], a0); a1 = fmaf(wv[5], sv[5], a1); a2 = fmaf(wv[6], sv[6], a2); a3 = fmaf(wv[7], sv[7], a3);
        }
        for (; k < S; ++k) a0 = fmaf(wc[(size_t)k * H], lds[Ls + k], a0);
        const float h = act_fwd((a0 + a1) + (a2 + a3), act);
        lds[Lh1 + tid] = h;
        if (member == 0 && io.sv_h1) io.sv_h1[bt * H + tid] = h;
      }
      lds_barrier();

      MTRSSM_CLU_STAMP(1);
      // (2) partial dot products of this thread's pieces of gi (fused W_ih W2, times h1) and gh (W_hh, times d_prev)
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const float* xv = lds + ((xsel[i] & 1) ? cur : Lh1) + (xsel[i] >> 1);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < KP; k += 4) {
          const float4 x = *reinterpret_cast<const float4*>(xv + k);
          a0 = fmaf(wreg[i][k], x.x, a0);
          a1 = fmaf(wreg[i][k + 1], x.y, a1);
          a2 = fmaf(wreg[i][k + 2], x.z, a2);
          a3 = fmaf(wreg[i][k + 3], x.w, a3);
        }
        if (tid * PPT + i < NPIECE) lds[Lred + tid * PPT + i] = (a0 + a1) + (a2 + a3);
      }
      lds_barrier();

      MTRSSM_CLU_STAMP(2);
      // (3) gates of the own deter units; publish them                          networks.py:170 (nn.GRUCell)
      ++epoch;
      if (tid < UD) {
        float gs[6];
#pragma unroll
        for (int g = 0; g < 6; ++g) {  // g < 3: gi of gate g (column g * UD + tid); else gh (column NG + ...)
          const float* pr = lds + Lred + ((g < 3 ? g : g - 3) * UD + tid + (g < 3 ? 0 : NG)) * NP;
          float a = bias_g[g];
#pragma unroll
          for (int q = 0; q < NP; ++q) a += pr[q];
          gs[g] = a;
        }
        // hardware exp (v_exp_f32, ~1e-6 relative) on the step's critical path: sigmoid = 1 / (1 + e^-x), tanh = 1 - 2 / (e^2x + 1)
        const float rg = __fdividef(1.f, 1.f + __expf(-(gs[3] + gs[0])));
        const float zg = __fdividef(1.f, 1.f + __expf(-(gs[4] + gs[1])));
        const float ng = 1.f - __fdividef(2.f, __expf(2.f * (gs[2] + gs[5] * rg)) + 1.f);
        const int unit = member * UD + tid;
        const float dnew = (lds[cur + unit] - ng) * zg + ng;
        lds[nxt + unit] = dnew;
        granule_store(gpar + (size_t)member * slotsD + tid, epoch, dnew);
        io.deter[bt * D + unit] = dnew;
        if (io.sv_gates) {
          float* gsv = io.sv_gates + bt * 4 * D;
          gsv[unit] = rg; gsv[D + unit] = zg; gsv[2 * D + unit] = ng; gsv[3 * D + unit] = gs[5];
        }
      }
      if (wave == 3) {  // exchange 1: all D deter units, in unit order (the own ones come back unchanged)
        const bool ok = wave_gather<4>(gpar, lds + nxt, D, epoch, lane);
        if (!ok && lane == 0) { *abort_flag = 1; atomicExch(status, 1 + 2 * t); }
      }
      lds_barrier();
      if (*abort_flag) return;

      MTRSSM_CLU_STAMP(3);
      // (4) head layer 0 for the own units of the three heads (matrix slice resident in LDS), cut into the same NP parts
      for (int pc = tid; pc < NH * NP; pc += kCluThreads) {
        const int oh = pc / NP, part = pc - oh * NP;
        const float* wcol = lds + Lwh + (size_t)(part * KP) * NH + oh;
        const float* dv = lds + nxt + part * KP;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int k = 0; k < KP; k += 8) {  // eight matrix reads in flight, four chains
          float wv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) wv[i] = k + i < KP ? wcol[(size_t)(k + i) * NH] : 0.f;
          const float4 d0 = *reinterpret_cast<const float4*>(dv + k);
          const float4 d1 = k + 4 < KP ? *reinterpret_cast<const float4*>(dv + k + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
          p0 = fmaf(wv[0], d0.x, p0); p1 = fmaf(wv[1], d0.y, p1); p2 = fmaf(wv[2], d0.z, p2); p3 = fmaf(wv[3], d0.w, p3);
          p0 = fmaf(wv[4], d1.x, p0); p1 = fmaf(wv[5], d1.y, p1); p2 = fmaf(wv[6], d1.z, p2); p3 = fmaf(wv[7], d1.w, p3);
        }
        lds[Lred + pc] = (p0 + p1) + (p2 + p3);
      }
      lds_barrier();
      if (tid < NH) {
        float z = tid < UH ? bias_h : pav;
#pragma unroll
        for (int q = 0; q < NP; ++q) z += lds[Lred + tid * NP + q];
        const float h = act_fwd(z, act);
        const int which = tid / UH, u = tid - which * UH;
        lds[Lhd + which * 64 + u] = h;
        if (io.sv_heads) io.sv_heads[bt * 3 * H + which * H + member * UH + u] = h;
      }
      lds_barrier();

      MTRSSM_CLU_STAMP(4);
      // (5) this member's share of the 3 S logit dot products (rows resident in registers); publish; gather the others
      ++epoch;
      unsigned long long* gl = gpar + (size_t)kClu * slotsD;
      if (tid < 3 * S) {
        const float* hv = lds + Lhd + (tid / S) * 64;
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int u = 0; u < 64; u += 4) {  // UH <= 64 (mrssm_cluster_supported); hv is zero beyond UH
          const float4 x = *reinterpret_cast<const float4*>(hv + u);
          p0 = fmaf(wrow[u], x.x, p0);
          p1 = fmaf(wrow[u + 1], x.y, p1);
          p0 = fmaf(wrow[u + 2], x.z, p0);
          p1 = fmaf(wrow[u + 3], x.w, p1);
        }
        const float p = p0 + p1;
        lds[Lpart + member * 3 * S + tid] = p;
        granule_store(gl + (size_t)member * slotsL + tid, epoch, p);
      }
      if (wave == 3) {  // exchange 2: the four members' 3 S partial sums, [member][3S]
        const bool ok = wave_gather<6>(gl, lds + Lpart, kClu * 3 * S, epoch, lane);
        if (!ok && lane == 0) { *abort_flag = 1; atomicExch(status, 2 + 2 * t); }
      }
      lds_barrier();
      if (*abort_flag) return;

      MTRSSM_CLU_STAMP(5);
      // (6) logits = bias + partials in member order (the same bits on every member), then fusion, per-categorical softmax,
      //     KL, sampling: one wave (every member; member 0 writes the outputs)
      if (wave == 3) {
        const bool writer = member == 0;
        for (int i = lane; i < 3 * S; i += kWave) {
          const int which = i / S, s2 = i - which * S;
          float v = lds[Lbl + i];
#pragma unroll
          for (int m2 = 0; m2 < kClu; ++m2) v += lds[Lpart + m2 * 3 * S + i];
          lds[(which == 0 ? Llp : (which == 1 ? Lla : Llv)) + s2] = v;
        }
        wave_mopoe_mix<true>(lds + Lla, lds + Llv, lds + Lmx, S, lane);
        for (int s = lane; s < S; s += kWave) {
          if (writer) {
            io.prior_logits[bt * S + s] = lds[Llp + s];
            io.post_logits[bt * S + s] = lds[Lmx + s];
            if (io.sv_la) { io.sv_la[bt * S + s] = lds[Lla + s]; io.sv_lv[bt * S + s] = lds[Llv + s]; }
          }
        }
        float kl = C <= 8 ? cat_block_fwd_fast8(lds + Lmx, lds + Llp, K, C, lane, lds + Lu, io.u_prior ? lds + Lu + 64 : nullptr, lds + Ls,
                                                io.post_stoch + bt * S, io.prior_stoch ? io.prior_stoch + bt * S : nullptr, writer)
                          : cat_block_fwd<true, true>(lds + Lmx, lds + Llp, K, C, lane, lds + Lu, io.u_prior ? lds + Lu + 64 : nullptr, lds + Ls,
                                                      io.post_stoch + bt * S, io.prior_stoch ? io.prior_stoch + bt * S : nullptr, writer);
        if (io.kl) {
          kl = wave_sum(kl);
          if (lane == 0 && writer) io.kl[bt] = kl;
        }
      }
      lds_barrier();
      MTRSSM_CLU_STAMP(7);
      const int tmp = cur; cur = nxt; nxt = tmp;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward (reverse-time scan), same cluster layout.  Member m owns deter units [m UD, (m+1) UD) and UH units of each head.
//   every member   categorical block, MoPoE mix, flat log-softmaxes backward  -> dlp, dla, dlv          (redundant)
//   member m       dzh[own units] = act'(hd) * W2nd^T dl                        (columns of the second layers: registers)
//   member m       partial dd[i] = sum_{own j} Wh1[j][i] dzh[j] for ALL i       (head matrix slice: LDS)   -> exchange 1:
//                  publish D partials, gather the partials of the own units (4 x UD)
//   member m       gates backward for the own units -> dgi, dgh (own 3 UD each)
//   member m       partial carry_d[i] = sum_{own j} W_hh[j][i] dgh[j],  partial dh1[k] = sum_{own j} (W_ih W2)[j][k] dgi[j]
//                  for ALL i, k (rows of the own columns: registers, pieces over all threads)               -> exchange 2:
//                  publish 2 D partials, gather carry partials of the own units (4 x UD) and all dh1 partials (4 x D)
//   every member   dz1 = act'(h1) * dh1,  carry_s = W1s dz1                                             (redundant)
// d_h2 is NOT written (the fused input path has no h2 inside the scan): the caller forms it as d_gi . W_ih (one GEMM).
// ------------------------------------------------------------------------------------------------
template <int NJ, typename Map>
__device__ __forceinline__ bool wave_gather_map(int total, unsigned epoch, int lane, Map map) {  // map(i, &slot, &dst)
  unsigned done = 0;
  const unsigned full = (1u << NJ) - 1u;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (lane + 64 * j >= total) done |= 1u << j;
  for (unsigned spins = 0;; ++spins) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!(done & (1u << j))) {
        const unsigned long long* slot;
        float* dst;
        map(lane + 64 * j, slot, dst);
        const unsigned long long x = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == epoch) {
          *dst = __uint_as_float((unsigned)x);
          done |= 1u << j;
        }
      }
    }
    if (__all(done == full)) return true;
    if (spins > kSpinLimit) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// NPB parts of KPB = 3 UD / NPB own gate columns per row piece; thread t holds row pieces t * PPT .. (row = piece / NPB of
// the stacked [W_hh^T rows ; (W_ih W2)^T rows] (2 DH rows), part = piece % NPB).
template <int DH, int NPB, int PPT>
__global__ __launch_bounds__(kCluThreads) void mrssm_bwd_cluster_kernel(const MtrssmMrssmDims dm, const MtrssmMrssmClusterWeights w,
                                                                        const MtrssmMrssmBwdIO io, unsigned long long* __restrict__ gran,
                                                                        int* __restrict__ status, int nclusters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int D = DH, H = DH;
  constexpr int UD = D / kClu, UH = H / kClu;
  constexpr int NG = 3 * UD, NH = 3 * UH;
  constexpr int NHP = (NH + 3) & ~3;       // padded row of the head-matrix slice (16-byte aligned rows)
  constexpr int KPB = NG / NPB;            // own gate columns per row piece
  constexpr int NPIECE = 2 * DH * NPB;
  constexpr int W1P = 8;                   // parts of the carry_s reduction over H
  constexpr int KW1 = (DH + W1P - 1) / W1P;
  static_assert(NG % NPB == 0 && KPB % 2 == 0, "row pieces are whole float2 runs");
  static_assert(PPT * kCluThreads >= NPIECE, "every row piece has a thread");
  const int K = dm.K, C = dm.C, S = K * C, T = dm.T, act = dm.act;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int blk = blockIdx.x;
  const int member = (blk >> 3) & (kClu - 1);
  const int cluster = (blk & 7) + 8 * (blk >> 5);
  if (cluster >= nclusters) return;

  int o = 0;
  auto take = [&](int n) { const int r = o; o += (n + 3) & ~3; return r; };
  const int Lla = take(S), Llv = take(S), Lmx = take(S), Llp = take(S), Ldmx = take(S), Ldlp = take(S), Ldla = take(S), Ldlv = take(S);
  const int Lgps = take(S), Lcs = take(S);                   // g_post_stoch row, carry_s
  const int Lhd = take(NHP), Ldzh = take(NHP);               // own head units: saved act, pre-activation gradient
  const int Lgate = take(4 * UD), Ldprev = take(UD), Lgd = take(UD), Lcd = take(UD);   // own deter units
  const int Ldg = take(2 * NG);                              // [dgi own | dgh own]
  const int Lh1 = take(DH), Ldz1 = take(DH);
  const int Lpdd = take(kClu * UD);                          // gathered dd partials of the own units
  const int Lpcd = take(kClu * UD), Lpdh = take(kClu * DH);  // gathered carry / dh1 partials
  const int Lred = take(NPIECE > DH ? NPIECE : DH);
  const int Lred2 = take(W1P * 64 > 3 * 64 ? W1P * 64 : 3 * 64);
  const int Lwh = take(D * NHP);                             // [i][own head unit]: this member's columns of the head layer 0
  const int Lw2 = take(32 * NHP);                            // [s][own head unit]: this member's columns of the second layers
  const int Lflag = take(4);
  (void)o;
  int* abort_flag = reinterpret_cast<int*>(lds + Lflag);
  float* gk_lds = lds + Lflag + 1;

  // ---- resident weights
  // (a) row pieces of the own gate columns: rows of W_hh^T (carry_d) then rows of (W_ih W2)^T (dh1)
  float wreg[PPT][KPB];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int pc = tid * PPT + i;
    const int pcc = pc < NPIECE ? pc : NPIECE - 1;
    const int rowi = pcc / NPB, part = pcc - rowi * NPB;
    const bool is_f = rowi >= DH;                                   // second half: rows of wf_t
    const float* src = (is_f ? w.wf_t : w.whh_t) + (size_t)(is_f ? rowi - DH : rowi) * 3 * D;
#pragma unroll
    for (int k = 0; k < KPB; ++k) {
      const int og = part * KPB + k;                                // own gate output g * UD + u
      wreg[i][k] = src[(og / UD) * D + member * UD + (og % UD)];
    }
  }
  // (b) the second layers' columns of the own head units, [s][unit] in LDS (thread j reads column j: conflict-free)
  for (int idx = tid; idx < 32 * NHP; idx += kCluThreads) {
    const int s2 = idx / NHP, j = idx - s2 * NHP;
    float v = 0.f;
    if (s2 < S && j < NH) {
      const int which = j / UH, u = j - which * UH;
      v = (which == 0 ? w.w4 : (which == 1 ? w.wa2 : w.wv2))[(size_t)s2 * H + member * UH + u];
    }
    lds[Lw2 + idx] = v;
  }
  // (c) thread (s, part) = (tid / W1P, tid % W1P), s < S: KW1 values of row s of W1s^T
  float w1r[KW1];
  {
    const int srow = tid / W1P < S ? tid / W1P : 0, part = tid % W1P;
#pragma unroll
    for (int k = 0; k < KW1; ++k) {
      const int kk = part * KW1 + k;
      w1r[k] = w.w1s_t[(size_t)srow * H + (kk < H ? kk : H - 1)];
    }
  }
  for (int idx = tid; idx < D * NHP; idx += kCluThreads) {
    const int k = idx / NHP, oh = idx - k * NHP;
    lds[Lwh + idx] = oh < NH ? w.wh1_t[(size_t)k * 3 * H + (oh / UH) * H + member * UH + (oh % UH)] : 0.f;
  }
  for (int i = tid; i < NHP; i += kCluThreads) { lds[Lhd + i] = 0.f; lds[Ldzh + i] = 0.f; }
  if (tid == 0) *abort_flag = 0;

  // granule slots of this cluster: [parity][ exchange 1: member x D | exchange 2: member x 2D ]
  const int per_parity = kClu * 3 * D;
  unsigned long long* gbase = gran + (size_t)cluster * 2 * per_parity;
  unsigned epoch = 0;

  for (int row = cluster; row < dm.B; row += nclusters) {
    lds_barrier();
    for (int i = tid; i < UD; i += kCluThreads) lds[Lcd + i] = 0.f;
    for (int i = tid; i < S; i += kCluThreads) lds[Lcs + i] = 0.f;
    lds_barrier();

    // staged values of one step, five registers per thread: r0 logits (4 S), r1 head units (NH) | g_post_stoch (S, threads
    // NH..), r2 gates (4 UD) | g_kl (thread 4 UD), r3 h1 (H), r4 d_prev (UD) | g_deter (UD, threads 64..)
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f;
    auto stage_load = [&](int tt) {
      const size_t q = (size_t)row * T + tt;
      if (tid < 4 * S) {
        const int which = tid / S, s2 = tid - which * S;
        const float* src = which == 0 ? io.sv_la : (which == 1 ? io.sv_lv : (which == 2 ? io.post_logits : io.prior_logits));
        r0 = src[q * S + s2];
      }
      if (tid < NH) {
        const int which = tid / UH, u = tid - which * UH;
        r1 = io.sv_heads[q * 3 * H + which * H + member * UH + u];
      } else if (tid < NH + S) {
        r1 = io.g_post_stoch ? io.g_post_stoch[q * S + (tid - NH)] : 0.f;
      }
      if (tid < 4 * UD) {
        const int g = tid / UD, u = tid - g * UD;
        r2 = io.sv_gates[q * 4 * D + g * D + member * UD + u];
      } else if (tid == 4 * UD) {
        r2 = io.g_kl ? io.g_kl[q] : 0.f;
      }
      if (tid < H) r3 = io.sv_h1[q * H + tid];
      if (tid < UD) {
        const int unit = member * UD + tid;
        r4 = tt > 0 ? io.deter[(q - 1) * D + unit] : io.deter0[(size_t)row * D + unit];
      } else if (tid >= 64 && tid < 64 + UD) {
        r4 = io.g_deter ? io.g_deter[q * D + member * UD + (tid - 64)] : 0.f;
      }
    };
    auto stage_store = [&]() {
      if (tid < 4 * S) {
        const int which = tid / S, s2 = tid - which * S;
        lds[(which == 0 ? Lla : (which == 1 ? Llv : (which == 2 ? Lmx : Llp))) + s2] = r0;
      }
      if (tid < NH) lds[Lhd + tid] = r1;
      else if (tid < NH + S) lds[Lgps + tid - NH] = r1;
      if (tid < 4 * UD) lds[Lgate + tid] = r2;
      else if (tid == 4 * UD) *gk_lds = r2;
      if (tid < H) lds[Lh1 + tid] = r3;
      if (tid < UD) lds[Ldprev + tid] = r4;
      else if (tid >= 64 && tid < 64 + UD) lds[Lgd + tid - 64] = r4;
    };
    stage_load(T - 1);

    for (int t = T - 1; t >= 0; --t) {
      const size_t bt = (size_t)row * T + t;
      unsigned long long* gpar = gbase + (size_t)(t & 1) * per_parity;

      // (a) this step's saved vectors and incoming gradients (own parts) were loaded one step ahead into registers
      //     (stage_load below): to LDS now, then the loads of step t - 1 go out and fly during this whole step
      stage_store();
      if (t > 0) stage_load(t - 1);
      lds_barrier();

      // (b) categorical block: straight-through sample, KL, per-categorical softmax, MoE/PoE, flat log-softmax (one wave)
      if (wave == 3) {
        if (C <= 8)
          cat_block_bwd_fast8(lds + Lmx, lds + Llp, K, C, lane, lds + Lgps, lds + Lcs, io.g_prior_stoch ? io.g_prior_stoch + bt * S : nullptr,
                              io.g_post_logits ? io.g_post_logits + bt * S : nullptr, io.g_prior_logits ? io.g_prior_logits + bt * S : nullptr,
                              *gk_lds, dm.kl_w_post, dm.kl_w_prior, lds + Ldmx, lds + Ldlp);
        else
          cat_block_bwd<true>(lds + Lmx, lds + Llp, K, C, lane, lds + Lgps, lds + Lcs, io.g_prior_stoch ? io.g_prior_stoch + bt * S : nullptr,
                              io.g_post_logits ? io.g_post_logits + bt * S : nullptr, io.g_prior_logits ? io.g_prior_logits + bt * S : nullptr,
                              *gk_lds, dm.kl_w_post, dm.kl_w_prior, lds + Ldmx, lds + Ldlp);
        wave_mopoe_mix_bwd<true>(lds + Lla, lds + Llv, lds + Lmx, lds + Ldmx, lds + Ldla, lds + Ldlv, S, lane);
        if (member == 0) {
          for (int s2 = lane; s2 < S; s2 += kWave) {
            io.d_la[bt * S + s2] = lds[Ldla + s2];
            io.d_lv[bt * S + s2] = lds[Ldlv + s2];
            io.d_lp[bt * S + s2] = lds[Ldlp + s2];
          }
        }
      }
      lds_barrier();

      // (c) head layer 1 transposed, own units: dzh[j] = act'(hd[j]) * sum_s W[s][j] dl[s]
      if (tid < NH) {
        const int which = tid / UH;
        const float* dl = lds + (which == 0 ? Ldlp : (which == 1 ? Ldla : Ldlv));
        const float* wc2 = lds + Lw2 + tid;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 32; s2 += 2) {  // rows s >= S of the LDS image are zero
          a0 = fmaf(wc2[s2 * NHP], s2 < S ? dl[s2] : 0.f, a0);
          a1 = fmaf(wc2[(s2 + 1) * NHP], s2 + 1 < S ? dl[s2 + 1] : 0.f, a1);
        }
        const float g = (a0 + a1) * act_grad_from_out(lds[Lhd + tid], act);
        lds[Ldzh + tid] = g;
        const int u = tid - which * UH;
        io.d_zh[bt * 3 * H + which * H + member * UH + u] = g;
      }
      lds_barrier();

      // (d) partial dd[i] over the own head units, for ALL i; publish (exchange 1)
      ++epoch;
      if (tid < D) {
        const float* wr = lds + Lwh + (size_t)tid * NHP;
        const float* dz = lds + Ldzh;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int j = 0; j < NHP; j += 4) {
          const float4 a = *reinterpret_cast<const float4*>(wr + j);
          const float4 b = *reinterpret_cast<const float4*>(dz + j);
          p0 = fmaf(a.x, b.x, p0); p1 = fmaf(a.y, b.y, p1); p2 = fmaf(a.z, b.z, p2); p3 = fmaf(a.w, b.w, p3);
        }
        granule_store(gpar + (size_t)member * D + tid, epoch, (p0 + p1) + (p2 + p3));
      }
      if (wave == 3) {  // the four members' partials for the own units
        const bool ok = wave_gather_map<4>(kClu * UD, epoch, lane, [&](int i, const unsigned long long*& slot, float*& dst) {
          const int m2 = i / UD, u = i - m2 * UD;
          slot = gpar + (size_t)m2 * D + member * UD + u;
          dst = lds + Lpdd + i;
        });
        if (!ok && lane == 0) { *abort_flag = 1; atomicExch(status, 1 + 2 * t); }
      }
      lds_barrier();
      if (*abort_flag) return;

      // (e) GRU gate gradients of the own units
      if (tid < UD) {
        float dd = lds[Lgd + tid] + lds[Lcd + tid];
#pragma unroll
        for (int m2 = 0; m2 < kClu; ++m2) dd += lds[Lpdd + m2 * UD + tid];
        const float rg = lds[Lgate + tid], zg = lds[Lgate + UD + tid], ng = lds[Lgate + 2 * UD + tid], ghn = lds[Lgate + 3 * UD + tid];
        const float dprev = lds[Ldprev + tid];
        const float dn = dd * (1.f - zg);
        const float dz = dd * (dprev - ng);
        const float dn_pre = dn * (1.f - ng * ng);
        const float dr = dn_pre * ghn;
        const float dr_pre = dr * rg * (1.f - rg);
        const float dz_pre = dz * zg * (1.f - zg);
        lds[Lcd + tid] = dd * zg;  // direct path into d_prev; the W_hh path is added after exchange 2
        lds[Ldg + tid] = dr_pre; lds[Ldg + UD + tid] = dz_pre; lds[Ldg + 2 * UD + tid] = dn_pre;
        lds[Ldg + NG + tid] = dr_pre; lds[Ldg + NG + UD + tid] = dz_pre; lds[Ldg + NG + 2 * UD + tid] = dn_pre * rg;
        const int unit = member * UD + tid;
        float* gi = io.d_gi + bt * 3 * D;
        float* gh = io.d_gh + bt * 3 * D;
        gi[unit] = dr_pre; gi[D + unit] = dz_pre; gi[2 * D + unit] = dn_pre;
        gh[unit] = dr_pre; gh[D + unit] = dz_pre; gh[2 * D + unit] = dn_pre * rg;
      }
      lds_barrier();

      // (f) row pieces: partial carry_d[i] (W_hh^T rows times dgh) and partial dh1[k] ((W_ih W2)^T rows times dgi)
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int pc = tid * PPT + i;
        const int pcc = pc < NPIECE ? pc : NPIECE - 1;
        const int rowi = pcc / NPB, part = pcc - rowi * NPB;
        const float* xv = lds + Ldg + (rowi >= DH ? 0 : NG) + part * KPB;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < KPB; k += 2) {
          const float2 x = *reinterpret_cast<const float2*>(xv + k);
          a0 = fmaf(wreg[i][k], x.x, a0);
          a1 = fmaf(wreg[i][k + 1], x.y, a1);
        }
        if (pc < NPIECE) lds[Lred + pc] = a0 + a1;
      }
      lds_barrier();
      ++epoch;
      unsigned long long* g2 = gpar + (size_t)kClu * D;
      for (int r2 = tid; r2 < 2 * DH; r2 += kCluThreads) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < NPB; ++q) a += lds[Lred + r2 * NPB + q];
        granule_store(g2 + (size_t)member * 2 * D + r2, epoch, a);
      }
      // exchange 2: wave 1 the carry partials of the own units, waves 2 and 3 the dh1 partials of two members each
      if (wave >= 1) {
        bool ok;
        if (wave == 1) {
          ok = wave_gather_map<4>(kClu * UD, epoch, lane, [&](int i, const unsigned long long*& slot, float*& dst) {
            const int m2 = i / UD, u = i - m2 * UD;
            slot = g2 + (size_t)m2 * 2 * D + member * UD + u;
            dst = lds + Lpcd + i;
          });
        } else {
          const int mb = (wave - 2) * 2;
          ok = wave_gather_map<(2 * DH + 63) / 64>(2 * DH, epoch, lane, [&](int i, const unsigned long long*& slot, float*& dst) {
            const int m2 = mb + i / DH, k = i % DH;
            slot = g2 + (size_t)m2 * 2 * D + D + k;
            dst = lds + Lpdh + m2 * DH + k;
          });
        }
        if (!ok && lane == 0) { *abort_flag = 1; atomicExch(status, 2 + 2 * t); }
      }
      lds_barrier();
      if (*abort_flag) return;

      // (g) carry_d of the own units; dz1 = act'(h1) * dh1 (all H, every member)
      if (tid < UD) {
        float a = lds[Lcd + tid];
#pragma unroll
        for (int m2 = 0; m2 < kClu; ++m2) a += lds[Lpcd + m2 * UD + tid];
        lds[Lcd + tid] = a;
      }
      if (tid < H) {
        float a = 0.f;
#pragma unroll
        for (int m2 = 0; m2 < kClu; ++m2) a += lds[Lpdh + m2 * DH + tid];
        const float g = a * act_grad_from_out(lds[Lh1 + tid], act);
        lds[Ldz1 + tid] = g;
        if (member == 0) io.d_z1[bt * H + tid] = g;
      }
      lds_barrier();
      // (h) carry_s[s] = sum_k W1s[s][k] dz1[k]: W1P partial sums per s, then their sum
      {
        const int srow = tid / W1P, part = tid % W1P;
        if (srow < S) {
          float a0 = 0.f;
#pragma unroll
          for (int k = 0; k < KW1; ++k) {
            const int kk = part * KW1 + k;
            a0 = fmaf(w1r[k], kk < H ? lds[Ldz1 + kk] : 0.f, a0);
          }
          lds[Lred2 + srow * W1P + part] = a0;
        }
      }
      lds_barrier();
      if (tid < S) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < W1P; ++q) a += lds[Lred2 + tid * W1P + q];
        lds[Lcs + tid] = a;
      }
      lds_barrier();
    }
    if (tid < UD) io.g_deter0[(size_t)row * D + member * UD + tid] = lds[Lcd + tid];
    if (member == 0 && tid < S) io.g_stoch0[(size_t)row * S + tid] = lds[Lcs + tid];
  }
}

static int cluster_npb(int DH) { return (DH == 200 || DH == 128) ? 3 : 1; }

static size_t cluster_bwd_lds_floats(int DH, int S) {
  const int UD = DH / kClu, NG = 3 * UD, NH = NG, NHP = (NH + 3) & ~3, NPB = cluster_npb(DH), NPIECE = 2 * DH * NPB;
  size_t o = 0;
  auto take = [&](size_t n) { o += (n + 3) & ~(size_t)3; };
  for (int i = 0; i < 10; ++i) take(S);
  take(NHP); take(NHP);
  take((size_t)4 * UD); take(UD); take(UD); take(UD);
  take((size_t)2 * NG);
  take(DH); take(DH);
  take((size_t)kClu * UD);
  take((size_t)kClu * UD); take((size_t)kClu * DH);
  take((size_t)(NPIECE > DH ? NPIECE : DH));
  take((size_t)(8 * 64));
  take((size_t)DH * NHP);
  take((size_t)32 * NHP);
  take(4);
  return o;
}

size_t mrssm_cluster_bwd_workspace_bytes(const MtrssmMrssmDims* d) {
  if (!d || d->B <= 0 || d->D <= 0) return 0;
  const int nclusters = d->B < 64 ? d->B : 64;
  return 16 + (size_t)nclusters * 2 * (size_t)kClu * 3 * d->D * sizeof(unsigned long long);
}

int mrssm_cluster_supported(const MtrssmMrssmDims* d);

int mrssm_bwd_cluster_launch(const MtrssmMrssmDims* d, const MtrssmMrssmClusterWeights* w, const MtrssmMrssmBwdIO* io, void* workspace,
                             size_t workspace_bytes, hipStream_t stream) {
  if (!mrssm_cluster_supported(d) || d->K * d->C > 32) {
    set_error("mrssm_rollout_bwd_cluster: dims outside the cluster kernel's regime (ask mtrssm_mrssm_cluster_supported first; S <= 32)");
    return MTRSSM_EINVAL;
  }
  if (!w || !io || !workspace || !w->w1s_t || !w->wf_t || !w->whh_t || !w->wh1_t || !w->w4 || !w->wa2 || !w->wv2 || !io->deter0 ||
      !io->deter || !io->prior_logits || !io->post_logits || !io->sv_h1 || !io->sv_gates || !io->sv_heads || !io->sv_la || !io->sv_lv ||
      !io->g_deter0 || !io->g_stoch0 || !io->d_z1 || !io->d_gi || !io->d_gh || !io->d_zh || !io->d_lp || !io->d_la || !io->d_lv) {
    set_error("mrssm_rollout_bwd_cluster: null required pointer");
    return MTRSSM_EINVAL;
  }
  if (workspace_bytes < mrssm_cluster_bwd_workspace_bytes(d) || ((uintptr_t)workspace & 15)) {
    set_error("mrssm_rollout_bwd_cluster: workspace too small (%zu < %zu) or not 16-byte aligned", workspace_bytes,
              mrssm_cluster_bwd_workspace_bytes(d));
    return MTRSSM_EINVAL;
  }
  const int nclusters = d->B < 64 ? d->B : 64;
  const int grid = ((nclusters + 7) / 8) * 32;
  const size_t lds = cluster_bwd_lds_floats(d->D, d->K * d->C) * sizeof(float);
  if (lds > 160 * 1024) { set_error("mrssm_rollout_bwd_cluster: %zu bytes of LDS", lds); return MTRSSM_ELDS; }
  // granules zeroed every launch (first node of the launch under a graph); the status word in front of them is sticky
  if (int rc = clear_async(static_cast<char*>(workspace) + 16, mrssm_cluster_bwd_workspace_bytes(d) - 16, stream)) return rc;
  hipError_t e = hipSuccess;
  int* status = reinterpret_cast<int*>(workspace);
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + 16);
#define MTRSSM_CLUB_LAUNCH(DHV, NPV, PPTV)                                                                                         \
  {                                                                                                                               \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mrssm_bwd_cluster_kernel<DHV, NPV, PPTV>),                              \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                               \
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e)); return MTRSSM_ELAUNCH; } \
    set_last_kernel("mtrssm::mrssm_bwd_cluster_kernel<" #DHV ", " #NPV ", " #PPTV ">");                                          \
    hipLaunchKernelGGL((mrssm_bwd_cluster_kernel<DHV, NPV, PPTV>), dim3(grid), dim3(kCluThreads), lds, stream, *d, *w, *io, gran,   \
                       status, nclusters);                                                                                        \
  }
  // row pieces per thread = ceil(2 DH NPB / 256)
  if (d->D == 32) MTRSSM_CLUB_LAUNCH(32, 1, 1)
  else if (d->D == 64) MTRSSM_CLUB_LAUNCH(64, 1, 1)
  else if (d->D == 128) MTRSSM_CLUB_LAUNCH(128, 3, 3)
  else MTRSSM_CLUB_LAUNCH(200, 3, 5)
#undef MTRSSM_CLUB_LAUNCH
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("cluster backward scan launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  return MTRSSM_OK;
}

int debug_set_cluster_profile(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_cluster_prof), &p, sizeof(p)) == hipSuccess ? MTRSSM_OK : MTRSSM_ELAUNCH;
}

static int cluster_kw(int D, int H) { return D > H ? D : H; }

static int cluster_np(int DH) { return DH == 200 ? 5 : (DH == 128 ? 4 : (DH == 64 ? 2 : 1)); }

static size_t cluster_lds_floats(int D, int H, int S) {
  const int UD = D / kClu, UH = H / kClu, NG = 3 * UD, NH = 3 * UH, KW = cluster_kw(D, H), NP = cluster_np(KW);
  size_t o = 0;
  auto take = [&](size_t n) { o += (n + 3) & ~(size_t)3; };
  take(S); take(KW); take(KW); take(KW); take(3 * 64);
  take(S); take(S); take(S); take(S);
  take((size_t)kClu * 3 * S);
  take((size_t)(2 * NG * NP > NH * NP ? 2 * NG * NP : NH * NP));
  take((size_t)S * H);
  take((size_t)D * NH);
  take((size_t)3 * S);
  take(2 * 64);
  take(4);
  return o;
}

size_t mrssm_cluster_workspace_bytes(const MtrssmMrssmDims* d) {
  if (!d || d->B <= 0 || d->D <= 0 || d->K <= 0 || d->C <= 0) return 0;
  const int nclusters = d->B < 64 ? d->B : 64;
  const size_t per_parity = (size_t)kClu * (d->D / kClu + 3 * d->K * d->C);
  return 16 + (size_t)nclusters * 2 * per_parity * sizeof(unsigned long long);
}

// 1 if the cluster kernel takes these dims (the caller then provides the workspace), else 0
int mrssm_cluster_supported(const MtrssmMrssmDims* d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->D <= 0 || d->H <= 0 || d->K <= 0 || d->C <= 0 || !d->post) return 0;
  const int S = d->K * d->C, D = d->D, H = d->H;
  if (D != H || (D != 32 && D != 64 && D != 128 && D != 200)) return 0;  // the instantiated square sizes
  if (3 * S > kCluThreads || H / kClu > 64 || 3 * H / kClu > kCluThreads || d->K > 64) return 0;  // logit rows / head units: one thread each
  if (D > 4 * kWave || kClu * 3 * S > 6 * kWave) return 0;  // granules per gather (wave_gather<4> / <6>)
  if (cluster_lds_floats(D, H, S) * sizeof(float) > 160 * 1024) return 0;
  // every workgroup takes a whole CU's LDS and spins on its partners: the grid must fit the device's CUs (a partitioned or
  // CU-masked device reports fewer than 256)
  const int nclusters = d->B < 64 ? d->B : 64;
  if (device_cu_count() < ((nclusters + 7) / 8) * 32) return 0;
  return 1;
}

int mrssm_fwd_cluster_launch(const MtrssmMrssmDims* d, const MtrssmMrssmClusterWeights* w, const MtrssmMrssmFwdIO* io, void* workspace,
                             size_t workspace_bytes, hipStream_t stream) {
  if (!mrssm_cluster_supported(d)) {
    set_error("mrssm_rollout_fwd_cluster: dims outside the cluster kernel's regime (ask mtrssm_mrssm_cluster_supported first)");
    return MTRSSM_EINVAL;
  }
  if (!w || !io || !workspace || !w->w1s_t || !w->wf_t || !w->bf || !w->whh_t || !w->bhh || !w->wh1_t || !w->b3 || !w->w4 || !w->b4 ||
      !w->wa2 || !w->ba2 || !w->wv2 || !w->bv2 || !io->xa || !io->pa || !io->pv || !io->deter0 || !io->stoch0 || !io->u_post ||
      !io->deter || !io->prior_logits || !io->post_logits || !io->post_stoch) {
    set_error("mrssm_rollout_fwd_cluster: null required pointer");
    return MTRSSM_EINVAL;
  }
  if (workspace_bytes < mrssm_cluster_workspace_bytes(d) || ((uintptr_t)workspace & 15)) {
    set_error("mrssm_rollout_fwd_cluster: workspace too small (%zu < %zu) or not 16-byte aligned", workspace_bytes,
              mrssm_cluster_workspace_bytes(d));
    return MTRSSM_EINVAL;
  }
  const int nclusters = d->B < 64 ? d->B : 64;
  const int groups = (nclusters + 7) / 8;            // 8 clusters per 32 consecutive blocks
  const int grid = groups * 32;
  const size_t lds = cluster_lds_floats(d->D, d->H, d->K * d->C) * sizeof(float);
  // granules zeroed every launch (first node of the launch under a graph); the status word in front of them is STICKY: the
  // caller zeroes it when allocating, a failed exchange sets it, nothing clears it
  if (int rc = clear_async(static_cast<char*>(workspace) + 16, mrssm_cluster_workspace_bytes(d) - 16, stream)) return rc;
  hipError_t e = hipSuccess;
  int* status = reinterpret_cast<int*>(workspace);
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + 16);
  const int kw = cluster_kw(d->D, d->H);
#define MTRSSM_CLU_LAUNCH(DHV, NPV, PPTV)                                                                                          \
  {                                                                                                                               \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mrssm_fwd_cluster_kernel<DHV, NPV, PPTV>),                              \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                               \
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e)); return MTRSSM_ELAUNCH; } \
    set_last_kernel("mtrssm::mrssm_fwd_cluster_kernel<" #DHV ", " #NPV ", " #PPTV ">");                                          \
    hipLaunchKernelGGL((mrssm_fwd_cluster_kernel<DHV, NPV, PPTV>), dim3(grid), dim3(kCluThreads), lds, stream, *d, *w, *io, gran,   \
                       status, nclusters);                                                                                        \
  }
  // pieces per thread = ceil(2 * (3 DH / 4) * NP / 256)
  if (kw == 32) MTRSSM_CLU_LAUNCH(32, 1, 1)
  else if (kw == 64) MTRSSM_CLU_LAUNCH(64, 2, 1)
  else if (kw == 128) MTRSSM_CLU_LAUNCH(128, 4, 3)
  else MTRSSM_CLU_LAUNCH(200, 5, 6)
#undef MTRSSM_CLU_LAUNCH
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("cluster scan launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  return MTRSSM_OK;
}

}  // namespace mtrssm
