// MoPoE-MRSSM forward scan, one batch row on a CLUSTER of four compute units (gfx950).
//
// The single-CU scan (mrssm_scan.hip) streams the step's 1.5 MB of fp32 weights from L2 through one CU's 64 B/clk load
// path every timestep: 31 us per step against 0.4 us of arithmetic.  Here a row's step is split over kClu = 4 workgroups
// (= 4 CUs: each workgroup takes a whole CU's LDS), each of which keeps ITS QUARTER of the weights resident for all T
// steps -- one column of the fused GRU input matrix or of W_hh per thread in registers (up to 200 floats), the head matrix
// and W1s in LDS (144 KB) -- so nothing is streamed and the step costs its arithmetic plus two exchanges:
//
//   every member   h1 = act(xa + W1s s)                                    (redundant: H x S, W1s in LDS)
//   member m       gi | gh for its D/4 deter units (threads own whole columns: no reduction), gates, d_new[own]
//                                                                           -> publish D/4 values, gather 3 D/4   (exchange 1)
//   member m       head layer 0 for its H/4 units of the prior / audio / vision heads, then its share of the 3 S logit
//                  dot products (partial sums over its units)              -> publish 3 S partials, gather 9 S   (exchange 2)
//   every member   logits = bias + the four partials in member order, MoPoE mix, KL, sample   (redundant, identical bits)
//
// Thread roles (512 threads, 8 waves, <= 256 VGPRs): waves 0-2 one column of (W_ih W2)^T each (gi), waves 3-5 one column of
// W_hh^T each (gh), both groups split head layer 0's reduction between them; threads 384.. hold one row of the second-layer
// slices (logit partials); wave 7 gathers the exchanges and runs the categorical block.
//
// Exchanges are 8-byte {epoch, value} granules written by ONE sc1 (write-through) store and polled by ONE wave with sc1 loads:
// the data is the flag, no fence, no ordering requirement (cdna_hip_programming.md Guideline 16, form R2); granule words are
// zeroed by a memset node in the launch function, epochs count from 1 inside the launch.  Every spin is bounded: a wave that
// gives up writes a code to the launch's status word and the whole workgroup leaves (the host reads the word after the
// launch; results are then invalid).  The grid is 4 x min(B, 64) workgroups of 512 threads with ~150 KB of LDS each: one per
// CU, all resident on the 256 CUs (a cluster loops over rows c, c + clusters, ...), which the spins rely on.
//
// The GRU's input path is fused: gi = W_ih (W2 h1 + b2) + b_ih = (W_ih W2) h1 + (W_ih b2 + b_ih); the caller hands in the
// product (one small GEMM per launch).  Rounding differs from the two-step form by ~1e-7 relative; h2 itself (needed by
// dW_ih) is recomputed after the scan as one batched GEMM.
#include "scan_common.h"

namespace mtrssm {

void set_error(const char* fmt, ...);
void set_last_kernel(const char* name);

constexpr int kClu = 4;            // workgroups (CUs) per row
constexpr int kCluThreads = 512;
constexpr int kCluGroup = 192;     // threads per column group (3 waves): own gate outputs / own head units <= 192
constexpr int kCluLogit0 = 384;    // first thread of the logit-partial rows
constexpr unsigned kSpinLimit = 1u << 22;

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

template <int DH>  // D = H = DH at compile time: the resident columns are DH registers, every stride an immediate
__global__ __launch_bounds__(kCluThreads) void mrssm_fwd_cluster_kernel(const MtrssmMrssmDims dm, const MtrssmMrssmClusterWeights w,
                                                                        const MtrssmMrssmFwdIO io, unsigned long long* __restrict__ gran,
                                                                        int* __restrict__ status, int nclusters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int D = DH, H = DH, KW = DH;
  constexpr int UD = D / kClu, UH = H / kClu;
  constexpr int NG = 3 * UD, NH = 3 * UH;  // own gate outputs / own head units
  const int K = dm.K, C = dm.C, S = K * C, T = dm.T, act = dm.act;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  // blocks b, b + 8, b + 16, b + 24 share an XCD under round-robin placement (speed only)
  const int blk = blockIdx.x;
  const int member = (blk >> 3) & (kClu - 1);
  const int cluster = (blk & 7) + 8 * (blk >> 5);
  if (cluster >= nclusters) return;  // whole clusters drop out together

  // ---- LDS carve-up (floats).  The vectors the register-resident columns are multiplied with are KW long, zero padded.
  int o = 0;
  auto take = [&](int n) { const int r = o; o += (n + 3) & ~3; return r; };
  const int Ls = take(S), Lh1 = take(KW), Ld0 = take(KW), Ld1 = take(KW), Lhd = take(3 * 64);  // head units [which][64], zero padded
  const int Llp = take(S), Lla = take(S), Llv = take(S), Lmx = take(S);
  const int Lpart = take(kClu * 3 * S);   // logit partial sums of the four members
  const int Lgi = take(NG), Lgh = take(NG), Lred = take(2 * NH);
  const int Lw1 = take(S * H);            // W1s^T [S][H]
  const int Lwh = take(D * NH);           // [k][o]: this member's columns of the head layer 0
  const int Lflag = take(4);
  (void)o;
  int* abort_flag = reinterpret_cast<int*>(lds + Lflag);

  // ---- roles and resident weights
  const bool roleG = tid < kCluGroup && tid < NG;                                  // a column of (W_ih W2)^T: gi
  const bool roleH = tid >= kCluGroup && tid < 2 * kCluGroup && tid - kCluGroup < NG;   // a column of W_hh^T: gh
  const int og = tid < kCluGroup ? tid : tid - kCluGroup;                          // own output inside the group
  const bool roleL = tid >= kCluLogit0 && tid - kCluLogit0 < 3 * S;                // a row of the second-layer slices
  const int lrow = tid - kCluLogit0;
  // own gate output og = g * UD + u  <->  column g * D + member * UD + u of the [.][3D] matrices
  const int gcol = (og / UD) * D + member * UD + (og % UD);
  // one resident vector per thread (no selects, immediate strides: 200 masked loads spilled the register file):
  //   group G / H: column gcol of wf_t / whh_t, DH elements, stride 3 DH;  logit rows: UH consecutive elements, then zeros
  float wreg[KW];
  {
    // Column roles (waves 0-5): wave-uniform matrix base + compile-time k stride + one 32-bit lane offset, so the 200 loads
    // need no per-load address registers.  Row roles (waves 6-7): one per-lane base, immediate offsets (UH <= 64 elements).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    if (wave_u < 6) {
      const float* ub = wave_u < 3 ? w.wf_t : w.whh_t;
      const int gc = (roleG || roleH) ? gcol : 0;
#pragma unroll
      for (int k = 0; k < KW; ++k) wreg[k] = ub[(size_t)k * 3 * D + gc];
    } else {
      const int which = roleL ? lrow / S : 0, srow = roleL ? lrow - which * S : 0;
      const float* rowsrc = (which == 0 ? w.w4 : (which == 1 ? w.wa2 : w.wv2)) + (size_t)srow * H + member * UH;
#pragma unroll
      for (int k = 0; k < KW; ++k) wreg[k] = k < UH ? rowsrc[k] : 0.f;
    }
  }
  for (int idx = tid; idx < S * H; idx += kCluThreads) lds[Lw1 + idx] = w.w1s_t[idx];
  // head layer 0: own head unit oh = which * UH + u  <->  column which * H + member * UH + u of wh1_t [D][3H]
  for (int idx = tid; idx < D * NH; idx += kCluThreads) {
    const int k = idx / NH, oh = idx - k * NH;
    lds[Lwh + idx] = w.wh1_t[(size_t)k * 3 * H + (oh / UH) * H + member * UH + (oh % UH)];
  }
  for (int i = tid; i < KW; i += kCluThreads) { lds[Lh1 + i] = 0.f; lds[Ld0 + i] = 0.f; lds[Ld1 + i] = 0.f; }
  for (int i = tid; i < 3 * 64; i += kCluThreads) lds[Lhd + i] = 0.f;
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
  float bias_l = 0.f;                                                // thread i < 3S: bias of logit i
  if (tid < 3 * S) {
    const int which = tid / S, s = tid - which * S;
    bias_l = (which == 0 ? w.b4 : (which == 1 ? w.ba2 : w.bv2))[s];
  }
  if (tid == 0) *abort_flag = 0;

  // granule slots of this cluster: [parity][ D-exchange: member x UD | logit exchange: member x 3S ]
  const int slotsD = UD, slotsL = 3 * S;
  const int per_parity = kClu * (slotsD + slotsL);
  unsigned long long* gbase = gran + (size_t)cluster * 2 * per_parity;
  unsigned epoch = 0;
  const int khalf = ((D / 2) + 3) & ~3;  // head layer 0: group G reduces k < khalf, group H the rest

  for (int row = cluster; row < dm.B; row += nclusters) {
    int cur = Ld0, nxt = Ld1;
    __syncthreads();
    for (int i = tid; i < D; i += kCluThreads) lds[cur + i] = io.deter0[(size_t)row * D + i];
    for (int i = tid; i < S; i += kCluThreads) lds[Ls + i] = io.stoch0[(size_t)row * S + i];
    __syncthreads();

    for (int t = 0; t < T; ++t) {
      const size_t bt = (size_t)row * T + t;
      unsigned long long* gpar = gbase + (size_t)(t & 1) * per_parity;
      // early loads of this step's streamed inputs (consumed several barriers later)
      const float xa_v = tid < H ? io.xa[bt * H + tid] : 0.f;
      float pav = 0.f;
      if (tid >= UH && tid < NH) {
        const int which = tid / UH, u = tid - which * UH;
        pav = (which == 1 ? io.pa : io.pv)[bt * H + member * UH + u];
      }

      // (1) h1 = act(xa + W1s s): every member, all H outputs                    networks.py:165-166
      if (tid < H) {
        float a = xa_v;
        const float* wc = lds + Lw1 + tid;
        for (int k = 0; k < S; ++k) a = fmaf(wc[(size_t)k * H], lds[Ls + k], a);
        const float h = act_fwd(a, act);
        lds[Lh1 + tid] = h;
        if (member == 0 && io.sv_h1) io.sv_h1[bt * H + tid] = h;
      }
      __syncthreads();

      // (2) gi (fused W_ih W2) by group G, gh (W_hh) by group H: one whole column per thread, no reduction
      if (roleG || roleH) {
        const float* xv = lds + (roleG ? Lh1 : cur);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < KW; k += 4) {
          const float4 x = *reinterpret_cast<const float4*>(xv + k);
          a0 = fmaf(wreg[k], x.x, a0);
          a1 = fmaf(wreg[k + 1], x.y, a1);
          a2 = fmaf(wreg[k + 2], x.z, a2);
          a3 = fmaf(wreg[k + 3], x.w, a3);
          // registers are the scarce resource here (the column itself holds up to 200): an empty asm that "uses" the
          // accumulators and clobbers memory keeps the operand reads of the unrolled loop from being hoisted to its top
          asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "memory");
        }
        lds[(roleG ? Lgi : Lgh) + og] = (a0 + a1) + (a2 + a3);
      }
      __syncthreads();

      // (3) gates of the own deter units; publish them                          networks.py:170 (nn.GRUCell)
      ++epoch;
      if (tid < UD) {
        const float gi0 = lds[Lgi + tid] + bias_g[0], gi1 = lds[Lgi + UD + tid] + bias_g[1], gi2 = lds[Lgi + 2 * UD + tid] + bias_g[2];
        const float gh0 = lds[Lgh + tid] + bias_g[3], gh1 = lds[Lgh + UD + tid] + bias_g[4], gh2 = lds[Lgh + 2 * UD + tid] + bias_g[5];
        const float rg = sigmoidf_(gh0 + gi0);
        const float zg = sigmoidf_(gh1 + gi1);
        const float ng = tanhf(gi2 + gh2 * rg);
        const int unit = member * UD + tid;
        const float dnew = (lds[cur + unit] - ng) * zg + ng;
        lds[nxt + unit] = dnew;
        granule_store(gpar + (size_t)member * slotsD + tid, epoch, dnew);
        io.deter[bt * D + unit] = dnew;
        if (io.sv_gates) {
          float* gsv = io.sv_gates + bt * 4 * D;
          gsv[unit] = rg; gsv[D + unit] = zg; gsv[2 * D + unit] = ng; gsv[3 * D + unit] = gh2;
        }
      }
      if (wave == 7) {  // exchange 1: all D deter units, in unit order (the own ones come back unchanged)
        const bool ok = wave_gather<4>(gpar, lds + nxt, D, epoch, lane);
        if (!ok && lane == 0) { *abort_flag = 1; atomicExch(status, 1 + 2 * t); }
      }
      __syncthreads();
      if (*abort_flag) return;

      // (4) head layer 0 for the own units of the three heads (matrix slice resident in LDS): the two column groups split
      //     the reduction                                                        networks.py:171, 82
      if ((tid < kCluGroup || (tid >= kCluGroup && tid < 2 * kCluGroup)) && og < NH) {
        const bool first = tid < kCluGroup;
        const int k0 = first ? 0 : khalf, k1 = first ? (khalf < D ? khalf : D) : D;
        const float* wcol = lds + Lwh + og;
        const float* dv = lds + nxt;
        float p0 = 0.f, p1 = 0.f;
        int k = k0;
        for (; k + 2 <= k1; k += 2) {
          p0 = fmaf(wcol[(size_t)k * NH], dv[k], p0);
          p1 = fmaf(wcol[(size_t)(k + 1) * NH], dv[k + 1], p1);
        }
        if (k < k1) p0 = fmaf(wcol[(size_t)k * NH], dv[k], p0);
        lds[Lred + (first ? 0 : NH) + og] = p0 + p1;
      }
      __syncthreads();
      if (tid < NH) {
        const float z = (tid < UH ? bias_h : pav) + (lds[Lred + tid] + lds[Lred + NH + tid]);
        const float h = act_fwd(z, act);
        const int which = tid / UH, u = tid - which * UH;
        lds[Lhd + which * 64 + u] = h;
        if (io.sv_heads) io.sv_heads[bt * 3 * H + which * H + member * UH + u] = h;
      }
      __syncthreads();

      // (5) this member's share of the 3 S logit dot products (rows resident in registers); publish; gather the others
      ++epoch;
      unsigned long long* gl = gpar + (size_t)kClu * slotsD;
      if (roleL) {
        const float* hv = lds + Lhd + (lrow / S) * 64;
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int u = 0; u < 64; u += 4) {  // UH <= 64 (mrssm_cluster_supported); hv is zero beyond UH
          const float4 x = *reinterpret_cast<const float4*>(hv + u);
          p0 = fmaf(wreg[u], x.x, p0);
          p1 = fmaf(wreg[u + 1], x.y, p1);
          p0 = fmaf(wreg[u + 2], x.z, p0);
          p1 = fmaf(wreg[u + 3], x.w, p1);
          asm volatile("" : "+v"(p0), "+v"(p1) :: "memory");
        }
        const float p = p0 + p1;
        lds[Lpart + member * 3 * S + lrow] = p;
        granule_store(gl + (size_t)member * slotsL + lrow, epoch, p);
      }
      if (wave == 7) {  // exchange 2: the four members' 3 S partial sums, [member][3S]
        const bool ok = wave_gather<6>(gl, lds + Lpart, kClu * 3 * S, epoch, lane);
        if (!ok && lane == 0) { *abort_flag = 1; atomicExch(status, 2 + 2 * t); }
      }
      __syncthreads();
      if (*abort_flag) return;

      // (6) logits = bias + partials in member order (the same bits on every member)
      if (tid < 3 * S) {
        float v = bias_l;
#pragma unroll
        for (int m2 = 0; m2 < kClu; ++m2) v += lds[Lpart + m2 * 3 * S + tid];
        const int which = tid / S, s = tid - which * S;
        lds[(which == 0 ? Llp : (which == 1 ? Lla : Llv)) + s] = v;
      }
      __syncthreads();

      // (7) fusion, per-categorical softmax, KL, sampling: one wave (every member; member 0 writes the outputs)
      if (wave == 7) {
        const bool writer = member == 0;
        wave_mopoe_mix(lds + Lla, lds + Llv, lds + Lmx, S, lane);
        for (int s = lane; s < S; s += kWave) {
          if (writer) {
            io.prior_logits[bt * S + s] = lds[Llp + s];
            io.post_logits[bt * S + s] = lds[Lmx + s];
            if (io.sv_la) { io.sv_la[bt * S + s] = lds[Lla + s]; io.sv_lv[bt * S + s] = lds[Llv + s]; }
          }
        }
        float kl = cat_block_fwd<true>(lds + Lmx, lds + Llp, K, C, lane, io.u_post + bt * K, io.u_prior ? io.u_prior + bt * K : nullptr,
                                       lds + Ls, io.post_stoch + bt * S, io.prior_stoch ? io.prior_stoch + bt * S : nullptr, writer);
        if (io.kl) {
          kl = wave_sum(kl);
          if (lane == 0 && writer) io.kl[bt] = kl;
        }
      }
      __syncthreads();
      const int tmp = cur; cur = nxt; nxt = tmp;
    }
  }
}

static int cluster_kw(int D, int H) { return D > H ? D : H; }

static size_t cluster_lds_floats(int D, int H, int S) {
  const int UD = D / kClu, UH = H / kClu, NG = 3 * UD, NH = 3 * UH, KW = cluster_kw(D, H);
  size_t o = 0;
  auto take = [&](size_t n) { o += (n + 3) & ~(size_t)3; };
  take(S); take(KW); take(KW); take(KW); take(3 * 64);
  take(S); take(S); take(S); take(S);
  take((size_t)kClu * 3 * S);
  take(NG); take(NG); take((size_t)2 * NH);
  take((size_t)S * H);
  take((size_t)D * NH);
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
  if (3 * D / kClu > kCluGroup || 3 * H / kClu > kCluGroup) return 0;
  if (3 * S > kCluThreads - kCluLogit0 || H / kClu > 64) return 0;  // logit rows sit in threads 384..511, <= 64 units each
  if (D > 4 * kWave || kClu * 3 * S > 6 * kWave) return 0;  // granules per gather (wave_gather<4> / <6>)
  if (cluster_lds_floats(D, H, S) * sizeof(float) > 160 * 1024) return 0;
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
  // status word + granules: zeroed every launch (a memset node: replayed first under a graph)
  hipError_t e = hipMemsetAsync(workspace, 0, mrssm_cluster_workspace_bytes(d), stream);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  int* status = reinterpret_cast<int*>(workspace);
  unsigned long long* gran = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + 16);
  const int kw = cluster_kw(d->D, d->H);
#define MTRSSM_CLU_LAUNCH(KWV)                                                                                                     \
  {                                                                                                                               \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mrssm_fwd_cluster_kernel<KWV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                                            \
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(max dynamic LDS=%zu): %s", lds, hipGetErrorString(e)); return MTRSSM_ELAUNCH; } \
    set_last_kernel("mtrssm::mrssm_fwd_cluster_kernel<" #KWV ">");                                                               \
    hipLaunchKernelGGL(mrssm_fwd_cluster_kernel<KWV>, dim3(grid), dim3(kCluThreads), lds, stream, *d, *w, *io, gran, status, nclusters); \
  }
  if (kw == 32) MTRSSM_CLU_LAUNCH(32)
  else if (kw == 64) MTRSSM_CLU_LAUNCH(64)
  else if (kw == 128) MTRSSM_CLU_LAUNCH(128)
  else MTRSSM_CLU_LAUNCH(200)
#undef MTRSSM_CLU_LAUNCH
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("cluster scan launch failed: %s", hipGetErrorString(e)); return MTRSSM_ELAUNCH; }
  return MTRSSM_OK;
}

}  // namespace mtrssm
