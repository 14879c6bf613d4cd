/* mtrssm.h -- C-ABI of libmtrssm_hip.so: MI355X (gfx950) kernels for the MoPoE-M(MT)RSSM
 * sequential-rollout train step.
 *
 * The reference (Mamo1031/Multimodal-MTRSSM) is pure Python and has no FFI of its own; its boundary
 * for this path is the Python class API.  This header is the build-defined C
 * boundary underneath the drop-in Python classes of `multimodal_mtrssm_amd/`: every entry point
 * names the reference code it replaces.
 *
 * Conventions
 *   - plain C: pointers and sizes only, no torch / HIP types.  `stream` is a hipStream_t passed as
 *     void* (NULL = the null stream).  Kernels are launched asynchronously on it.
 *   - all device buffers are caller-owned, fp32, row-major contiguous; sequence tensors are
 *     [B, T, dim] (a row's T steps contiguous).  The library never allocates, frees or keeps
 *     global mutable state and is re-entrant across streams.
 *   - return 0 on success, a negative MTRSSM_E* code otherwise; mtrssm_last_error() gives the
 *     thread-local message.  There is NO CPU fallback.
 *   - S = K*C flat stochastic size, K = category_size (number of categoricals), C = class_size.
 *   - a matrix-vector product with a WIDE output streams its matrix reduction-major (row r of the
 *     buffer = all outputs for input r; threads own outputs, loads coalesce); one with a NARROW
 *     output (the S-wide heads) streams it output-major (one wave per output row).  Hence the
 *     forward scan takes "_t" (= W^T, [in][out]) copies of the wide layers and the PyTorch
 *     [out][in] layout of the narrow ones, and the backward scan the opposite.
 */
#ifndef MTRSSM_H
#define MTRSSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTRSSM_VERSION 100 /* 0.1.0 */

#define MTRSSM_OK 0
#define MTRSSM_EINVAL (-1)  /* bad dims / null pointer */
#define MTRSSM_ELAUNCH (-2) /* HIP launch error */
#define MTRSSM_ELDS (-3)    /* dims do not fit the 160 KiB LDS of a CU in this kernel regime */

/* activation ids (torch.nn names the reference YAML uses: Identity, ReLU, ELU, Tanh) */
#define MTRSSM_ACT_IDENTITY 0
#define MTRSSM_ACT_RELU 1
#define MTRSSM_ACT_ELU 2
#define MTRSSM_ACT_TANH 3

int mtrssm_version(void);
const char* mtrssm_last_error(void);
/* name of the device kernel the calling thread's most recent entry-point call launched (as rocprofv3 prints it,
 * without the argument list), e.g. "mtrssm::conv_gather_gemm_patch_kernel<2>": lets a caller key its own HIP-event
 * timings exactly like a kernel trace. */
const char* mtrssm_last_kernel(void);

/* ------------------------------------------------------------------------------------------
 * MoPoE-MRSSM scan.  Replaces the T loop of MoPoE_MRSSM.rollout_representation
 * (mrssm/mopoe_mrssm/core.py:221-256): Transition.forward (networks.py:151-173), the two posterior
 * heads (core.py:62-84), flat log-softmax + PoE + MoE (core.py:241-243, 112-163), State sampling
 * (state.py:17) and the per-step KL term of BaseRSSM.shared_step (core.py:212-216).
 * With post == 0 it is BaseRSSM.rollout_transition (core.py:170-185): prior-only.
 *
 * Contractions that do not depend on the recurrence are hoisted out of the loop by the caller:
 *   xa[b,t,:] = W1[:, :A] a[b,t] + b1          (action part of action_state_projector.0)
 *   pa[b,t,:] = Wa1[:, D:] ea[b,t] + ba1       (embedding part of audio rnn_to_post_projector.0)
 *   pv[b,t,:] = Wv1[:, D:] ev[b,t] + bv1
 * ------------------------------------------------------------------------------------------ */
typedef struct MtrssmMrssmDims {
  int32_t B, T;          /* sequences, timesteps */
  int32_t D, H;          /* deterministic size, hidden (num_cells) */
  int32_t K, C;          /* categoricals, classes */
  int32_t act;           /* MTRSSM_ACT_* of the three MLPs */
  int32_t post;          /* 1: posterior rollout; 0: prior-only rollout */
  float kl_w_post;       /* d(loss)/d(KL) share sent to the posterior (1-alpha with balancing, else 1) */
  float kl_w_prior;      /* share sent to the prior (alpha with balancing, else 1) */
  int32_t rows_per_block; /* 0 = library default */
  int32_t threads;        /* 0 = library default */
} MtrssmMrssmDims;

typedef struct MtrssmMrssmFwdWeights {
  const float* w1s_t;  /* [S][H]   action_state_projector.0.weight[:, A:]^T */
  const float* w2_t;   /* [H][H]   action_state_projector.2.weight^T */
  const float* b2;     /* [H] */
  const float* wih_t;  /* [H][3D]  rnn_cell.weight_ih^T (gate order r,z,n) */
  const float* bih;    /* [3D] */
  const float* whh_t;  /* [D][3D]  rnn_cell.weight_hh^T */
  const float* bhh;    /* [3D] */
  const float* wh1_t;  /* [D][3H]  [prior.0.weight ; audio post.0.weight[:, :D] ; vision post.0.weight[:, :D]]^T
                                   (prior-only: [D][H]) */
  const float* b3;     /* [H]      rnn_to_prior_projector.0.bias */
  const float* w4;     /* [S][H]   rnn_to_prior_projector.2.weight (PyTorch layout: narrow output, one wave per row) */
  const float* b4;     /* [S] */
  const float* wa2;    /* [S][H]   audio rnn_to_post_projector.2.weight */
  const float* ba2;    /* [S] */
  const float* wv2;    /* [S][H]   vision rnn_to_post_projector.2.weight */
  const float* bv2;    /* [S] */
} MtrssmMrssmFwdWeights;

typedef struct MtrssmMrssmFwdIO {
  /* inputs */
  const float* xa;      /* [B,T,H] */
  const float* pa;      /* [B,T,H]  (post only) */
  const float* pv;      /* [B,T,H]  (post only) */
  const float* deter0;  /* [B,D] */
  const float* stoch0;  /* [B,S] */
  const float* u_post;  /* [B,T,K] uniforms for the posterior sample (post only) */
  const float* u_prior; /* [B,T,K] uniforms for the prior sample (NULL: not sampled when post=1) */
  /* outputs */
  float* deter;         /* [B,T,D] */
  float* prior_logits;  /* [B,T,S] raw prior logits */
  float* prior_stoch;   /* [B,T,S] one-hot prior sample (NULL allowed when post=1) */
  float* post_logits;   /* [B,T,S] mixed log-probs (post only) */
  float* post_stoch;    /* [B,T,S] one-hot posterior sample (post only) */
  float* kl;            /* [B,T]   sum_K KL(q_k || p_k) (post only, NULL allowed) */
  /* activations saved for the backward scan (all NULL = inference) */
  float* sv_h1;         /* [B,T,H]  act(MLP1 layer 0) */
  float* sv_h2;         /* [B,T,H]  MLP1 output */
  float* sv_gates;      /* [B,T,4D] r, z, n, (W_hn h + b_hn) */
  float* sv_heads;      /* [B,T,3H] act of prior / audio / vision head layer 0 */
  float* sv_la;         /* [B,T,S]  audio logits */
  float* sv_lv;         /* [B,T,S]  vision logits */
} MtrssmMrssmFwdIO;

int mtrssm_mrssm_rollout_fwd(const MtrssmMrssmDims* dims, const MtrssmMrssmFwdWeights* w,
                             const MtrssmMrssmFwdIO* io, void* stream);

/* The same forward scan with one batch row on a CLUSTER of four compute units (csrc/mrssm_cluster.hip): every workgroup
 * keeps its quarter of the weights resident in registers / LDS for all T steps and the four exchange D/4 deter values and
 * 3S logit partial sums per step through 8-byte {epoch, value} granules (no weight is re-streamed; 31 -> ~5 us per step at
 * D = H = 200).  Posterior rollout only (post = 1).  The GRU input path is fused by the caller:
 *   wf_t = w2_t . wih_t  ([H][3D] = (W_ih W2)^T),   bf = W_ih b2 + b_ih,
 * so sv_h2 is NOT written (recompute h2 = W2 h1 + b2 as one batched GEMM where dW_ih needs it).
 * workspace: caller-owned device memory of mtrssm_mrssm_cluster_workspace_bytes() bytes, 16-byte aligned; the call zeroes
 * everything but its first 16 bytes.  Its first int32 is a STICKY status word: the caller zeroes it once when allocating; a
 * launch in which a spin gave up stores a non-zero code there (results invalid) and no launch clears it, so one check at any
 * later synchronisation point sees every earlier failure (mtrssm_adamw_apply can be given the word: it then skips the update).
 * The grid is 4 x min(B, 64) workgroups that must be co-resident (one per CU): launch it on a GPU this process has to
 * itself.  mtrssm_mrssm_cluster_supported() says whether the dims AND the current device (CU count >= grid) fit this regime
 * (else use mtrssm_mrssm_rollout_fwd). */
typedef struct MtrssmMrssmClusterWeights {
  const float* w1s_t;  /* [S][H] */
  const float* wf_t;   /* [H][3D]  (W_ih W2)^T */
  const float* bf;     /* [3D]     W_ih b2 + b_ih */
  const float* whh_t;  /* [D][3D] */
  const float* bhh;    /* [3D] */
  const float* wh1_t;  /* [D][3H] */
  const float* b3;     /* [H] */
  const float* w4;     /* [S][H] */
  const float* b4;     /* [S] */
  const float* wa2;    /* [S][H] */
  const float* ba2;    /* [S] */
  const float* wv2;    /* [S][H] */
  const float* bv2;    /* [S] */
} MtrssmMrssmClusterWeights;
int mtrssm_mrssm_cluster_supported(const MtrssmMrssmDims* dims);
int64_t mtrssm_mrssm_cluster_workspace_bytes(const MtrssmMrssmDims* dims);
int mtrssm_mrssm_rollout_fwd_cluster(const MtrssmMrssmDims* dims, const MtrssmMrssmClusterWeights* weights, const MtrssmMrssmFwdIO* io,
                                     void* workspace, int64_t workspace_bytes, void* stream);

typedef struct MtrssmMrssmBwdWeights {
  const float* w1s_t; /* [S][H]  action_state_projector.0.weight[:, A:]^T (same buffer as the forward's) */
  const float* w2;   /* [H][H] */
  const float* wih;  /* [3D][H] */
  const float* whh;  /* [3D][D] */
  const float* wh1;  /* [3H][D]  [prior.0.weight ; audio post.0.weight[:, :D] ; vision post.0.weight[:, :D]] */
  const float* w4;   /* [S][H] */
  const float* wa2;  /* [S][H] */
  const float* wv2;  /* [S][H] */
} MtrssmMrssmBwdWeights;

typedef struct MtrssmMrssmBwdIO {
  /* forward inputs / outputs / saved activations */
  const float* deter0;       /* [B,D] */
  const float* deter;        /* [B,T,D] */
  const float* prior_logits; /* [B,T,S] */
  const float* post_logits;  /* [B,T,S] */
  const float* sv_h1;
  const float* sv_h2;
  const float* sv_gates;
  const float* sv_heads;
  const float* sv_la;
  const float* sv_lv;
  /* incoming gradients (any may be NULL = zero) */
  const float* g_deter;        /* [B,T,D] */
  const float* g_post_stoch;   /* [B,T,S] straight-through */
  const float* g_prior_stoch;  /* [B,T,S] straight-through */
  const float* g_post_logits;  /* [B,T,S] */
  const float* g_prior_logits; /* [B,T,S] */
  const float* g_kl;           /* [B,T] */
  /* outgoing gradients */
  float* g_deter0;  /* [B,D] */
  float* g_stoch0;  /* [B,S] */
  float* d_z1;      /* [B,T,H]  = d xa        (pre-activation of MLP1 layer 0) */
  float* d_h2;      /* [B,T,H]  grad at MLP1 output */
  float* d_gi;      /* [B,T,3D] */
  float* d_gh;      /* [B,T,3D] */
  float* d_zh;      /* [B,T,3H] pre-activation grads of prior / audio / vision head layer 0 (audio = d pa, vision = d pv) */
  float* d_lp;      /* [B,T,S]  grad at prior logits */
  float* d_la;      /* [B,T,S]  grad at audio logits */
  float* d_lv;      /* [B,T,S]  grad at vision logits */
} MtrssmMrssmBwdIO;

/* The reverse-time scan on the same four-CU clusters (csrc/mrssm_cluster.hip: mrssm_bwd_cluster_kernel): weights as in
 * MtrssmMrssmClusterWeights (bias fields unused).  io->sv_h2 is not read and io->d_h2 is NOT written (the fused input path
 * has no h2 inside the scan): form d_h2 = d_gi . W_ih afterwards (one GEMM).  Workspace / status word / co-residency as for
 * the forward cluster call; S <= 32. */
int64_t mtrssm_mrssm_cluster_bwd_workspace_bytes(const MtrssmMrssmDims* dims);
int mtrssm_mrssm_rollout_bwd_cluster(const MtrssmMrssmDims* dims, const MtrssmMrssmClusterWeights* weights, const MtrssmMrssmBwdIO* io,
                                     void* workspace, int64_t workspace_bytes, void* stream);

/* The same scans for LARGE deterministic / hidden sizes (D or H >= 256; BASELINE configs[4]: D = H = 1024, S = 128), ALL compute
 * units of the chip on one tile of 32 batch rows (csrc/mrssm_wide.hip): every matrix product of a timestep is cut into
 * 16-column output tiles, each streamed by one CU per step as bf16 pieces in MFMA operand order (packed by the call, once per
 * launch), the batch rows are the MFMA N dimension (v_mfma_f32_16x16x32_bf16), consecutive layers meet through exchange vectors
 * in L2 and a grid-wide barrier (4 per forward timestep, 5 per backward one).  pieces = 3: every fp32 operand as three bf16
 * pieces, six products per k-block (exact to 2^-24: fp32-grade; the default of the Python layer); 2: two pieces, three products
 * (16 significant bits).  Weights / fused input path / sv_h2 / d_h2 exactly as for the cluster calls above.
 * workspace: caller-owned, 256-byte aligned, mtrssm_mrssm_wide_workspace_bytes() / _bwd_workspace_bytes() bytes.  Its first
 * int32 is a STICKY status word: the caller zeroes it once when allocating; a launch whose barrier gave up stores a non-zero
 * code there (results invalid) and no launch ever clears it (mtrssm_adamw_apply can be given the word and then skips the update).
 * The grid is one workgroup per CU, all of which must be resident: a GPU this process has to itself.
 * mtrssm_mrssm_wide_supported() = 1 when dims and device fit (D, H multiples of 16, K <= 64, D/16 + H/16 <= CU count). */
int mtrssm_mrssm_wide_supported(const MtrssmMrssmDims* dims, int32_t pieces);
int64_t mtrssm_mrssm_wide_workspace_bytes(const MtrssmMrssmDims* dims, int32_t pieces);
int64_t mtrssm_mrssm_wide_bwd_workspace_bytes(const MtrssmMrssmDims* dims, int32_t pieces);
int mtrssm_mrssm_rollout_fwd_wide(const MtrssmMrssmDims* dims, const MtrssmMrssmClusterWeights* weights, const MtrssmMrssmFwdIO* io,
                                  int32_t pieces, void* workspace, int64_t workspace_bytes, void* stream);
int mtrssm_mrssm_rollout_bwd_wide(const MtrssmMrssmDims* dims, const MtrssmMrssmClusterWeights* weights, const MtrssmMrssmBwdIO* io,
                                  int32_t pieces, void* workspace, int64_t workspace_bytes, void* stream);

/* Reverse-time scan (BPTT).  Weight gradients are NOT accumulated inside the serial loop: the
 * kernel emits the per-step pre-activation gradients above and the caller forms every dW as one
 * batched [out x B*T] . [B*T x in] library GEMM (rocBLAS), which is where MFMA belongs. */
int mtrssm_mrssm_rollout_bwd(const MtrssmMrssmDims* dims, const MtrssmMrssmBwdWeights* w,
                             const MtrssmMrssmBwdIO* io, void* stream);

/* ------------------------------------------------------------------------------------------
 * MoPoE-MMTRSSM scan (two-timescale MTState variant).  Replaces the T loop of
 * MoPoE_MMTRSSM.rollout_representation (mmtrssm/mopoe_mmtrssm/core.py:405-490): MTRNN (:59-60),
 * _compute_lower_prior (:263-287), the two posterior heads (:241-261), inline MoPoE (:436-453),
 * _compute_higher_prior_posterior (:289-319), MTState sampling (mmtrssm/state.py:47-49) and the
 * per-step kl / kl_h terms of shared_step (:586-600).  post == 0: rollout_transition (:496-544).
 * Hoisted by the caller:  xl[b,t,:] = Wx_l[:, :A] a[b,t] + bx_l + bd_l ; pa / pv as above.
 * ------------------------------------------------------------------------------------------ */
typedef struct MtrssmMmtrssmDims {
  int32_t B, T;
  int32_t LD, HD;        /* lower / higher deterministic sizes */
  int32_t H;             /* num_cells of l_prior, h_prior, h_posterior and the posterior heads */
  int32_t KL, CL;        /* lower categoricals, classes  (ls = KL*CL) */
  int32_t KH, CH;        /* higher categoricals, classes (hs = KH*CH) */
  int32_t act;
  int32_t post;
  float tau_l, tau_h;    /* MTRNN time constants (> 1) */
  float keep_l, keep_h;  /* (float)(1.0 - 1.0/tau), computed in double by the caller (core.py:59) */
  float kl_w_post, kl_w_prior;
  int32_t rows_per_block;
  int32_t threads;
} MtrssmMmtrssmDims;

typedef struct MtrssmMmtrssmFwdWeights {
  const float* wxl_s_t;  /* [LS+HS][LD]  l_rnn._input2h.weight[:, A:]^T */
  const float* wdl_t;    /* [LD][LD]     l_rnn._d2h.weight^T (bias folded into xl) */
  const float* wxh_t;    /* [HS][HD]     h_rnn._input2h.weight^T */
  const float* wdh_t;    /* [HD][HD]     h_rnn._d2h.weight^T */
  const float* bh;       /* [HD]         h_rnn._input2h.bias + h_rnn._d2h.bias */
  const float* wl1_t;    /* [LD][4H]     [l_prior.0 ; audio post.0[:, :LD] ; vision post.0[:, :LD] ; h_posterior.0[:, :LD]]^T
                                         (prior-only: [LD][H]) */
  const float* bl1;      /* [H]          l_prior.0.bias */
  const float* wh1_t;    /* [HD][2H]     [h_prior.0 ; h_posterior.0[:, LD:]]^T (prior-only: [HD][H]) */
  const float* bh1;      /* [2H]         [h_prior.0.bias ; h_posterior.0.bias] */
  const float* wlp2;     /* [LS][H]  l_prior.2.weight (PyTorch layout) */
  const float* blp2;     /* [LS] */
  const float* wa2;      /* [LS][H]  audio rnn_to_post_projector.2.weight */
  const float* ba2;
  const float* wv2;      /* [LS][H] */
  const float* bv2;
  const float* whp2;     /* [HS][H]  h_prior.2.weight */
  const float* bhp2;
  const float* whq2;     /* [HS][H]  h_posterior.2.weight */
  const float* bhq2;
} MtrssmMmtrssmFwdWeights;

typedef struct MtrssmMmtrssmFwdIO {
  const float* xl;        /* [B,T,LD] */
  const float* pa;        /* [B,T,H] (post only) */
  const float* pv;        /* [B,T,H] (post only) */
  const float* deter_l0;  /* [B,LD] */
  const float* deter_h0;  /* [B,HD] */
  const float* hidden_l0; /* [B,LD] */
  const float* hidden_h0; /* [B,HD] */
  const float* stoch_l0;  /* [B,LS] */
  const float* stoch_h0;  /* [B,HS] */
  const float* u_post_l;  /* [B,T,KL] */
  const float* u_post_h;  /* [B,T,KH] */
  const float* u_prior_l; /* [B,T,KL] (NULL allowed when post=1) */
  const float* u_prior_h; /* [B,T,KH] */
  float* deter_l;         /* [B,T,LD] */
  float* deter_h;         /* [B,T,HD] */
  float* hidden_l;        /* [B,T,LD] */
  float* hidden_h;        /* [B,T,HD] */
  float* prior_logits_l;  /* [B,T,LS] */
  float* prior_logits_h;  /* [B,T,HS] */
  float* prior_stoch_l;   /* [B,T,LS] (NULL allowed when post=1) */
  float* prior_stoch_h;   /* [B,T,HS] */
  float* post_logits_l;   /* [B,T,LS] mixed log-probs */
  float* post_logits_h;   /* [B,T,HS] raw h_posterior logits */
  float* post_stoch_l;    /* [B,T,LS] */
  float* post_stoch_h;    /* [B,T,HS] */
  float* kl_l;            /* [B,T] */
  float* kl_h;            /* [B,T] */
  float* sv_l1;           /* [B,T,4H] act of l_prior.0 / audio post.0 / vision post.0 / h_posterior.0 */
  float* sv_h1;           /* [B,T,H]  act of h_prior layer 0 */
  float* sv_la;           /* [B,T,LS] */
  float* sv_lv;           /* [B,T,LS] */
} MtrssmMmtrssmFwdIO;

int mtrssm_mmtrssm_rollout_fwd(const MtrssmMmtrssmDims* dims, const MtrssmMmtrssmFwdWeights* w,
                               const MtrssmMmtrssmFwdIO* io, void* stream);

typedef struct MtrssmMmtrssmBwdWeights {
  const float* wxl_s_t; /* [LS+HS][LD] (the forward's buffer: narrow output) */
  const float* wdl;    /* [LD][LD] */
  const float* wxh_t;  /* [HS][HD]    (the forward's buffer: narrow output) */
  const float* wdh;    /* [HD][HD] */
  const float* wl1;    /* [4H][LD] */
  const float* wh1;    /* [2H][HD] */
  const float* wlp2;   /* [LS][H] */
  const float* wa2;    /* [LS][H] */
  const float* wv2;    /* [LS][H] */
  const float* whp2;   /* [HS][H] */
  const float* whq2;   /* [HS][H] */
} MtrssmMmtrssmBwdWeights;

typedef struct MtrssmMmtrssmBwdIO {
  const float* deter_l0;
  const float* deter_h0;
  const float* deter_l;
  const float* deter_h;
  const float* prior_logits_l;
  const float* prior_logits_h;
  const float* post_logits_l;
  const float* post_logits_h;
  const float* sv_l1;
  const float* sv_h1;
  const float* sv_la;
  const float* sv_lv;
  /* incoming gradients (NULL = zero) */
  const float* g_deter_l;
  const float* g_deter_h;
  const float* g_hidden_l;
  const float* g_hidden_h;
  const float* g_post_stoch_l;
  const float* g_post_stoch_h;
  const float* g_prior_stoch_l;
  const float* g_prior_stoch_h;
  const float* g_post_logits_l;
  const float* g_post_logits_h;
  const float* g_prior_logits_l;
  const float* g_prior_logits_h;
  const float* g_kl_l;
  const float* g_kl_h;
  /* outgoing gradients */
  float* g_deter_l0;
  float* g_deter_h0;
  float* g_hidden_l0;
  float* g_hidden_h0;
  float* g_stoch_l0;
  float* g_stoch_h0;
  float* d_ul;    /* [B,T,LD] grad at the lower MTRNN pre-activation sum (already / tau_l) = d xl */
  float* d_uh;    /* [B,T,HD] grad at the higher MTRNN pre-activation sum (already / tau_h) */
  float* d_zl1;   /* [B,T,4H] pre-activation grads: l_prior.0, audio.0 (= d pa), vision.0 (= d pv), h_posterior.0 */
  float* d_zh1;   /* [B,T,H]  pre-activation grad of h_prior.0 */
  float* d_lpl;   /* [B,T,LS] grad at lower prior logits */
  float* d_la;    /* [B,T,LS] */
  float* d_lv;    /* [B,T,LS] */
  float* d_lph;   /* [B,T,HS] grad at higher prior logits */
  float* d_lqh;   /* [B,T,HS] grad at higher posterior logits */
} MtrssmMmtrssmBwdIO;

int mtrssm_mmtrssm_rollout_bwd(const MtrssmMmtrssmDims* dims, const MtrssmMmtrssmBwdWeights* w,
                               const MtrssmMmtrssmBwdIO* io, void* stream);

/* The same two scans on ALL compute units (csrc/mmtrssm_wide.hip; regime of mtrssm_mrssm_rollout_*_wide above): up to 64 batch rows
 * per pass as the MFMA N dimension, every product of a timestep cut into 16-column output tiles streamed by one CU each, four grid
 * barriers per timestep in either direction (one-CU form at ld = hd = H = 200: 38 / 46 us per timestep).  Weights, io and results
 * exactly as for mtrssm_mmtrssm_rollout_fwd / _bwd (posterior rollout, post = 1); pieces / workspace / sticky status word /
 * co-residency as for the MRSSM wide calls.  mtrssm_mmtrssm_wide_supported() = 1 when dims and device fit (LD, HD, H multiples of 4,
 * LD or HD >= 128, KL, KH <= 64, (LD + HD + LS + HS) / 16 column tiles <= CU count). */
int mtrssm_mmtrssm_wide_supported(const MtrssmMmtrssmDims* dims, int32_t pieces);
int64_t mtrssm_mmtrssm_wide_workspace_bytes(const MtrssmMmtrssmDims* dims, int32_t pieces);
int64_t mtrssm_mmtrssm_wide_bwd_workspace_bytes(const MtrssmMmtrssmDims* dims, int32_t pieces);
int mtrssm_mmtrssm_rollout_fwd_wide(const MtrssmMmtrssmDims* dims, const MtrssmMmtrssmFwdWeights* w, const MtrssmMmtrssmFwdIO* io,
                                    int32_t pieces, void* workspace, int64_t workspace_bytes, void* stream);
int mtrssm_mmtrssm_rollout_bwd_wide(const MtrssmMmtrssmDims* dims, const MtrssmMmtrssmBwdWeights* w, const MtrssmMmtrssmBwdIO* io,
                                    int32_t pieces, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Conv encoder / decoder kernels (fp32 MFMA implicit GEMM, NCHW).  Replace the layers of the
 * `cnn.Encoder` / `cnn.Decoder` stacks the reference YAML instantiates
 * (mrssm/mopoe_mrssm/configs/default.yaml:31-92; called at mrssm core.py:179-180,215-216,272-273).
 *
 * mtrssm_conv_gather_gemm computes, for every frame n and output channel co,
 *   out[n, co, oy*OS+QY, ox*OS+QX] = (bias[co] + sum_{ty<KH, tx<KW} sum_{c<C+C2} wp[co][ty*KW+tx][c] * pre(S[n,c,sy,sx])) * egrad + add
 *   sy = oy*SS + ty*TS + OFFY, sx = ox*SS + tx*TS + OFFX (zero outside [0,Hs)x[0,Ws)),  oy < Hq, ox < Wq
 * where S = src for c < C and the frame-independent src2 (coordinate channels) for C <= c < C+C2,
 * pre() = act() if pre_act, egrad = act'(actgrad_in[same element as out]) when actgrad_in != NULL, and
 * add = add_in[same element as out] when add_in != NULL (residual skip connection forward; its gradient backward).
 *   Conv2d forward (k,s,p):            KH=KW=k, SS=s, TS=+1, OFF=-p, OS=1, Hq=Ho
 *   Conv2d backward-data / ConvTranspose2d forward: one call per output parity class (qy,qx) in [0,s)^2 with
 *     the taps ky = ky0 + s*ty (ky0 = (qy+p) mod s): SS=1, TS=-1, OFFY=(qy+p-ky0)/s, OS=s, QY=qy.
 * wp is the packed weight matrix [CoutPad][KH*KW][Cpad], zero padded (Cpad % 16 == 0, CoutPad % 32 == 0,
 * CoutPad % 64 == 0 when Cout > 32), 16-byte aligned; wq: its bf16 pieces (see mtrssm_pack_conv_weight), 16-byte aligned, or NULL.
 *
 * mtrssm_conv_weight_grad accumulates (the caller zeroes dwp)
 *   dwp[co][ty*KW+tx][c] += sum_{n, y<Hq, x<Wq} preA(a[n,co,y,x]) * pre(S[n,c,y*SS+ty*TS+OFFY,x*SS+tx*TS+OFFX])
 * with a of shape [N, Cout, Hq, Wq] (geometry must have OS=1, QY=QX=0).  When dbias != NULL (allowed only with
 * pre_act_a == 0) it also accumulates the bias gradient dbias[co] += sum_{n,y,x} a[n,co,y,x] in the same pass.
 * Kernel selection (mfma_split > 0; csrc/conv_split.h): 1x1 layers and 3x3 / stride 1 / pad 1 layers on 8- or 4-pixel-wide
 * planes read both operands straight from HBM into the MFMA register layout (k = pixels is contiguous in NCHW); layers with
 * <= 32 output channels and <= 32 (tap, channel) columns on power-of-two-wide planes run one MFMA tile with a per-lane
 * gathered operand; everything else stages 64-pixel groups through LDS (persistent workgroups); mfma_split = 0 and odd
 * geometries use the fp32 kernels of csrc/conv.hip.  Those accumulate with fp32 atomics: results are equal up to the
 * arrival order of the partial sums.
 * The layer shapes of the reference's default encoders / decoders (mfma_split 1 or 2, pre_act_a as those layers set it: the
 * 3x3 and 1x1 layers of the residual stacks on 64-pixel planes, the three 3x3 / stride-2 convolutions, the three k = 4 /
 * stride-2 transposed convolutions; csrc/conv_wgrad_resident.h) run staged kernels instead: every workgroup stores ONE
 * partial set of its tiles and bias sums into `workspace` -- caller-owned device memory, 256-byte aligned, at least
 * mtrssm_conv_weight_grad_workspace_bytes() bytes, free for reuse as soon as the call's kernels have run (calls on one stream may
 * share one buffer) -- and a second kernel on the same stream adds the sets to dwp / dbias by plain read-modify-write: bitwise
 * reproducible from run to run.  The caller must therefore not accumulate into the same dwp / dbias from ANOTHER stream at the
 * same time (the atomics of the other kernels allow that); calls on one stream are ordered.  With workspace NULL or too small
 * (and with MTRSSM_WGRAD_PARTIALS=0) the same kernels add their tiles by atomics.  The library itself allocates nothing.
 * ------------------------------------------------------------------------------------------ */
typedef struct MtrssmConvGeom {
  int32_t N;                    /* frames (B*T) */
  int32_t C, Hs, Ws;            /* gathered tensor [N, C, Hs, Ws] */
  int32_t C2;                   /* extra frame-independent channels src2 [C2, Hs, Ws] appended after C */
  int32_t Cpad;                 /* channel extent of wp */
  int32_t KH, KW;
  int32_t SS, TS, OFFY, OFFX;
  int32_t Hq, Wq;               /* output sub-grid enumerated by this call */
  int32_t OS, QY, QX;
  int32_t Ho, Wo;               /* full output plane */
  int32_t Cout, CoutPad;
  int32_t pre_act;              /* apply act() to gathered values */
  int32_t act;                  /* MTRSSM_ACT_* */
  int32_t mfma_split;           /* MFMA operand format of the patch-staged kernels: 0 = fp32 (v_mfma_f32_32x32x2_f32, exact);
                                   3 = three bf16 pieces per operand, six v_mfma_f32_32x32x16_bf16 products (fp32-grade, ~2^-24);
                                   1 = plain bf16 operands.  Accumulation is fp32.  Layers the split kernels do not cover
                                   (thin layers, odd geometries) run the fp32 kernels whatever this says. */
} MtrssmConvGeom;

int mtrssm_conv_gather_gemm(const MtrssmConvGeom* g, const float* src, const float* src2, const float* wp, const uint16_t* wq,
                            const float* bias, const float* actgrad_in, const float* add_in, float* out, void* stream);
/* Two independent gather problems (the same layer of the audio and of the vision stack: same channels and taps, different
 * planes) in ONE launch when both map to the same split-bf16 kernel, else two launches: a layer is ~2 rounds of workgroup tiles
 * with a nearly empty last one, two of them back to back in one grid waste one round instead of two.  Arguments as
 * mtrssm_conv_gather_gemm, once per problem. */
int mtrssm_conv_gather_gemm_pair(const MtrssmConvGeom* ga, const float* srca, const float* src2a, const float* wpa, const uint16_t* wqa,
                                 const float* biasa, const float* actgrada, const float* adda, float* outa,
                                 const MtrssmConvGeom* gb, const float* srcb, const float* src2b, const float* wpb, const uint16_t* wqb,
                                 const float* biasb, const float* actgradb, const float* addb, float* outb, void* stream);
/* 1 if mtrssm_conv_gather_gemm_pair would run the two problems as one grid (same split-bf16 kernel for both; has_wq: both
 * have bf16 pieces), 0 if it would fall back to two launches -- a host-side query, nothing is launched. */
int mtrssm_conv_gather_pair_merges(const MtrssmConvGeom* ga, const MtrssmConvGeom* gb, int32_t has_wq);
/* Forward of a whole residual block of the encoders' / decoders' stacks (the builder's `cnn` stand-in, oracle/ref_cnn.py:57-58:
 * `x + conv1(act(conv3(act(x))))`; called through cnn.Encoder / cnn.Decoder at mrssm core.py:179-180,215-216) in ONE launch:
 *   h[n, m, p] = b3[m] + sum_{tap, c} wq3[m][tap][c] * act(x)[n, c, p + tap]          (stored: the backward pass reads it)
 *   y[n, c, p] = x[n, c, p] + b1[c] + sum_m w1[c][m] * act(h[n, m, p])
 * g3 is the 3x3's geometry exactly as mtrssm_conv_gather_gemm takes it (pre_act = 1, mfma_split = 2), wq3 its packed bf16 pieces
 * (mtrssm_pack_conv_weight(s)), w1 [C][Cout] and b1 [C] the 1x1 module's own fp32 parameters (contiguous); two bf16 pieces per
 * operand and fp32 accumulation in both products, as the two separate launches compute.  A second block on another tensor
 * (gb != NULL: the other modality) rides in the same grid.  _supported: 1 when the shape has a fused instance (64 channels,
 * 128 intermediate channels -- or 64 with an even frame count --, 64-pixel planes), else 0 -- a host-side query; mtrssm_residual_block_fwd on an unsupported shape
 * returns MTRSSM_EINVAL. */
int mtrssm_residual_block_fwd_supported(const MtrssmConvGeom* g3);
int mtrssm_residual_block_fwd(const MtrssmConvGeom* ga, const float* xa, const uint16_t* wq3a, const float* b3a, const float* w1a,
                              const float* b1a, float* ha, float* ya, const MtrssmConvGeom* gb, const float* xb, const uint16_t* wq3b,
                              const float* b3b, const float* w1b, const float* b1b, float* hb, float* yb, void* stream);
/* Packs a conv weight view w[O][I][KH][KW] (element strides so, si, sh, sw: any permuted / strided view of the module's
 * parameter, e.g. the per-parity-class tap subset of a ConvTranspose2d weight) into the kernels' layout:
 *   wp fp32 [OPad][KH*KW][IPad], zero padded;
 *   wq (pieces = mfma_split > 0) bf16 bit patterns [pieces][OPad][KH*KW][IPad]: piece s of each wp element, wp = sum_s wq_s
 *   up to 2^-(8 pieces) relative.  mtrssm_conv_gather_gemm reads wq when g->mfma_split > 0 and wq != NULL, wp otherwise
 *   (thin / odd-geometry layers always read wp). */
int mtrssm_pack_conv_weight(const float* w, int32_t O, int32_t I, int32_t KH, int32_t KW, int64_t so, int64_t si, int64_t sh,
                            int64_t sw, int32_t OPad, int32_t IPad, int32_t pieces, float* wp, uint16_t* wq, void* stream);
/* The same for `count` weights in one launch (the train step packs every conv weight once, in both the forward and the
 * backward-data layout, right after the optimizer changed them).  `table` is DEVICE memory: MTRSSM_PACK_DESC_WORDS int64 words
 * per weight = { w, wp, wq (addresses), O, I, KH, KW, so, si, sh, sw, OPad, IPad, pieces, VH, VW } with the meanings above;
 * VH, VW > 0: the view holds only the leading VH x VW taps of the KH x KW tap grid, the rest are packed as zeros (the
 * parity-class sub-kernels of a 3 x 3 kernel run as a zero-padded 4 x 4 transposed convolution); 0, 0: every tap.
 * blocks_per_weight workgroups of 256 threads stride over each weight. */
#define MTRSSM_PACK_DESC_WORDS 16
int mtrssm_pack_conv_weights(const int64_t* table, int32_t count, int32_t blocks_per_weight, void* stream);
int mtrssm_conv_weight_grad(const MtrssmConvGeom* g, const float* a, const float* src, const float* src2,
                            int32_t pre_act_a, float* dwp, float* dbias, void* workspace, int64_t workspace_bytes, void* stream);
/* mtrssm_conv_weight_grad with the sum of its partial tile sets DEFERRED: the main kernel is launched, the small launch that
 * adds the sets into dwp / dbias is recorded for `stream` instead (38 such launches of 6-15 us per train step otherwise).
 * mtrssm_conv_weight_grad_reduce(stream) then runs every recorded sum in one launch.  Contract: the workspace of a deferred
 * call stays untouched -- no other call may be given the same bytes -- and dwp / dbias are not read until the reduce; a
 * second deferred call on the same dwp or workspace first flushes the recorded ones (order is kept).  Kernels without
 * partial sets (general geometry, atomics mode) behave exactly like mtrssm_conv_weight_grad. */
int mtrssm_conv_weight_grad_deferred(const MtrssmConvGeom* g, const float* a, const float* src, const float* src2,
                                     int32_t pre_act_a, float* dwp, float* dbias, void* workspace, int64_t workspace_bytes, void* stream);
int mtrssm_conv_weight_grad_reduce(void* stream);
/* Bytes of workspace the kernel chosen for this geometry wants for its partial tile sets (0: none; -1: invalid geometry; up to
 * ~38 MB for the reference's layer shapes at B*T = 3200 frames).  A host-side query, nothing is launched. */
int64_t mtrssm_conv_weight_grad_workspace_bytes(const MtrssmConvGeom* g, int32_t pre_act_a);
/* End of backward: add every packed conv weight gradient [OPad][taps][IPad] of the step into its parameter-layout target
 * [O][I][KH][KW] (a view of the flat gradient buffer) and clear the packed buffers, in ONE launch.  table (device memory):
 * `count` rows of 8 int64 = { packed pointer, target pointer, O, I, taps = KH*KW, IPad, 0, 0 }. */
int mtrssm_unpack_conv_grads(const int64_t* table, int32_t count, int32_t blocks_per_entry, void* stream);
/* Decoder ConvTranspose2d (k = 4, s = 2, p = 1; default.yaml:61-92) forward, ALL FOUR output parity classes in one pass over the
 * source (mtrssm_conv_gather_gemm takes one launch per class: each re-reads the source).  g4[q], wq4[q]: geometry and packed
 * sub-kernel (mtrssm_pack_conv_weight(s), two bf16 pieces) of class q = 2 * QY + QX exactly as for mtrssm_conv_gather_gemm
 * (KH = KW = 2, TS = -1, SS = 1, OS = 2, same source / output).  Optionally a second tensor (gb4 != NULL: the other modality) in
 * the same launch.  actgrad (may be NULL): out *= act'(actgrad[same index]) -- the backward-data of a Conv2d(k = 3, s = 2, p = 1)
 * is this transposed conv with the kernel zero-padded to 4 x 4 (the encoders' third conv).
 * mtrssm_convt_quad_supported: 1 / 2 = the instantiated forward shapes (64 -> 32 on 64-pixel planes, 32 -> 16 on 256-pixel planes),
 * 3 = 32 -> 16 on 64-pixel planes WITH actgrad, 4 = 16 -> 8 on 256-pixel planes WITH actgrad; 0 = use the general path.
 * Kernel selection inside: shapes 1 and 3 run one wave per parity class (csrc/conv_resident.h: convt_quad_resident_kernel),
 * shapes 2 and 4 the classes as rows of the MFMA tile (csrc/conv_s2_band.h: convt4s2_rows_kernel); same arithmetic (two bf16
 * pieces per operand, fp32 accumulation), another summation order. */
int mtrssm_convt_quad_supported(const MtrssmConvGeom* g4);
int mtrssm_convt_quad(const MtrssmConvGeom* ga4, const float* srca, const uint16_t* const* wqa4, const float* biasa, const float* actgrada,
                      float* outa, const MtrssmConvGeom* gb4, const float* srcb, const uint16_t* const* wqb4, const float* biasb,
                      const float* actgradb, float* outb, void* stream);
/* Last decoder layer (default.yaml:70-74, channels [.., 1]): out[N, Cout<=2, 2Hs, 2Ws] = bias + ConvTranspose2d_{k=4,s=2,p=1}(pre(src[N,C,Hs,Ws]))
 * with w in the ConvTranspose2d layout [C][Cout][4][4].  All four output parity classes in one pass, one activation per
 * source element. */
int mtrssm_convt_k4s2_thin(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t Cout, const float* src, const float* w,
                           const float* bias, int32_t pre_act, int32_t act, float* out, void* stream);
/* The same layer on the MFMA for the reference's shapes (16 input channels, Cout <= 2, frames of 1024 positions: 64 x 16 or
 * 32 x 32; two bf16 pieces per operand, fp32 accumulation = the bf16x2 conv mode): the four output parity classes of an input
 * position are rows of an MFMA tile, a frame is staged once (csrc/conv_s2_band.h: convt4s2_band_kernel).  _supported: 1 / 0, a
 * host-side query. */
int mtrssm_convt_k4s2_band_supported(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t Cout);
int mtrssm_convt_k4s2_band(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t Cout, const float* src, const float* w,
                           const float* bias, int32_t pre_act, int32_t act, float* out, void* stream);
/* Conv2d backward-data / ConvTranspose2d forward with <= 8 output channels, all output parity classes in ONE pass (the general
 * path, mtrssm_conv_gather_gemm, takes one launch per parity class of a strided layer; replaces the backward-data of the
 * encoders' second conv, cnn.Encoder at mrssm core.py:179-180):
 *   out[n, c, iy, ix] = (bias[c] + sum_{o, ky, kx} w[o][c][ky][kx] * pre(y)[n, o, (iy + pad - ky) / stride, (ix + pad - kx) / stride])
 *                       * act'(actgrad_in[n, c, iy, ix]) + add_in[n, c, iy, ix]
 * over the taps whose (iy + pad - ky), (ix + pad - kx) are non-negative multiples of the stride inside the source plane.
 * y [N, O, Hs, Ws], w [O][Cout][KH][KW] contiguous (the Conv2d weight), out [N, Cout, Ho, Wo]; KH * KW * O <= 256; stride 2, even Ho and Wo. */
int mtrssm_conv_tgather_thin(int32_t N, int32_t O, int32_t Hs, int32_t Ws, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                             int32_t Ho, int32_t Wo, const float* y, const float* w, const float* bias, int32_t pre_act, int32_t act,
                             const float* actgrad_in, const float* add_in, float* out, void* stream);
/* out[c] += sum_{n, i<HW} x[n, c, i]   (bias gradients; the caller zeroes out) */
int mtrssm_channel_sum(const float* x, int32_t N, int32_t C, int32_t HW, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Episode feed.  Replaces EpisodeDataset.__getitem__ + the default collate of the 6-tuple
 * StackDataset (models/dataset.py:84-112, models/mrssm/dataset.py:155-183) for the YAML's transform chains
 * TakeFirstN(n) [+ GaussianNoise(std)] (models/transform.py:31-72, default.yaml:176-220): from the HBM-resident store
 * [n_episodes, Tfull, E] (fp32, already preprocessed) it writes, for the B episodes idx[0..B) (int64, device memory),
 *   target[b, t, :] = store[idx[b], t, :]                t < T <= Tfull
 *   input [b, t, :] = target[b, t, :] + noise[b, t, :] * std     (mul then add, each rounded: bitwise torch's expression)
 * noise (caller-drawn standard normals [B, T, E]) may be NULL (input = target); input or target may be NULL.
 * E % 4 == 0; every buffer 16-byte aligned.  idx values are NOT range-checked on the device.
 * ------------------------------------------------------------------------------------------ */
int mtrssm_episode_gather(const float* store, const int64_t* idx, const float* noise, int64_t n_episodes, int64_t B, int64_t T,
                          int64_t Tfull, int64_t E, float std_, float* input, float* target, void* stream);

/* The scalar end of shared_step (core.py:187-221; mmtrssm core.py:563-606) in one launch each way:
 *   recon = nll_a + nll_v;  kl_j = c_j * mean_i kl_j[i], i < n (kl1 may be NULL);  loss = recon + kl_0 + kl_1, written to four scalars (o_k1 may be NULL).
 * bwd: scalar gradients of those four (each may be NULL = 0) -> g_nll_a = g_nll_v = g_recon + g_loss, g_kl_j[i] = (g_kl_j + g_loss) c_j / n. */
int mtrssm_elbo_combine_fwd(const float* nll_a, const float* nll_v, const float* kl0, const float* kl1, int64_t n, float c0, float c1,
                            float* o_recon, float* o_k0, float* o_k1, float* o_loss, void* stream);
int mtrssm_elbo_combine_bwd(const float* g_recon, const float* g_k0, const float* g_k1, const float* g_loss, int64_t n, float c0, float c1,
                            float* g_nll_a, float* g_nll_v, float* g_kl0, float* g_kl1, void* stream);
/* Categorical head of the initial state (core.py:121-135, mmtrssm core.py:321-362): `logits` [rows][K * C] flat (K categoricals
 * of C classes, softmax over classes), `u` [rows][K] uniforms -> logp, probs [rows][K][C] and the inverse-CDF one-hot sample
 * onehot [rows][K * C] (index = #{c <= C - 2 : cumulative probability <= u}, the cumulative sum a left fold).  The straight-through
 * sample of the reference is onehot + probs - probs.detach(): its gradient arrives as g_probs of the backward call,
 *   d_logits[c] = p_c (g_probs[c] - sum_j p_j g_probs[j]) + g_logp[c] - p_c sum_j g_logp[j]   (either gradient may be NULL). */
int mtrssm_categorical_sample_fwd(const float* logits, const float* u, int64_t rows, int32_t K, int32_t C, float* logp, float* probs,
                                  float* onehot, void* stream);
int mtrssm_categorical_sample_bwd(const float* probs, const float* g_probs, const float* g_logp, int64_t rows, int32_t K, int32_t C,
                                  float* d_logits, void* stream);
/* ------------------------------------------------------------------------------------------
 * Gaussian NLL with unit scale, fused reduction.  Replaces objective.likelihood
 * (objective.py:7-23) as used by compute_reconstruction_loss (mrssm/mopoe_mrssm/core.py:279-308):
 *   nll = mean_n sum_e [ 0.5 (target - pred)^2 + 0.5 log(2 pi) ],  n = B*T frames, e = C*H*W.
 * `act` (MTRSSM_ACT_IDENTITY or MTRSSM_ACT_TANH) is the decoder's out_activation (default.yaml: Tanh), applied to `pred`
 * while it is read: pred then holds the decoder's RAW last-layer output and the activated reconstruction is never written.
 * fwd zeroes `out` and accumulates the scalar there; bwd writes d pred = g_out * (act(pred) - target) * act'(pred) / n.
 * ------------------------------------------------------------------------------------------ */
int mtrssm_gaussian_nll_fwd(const float* pred, const float* target, int64_t frames, int64_t event, int32_t act,
                            float* out, void* stream);
int mtrssm_gaussian_nll_bwd(const float* pred, const float* target, const float* g_out,
                            int64_t frames, int64_t event, int32_t act, float* g_pred, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused AdamW over one flat fp32 parameter buffer, with global-norm gradient clipping
 * (yaml: torch.optim.AdamW lr 1e-3; trainer.gradient_clip_val 10 -- default.yaml:103-107,119).
 * sumsq: device scalar holding sum(grad^2) (mtrssm_sumsq writes it); clip <= 0 disables clipping.
 * ------------------------------------------------------------------------------------------ */
int mtrssm_sumsq(const float* x, int64_t n, float* out, void* stream);
int mtrssm_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const float* sumsq, float clip_norm, float grad_scale, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int32_t step, void* stream);
/* The same step with its scalars in device memory, so that it can sit inside a captured hipGraph (FlatAdamW's path):
 *   state[4] (device floats): [0] learning rate (the host writes it when a scheduler changes it), [1] steps taken,
 *   [2] 1 - beta1^step, [3] sqrt(1 - beta2^step).  mtrssm_adamw_prepare = zero `sumsq`, sum(grad^2) into it, step += 1 and the
 *   two bias corrections; mtrssm_adamw_apply = clip + AdamW reading `state`.
 *   active (n bytes, may be NULL = all): 0 marks elements of parameters that never received a gradient; they are left
 *   untouched (no decay, no moments) as torch.optim.AdamW does for `.grad is None` -- MMTRSSM's l_posterior and dummy
 *   transition (mmtrssm/mopoe_mmtrssm/core.py:143-151,188). */
/* Zero `bytes` (multiple of 4, 4-byte aligned) of device memory with a plain kernel: what FlatParameters.zero_grad() uses, so that a
 * captured train step records no memset node (a captured multi-megabyte hipMemsetAsync left foreign bytes at the buffer's head
 * on replay under ROCm 7.2). */
int mtrssm_clear(void* p, int64_t bytes, void* stream);
/* status (device int32, may be NULL): the sticky status word of the step's cooperative scan kernels (first word of the workspace of
 * mtrssm_mrssm_rollout_*_cluster / _wide).  Non-zero = an exchange of the step gave up, its gradients are invalid: prepare then
 * does not count the step and apply leaves parameters and moments untouched (the host raises when it next polls the word). */
int mtrssm_adamw_prepare(const float* grad, int64_t n, float* sumsq, float* state, const int32_t* status, float beta1, float beta2,
                         void* stream);
int mtrssm_adamw_apply(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* active, int64_t n,
                       const float* sumsq, const float* state, const int32_t* status, float clip_norm, float grad_scale, float beta1,
                       float beta2, float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense fp32 GEMM on the fp32 MFMA with the elementwise neighbours fused in.  Replaces the nn.Linear / MLP calls around
 * the recurrence and their autograd (torchrl MLP at networks.py:57-64,130-145; cnn's Linear layers; init_proj,
 * core.py:132-133) and the weight gradients of the scan:
 *   C[i][j] (+)= act_out( bias[j] + sum_r actA(A'(i, r)) * actB(B'(j, r)) ) * act_z'(zgrad[i][j])
 *   A'(i, r) = a_rmajor ? A[r * lda + i] : A[i * lda + r];   B'(j, r) = b_rmajor ? B[r * ldb + j] : B[j * ldb + r]
 *   forward Y = act(X) W^T + b: A = X, B = W;  data gradient dX = (dY W) * act'(X): A = dY, B = W with b_rmajor;
 *   weight gradient dW += dY^T act(X), db += column sums of dY: A = dY, B = X, both r-major, accumulate, colsum = db.
 * bias, zgrad, colsum may be NULL.  split_r: 0 = choose (splits only when accumulate is set and no output epilogue), else
 * the number of reduction slices (> 1 meets in C by fp32 atomics: accumulate = 1, or a dense C that the call zeroes
 * first).  Numerics: each output element is a k-ordered fp32 fma chain per slice (v_mfma_f32_32x32x2_f32).  Rows may be strided views (lda / ldb / ldc / ldz).
 * ------------------------------------------------------------------------------------------ */
typedef struct MtrssmGemm {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* zgrad;
  float* colsum;
  int32_t M, N, R, lda, ldb, ldc, ldz;
  int32_t a_rmajor, b_rmajor, act_a, act_b, act_out, act_z, accumulate, split_r;
  int32_t* tickets;   /* optional: >= n_tickets device ints (zeroed by the call).  Lets a reduction be split although the output */
  int32_t n_tickets;  /* has an epilogue (act' of a data gradient): the last slice to arrive at a 64 x 64 tile finishes it */
  int32_t mfma_split; /* 0: fp32 MFMA (above); 2: every operand value as two bf16 pieces (16 significant bits, made while staging),
                         three bf16 MFMA products per k-block, fp32 accumulation -- the conv kernels' default arithmetic, for the
                         large Linear layers inside the conv stacks (cnn.Encoder head, cnn.Decoder stem) */
} MtrssmGemm;
int mtrssm_gemm(const MtrssmGemm* g, void* stream);
/* `count` INDEPENDENT problems (no output of one is an operand or the output of another; mfma_split = 0) in as few launches as
 * their operand layouts allow -- problems with equal (a_rmajor, b_rmajor) share one grid.  The weight gradients of the scan
 * (scan.py: ten to fifteen [out, B*T] x [B*T, in] GEMMs per backward, each a latency-bound launch of its own before) go
 * through this call.  Per problem exactly the arithmetic of mtrssm_gemm. */
int mtrssm_gemm_group(const MtrssmGemm* problems, int32_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTRSSM_H */
