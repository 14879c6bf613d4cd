// extern "C" boundary of libmtrssm_hip.so (declared in include/mtrssm.h).
#include <stdarg.h>
#include <stdio.h>

#include "scan_common.h"

namespace mtrssm {

static thread_local char g_err[512] = "";
static thread_local const char* g_kernel = "";

void set_last_kernel(const char* name) { g_kernel = name; }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mrssm_fwd_launch(const MtrssmMrssmDims*, const MtrssmMrssmFwdWeights*, const MtrssmMrssmFwdIO*, hipStream_t);
int mrssm_bwd_launch(const MtrssmMrssmDims*, const MtrssmMrssmBwdWeights*, const MtrssmMrssmBwdIO*, hipStream_t);
int mmtrssm_fwd_launch(const MtrssmMmtrssmDims*, const MtrssmMmtrssmFwdWeights*, const MtrssmMmtrssmFwdIO*, hipStream_t);
int mmtrssm_bwd_launch(const MtrssmMmtrssmDims*, const MtrssmMmtrssmBwdWeights*, const MtrssmMmtrssmBwdIO*, hipStream_t);
int conv_gather_gemm_launch(const MtrssmConvGeom*, const float*, const float*, const float*, const unsigned short*, const float*, const float*, const float*, float*, hipStream_t);
int episode_gather_launch(const float*, const int64_t*, const float*, int64_t, int64_t, int64_t, int64_t, int64_t, float, float*, float*, hipStream_t);
int conv_gather_gemm_pair_launch(const MtrssmConvGeom*, const float*, const float*, const float*, const unsigned short*, const float*, const float*, const float*, float*,
                                 const MtrssmConvGeom*, const float*, const float*, const float*, const unsigned short*, const float*, const float*, const float*, float*, hipStream_t);
int conv_residual_fwd_supported(const MtrssmConvGeom*);
int conv_residual_fwd_launch(const MtrssmConvGeom*, const float*, const unsigned short*, const float*, const float*, const float*, float*, float*,
                             const MtrssmConvGeom*, const float*, const unsigned short*, const float*, const float*, const float*, float*, float*, hipStream_t);
int pack_conv_weight_launch(const float*, int, int, int, int, long, long, long, long, int, int, int, float*, unsigned short*, hipStream_t);
int pack_conv_weights_launch(const int64_t*, int, int, hipStream_t);
int conv_gather_pair_merges(const MtrssmConvGeom*, const MtrssmConvGeom*, bool);
int conv_weight_grad_launch(const MtrssmConvGeom*, const float*, const float*, const float*, int, float*, float*, void*, size_t, size_t*, hipStream_t);
int conv_weight_grad_deferred_launch(const MtrssmConvGeom*, const float*, const float*, const float*, int, float*, float*, void*, size_t, hipStream_t);
int conv_weight_grad_reduce_flush(hipStream_t);
int channel_sum_launch(const float*, int, int, int, float*, hipStream_t);
int convt_k4s2_band_supported(int, int, int, int, int);
int convt_k4s2_band_launch(int, int, int, int, int, const float*, const float*, const float*, int, int, float*, hipStream_t);
int convt_k4s2_thin_launch(int, int, int, int, int, const float*, const float*, const float*, int, int, float*, hipStream_t);
int conv_convt_quad_supported(const MtrssmConvGeom*);
int conv_convt_quad_launch(const MtrssmConvGeom*, const float*, const unsigned short* const*, const float*, const float*, float*,
                           const MtrssmConvGeom*, const float*, const unsigned short* const*, const float*, const float*, float*, hipStream_t);
int conv_tgather_thin_launch(int, int, int, int, int, int, int, int, int, int, int, const float*, const float*, const float*, int, int, const float*,
                             const float*, float*, hipStream_t);
int categorical_sample_fwd_launch(const float*, const float*, int64_t, int, int, float*, float*, float*, hipStream_t);
int categorical_sample_bwd_launch(const float*, const float*, const float*, int64_t, int, int, float*, hipStream_t);
int elbo_combine_fwd_launch(const float*, const float*, const float*, const float*, int64_t, float, float, float*, float*, float*, float*, hipStream_t);
int elbo_combine_bwd_launch(const float*, const float*, const float*, const float*, int64_t, float, float, float*, float*, float*, float*, hipStream_t);
int nll_fwd_launch(const float*, const float*, int64_t, int64_t, int, float*, hipStream_t);
int nll_bwd_launch(const float*, const float*, const float*, int64_t, int64_t, int, float*, hipStream_t);
int sumsq_launch(const float*, int64_t, float*, hipStream_t);
int adamw_launch(float*, const float*, float*, float*, int64_t, const float*, float, float, float, float, float, float, float, int, hipStream_t);
int gemm_launch(const MtrssmGemm*, hipStream_t);
int gemm_group_launch(const MtrssmGemm*, int, hipStream_t);
int debug_set_cluster_profile(void*);
int debug_set_resident_profile(void*);
int debug_set_wide_profile(void*);
int debug_set_mmt_profile(void*);
int mrssm_cluster_supported(const MtrssmMrssmDims*);
size_t mrssm_cluster_workspace_bytes(const MtrssmMrssmDims*);
size_t mrssm_cluster_bwd_workspace_bytes(const MtrssmMrssmDims*);
int mrssm_bwd_cluster_launch(const MtrssmMrssmDims*, const MtrssmMrssmClusterWeights*, const MtrssmMrssmBwdIO*, void*, size_t, hipStream_t);
int mrssm_fwd_cluster_launch(const MtrssmMrssmDims*, const MtrssmMrssmClusterWeights*, const MtrssmMrssmFwdIO*, void*, size_t, hipStream_t);
int unpack_conv_grads_launch(const int64_t*, int, int, hipStream_t);
int mrssm_wide_supported(const MtrssmMrssmDims*, int);
size_t mrssm_wide_workspace_bytes(const MtrssmMrssmDims*, int);
size_t mrssm_wide_bwd_workspace_bytes(const MtrssmMrssmDims*, int);
int mrssm_wide_fwd_launch(const MtrssmMrssmDims*, const MtrssmMrssmClusterWeights*, const MtrssmMrssmFwdIO*, int, void*, size_t, hipStream_t);
int mrssm_wide_bwd_launch(const MtrssmMrssmDims*, const MtrssmMrssmClusterWeights*, const MtrssmMrssmBwdIO*, int, void*, size_t, hipStream_t);
int adamw_prepare_launch(const float*, int64_t, float*, float*, const int*, float, float, hipStream_t);
int adamw_apply_launch(float*, const float*, float*, float*, const unsigned char*, int64_t, const float*, const float*, const int*, float, float, float,
                       float, float, float, hipStream_t);

// Compute units of the calling thread's current device (0 when there is no device): the kernels whose workgroups wait for each
// other size their grids by it.  A cache of an immutable device property, not state.
int mmtrssm_wide_supported(const MtrssmMmtrssmDims*, int);
size_t mmtrssm_wide_workspace_bytes(const MtrssmMmtrssmDims*, int);
size_t mmtrssm_wide_bwd_workspace_bytes(const MtrssmMmtrssmDims*, int);
int mmtrssm_wide_fwd_launch(const MtrssmMmtrssmDims*, const MtrssmMmtrssmFwdWeights*, const MtrssmMmtrssmFwdIO*, int, void*, size_t, hipStream_t);
int mmtrssm_wide_bwd_launch(const MtrssmMmtrssmDims*, const MtrssmMmtrssmBwdWeights*, const MtrssmMmtrssmBwdIO*, int, void*, size_t, hipStream_t);

int device_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (dev < 0 || dev >= 64) return 0;
  if (cached[dev] > 0) return cached[dev];
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); return 0; }
  cached[dev] = v;
  return v;
}

}  // namespace mtrssm

using namespace mtrssm;
#define MTRSSM_API extern "C" __attribute__((visibility("default")))

MTRSSM_API int mtrssm_version(void) { return MTRSSM_VERSION; }
MTRSSM_API const char* mtrssm_last_error(void) { return g_err; }
MTRSSM_API const char* mtrssm_last_kernel(void) { return g_kernel; }

MTRSSM_API int mtrssm_mrssm_rollout_fwd(const MtrssmMrssmDims* d, const MtrssmMrssmFwdWeights* w, const MtrssmMrssmFwdIO* io, void* stream) {
  return mrssm_fwd_launch(d, w, io, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_mrssm_rollout_bwd(const MtrssmMrssmDims* d, const MtrssmMrssmBwdWeights* w, const MtrssmMrssmBwdIO* io, void* stream) {
  return mrssm_bwd_launch(d, w, io, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_mmtrssm_rollout_fwd(const MtrssmMmtrssmDims* d, const MtrssmMmtrssmFwdWeights* w, const MtrssmMmtrssmFwdIO* io, void* stream) {
  return mmtrssm_fwd_launch(d, w, io, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_mmtrssm_rollout_bwd(const MtrssmMmtrssmDims* d, const MtrssmMmtrssmBwdWeights* w, const MtrssmMmtrssmBwdIO* io, void* stream) {
  return mmtrssm_bwd_launch(d, w, io, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_categorical_sample_fwd(const float* logits, const float* u, int64_t rows, int32_t K, int32_t C, float* logp, float* probs,
                                            float* onehot, void* stream) {
  return categorical_sample_fwd_launch(logits, u, rows, K, C, logp, probs, onehot, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_categorical_sample_bwd(const float* probs, const float* g_probs, const float* g_logp, int64_t rows, int32_t K, int32_t C,
                                            float* d_logits, void* stream) {
  return categorical_sample_bwd_launch(probs, g_probs, g_logp, rows, K, C, d_logits, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_elbo_combine_fwd(const float* nll_a, const float* nll_v, const float* kl0, const float* kl1, int64_t n, float c0, float c1,
                                      float* o_recon, float* o_k0, float* o_k1, float* o_loss, void* stream) {
  return elbo_combine_fwd_launch(nll_a, nll_v, kl0, kl1, n, c0, c1, o_recon, o_k0, o_k1, o_loss, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_elbo_combine_bwd(const float* g_recon, const float* g_k0, const float* g_k1, const float* g_loss, int64_t n, float c0,
                                      float c1, float* g_nll_a, float* g_nll_v, float* g_kl0, float* g_kl1, void* stream) {
  return elbo_combine_bwd_launch(g_recon, g_k0, g_k1, g_loss, n, c0, c1, g_nll_a, g_nll_v, g_kl0, g_kl1, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_gaussian_nll_fwd(const float* pred, const float* target, int64_t frames, int64_t event, int32_t act, float* out, void* stream) {
  return nll_fwd_launch(pred, target, frames, event, act, out, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_gaussian_nll_bwd(const float* pred, const float* target, const float* g_out, int64_t frames, int64_t event, int32_t act,
                                       float* g_pred, void* stream) {
  return nll_bwd_launch(pred, target, g_out, frames, event, act, g_pred, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_sumsq(const float* x, int64_t n, float* out, void* stream) {
  return sumsq_launch(x, n, out, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* sumsq,
                                 float clip_norm, float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay,
                                 int32_t step, void* stream) {
  return adamw_launch(param, grad, exp_avg, exp_avg_sq, n, sumsq, clip_norm, grad_scale, lr, beta1, beta2, eps, weight_decay, step,
                      static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_unpack_conv_grads(const int64_t* table, int32_t count, int32_t blocks_per_entry, void* stream) {
  return unpack_conv_grads_launch(table, count, blocks_per_entry, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_mrssm_cluster_supported(const MtrssmMrssmDims* d) { return mrssm_cluster_supported(d); }
MTRSSM_API int64_t mtrssm_mrssm_cluster_workspace_bytes(const MtrssmMrssmDims* d) { return (int64_t)mrssm_cluster_workspace_bytes(d); }
MTRSSM_API int mtrssm_mrssm_rollout_fwd_cluster(const MtrssmMrssmDims* d, const MtrssmMrssmClusterWeights* w, const MtrssmMrssmFwdIO* io,
                                                void* workspace, int64_t workspace_bytes, void* stream) {
  return mrssm_fwd_cluster_launch(d, w, io, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, static_cast<hipStream_t>(stream));
}
/* development aid, not part of the documented ABI (tools/cluster_probe.py) */
extern "C" __attribute__((visibility("default"))) int mtrssm_debug_set_cluster_profile(void* buf) { return debug_set_cluster_profile(buf); }
extern "C" __attribute__((visibility("default"))) int mtrssm_debug_set_resident_profile(void* buf) { return debug_set_resident_profile(buf); }
extern "C" __attribute__((visibility("default"))) int mtrssm_debug_set_wide_profile(void* buf) { return debug_set_wide_profile(buf); }
extern "C" __attribute__((visibility("default"))) int mtrssm_debug_set_mmt_profile(void* buf) { return debug_set_mmt_profile(buf); }
MTRSSM_API int64_t mtrssm_mrssm_cluster_bwd_workspace_bytes(const MtrssmMrssmDims* d) { return (int64_t)mrssm_cluster_bwd_workspace_bytes(d); }
MTRSSM_API int mtrssm_mrssm_rollout_bwd_cluster(const MtrssmMrssmDims* d, const MtrssmMrssmClusterWeights* w, const MtrssmMrssmBwdIO* io,
                                                void* workspace, int64_t workspace_bytes, void* stream) {
  return mrssm_bwd_cluster_launch(d, w, io, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_mrssm_wide_supported(const MtrssmMrssmDims* d, int32_t pieces) { return mrssm_wide_supported(d, pieces); }
MTRSSM_API int64_t mtrssm_mrssm_wide_workspace_bytes(const MtrssmMrssmDims* d, int32_t pieces) { return (int64_t)mrssm_wide_workspace_bytes(d, pieces); }
MTRSSM_API int64_t mtrssm_mrssm_wide_bwd_workspace_bytes(const MtrssmMrssmDims* d, int32_t pieces) {
  return (int64_t)mrssm_wide_bwd_workspace_bytes(d, pieces);
}
MTRSSM_API int mtrssm_mrssm_rollout_fwd_wide(const MtrssmMrssmDims* d, const MtrssmMrssmClusterWeights* w, const MtrssmMrssmFwdIO* io, int32_t pieces,
                                             void* workspace, int64_t workspace_bytes, void* stream) {
  return mrssm_wide_fwd_launch(d, w, io, pieces, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_mrssm_rollout_bwd_wide(const MtrssmMrssmDims* d, const MtrssmMrssmClusterWeights* w, const MtrssmMrssmBwdIO* io, int32_t pieces,
                                             void* workspace, int64_t workspace_bytes, void* stream) {
  return mrssm_wide_bwd_launch(d, w, io, pieces, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_mmtrssm_wide_supported(const MtrssmMmtrssmDims* d, int32_t pieces) { return mmtrssm_wide_supported(d, pieces); }
MTRSSM_API int64_t mtrssm_mmtrssm_wide_workspace_bytes(const MtrssmMmtrssmDims* d, int32_t pieces) { return (int64_t)mmtrssm_wide_workspace_bytes(d, pieces); }
MTRSSM_API int64_t mtrssm_mmtrssm_wide_bwd_workspace_bytes(const MtrssmMmtrssmDims* d, int32_t pieces) {
  return (int64_t)mmtrssm_wide_bwd_workspace_bytes(d, pieces);
}
MTRSSM_API int mtrssm_mmtrssm_rollout_fwd_wide(const MtrssmMmtrssmDims* d, const MtrssmMmtrssmFwdWeights* w, const MtrssmMmtrssmFwdIO* io, int32_t pieces,
                                               void* workspace, int64_t workspace_bytes, void* stream) {
  return mmtrssm_wide_fwd_launch(d, w, io, pieces, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_mmtrssm_rollout_bwd_wide(const MtrssmMmtrssmDims* d, const MtrssmMmtrssmBwdWeights* w, const MtrssmMmtrssmBwdIO* io, int32_t pieces,
                                               void* workspace, int64_t workspace_bytes, void* stream) {
  return mmtrssm_wide_bwd_launch(d, w, io, pieces, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_gemm(const MtrssmGemm* g, void* stream) { return gemm_launch(g, static_cast<hipStream_t>(stream)); }
MTRSSM_API int mtrssm_gemm_group(const MtrssmGemm* problems, int32_t count, void* stream) {
  return gemm_group_launch(problems, count, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_clear(void* p, int64_t bytes, void* stream) {
  return clear_async(p, bytes < 0 ? 0 : (size_t)bytes, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_adamw_prepare(const float* grad, int64_t n, float* sumsq, float* state, const int32_t* status, float beta1, float beta2,
                                    void* stream) {
  return adamw_prepare_launch(grad, n, sumsq, state, status, beta1, beta2, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_adamw_apply(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* active, int64_t n,
                                  const float* sumsq, const float* state, const int32_t* status, float clip_norm, float grad_scale, float beta1,
                                  float beta2, float eps, float weight_decay, void* stream) {
  return adamw_apply_launch(param, grad, exp_avg, exp_avg_sq, active, n, sumsq, state, status, clip_norm, grad_scale, beta1, beta2, eps,
                            weight_decay, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_conv_gather_gemm(const MtrssmConvGeom* g, const float* src, const float* src2, const float* wp, const uint16_t* wq,
                                       const float* bias, const float* actgrad_in, const float* add_in, float* out, void* stream) {
  return conv_gather_gemm_launch(g, src, src2, wp, wq, bias, actgrad_in, add_in, out, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_episode_gather(const float* store, const int64_t* idx, const float* noise, int64_t n_episodes, int64_t B, int64_t T,
                                     int64_t Tfull, int64_t E, float std_, float* input, float* target, void* stream) {
  return episode_gather_launch(store, idx, noise, n_episodes, B, T, Tfull, E, std_, input, target, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_conv_gather_gemm_pair(const MtrssmConvGeom* ga, const float* srca, const float* src2a, const float* wpa, const uint16_t* wqa,
                                            const float* biasa, const float* actgrada, const float* adda, float* outa,
                                            const MtrssmConvGeom* gb, const float* srcb, const float* src2b, const float* wpb, const uint16_t* wqb,
                                            const float* biasb, const float* actgradb, const float* addb, float* outb, void* stream) {
  return conv_gather_gemm_pair_launch(ga, srca, src2a, wpa, wqa, biasa, actgrada, adda, outa, gb, srcb, src2b, wpb, wqb, biasb, actgradb, addb,
                                      outb, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_residual_block_fwd_supported(const MtrssmConvGeom* g3) { return conv_residual_fwd_supported(g3) != 0; }
MTRSSM_API int mtrssm_residual_block_fwd(const MtrssmConvGeom* ga, const float* xa, const uint16_t* wq3a, const float* b3a, const float* w1a,
                                         const float* b1a, float* ha, float* ya, const MtrssmConvGeom* gb, const float* xb, const uint16_t* wq3b,
                                         const float* b3b, const float* w1b, const float* b1b, float* hb, float* yb, void* stream) {
  return conv_residual_fwd_launch(ga, xa, wq3a, b3a, w1a, b1a, ha, ya, gb, xb, wq3b, b3b, w1b, b1b, hb, yb, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_pack_conv_weight(const float* w, int32_t O, int32_t I, int32_t KH, int32_t KW, int64_t so, int64_t si, int64_t sh,
                                       int64_t sw, int32_t OPad, int32_t IPad, int32_t pieces, float* wp, uint16_t* wq, void* stream) {
  return pack_conv_weight_launch(w, O, I, KH, KW, so, si, sh, sw, OPad, IPad, pieces, wp, wq, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_conv_gather_pair_merges(const MtrssmConvGeom* ga, const MtrssmConvGeom* gb, int32_t has_wq) {
  return conv_gather_pair_merges(ga, gb, has_wq != 0);
}
MTRSSM_API int mtrssm_pack_conv_weights(const int64_t* table, int32_t count, int32_t blocks_per_weight, void* stream) {
  return pack_conv_weights_launch(table, count, blocks_per_weight, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_conv_weight_grad(const MtrssmConvGeom* g, const float* a, const float* src, const float* src2, int32_t pre_act_a,
                                       float* dwp, float* dbias, void* workspace, int64_t workspace_bytes, void* stream) {
  return conv_weight_grad_launch(g, a, src, src2, pre_act_a, dwp, dbias, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes, nullptr,
                                 static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_conv_weight_grad_deferred(const MtrssmConvGeom* g, const float* a, const float* src, const float* src2, int32_t pre_act_a,
                                               float* dwp, float* dbias, void* workspace, int64_t workspace_bytes, void* stream) {
  return conv_weight_grad_deferred_launch(g, a, src, src2, pre_act_a, dwp, dbias, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes,
                                          static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_conv_weight_grad_reduce(void* stream) { return conv_weight_grad_reduce_flush(static_cast<hipStream_t>(stream)); }
MTRSSM_API int64_t mtrssm_conv_weight_grad_workspace_bytes(const MtrssmConvGeom* g, int32_t pre_act_a) {
  size_t need = 0;
  if (conv_weight_grad_launch(g, nullptr, nullptr, nullptr, pre_act_a, nullptr, nullptr, nullptr, 0, &need, nullptr) != MTRSSM_OK) return -1;
  return (int64_t)need;
}
MTRSSM_API int mtrssm_channel_sum(const float* x, int32_t N, int32_t C, int32_t HW, float* out, void* stream) {
  return channel_sum_launch(x, N, C, HW, out, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_conv_tgather_thin(int32_t N, int32_t O, int32_t Hs, int32_t Ws, int32_t Cout, int32_t KH, int32_t KW, int32_t stride,
                                        int32_t pad, int32_t Ho, int32_t Wo, const float* y, const float* w, const float* bias, int32_t pre_act,
                                        int32_t act, const float* actgrad_in, const float* add_in, float* out, void* stream) {
  return conv_tgather_thin_launch(N, O, Hs, Ws, Cout, KH, KW, stride, pad, Ho, Wo, y, w, bias, pre_act, act, actgrad_in, add_in, out,
                                  static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_convt_quad_supported(const MtrssmConvGeom* g4) { return conv_convt_quad_supported(g4); }
MTRSSM_API int mtrssm_convt_quad(const MtrssmConvGeom* ga4, const float* srca, const uint16_t* const* wqa4, const float* biasa,
                                 const float* actgrada, float* outa, const MtrssmConvGeom* gb4, const float* srcb,
                                 const uint16_t* const* wqb4, const float* biasb, const float* actgradb, float* outb, void* stream) {
  return conv_convt_quad_launch(ga4, srca, reinterpret_cast<const unsigned short* const*>(wqa4), biasa, actgrada, outa, gb4, srcb,
                                reinterpret_cast<const unsigned short* const*>(wqb4), biasb, actgradb, outb,
                                static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_convt_k4s2_band_supported(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t Cout) {
  return convt_k4s2_band_supported(N, C, Hs, Ws, Cout);
}
MTRSSM_API int mtrssm_convt_k4s2_band(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t Cout, const float* src, const float* w,
                                     const float* bias, int32_t pre_act, int32_t act, float* out, void* stream) {
  return convt_k4s2_band_launch(N, C, Hs, Ws, Cout, src, w, bias, pre_act, act, out, static_cast<hipStream_t>(stream));
}
MTRSSM_API int mtrssm_convt_k4s2_thin(int32_t N, int32_t C, int32_t Hs, int32_t Ws, int32_t Cout, const float* src, const float* w,
                                      const float* bias, int32_t pre_act, int32_t act, float* out, void* stream) {
  return convt_k4s2_thin_launch(N, C, Hs, Ws, Cout, src, w, bias, pre_act, act, out, static_cast<hipStream_t>(stream));
}
