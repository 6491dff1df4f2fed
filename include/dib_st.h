/* dib_st.h - C ABI of the building blocks of the per-particle Distributed-IB SET TRANSFORMER on MI355X (part of
 * libdib_hip.so; conventions as in dib_hip.h: extern "C", raw device pointers + sizes, int return codes, never allocates
 * or synchronises, enqueues on the caller's stream, capture-safe).
 *
 * Reference (BASELINE config 5):
 *   complex_systems/InfoDecomp_Amorphous_plasticity_per_particle_measurements_and_set_transformer.ipynb, code cell 8 -
 *   particle encoder `tf.keras.Sequential([PositionalEncoding, Dense(128, LeakyReLU(0.1)) x2, Dense(2*32)])`, the
 *   `train_step` bottleneck (logvar - 3, reparameterised sample, KL summed over (particle, dim)), six blocks of
 *   MultiHeadAttention(12, 128)(x, x, x) -> Add -> LayerNormalization -> Dense(128, relu), Dense(32, relu) -> Add ->
 *   LayerNormalization, tf.reduce_mean over the particle axis, Dense(256, LeakyReLU(0.1)), Dense(1), BCE-from-logits.
 * The host-side mirror of that notebook cell is dib_amd/set_transformer.py; these are the device entry points it drives.
 */
#ifndef DIB_ST_H
#define DIB_ST_H
#include "dib_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DIB_ACT_LEAKY_RELU_01 7 /* tf.keras.layers.LeakyReLU(0.1) (the string 'leaky_relu' of the other notebooks is 0.2 = 2) */

/* One GEMM of a grouped launch.  Offsets are in ELEMENTS relative to the base pointers of dib_gemm_grouped; *_boff are
 * reserved (set 0).  bias_off < 0: no bias (mode 0) / no bias gradient (mode 2). */
typedef struct dib_gemm_desc {
  int64_t a_off, b_off, c_off, bias_off, aux_off;
  int64_t a_boff, b_boff, c_boff, aux_boff;
  int32_t M, N, K;
  int32_t lda, ldb, ldc, ldaux;
  int32_t flags;
} dib_gemm_desc;

/* n_groups independent fp32-MFMA GEMMs in one launch (blockIdx.z = group); dev_desc is a DEVICE array.
 *   mode 0: C[M,N] = act(A[M,K] @ B[K,N] + bias[N])              Dense forward / P @ V
 *   mode 1: C[M,N] = (A[M,K] @ B[N,K]^T) * act'(aux[M,N])        Dense dgrad (B = Keras kernel [in=N, out=K]) / Q @ K^T
 *   mode 2: C[M,N] = A[K,M]^T @ B[K,N]; bias_out[N] = column sums of B (groups with bias_off >= 0)   weight gradients,
 *           P^T @ dO, dS^T @ Q.  The contraction is split over nsplit slabs of rows_per_split rows written split_stride
 *           elements apart (nsplit = 1: directly into C); reduce with dib_reduce_splits.
 * max_m / max_n: largest M / N over the groups (tile-shape rule). */
int dib_gemm_grouped(int mode, int n_groups, const dib_gemm_desc* dev_desc, int max_m, int max_n, const float* A,
                     const float* B, float* C, const float* bias, const float* aux, float* bias_out, int act, int nsplit,
                     int rows_per_split, int64_t split_stride, dib_stream_t stream);
int dib_reduce_splits(const float* partial, int64_t n, int nsplit, int64_t stride, float* out, dib_stream_t stream);
/* out[i] += sum of the slabs (the split-K gradient of a residual branch added to the gradient already in `out`: one launch
 * instead of reduce + add) */
int dib_reduce_splits_add(const float* partial, int64_t n, int nsplit, int64_t stride, float* out, dib_stream_t stream);

/* The same products for a SKINNY contraction (K <= 32, K % 4 == 0) with a large output - the set transformer's q / k / v
 * projections of the 32-wide residual stream (dense layers of ...set_transformer.ipynb:332-389's MultiHeadAttention) and the
 * gradient of the attention context: a streaming kernel bound by its output stores (no LDS, B columns register-resident).
 * Every group has the same M, N, K (passed by value; the descriptors' M / N / K are ignored) and its own offsets / leading
 * dimensions.  mode 0: C = A[M,K] @ B[K,N] + bias[N];  mode 1: C = A[M,K] @ B[N,K]^T.  No activation.  N % 32 == 0.
 * DIB_E_UNSUPPORTED for shapes outside that (use dib_gemm_grouped). */
int dib_gemm_skinny_k(int mode, int n_groups, const dib_gemm_desc* dev_desc, int M, int N, int K, const float* A,
                      const float* B, float* C, const float* bias, dib_stream_t stream);

/* Keras MultiHeadAttention softmax over the key axis, in place: S[row][0..P) <- softmax(scale * S[row][0..P)); rows are ld
 * floats apart.  Backward (in place on dP): dS = scale * P * (dP - sum_j dP_j P_j), the gradient w.r.t. the unscaled q.k. */
int dib_softmax_rows_fwd(float* S, int64_t rows, int P, int ld, float scale, dib_stream_t stream);
int dib_softmax_rows_bwd(const float* P_probs, float* dP, int64_t rows, int P, int ld, float scale, dib_stream_t stream);

/* Flash-style self-attention over the particle axis, Keras MultiHeadAttention(heads, key_dim = 128)(x, x, x) semantics:
 * o[b, p, h, :] = sum_q softmax_q(scale * q[b, p, h, :] . k[b, q, h, :]) v[b, q, h, :].  q, k, v, o (and gradients) are
 * [B * P, ld] row-major, head h in columns [h * 128, (h + 1) * 128), 16-byte aligned.  The [P, P] probabilities never reach
 * HBM (online softmax forward; the backward rebuilds them from the scores and lse [B, H, P]); deterministic (no atomics: the
 * dQ contributions of the 128-key blocks go through a partial buffer inside `ws` and a fixed-order reduce).
 * s_stash (optional, dib_attention_stash_bytes; both calls get the same buffer or both NULL): the forward leaves the raw
 * score tiles there and the backward reads them back instead of recomputing S = scale q k^T - 4 tile products per tile pair
 * instead of 5 for 4 * ceil(P/32)^2 * 4 KB per (neighbourhood, head) of HBM (3.2 GB at 4 x 4096 x 12).  NULL: recompute.
 * ws: dib_attention_bwd_workspace_bytes.
 * P <= 64 (the reference notebook's neighbourhoods hold 50 particles): one workgroup per (neighbourhood, head) keeps q, k, v
 * (dO) and the [P, P] scores in LDS - one launch forward, one backward, no partial buffer; dib_attention_stash_bytes is 0 and
 * s_stash is ignored (csrc/dib_attn_small.h).
 * DIB_E_UNSUPPORTED for key_dim != 128 or P * ld >= 2^30 elements (row offsets inside one neighbourhood are 32-bit). */
int64_t dib_attention_stash_bytes(int B, int P, int H);
int dib_attention_fwd(const float* q, const float* k, const float* v, int B, int P, int H, int key_dim, int64_t ld,
                      float scale, float* o, float* lse, float* s_stash, dib_stream_t stream);
int64_t dib_attention_bwd_workspace_bytes(int B, int P, int H);
/* Round 6, neighbourhoods of at most 64 particles (the notebook's 50), model width 32: MultiHeadAttention's three input Dense
 * layers inside the attention forward - q / k / v [T, ld] are OUTPUTS (written for dib_attention_bwd), computed per
 * (neighbourhood, head) as x [P, 32] @ (params + w_off[i]) [32, H * key_dim] + (params + b_off[i]), i = q, k, v; ld == H * key_dim.
 * One launch instead of projection + attention.  dib_attention_fwd_proj_supported: P <= 64, key_dim == 128, model_dim == 32. */
int dib_attention_fwd_proj_supported(int P, int key_dim, int model_dim);
int dib_attention_fwd_proj(const float* x, int64_t ldx, const float* params, const int64_t* w_off, const int64_t* b_off, int B, int P,
                           int H, int key_dim, int model_dim, int64_t ld, float scale, float* q, float* k, float* v, float* o,
                           float* lse, dib_stream_t stream);
/* ... and their INPUT gradient inside the attention backward (8-wave kernel of csrc/dib_attn_small.h): besides dq / dk / dv
 * [T, ld] (still written: operands of the projections' weight gradients) every (neighbourhood, head) workgroup writes
 * dq_h W_q[:, head]^T + dk_h W_k[:, head]^T + dv_h W_v[:, head]^T [P, model_dim] into slab 1 + head of dx_slabs (slabs
 * slab_stride floats apart; slab 0 is the caller's: the residual share of the block input's gradient).  The consumer sums
 * slabs 0 .. H in order (dib_st_chain_bwd g_out_slabs = 1 + H, or dib_reduce_splits).  Same support rule as the forward. */
int dib_attention_bwd_proj(const float* q, const float* k, const float* v, const float* d_o, const float* lse, int B, int P, int H,
                           int key_dim, int model_dim, int64_t ld, float scale, float* dq, float* dk, float* dv, const float* params,
                           const int64_t* w_off, float* dx_slabs, int64_t slab_stride, dib_stream_t stream);
int dib_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                      const float* s_stash, int B, int P, int H, int key_dim, int64_t ld, float scale, float* dq, float* dk,
                      float* dv, void* ws, dib_stream_t stream);

/* tf.keras.layers.Add()([a, b]) -> LayerNormalization(epsilon): y = (s - mean)/sqrt(var + eps) * gamma + beta over the last
 * axis (D <= 256); xhat [T, D] and rstd [T] are stashed for the backward.  b may arrive as b_slabs >= 1 split-K partials
 * b_stride elements apart (the attention output projection at small token counts): they are summed here, in slab order,
 * instead of by a separate dib_reduce_splits launch.  Backward: ds [T, D] (gradient of BOTH addends)
 * and dgamma_dbeta = [dgamma (D) | dbeta (D)] (contiguous, Keras variable order gamma, beta). */
int dib_add_layernorm_fwd(const float* a, const float* b, int b_slabs, int64_t b_stride, int64_t T, int D, const float* gamma,
                          const float* beta, float eps, float* y, float* xhat, float* rstd, dib_stream_t stream);
int64_t dib_add_layernorm_bwd_workspace_bytes(int64_t T, int D);
int dib_add_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, int64_t T, int D,
                          float* ds, float* dgamma_dbeta, void* ws, dib_stream_t stream);
/* The same with two optional fusions: dy2 (a second gradient addend: dY = dy + dy2 - the residual branch's gradient) and
 * (act_src, act, dz): dz = ds * act'(act_src) for the branch that fed the Add through activation `act` (act_src = its
 * post-activation output).  NULL / NULL: dib_add_layernorm_bwd. */
/* The token-wise half of an attention block (...set_transformer.ipynb:332-389: output projection -> Add + LayerNorm ->
 * feed-forward Dense(relu)* -> Add + LayerNorm) as ONE launch per direction for up to 4096 tokens (csrc/dib_st_chain.h: a
 * workgroup owns 16 tokens).  Offsets are ELEMENT offsets into `params` (Keras orientation: o_w [heads*key_dim, D], ff_w[l]
 * [in, width]); D % 32 == 0, D <= 256; widths and HK % 16 == 0; ff_width[n_ff - 1] == D; act in {linear, relu, leaky_relu}.
 *   fwd: mha = ctx @ o_w + o_b; h = LN1(x_in + mha) (+ xhat1, rstd1); f_l = act(f_{l-1} ff_w[l] + ff_b[l]); x_out = LN2(h + f_last)
 *   bwd: from g_out = dL/dx_out (ABI 6: given as g_out_slabs >= 1 partial buffers g_out_stride floats apart, summed by the
 *        kernel in slab order - the next block's LN1-addend gradient and the split-K slabs of its q / k / v input gradient
 *        arrive that way, without a reduce launch per block): g_ff[l] = dL/d(pre-activation of layer l) (the dy operands of the ff weight gradients),
 *        g_in = dL/d(x_in + mha) (the residual share of the block input's gradient AND the dy of the o_w gradient),
 *        g_ctx = g_in @ o_w^T, and the four LayerNorm parameter gradients written to grads + ln{1,2}_{g,b}.
 * ws: dib_st_chain_workspace_bytes, zero-filled once by the caller (per-tile partials + one arrival counter). */
typedef struct dib_st_block_desc {
  int64_t o_w, o_b, ln1_g, ln1_b, ln2_g, ln2_b, ff_w[3], ff_b[3];
  int32_t n_ff, ff_width[3];
  int32_t D, HK;
  float eps;
  int32_t act;
} dib_st_block_desc;
int dib_st_chain_supported(const dib_st_block_desc* d, int64_t T);
int64_t dib_st_chain_workspace_bytes(int64_t T, int D);
int dib_st_chain_fwd(const dib_st_block_desc* d, int64_t T, const float* params, const float* ctx, const float* x_in, float* h,
                     float* xhat1, float* rstd1, float* const* ff, float* x_out, float* xhat2, float* rstd2, dib_stream_t stream);
int dib_st_chain_bwd(const dib_st_block_desc* d, int64_t T, const float* params, const float* g_out, int g_out_slabs,
                     int64_t g_out_stride, const float* xhat2, const float* rstd2, const float* const* ff, const float* xhat1,
                     const float* rstd1, float* const* g_ff, float* g_in, float* g_ctx, float* grads, void* ws, dib_stream_t stream);

int dib_add_layernorm_bwd_fused(const float* dy, const float* dy2, const float* xhat, const float* rstd, const float* gamma,
                                int64_t T, int D, float* ds, const float* act_src, int act, float* dz, float* dgamma_dbeta,
                                void* ws, dib_stream_t stream);

/* tf.reduce_mean(x, axis=-2): x [B, P, D] -> out [B, D]; backward dx = g / P broadcast over the particle axis */
int dib_mean_pool_fwd(const float* x, int B, int P, int D, float* out, dib_stream_t stream);
int dib_mean_pool_bwd(const float* g, int B, int P, int D, float* dx, dib_stream_t stream);
int dib_add_inplace(float* dst, const float* src, int64_t n, dib_stream_t stream);
/* out = g * act'(y), y = post-activation values (activation ids of dib_hip.h + DIB_ACT_LEAKY_RELU_01) */
int dib_act_grad_mul(const float* g, const float* y, int act, int64_t n, float* out, dib_stream_t stream);

/* The notebook's bottleneck on T = batch * particles tokens: enc_out [T, 2E] = (mu | raw logvar);
 *   logvar = raw + logvar_offset (-3); u = mu + exp(logvar/2) * eps; kl_sum = sum_{token, dim} 0.5 (mu^2 + e^logvar - logvar - 1)
 * (the caller divides by the number of neighbourhoods: "sum over dimension and particles, avg over batch").  eps is the
 * library's counter-based noise keyed by (seed, step, row0 + token, feature 0, dim).  Backward:
 *   d_enc_out = (g_u + k mu | g_u (u - mu) / 2 + k (e^logvar - 1)/2), k = beta_dev[0] * inv_batch, with u [T, E] the
 * samples the forward pass actually used (eps sigma = u - mu): the library's, the caller's own (a caller may overwrite u
 * before running the rest of the model) or mu itself (deterministic forward). */
int64_t dib_token_kl_workspace_bytes(int64_t T, int E);
int dib_token_reparam_kl_fwd(const float* enc_out, int64_t T, int E, float logvar_offset, uint64_t seed, uint32_t step,
                             const uint32_t* step_dev /* non-NULL: the noise step is read from device memory (hipGraph replay) */,
                             int64_t row0, int deterministic, float* u, float* kl_sum, void* ws, dib_stream_t stream);
int dib_token_reparam_kl_bwd(const float* enc_out, const float* g_u, const float* u, int64_t T, int E, float logvar_offset,
                             const float* beta_dev, float inv_batch, float* d_enc_out, dib_stream_t stream);

/* Per-particle information map (notebook cell 8, "Now use probe points along with a bunch of real points to get the info
 * for points on a grid"): enc_probe [M, 2E] / enc_data [N, 2E] = particle_encoder outputs (mu | raw logvar), logvar_offset
 * = -3.  One sample u_i ~ N(mu_i, sigma_i) per probe from the library's counter-based noise (seed, step, row = probe
 * index, feature); per probe lower = infonce_per (mean over the probe's own and the N data conditionals), upper = loo_per
 * (mean over the N data conditionals), nats, float64 with a log-sum-exp.  u_probe_out [M, E] (optional) returns the samples. */
int64_t dib_mi_probe_workspace_bytes(int n_probes, int n_data, int E);
int dib_mi_probe_bounds(const float* enc_probe, int n_probes, const float* enc_data, int n_data, int E, float logvar_offset,
                        uint64_t seed, uint32_t step, uint32_t feature, double* lower_rows, double* upper_rows,
                        double* u_probe_out, void* ws, dib_stream_t stream);

/* Keras loss on plain buffers (DIB_LOSS_* of dib_hip.h): out3 = {sum of per-row losses, #correct, rows};
 * g_pred = d(mean loss)/d(pred) * (inv_global_batch * batch). */
int64_t dib_loss_rows_workspace_bytes(int batch);
int dib_loss_rows(int loss_kind, const float* pred, int out_dim, const float* y, int64_t ldy, int batch,
                  float inv_global_batch, float* g_pred, float* out3, void* ws, dib_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
