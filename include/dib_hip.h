/*
 * dib_hip.h - C ABI of libdib_hip.so: the MI355X (gfx950) Distributed-IB training hot path.
 *
 * The reference (distributed-information-bottleneck.github.io @ 2024_10_08) has NO FFI on this
 * path: its boundary is the Python/Keras object surface.  This header is the
 * C boundary beneath the Python mirror of that surface; each entry point cites the reference
 * code (file:line under /root/reference) whose device math it replaces.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C"; every entry returns int: 0 = DIB_OK, negative = DIB_E_*, positive = hipError_t.
 *   - Never throws, never allocates or frees device memory: the caller owns every buffer,
 *     including the workspace (size from dib_workspace_bytes) and the descriptor tables
 *     (size from dib_layout_table_bytes).
 *   - Work is enqueued on the caller's hipStream_t; entry points return without synchronising
 *     and use no hipMalloc/hipFree/hipDeviceSynchronize (hipGraph-capturable).
 *   - All tensors are float32, row-major, contiguous unless a leading dimension is given.
 *   - Parameters live in ONE flat float32 buffer described by the dib_layout (Keras [in,out]
 *     kernel orientation, y = x @ W + b); gradients / Adam moments use the same layout, which
 *     is also the RCCL all-reduce bucket.
 *   - beta, learning-rate and the Adam step counter are DEVICE scalars (mirrors tf.Variable,
 *     models.py:86) so a captured step can be replayed while the host anneals beta.
 *
 * Threads ("thread-safe across distinct (device, stream, workspace) triples")
 *   - Every entry point that enqueues work may be called from several host threads at the same time as long as no two
 *     concurrent calls share a stream or write the same caller-owned buffer (workspace, gradient / parameter / Adam buffers,
 *     sync words, outputs).  The device a call runs on is the calling thread's current HIP device.  The results are the bits
 *     of the same calls made one after the other (tests/test_gpu_concurrency.py: two threads x two streams x two layouts).
 *   - A dib_layout is immutable after dib_layout_upload_tables / dib_layout_set_step_counter; from then on several threads
 *     may use ONE layout with distinct workspaces (dib_workspace_init takes the layout's internal lock for its per-batch
 *     descriptor cache).  dib_layout_create / _upload_tables / _set_step_counter / _destroy of one layout are not
 *     concurrent with anything else that uses it.
 *   - The library keeps no other per-call state in globals: the first-launch kernel attributes (dynamic LDS limits, per device
 *     ordinal) are set under a lock and published before the launch that needs them; the compute-unit count of the split rule
 *     is read from the calling thread's current device; dib_launch_count is a relaxed atomic sum over all threads.
 *   - dib_set_tuning is CONFIGURATION, not a per-call argument: it writes process-wide integers that every other entry
 *     point only reads.  Call it while no other entry point is running (start-up, or between steps of a single-threaded A/B);
 *     dib_get_tuning is always safe.  dib_profile_enable / dib_profile_summary are diagnostics with their own lock: spans
 *     from all threads land in one table.
 *   - Co-resident workgroups (round 6).  The row-tile integration kernel's cluster mode ("int_cluster") makes the 8 (or 4) workgroups
 *     of a row tile wait for each other inside the kernel.  They are consecutive in ONE XCD's dispatch order, so whatever share of the
 *     CUs a launch gets, the clusters at the front of its queue are complete and drain; a stall needs every queue on an XCD to hold
 *     only a PARTIAL cluster - five or more streams launching clustered steps onto the same 32 CUs at once.  The waits are bounded:
 *     after 2 s of wall clock the kernel traps (the process sees a HIP error at its next synchronisation) instead of hanging the
 *     device.  A program that steps many small models on more than four streams of one GPU sets "int_cluster" to 0.
 */
#ifndef DIB_HIP_H
#define DIB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dib_layout dib_layout; /* opaque, host-side */
typedef void* dib_stream_t;           /* hipStream_t */

#define DIB_OK 0
#define DIB_E_ARG (-1)
#define DIB_E_SHAPE (-2)
#define DIB_E_WORKSPACE (-3)
#define DIB_E_UNSUPPORTED (-4)
#define DIB_E_NODEVICE (-5)

/* activation ids (Keras names: None/'linear', 'relu', 'leaky_relu', 'tanh', 'sigmoid', 'elu', 'softplus') */
enum { DIB_ACT_LINEAR = 0, DIB_ACT_RELU = 1, DIB_ACT_LEAKY_RELU = 2, DIB_ACT_TANH = 3,
       DIB_ACT_SIGMOID = 4, DIB_ACT_ELU = 5, DIB_ACT_SOFTPLUS = 6 };

/* loss kinds (data.py:65 BinaryCrossentropy(from_logits=True); data.py:343 SparseCategoricalCrossentropy;
 * 'mse') */
enum { DIB_LOSS_BCE_LOGITS = 0, DIB_LOSS_BCE = 1, DIB_LOSS_SPARSE_CCE_LOGITS = 2, DIB_LOSS_MSE = 3 };

/* workspace sub-buffers addressable by the host (dib_workspace_offset) */
enum { DIB_WS_U = 0,        /* [B, F*E]  sampled embeddings, models.py:108,122 */
       DIB_WS_PRED = 1,     /* [B, out]  model output, models.py:122 */
       DIB_WS_ENC_OUT = 2,  /* [F][B][2E] feature-major (mu|logvar) per feature, models.py:106 */
       DIB_WS_G_U = 3,      /* [B, F*E]  dL/du */
       DIB_WS_STEP_OUT = 4, /* [F+3]     per-step scalars: KL_f local sums, task-loss sum, #correct, rows */
       DIB_WS_G_PRED = 5,   /* [B, out]  dL/dpred */
       DIB_WS_ENC_H0 = 16,  /* + l: [F][B][units_l] feature-major post-activation output of encoder hidden layer l (training
                               forward only; the backward's act' choices are exactly `value > 0`) */
       DIB_WS_INT_H0 = 32   /* + l: [B][units_l] post-activation output of integration hidden layer l */ };

const char* dib_version(void);
/* Integer revision of THIS header + dib_st.h: bumped on every change to an exported signature, to an argument's meaning or
 * to the size of a caller-provided array.  A binding compares it with the DIB_ABI_VERSION it was written against and
 * refuses a library that differs (dib_amd/_lib.py does): a stale variant build called with shifted arguments would corrupt
 * memory silently.  History: 3 = round 3 (attention stash arguments, 17 profile categories); 4 = round 4; 5 = round 5
 * (`flags` argument of dib_loss_fwd_bwd / dib_output_head_fused, dib_step_tail, dib_set_tuning; the workspace grew by the
 * tail's arrival counters, which dib_workspace_init zeroes - re-run it on workspaces kept from an older library; the
 * experimental bf16x6 GEMM entry points left the library); 6 = round 6 (dib_st_chain_bwd takes its incoming
 * gradient as g_out_slabs partial buffers; dib_mlp_small_head_{supported,workspace_bytes,step} and dib_attention_fwd_proj{,_supported}, dib_attention_bwd_proj added; dib_mlp_desc.act accepts
 * DIB_ACT_LEAKY_RELU_01; the tuning key "num_cus" = 0 now means "the calling thread's current device's own
 * count" and the library no longer writes it). */
#define DIB_ABI_VERSION 6
int dib_abi_version(void);
const char* dib_error_string(int code);

/* ---- layout ------------------------------------------------------------------------------
 * Mirrors DistributedIBNet.__init__ (models.py:56-86): F feature encoders
 * [PositionalEncoding] -> Dense(units,act)* -> Dense(2E), an integration MLP Dense(units,act)* ->
 * Dense(out,out_act).  n_freq is `number_positional_encoding_frequencies` (models.py:62,70:
 * frequencies 2**arange(1,n_freq)). */
int dib_layout_create(int num_features, const int* feature_dims, int n_enc_layers, const int* enc_units,
                      int embedding_dim, int n_int_layers, const int* int_units, int output_dim,
                      int use_positional_encoding, int n_freq, int activation, int output_activation,
                      dib_layout** out);
void dib_layout_destroy(dib_layout* l);
int64_t dib_layout_param_count(const dib_layout* l);
/* net: 0 = feature encoder bank, 1 = integration network. what: 0 = kernel [rows=in, cols=out], 1 = bias [cols]. */
int dib_layout_param_block(const dib_layout* l, int net, int layer, int feature, int what, int64_t* offset,
                           int* rows, int* cols);
/* device-resident GEMM group descriptor tables (batch-size independent) */
int64_t dib_layout_table_bytes(const dib_layout* l);
int dib_layout_upload_tables(dib_layout* l, void* dev_tables, dib_stream_t stream);
/* hipGraph replay: when step_dev is non-NULL every kernel keys its noise with *step_dev (device uint32, bumped by the
 * caller between replays) instead of the by-value `step` arguments below; NULL restores the by-value behaviour. */
int dib_layout_set_step_counter(dib_layout* l, const uint32_t* step_dev);
/* workspace (activations, activation gradients, split-batch wgrad partials) for local batch B.
 * CONTRACT: before its first use a workspace must be passed once to dib_workspace_init (zero-filling it is NOT enough since
 * ABI 5).  It zeroes the regions that need it - the split-batch weight-gradient slabs (for batch >= 1024 dib_grads_finalize
 * sums every slab of every parameter block, including slabs no launch writes), the per-step scalars, the arrival counters of
 * dib_step_tail (which clean themselves after every launch) - and writes the descriptor table of the step's merged
 * weight-gradient launch for THIS batch size.  The library never writes non-zero values into unwritten slabs, so one
 * initialisation per (workspace, layout, batch size) is enough; a workspace that is handed to ANOTHER layout or batch size
 * must be initialised again (different slabs stay unwritten, different table). */
int64_t dib_workspace_bytes(const dib_layout* l, int batch);
int dib_workspace_init(const dib_layout* l, int batch, void* ws, dib_stream_t stream);
int64_t dib_workspace_offset(const dib_layout* l, int batch, int which); /* byte offset, <0 on error */
int dib_layout_wgrad_splits(const dib_layout* l, int batch);

/* ---- forward -----------------------------------------------------------------------------
 * dib_encoder_bank_fwd replaces models.py:101-115: tf.split, PositionalEncoding.call (models.py:22-23),
 * the per-feature Dense chains (models.py:73-78,106), tf.split(.,2,-1), the reparameterised sample
 * (models.py:108) and the per-feature KL (models.py:111-112).
 *   x       : dataset matrix [*, ldx]; the batch is rows row_idx[0..B) (or row0..row0+B if row_idx NULL)
 *   eps     : keyed by (seed, step, GLOBAL row id, feature, dim) - Philox4x32-10 + Box-Muller, see
 *             dib_philox_normal_ref; global row id = row_idx[b] (or row0 + b).
 *   deterministic      : bit field.  DIB_FWD_DETERMINISTIC (1): u = mu (no noise) - dib_encode_deterministic-style
 *             evaluation.  DIB_FWD_INFERENCE (2): no backward pass follows (validation / predict): the fused forward
 *             skips the activation stashes it would write for it (h1, h2, act' bits: 70 % of its HBM writes).
 * Writes ws[ENC_OUT], ws[U] and the F local KL sums (sum over local rows, not yet divided) into
 * ws[STEP_OUT][0..F). */
#define DIB_FWD_DETERMINISTIC 1
#define DIB_FWD_INFERENCE 2
#define DIB_FWD_DEFER_SUMS 4   /* leave the KL column sums to dib_step_tail(DIB_TAIL_KL): ws[STEP_OUT][0..F) is not written */
int dib_encoder_bank_fwd(dib_layout* l, const float* x, int64_t ldx, const int32_t* row_idx, int64_t row0,
                         int batch, const float* params, uint64_t seed, uint32_t step, int deterministic,
                         void* ws, dib_stream_t stream);
/* models.py:122 integration_network(concat(u)) -> ws[PRED] */
int dib_integration_fwd(dib_layout* l, int batch, const float* params, void* ws, dib_stream_t stream);

/* ---- loss + backward ---------------------------------------------------------------------
 * Keras train_step (explicit form train.py:203-219): L = mean_b loss(y, pred) + beta * sum_f KL_f
 * (models.py:118).  inv_global_batch = 1/B_global so that data-parallel ranks produce partial
 * sums that all-reduce(sum) to the global-mean gradient.
 * dib_loss_fwd_bwd: ws[PRED] -> ws[G_PRED]; task-loss sum and #correct into ws[STEP_OUT][F], [F+1].
 * flags: DIB_HEAD_DEFER_SUMS = leave the sum of the per-workgroup loss partials to dib_step_tail(DIB_TAIL_LOSS). */
#define DIB_HEAD_DEFER_SUMS 1
#define DIB_HEAD_NO_GRAD 2     /* dib_output_head_fused only: validation - prediction and loss terms, no gradient (grads may be NULL) */
#define DIB_HEAD_DEFER_WGRAD 4 /* dib_integration_head_step only: the hidden layers' weight gradients are left to dib_backward */
int dib_loss_fwd_bwd(dib_layout* l, int loss_kind, const float* y, int64_t ldy, const int32_t* row_idx,
                     int64_t row0, int batch, float inv_global_batch, int flags, void* ws, dib_stream_t stream);
int dib_integration_bwd(dib_layout* l, int batch, const float* params, float* grads, void* ws,
                        dib_stream_t stream);
/* Fused 1-unit output head of a TRAINING step (one pass over the last hidden activation instead of four launches):
 *   dib_integration_fwd_hidden  = dib_integration_fwd without the output layer (models.py:83)
 *   dib_output_head_fused       = output Dense(1) forward -> ws[PRED], the loss (dib_loss_fwd_bwd's accounting into
 *                                 ws[STEP_OUT]) and the output layer's backward (its weight / bias gradient, ws[G_PRED], the
 *                                 gradient of the last hidden layer)
 *   dib_integration_bwd_hidden  = dib_integration_bwd without the output layer
 * Same results as the unfused sequence fwd -> dib_loss_fwd_bwd -> bwd.  dib_output_head_fused_supported: out_dim 1, linear
 * output, BCE-from-logits or MSE, >= 1 hidden layer of width % 4 == 0 and <= 1024; otherwise use the unfused sequence.
 * flags: DIB_HEAD_DEFER_SUMS = the reduce of the output layer's weight-gradient partials and of the loss partials is left to
 * dib_step_tail(DIB_TAIL_HEAD_WGRAD | DIB_TAIL_LOSS_HEAD); DIB_HEAD_NO_GRAD = a validation step's head. */
int dib_output_head_fused_supported(const dib_layout* l, int loss_kind);
int dib_integration_fwd_hidden(dib_layout* l, int batch, const float* params, void* ws, dib_stream_t stream);
int dib_output_head_fused(dib_layout* l, int loss_kind, const float* y, int64_t ldy, const int32_t* row_idx, int64_t row0,
                          int batch, float inv_global_batch, int flags, const float* params, float* grads, void* ws,
                          dib_stream_t stream);
int dib_integration_bwd_hidden(dib_layout* l, int batch, const float* params, float* grads, void* ws,
                               dib_stream_t stream);
/* The three calls above as ONE entry (what a training / validation step with the fused head uses):
 *   dib_integration_head_step = dib_integration_fwd_hidden + dib_output_head_fused(flags) [+ dib_integration_bwd_hidden unless
 *   DIB_HEAD_NO_GRAD].  For batches in the row-tile regime ("small_wgs" tuning key) the hidden layers, the head, the loss and the dgrad chain back to ws[G_U] run
 *   as one launch of 16-row tiles (csrc/dib_small.h; "small_batch" tuning key) instead of five GEMM launches. */
int dib_integration_head_step(dib_layout* l, int loss_kind, const float* y, int64_t ldy, const int32_t* row_idx, int64_t row0,
                              int batch, float inv_global_batch, int flags, const float* params, float* grads, void* ws,
                              dib_stream_t stream);
/* dib_backward: everything of a step's backward pass that follows the loss, for single-GPU callers (the data-parallel bucket
 * protocol uses the separate entries below): dib_integration_bwd (skipped with DIB_BWD_INTEGRATION_DONE: the step ran
 * dib_integration_head_step(DIB_HEAD_DEFER_WGRAD), only its hidden-layer weight gradients are outstanding) + dib_encoder_bank_bwd.
 * In the row-tile regime ("small_wgs") all weight gradients of the step are ONE grouped launch over the descriptor table that
 * dib_workspace_init wrote for this batch size.  Follow with dib_grads_finalize / dib_step_tail. */
#define DIB_BWD_INTEGRATION_DONE 1
int dib_backward(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev, float inv_global_batch,
                 int flags, void* ws, dib_stream_t stream);
/* dib_encoder_bank_bwd: tape.gradient through models.py:106-118 for the encoder bank (reparameterisation + beta*KL backward,
 * dgrads, weight gradients).  The noise term of d(logvar) is recovered from the forward's own sample, eps*sigma = ws[U] -
 * mu, so the backward needs neither the noise key nor the row ids, and it is the gradient of whatever forward wrote the
 * workspace (library noise or DIB_FWD_DETERMINISTIC). */
int dib_encoder_bank_bwd(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev,
                         float inv_global_batch, void* ws, dib_stream_t stream);
/* The same work in two stages, for the data-parallel caller that wants the encoder bank's gradients in two all-reduce
 * buckets: stage 1 = the gradient chain and every weight gradient except the last encoder layer's (then
 * dib_grads_finalize_part(2) and the all-reduce of part 2), stage 2 = the last layer's weight gradient, which needs nothing
 * of stage 1's weight gradients and runs under that all-reduce (then part 3).  Stage 1 then stage 2 == dib_encoder_bank_bwd. */
int dib_encoder_bank_bwd_stage(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev,
                               float inv_global_batch, int stage, void* ws, dib_stream_t stream);
/* reduce the split-batch wgrad partials into `grads` (fixed order => deterministic).
 * _part finalises one all-reduce bucket of the layer-major flat buffer: 0 = encoder bank, 1 = integration network (its
 * gradients are complete right after dib_integration_bwd, so its RCCL all-reduce can overlap the encoder-bank backward),
 * 2 = encoder layers before the last, 3 = last encoder layer (0 = 2 + 3, contiguous), -1 = everything.
 * dib_layout_part_range: (offset, count) of a part in floats. */
int dib_grads_finalize(dib_layout* l, int batch, float* grads, void* ws, dib_stream_t stream);
int dib_grads_finalize_part(dib_layout* l, int batch, int part, float* grads, void* ws, dib_stream_t stream);
int dib_layout_part_range(const dib_layout* l, int part, int64_t* offset, int64_t* count);
/* metrics_acc[F+3] += {KL_f_local_sum * inv_global_batch (F), task_sum + beta*sum_f KL_f_local_sum,
 * #correct, rows}: the History accounting of models.py:115,121 / train.py:169-172 without a host sync */
int dib_metrics_accumulate(dib_layout* l, int batch, const float* beta_dev, float inv_global_batch,
                           float* metrics_acc, void* ws, dib_stream_t stream);
/* dib_step_tail: the END of a step in ONE launch (csrc/dib_tail.h) - any subset of: the fixed-order reduce of gradient
 * bucket `part` (dib_grads_finalize_part, same summation order, bit-identical), the reduce of the fused output head's
 * weight-gradient partials, the KL column sums and the loss sums deferred by DIB_FWD_DEFER_SUMS / DIB_HEAD_DEFER_SUMS,
 * dib_metrics_accumulate, and Keras-Adam (dib_adam_step's expressions) or SGD applied to the bucket's parameters in the same
 * pass with the step count t_dev bumped ONCE by the launch that carries DIB_TAIL_BUMP (every launch of a step reads the
 * same t: the data-parallel caller steps buckets 1 and 2 while bucket 3 is on the wire and bumps with bucket 3).
 * Without DIB_TAIL_FINALIZE the optimizer reads the gradient from `grads` (e.g. after an all-reduce).
 * A 1-GPU training step ends with ONE call: part -1, FINALIZE | HEAD_WGRAD | KL | LOSS_HEAD | METRICS | ADAM | BUMP;
 * a validation step with KL | LOSS[_HEAD] | METRICS.  Pointers a flag does not need may be NULL. */
#define DIB_TAIL_FINALIZE 1     /* grads[part] = sum of the split-batch slabs (+ the fused backward's d(W1|b1) partials) */
#define DIB_TAIL_KL 2           /* ws[STEP_OUT][0..F) = column sums of the forward's KL partials */
#define DIB_TAIL_LOSS 4         /* ws[STEP_OUT][F..F+3) from dib_loss_fwd_bwd's partials */
#define DIB_TAIL_ADAM 8
#define DIB_TAIL_BUMP 16        /* *t_dev += 1 after every workgroup has read it */
#define DIB_TAIL_METRICS 32     /* metrics_acc += ... (needs KL and LOSS sums of this step: same launch or earlier) */
#define DIB_TAIL_SGD 64
#define DIB_TAIL_HEAD_WGRAD 128 /* output-layer (W|b) gradient from dib_output_head_fused's partials (parts -1 and 1) */
#define DIB_TAIL_LOSS_HEAD 256  /* ws[STEP_OUT][F..F+3) from dib_output_head_fused's partials */
int dib_step_tail(dib_layout* l, int batch, int part, int flags, float* params, float* grads, float* adam_m, float* adam_v,
                  const float* lr_dev, int64_t* t_dev, float beta1, float beta2, float eps, float grad_scale,
                  const float* beta_dev, float inv_global_batch, float* metrics_acc, void* ws, dib_stream_t stream);

/* ---- tuning: the library reads NO environment variable; these process-wide integer switches are its only hidden inputs.
 * Defaults are the measured choices (profiles/HISTORY.md).  Keys:
 *   "fwd_small_wgs"  (512)  forward / dgrad GEMMs with fewer 128-row workgroups than this use 64-row tiles
 *   "fwd_narrow_wgs" (1024) forward GEMMs with fewer 64 x 128 workgroups than this use 64 x 64 tiles
 *   "stream_rows"    (8192) GEMMs streaming at least this many rows load / store them non-temporally
 *   "split_policy"   (1)    1 = per-launch batch-split count of weight gradients (whole rounds of workgroup slots); 0 = layout-wide
 *   "split_overhead" (128)  per-workgroup fixed cost, in batch rows, of that rule's cost model
 *   "fused_encoder"  (1)    layouts created afterwards may use the fused encoder-bank kernels (0: grouped-GEMM path; A/B, tests)
 *   "fused_head"     (1)    dib_output_head_fused_supported may answer 1
 *   "small_batch"    (1)    small batches use the row-tile kernels of csrc/dib_small.h where the layout allows, namely while
 *   "small_wgs"      (512)  ceil(batch / 16) x number_features <= this (the measured crossover, profiles/r05x_*) and batch <= 2048
 *   "mlp_row_tiles"  (1)    ... and dib_mlp_small_supported may answer 1 (the custom loop's output encoder on the row-tile kernels)
 *   "infonce_one_launch" (1) dib_infonce_fwd_bwd at batch <= 128, dim <= 64 with l2sq / l2 / cosine: one launch instead of three
 *   "attn_small_bwd_waves" (8) dib_attention_bwd for neighbourhoods of <= 64 particles: 8 waves per workgroup (two per SIMD), or 4
 *                           (the round-4 kernel; bit-identical results)
 *   "attn_fwd_waves" (8)    dib_attention_fwd for >= 256 particles: 8-wave workgroups of 256 queries sharing one staged K / V tile (4: the
 *                           4-wave kernel of 128 queries, which shorter sets always take); bit-identical outputs
 *   "wgrad_max_splits" (32) most batch slabs of a layout's weight gradients (1 .. 32; read when a workspace is SIZED: set it before the
 *                           first dib_workspace_bytes of a layout).  Fewer slabs shrink the tail's reduce but starve the small
 *                           weight gradients: 32 / 24 / 16 / 8 -> 8.20 / 8.27 / 8.38 / 8.57 ms per config-3 step (profiles/r06j_*)
 *   "wgrad_flat_tile" (1)   weight gradients of a <= 32-row operand against >= 256 columns use the 32 x 256 tile (0: 64 x 128)
 *   "num_cus"        (0)    compute units the split rule prices rounds with; 0 = the calling thread's current device's own count
 *   "int_cluster"    (8)    the row-tile integration kernel puts each 16-row tile on this many co-resident workgroups, each a column
 *                           slice of every layer, slices exchanged through L2 (csrc/dib_small.h "cluster mode"; 0 / 1: one workgroup
 *                           per tile) while
 *   "int_cluster_wgs" (256) row tiles x cluster size <= this (one workgroup per CU; 8 per tile halves to 4 for more row tiles, the two
 *                           networks of a paired grid are sized together; profiles/r06u_int_cluster_sweep.txt) and
 *   "int_cluster_short_exchange" (1) a cluster whose workgroups all report one XCC_ID exchanges through that XCD's L2 (no L2 write-back /
 *                           invalidate); 0: always the agent-scope protocol - what a cluster placed across XCDs takes (tests)
 *   "int_cluster_min_weights" (65536) the network's hidden layers hold at least this many weights
 * Returns DIB_E_ARG for an unknown key or a negative value. */
int dib_set_tuning(const char* key, int value);
int dib_get_tuning(const char* key, int* value);

/* ---- optimizer ---------------------------------------------------------------------------
 * Keras Adam (train.py:128-129): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps), eps=1e-7.
 * t_dev is a device int64 counter holding the number of steps already applied; the call increments it. */
int dib_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, const float* lr_dev,
                  int64_t* t_dev, float beta1, float beta2, float eps, float grad_scale, dib_stream_t stream);
int dib_sgd_step(float* params, const float* grads, int64_t n, const float* lr_dev, float grad_scale,
                 dib_stream_t stream);
/* For a parameter buffer that is not a dib_layout (the InfoNCE output encoder, train.py:184-192): grads = fixed-order sum of
 * nsplit partial buffers `stride` floats apart (nsplit 0: grads as given), Keras-Adam and the step-count bump in ONE launch.
 * n % 4 == 0.  sync: DIB_SYNC_WORDS zero-initialised uint32 words owned by the caller (arrival counters, self-cleaning). */
#define DIB_SYNC_WORDS 1056
int dib_reduce_adam_step(const float* partial, int nsplit, int64_t stride, float* params, float* grads, float* adam_m,
                         float* adam_v, int64_t n, const float* lr_dev, int64_t* t_dev, float beta1, float beta2, float eps,
                         float grad_scale, uint32_t* sync, dib_stream_t stream);

/* ---- evaluation helpers ------------------------------------------------------------------
 * model.feature_encoders[f](x_f) (models.py:183, visualization.py:31): deterministic [N, 2E]. x_f is [N, d_f]. */
int dib_encode_deterministic(dib_layout* l, int feature, const float* x_f, int n, const float* params,
                             float* out, void* ws, dib_stream_t stream);
/* Bhattacharyya distance matrix between diagonal Gaussians (utils.py:177-212), closed form. */
int dib_bhattacharyya(const float* mu1, const float* lv1, int n, const float* mu2, const float* lv2, int m,
                      int dim, float* out, dib_stream_t stream);
/* InfoNCE custom-loop path (train.py:201-220 eval_batch_infonce; utils.get_scaled_similarity utils.py:131-175):
 * S = sim(emb_x, emb_y)/T [B,B]; loss = mean_i CE(i, S[i,:]) + mean_j CE(j, S[:,j]) (not halved, train.py:209-214);
 * writes the loss (device scalar) and, if g_x/g_y are non-NULL, dloss/d emb_x and dloss/d emb_y.
 * similarity: 0 'l2sq', 1 'l2' (eps 1e-9 inside the sqrt), 2 'l1', 3 'linf', 4 'cosine'. */
int64_t dib_infonce_workspace_bytes(int batch);
int dib_infonce_fwd_bwd(const float* emb_x, const float* emb_y, int batch, int dim, int similarity, float temperature,
                        float* g_x, float* g_y, float* loss_out, void* ws, dib_stream_t stream);
/* PositionalEncoding.call (models.py:22-23) of one dense [n, d] matrix -> [n, d*n_freq] (train.py:186-188: Y encoder) */
int dib_positional_encoding(const float* x, int64_t ldx, int n, int d, int n_freq, float* out, dib_stream_t stream);
/* the same of rows row_idx[0..n) of x (the shuffled batch of the custom loop, train.py:226-227); n_freq <= 1: a plain gather */
int dib_positional_encoding_rows(const float* x, int64_t ldx, const int32_t* row_idx, int n, int d, int n_freq, float* out,
                                 dib_stream_t stream);

/* The custom loop's output encoder (train.py:184-192: [PositionalEncoding ->] Dense(units, act)* -> Dense(out)) at ITS batch
 * sizes (128 .. 2048 rows): the whole layer chain of 16 batch rows in one workgroup (csrc/dib_small.h), one launch for the
 * forward (gather + positional encoding + every layer) and one for the dgrad chain, instead of 1 + L and L - 1 launches.
 * The weight gradients stay one grouped GEMM on the stashes these write (dib_gemm_grouped, include/dib_st.h).
 * Layer i: kernel [in_i][width[i]] row-major at params + w_off[i], bias at params + b_off[i]; in_0 = in_dim * max(n_freq, 1);
 * width[n_hidden] = the output width (linear).  act: DIB_ACT_* of the hidden layers (linear / relu / leaky_relu, and dib_st.h's
 * DIB_ACT_LEAKY_RELU_01 = LeakyReLU(0.1), the set transformer's particle encoder). */
typedef struct dib_mlp_desc {
  int64_t w_off[4], b_off[4];
  int32_t n_hidden, width[4];
  int32_t in_dim, n_freq, act;
} dib_mlp_desc;   /* 96 bytes */
/* 1 if (desc, batch) can take the row-tile kernels: 1-3 hidden layers and the output of widths % 16 == 0 (<= 1024), a
 * piecewise-linear activation, batch <= 2048, "small_batch" and "mlp_row_tiles" tuning on */
int dib_mlp_small_supported(const dib_mlp_desc* d, int batch);
/* x: [rows][ldx] device matrix; row_idx (may be NULL: rows 0 .. n-1): the batch's rows.  Writes a0 [n][in_0] (the encoded
 * input; may be NULL when no backward follows), h[i] [n][width[i]] post-activation stashes, i < n_hidden (NULL entries
 * allowed together with a0 == NULL), out [n][width[n_hidden]]. */
int dib_mlp_small_fwd(const dib_mlp_desc* d, const float* params, const float* x, int64_t ldx, const int32_t* row_idx, int n,
                      float* a0, float* const* h, float* out, dib_stream_t stream);
/* g_out [n][width[n_hidden]] = dL/d out; h: the forward's stashes; writes g[i] [n][width[i]] = dL/d(pre-activation of hidden
 * layer i), i < n_hidden - with g_out and a0 / h the operands of every layer's weight gradient. */
int dib_mlp_small_bwd(const dib_mlp_desc* d, const float* params, const float* g_out, float* const* h, float* const* g, int n,
                      dib_stream_t stream);
/* A plain MLP with a 1-unit linear output and a BCE-from-logits / 'mse' loss - the set transformer's head,
 * Dense(256, LeakyReLU(0.1)) -> Dense(1) on the pooled neighbourhood (...set_transformer.ipynb:378-389, train_step :419-445) -
 * as ONE launch for its whole share of a training step (round 6): hidden layers (stashed in h), z = h_last . w + b -> pred, the
 * loss, g_pred = dL/dz * inv_global_batch, the dgrad chain g[i] = dL/d(pre-activation of hidden layer i) and g_x = dL/dx
 * [n][in_dim] (may be NULL), and - summed in tile order by the last workgroup to arrive - the output layer's gradient into
 * grads + w_off[n_hidden] / b_off[n_hidden] and sums3 = {loss sum, #correct (z > 0.5 == y), loss sum * inv_global_batch}.
 * The hidden layers' weight gradients stay grouped GEMMs on (x, h, g).  d: width[n_hidden] == 1, n_freq <= 1, in_dim and hidden
 * widths % 16 == 0; x [n][in_dim] contiguous; ws: dib_mlp_small_head_workspace_bytes, zero-filled once by the caller. */
int dib_mlp_small_head_supported(const dib_mlp_desc* d, int n);
int64_t dib_mlp_small_head_workspace_bytes(const dib_mlp_desc* d, int n);
int dib_mlp_small_head_step(const dib_mlp_desc* d, const float* params, const float* x, int n, const float* y, int64_t ldy,
                            int loss_kind, float inv_global_batch, float* const* h, float* const* g, float* pred, float* g_pred,
                            float* g_x, float* grads, float* sums3, void* ws, dib_stream_t stream);
/* The custom loop runs TWO independent networks between the encoder bank and the loss, and again between the loss and the
 * encoder bank's backward (train.py:203-219: model(x) and output_encoder(y); tape.gradient of both).  At its batch sizes
 * each of them is a handful of workgroups, so the two entry points below give the pairs ONE grid each:
 *   dib_integration_fwd_and_mlp_fwd = dib_integration_fwd(l, batch, params, ws) ; dib_mlp_small_fwd(d, mlp_params, x, ...)
 *   dib_backward_and_mlp_bwd        = dib_backward(l, batch, params, grads, ...) ; dib_mlp_small_bwd(d, mlp_params, g_out, ...)
 * with the same results (the same workgroup code runs on the same tiles); where the model's path has no row-tile launch
 * to share (large batch, "small_batch" tuning off on its side) the MLP pass is launched on its own after it.
 * DIB_E_UNSUPPORTED (nothing launched) if !dib_mlp_small_supported(d, n). */
int dib_integration_fwd_and_mlp_fwd(dib_layout* l, int batch, const float* params, void* ws, const dib_mlp_desc* d,
                                    const float* mlp_params, const float* x, int64_t ldx, const int32_t* row_idx, int n, float* a0,
                                    float* const* h, float* out, dib_stream_t stream);
int dib_backward_and_mlp_bwd(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev,
                             float inv_global_batch, int flags, void* ws, const dib_mlp_desc* d, const float* mlp_params,
                             const float* g_out, float* const* h, float* const* g, int n, dib_stream_t stream);

/* Mutual-information sandwich bounds (utils.estimate_mi_sandwich_bounds, utils.py:10-73; used by
 * InfoPerFeatureCallback models.py:188-223): per-row InfoNCE-lower / leave-one-out-upper terms (nats, float64,
 * log-sum-exp) for one batch of n encoded points enc_out[n, 2E] = (mu|logvar); u_i = mu_i + sigma_i*eps with eps
 * from the Philox generator keyed (seed, step, row i, feature).  The bounds are the means of the two row arrays. */
int64_t dib_mi_workspace_bytes(int n, int embedding_dim);
int dib_mi_sandwich_rows(const float* enc_out, int n, int embedding_dim, uint64_t seed, uint32_t step,
                         uint32_t feature, double* lower_rows, double* upper_rows, void* ws, dib_stream_t stream);
/* fill eps[B, F, E] exactly as the fused kernels generate it (test / oracle cross-check) */
int dib_philox_normal_fill(float* eps, const int32_t* row_idx, int64_t row0, int batch, int num_features,
                           int embedding_dim, uint64_t seed, uint32_t step, dib_stream_t stream);
/* host-side reference of the same generator (float32 evaluation) - no GPU needed */
float dib_philox_normal_ref(uint64_t seed, uint32_t step, uint32_t row, uint32_t feature, uint32_t e);

/* ---- live kernel timing for bench.py's roofline: HIP events around every launch on the launch stream.
 * 17 categories = kernel symbols: 0..11 dib_gemm_kernel<MODE,NI,NJ> at MODE*4 + (NI-1)*2 + (NJ-1); 12 = fused
 * encoder forward; 13 = fused encoder backward; 14 = all other (HBM-bound) kernels (not bracketed); 15 = dib_attn_fwd_kernel;
 * 16 = dib_attn_bwd_kernel (include/dib_st.h).  summary() synchronises. */
/* number of kernel launches the library has issued in this process so far (host counter, no synchronisation): the launch
 * inventory of a step = the difference around it */
int64_t dib_launch_count(void);
#define DIB_PROFILE_CATEGORIES 17
int dib_profile_enable(int on);
int dib_profile_summary(double* ms_by_category /*[17]*/, int* launches_by_category /*[17]*/);

/* ---- raw grouped GEMM (exposed for tests/benchmarks of the dominant kernel) ----------------
 * mode 0: C[M,N] = act(A[M,K] @ B[K,N] + bias)    mode 1: C[M,N] = (A[M,K] @ B[N,K]^T) * act'(aux)
 * mode 2: C[K... see DESIGN.md; single group, fp32 MFMA (v_mfma_f32_32x32x2_f32). */
int dib_gemm(int mode, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
             int ldc, const float* bias, const float* aux, int ldaux, int act, void* dev_desc /* >=128 B */,
             dib_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIB_HIP_H */
