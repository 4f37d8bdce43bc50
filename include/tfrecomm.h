/* tfrecomm.h - C-ABI of libtfrecomm_hip.so: the MI355X (gfx950) implementation of the
 * SVD matrix-factorisation minibatch training step of jilljenn/TF-recomm.
 *
 * The reference has no native/FFI interface for this path: its boundary is the TensorFlow
 * 1.x Python API.  Each entry point below names the reference call it stands in for
 * (file:line into the reference tree); INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions across the boundary;
 *   - return 0 (TFR_OK) or a negative tfr_status; tfr_last_error() is thread-local text;
 *   - the library owns all device memory; host pointers are borrowed for the duration of
 *     the call and never freed by the library; outputs are caller-allocated;
 *   - one host thread drives one model; work is stream-ordered on the model's HIP stream;
 *     any call with a non-NULL host output pointer synchronises that stream before it
 *     returns; `_dev` entry points take device pointers, return without synchronising
 *     and report id errors at the next synchronising call;
 *   - ids are validated on the device (the reference relies on TensorFlow's CPU gather
 *     raising on out-of-range ids): an out-of-range id makes the step a no-op for every
 *     table and the next synchronising call returns TFR_ERR_OOB;
 *   - there is NO CPU fallback: without a usable HIP device tfr_create fails.
 */
#ifndef TFRECOMM_H
#define TFRECOMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFR_ABI_VERSION 3

typedef struct tfr_model tfr_model;

typedef enum {
    TFR_OK = 0,
    TFR_ERR_ARG = -1,     /* bad argument / unsupported shape */
    TFR_ERR_OOB = -2,     /* user or item id outside [0, user_num) / [0, item_num) */
    TFR_ERR_HIP = -3,     /* HIP runtime error (text in tfr_last_error) */
    TFR_ERR_STATE = -4,   /* call not valid in this state (e.g. resident step before upload) */
    TFR_ERR_NOMEM = -5
} tfr_status;

/* which-table ids for set/get/frozen.  Values 0..4 are the five trainables of
 * ops.py:8-12,29-32; +8 / +16 select the Adam first / second moment slot of that table. */
enum {
    TFR_MU = 0,           /* bias_global   []      ops.py:8     */
    TFR_BU = 1,           /* user_bias     [U]     ops.py:9-10  */
    TFR_BI = 2,           /* item_bias     [I]     ops.py:11-12 */
    TFR_P = 3,            /* user_features [U,D]   ops.py:29-30 */
    TFR_Q = 4,            /* item_features [I,D]   ops.py:31-32 */
    TFR_SLOT_M = 8,
    TFR_SLOT_V = 16
};

enum { TFR_LOSS_MSE = 0, TFR_LOSS_NLL = 1 };       /* ops.py:124 (canonical) / ops.py:125-126 (fork) */
enum { TFR_OPT_ADAM = 0, TFR_OPT_SGD = 1 };        /* ops.py:144,148 (canonical) / ops.py:145,149 (fork) */
enum { TFR_ADAM_TF1 = 0, TFR_ADAM_LAZY = 1 };      /* dense-moment TF1 semantics / touched rows only */

typedef struct tfr_opts {
    int32_t loss;         /* TFR_LOSS_*                                                    */
    int32_t item_abs;     /* 1: dot uses |item_features| (ops.py:44)                        */
    int32_t reg_bias;     /* 1: regulariser also has the two bias l2 terms (ops.py:85-89)   */
    int32_t optimizer;    /* TFR_OPT_*                                                     */
    int32_t adam_mode;    /* TFR_ADAM_*                                                    */
    int32_t device;       /* HIP device ordinal                                            */
    float lr;             /* learning_rate (ops.py:118)                                    */
    float reg;            /* reg = lambda  (ops.py:118,137)                                */
    float beta1, beta2, eps;  /* tf.train.AdamOptimizer defaults 0.9, 0.999, 1e-8          */
    int32_t reserved[5];  /* must be zero                                                  */
} tfr_opts;

/* ---- lifetime -------------------------------------------------------------------------- */
/* ops.inference_svd(..., user_num, item_num, dim) variable creation, ops.py:6-12,29-32.
 * Tables start at zero; the host side initialises and uploads them (tfr_set_table). */
int tfr_create(tfr_model** out, int64_t user_num, int64_t item_num, int32_t dim, const tfr_opts* opts);
int tfr_destroy(tfr_model* m);
void tfr_default_opts(tfr_opts* opts);

/* tf.global_variables_initializer() (svd_train_val.py:53,56) on the device: user/item features
 * ~ truncated normal(stddev 0.02), biases ~ truncated normal(stddev 1) (ops.py:9-12,29-32),
 * bias_global ~ U(-sqrt 3, sqrt 3) (TF's default glorot-uniform for a scalar); Adam slots,
 * global_step and the beta powers are reset.  The stream differs from TensorFlow's Philox,
 * so initial values are not a parity target. */
int tfr_init_tables(tfr_model* m, uint64_t seed, float feature_stddev, float bias_stddev);

/* ---- variables: tf.Variable.load / .eval and tf.train.Saver (svd_train_val.py:54,197-198) */
int tfr_set_table(tfr_model* m, int32_t which, const float* host, int64_t n);
int tfr_get_table(tfr_model* m, int32_t which, float* host, int64_t n);
/* var_list of Optimizer.minimize (ops.py:146-149; adaptive_test.py:28): bit (1<<TFR_x) set
 * = table x receives no update. */
int tfr_set_frozen(tfr_model* m, uint32_t mask);
/* global_step (svd_train_val.py:48, ops.py:119-120) and the Adam beta-power accumulators. */
int tfr_get_step(tfr_model* m, int64_t* step, float* beta1_power, float* beta2_power);
int tfr_set_step(tfr_model* m, int64_t step, float beta1_power, float beta2_power);
/* change lr / reg between steps (the reference rebuilds the graph for that; ops.py:118). */
int tfr_set_hyper(tfr_model* m, float lr, float reg);

/* ---- forward: sess.run([logits, infer], feed_dict) - svd_train_val.py:120-122; ops.py:13-14,37-47 */
int tfr_forward(tfr_model* m, const int32_t* user, const int32_t* item, int64_t batch,
                float* logits_out);
/* device-side validation metric: sum_k (infer_k - rate_k)^2 and count of infer==rate
 * (svd_train_val.py:144-149).  host id/rate pointers. */
int tfr_eval(tfr_model* m, const int32_t* user, const int32_t* item, const float* rate,
             int64_t batch, double* sum_sq_err_out, int64_t* n_equal_out);

/* the validation set kept in HBM (svd_train_val.py:33-38 feeds it whole, every epoch):
 * upload once, evaluate with no host data in the loop.  n_out = number of ratings. */
int tfr_upload_eval_triples(tfr_model* m, const int32_t* user, const int32_t* item, const float* rate,
                            int64_t n);
int tfr_eval_resident(tfr_model* m, double* sum_sq_err_out, int64_t* n_equal_out, int64_t* n_out);

/* the fork's epoch metrics on the device (svd_train_val.py:94-98,170-178: per-batch sklearn roc_auc_score and the fed-logits
 * NLL are the host bottleneck SURVEY 8f #1 names): count of round(sigmoid(logit)) == rate, summed sigmoid cross-entropy
 * (ops.py:125-126) and the AUC (rank sum over the radix-sorted logits, equal scores share their mean rank - what
 * roc_auc_score computes; NaN when one class is empty).  Binary-outcome model (loss = nll) only. */
int tfr_eval_binary(tfr_model* m, const int32_t* user, const int32_t* item, const float* rate, int64_t batch,
                    int64_t* n_equal_out, double* nll_sum_out, double* auc_out);
int tfr_eval_binary_resident(tfr_model* m, int64_t* n_equal_out, double* nll_sum_out, double* auc_out, int64_t* n_out);
/* roc_auc_score(rates, sigmoid(logits)) of the batch the last tfr_train_step ran on (svd_train_val.py:97) */
int tfr_last_batch_auc(tfr_model* m, double* auc_out);
/* roc_auc_score(label > 0.5, score) for device arrays */
int tfr_auc_dev(tfr_model* m, const float* d_score, const float* d_label, int64_t n, double* auc_out);

/* ---- one minibatch: sess.run([train_op, logits, infer], feed_dict) - svd_train_val.py:66-72;
 *      ops.py:81-89 (regulariser), ops.py:118-153 (loss, minimize).
 *      logits_out = pre-update logits of this batch; loss_out = data term only
 *      (ops.py:152-153); reg_out = regulariser value.  Any may be NULL. */
int tfr_train_step(tfr_model* m, const int32_t* user, const int32_t* item, const float* rate,
                   int64_t batch, float* logits_out, float* loss_out, float* reg_out);

/* ---- the same minibatch nsteps times: the inner loop of the per-user fine-tuning drivers - adaptive_test.py:104-116
 *      (EPOCH_MAX = 300 x sess.run(train_op) on the asked items of one user) and non_adaptive_test.py:82-87 (100 x on the
 *      user's history), usually with tfr_set_frozen = var_list=[user_bias, user_features] (adaptive_test.py:28).  The batch
 *      is uploaded once and the steps run with no host round trip in between.  logits_out[batch]: pre-update logits of the
 *      LAST step (what the last sess.run would fetch); loss_out[nsteps]: data term per step.  Either may be NULL.  Same
 *      result, bit for bit, as nsteps calls of tfr_train_step. */
int tfr_train_steps_repeat(tfr_model* m, const int32_t* user, const int32_t* item, const float* rate,
                           int64_t batch, int32_t nsteps, float* logits_out, float* loss_out);

/* ---- device-resident (user,item,rate) store: the feed of dataio.ShuffleIterator
 *      (dataio.py:98-103,114-117) kept in HBM; the host still draws the ids. */
int tfr_upload_triples(tfr_model* m, const int32_t* user, const int32_t* item, const float* rate,
                       int64_t n);
/* same, but the three columns are already in HBM (device pointers; copied into the library's
 * own 16-byte-record store, so the caller may free them after the call). */
int tfr_set_triples_dev(tfr_model* m, const int32_t* d_user, const int32_t* d_item,
                        const float* d_rate, int64_t n);
/* ids[step*batch + k] index the store: nsteps minibatches in one call.  loss_out[nsteps]
 * (data term per step) may be NULL (then the call does not synchronise). */
int tfr_train_steps_resident(tfr_model* m, const int64_t* ids, int64_t batch, int32_t nsteps,
                             float* loss_out);
/* upload pre-drawn ids once, then run steps [first, first+nsteps) from them with no host
 * data in the loop. */
int tfr_stage_ids(tfr_model* m, const int64_t* ids, int64_t n);
int tfr_train_steps_staged(tfr_model* m, int64_t first_step, int64_t batch, int32_t nsteps,
                           float* loss_out);
/* ---- the minibatch id draw itself on the device (dataio.py:113-117 `next`): NumPy's legacy global
 *      MT19937 stream (np.random.seed(13575), svd_train_val.py:15) and legacy randint's masked
 *      rejection, replayed bit for bit by one workgroup, so `ids = np.random.randint(0, N, (batch,))`
 *      never touches the host.  The state is what np.random.get_state() holds: key[624] and pos.
 *      high <= 2^32 (NumPy switches to 64-bit draws beyond that). */
int tfr_rng_seed(tfr_model* m, uint32_t seed);                               /* == np.random.seed(seed) */
int tfr_rng_set_state(tfr_model* m, const uint32_t* key624, int32_t pos);    /* np.random.get_state()[1:3] */
int tfr_rng_get_state(tfr_model* m, uint32_t* key624, int32_t* pos);         /* synchronises */
/* ids_out[count] (host) = np.random.randint(0, high, (count,)); advances the device state. */
int tfr_draw_ids(tfr_model* m, int64_t high, int64_t count, int64_t* ids_out);
/* the same draw into DEVICE memory, asynchronously on the draw stream: it starts once everything queued on the model's
 * stream so far has finished (which may still read d_ids_out), and tfr_join_draws makes the model's stream wait for every
 * draw issued so far.  For callers that run the step themselves, one call at a time (the data-parallel loop: every rank
 * draws the global batch's ids - the same stream on every rank - and takes its slice, dataio.py:115 with no host in it). */
int tfr_draw_ids_dev(tfr_model* m, int64_t high, int64_t count, int64_t* d_ids_out);
int tfr_join_draws(tfr_model* m);
/* ...for the first `ordinal` draws only (issued draws are counted from 1 and complete in issue order): a caller that keeps
 * several draws in flight waits for the one whose buffer it is about to read, not for the latest */
int tfr_join_draw(tfr_model* m, int64_t ordinal);
/* nsteps x { next(iter_train); sess.run(train_op) } (svd_train_val.py:66-72) with every part on the
 * device: ids drawn from randint(0, n_store_ratings) as above (on a side stream, ahead of the steps
 * that use them), rows gathered from the resident store, one training step each.  loss_out[nsteps]
 * may be NULL (then the call does not synchronise). */
int tfr_train_steps_drawn(tfr_model* m, int64_t batch, int32_t nsteps, float* loss_out);
/* host-drawn ids (the reference's own np.random.randint call) for ONE step on the resident store:
 * copied into a pinned ring slot, uploaded and trained on asynchronously - returns without waiting;
 * id errors surface at the next synchronising call. */
int tfr_train_step_ids(tfr_model* m, const int64_t* ids, int64_t batch);
/* forward over store rows [lo, hi): logits_out[hi-lo] host, may be NULL */
int tfr_forward_resident(tfr_model* m, int64_t lo, int64_t hi, float* logits_out);

/* ---- device-pointer entry points (plumbing for torch tensors / sharded multi-GPU) ------- */
int tfr_forward_dev(tfr_model* m, const int32_t* d_user, const int32_t* d_item, int64_t batch,
                    float* d_logits);
int tfr_train_step_dev(tfr_model* m, const int32_t* d_user, const int32_t* d_item,
                       const float* d_rate, int64_t batch, float* d_logits /* may be NULL */);
/* (item_features: the fused big-table step keeps updated rows in an alternate table until another reader asks - this call, like
 *  every reader, first brings them back; the pointer is current until the next big-table training step) */
int tfr_table_devptr(tfr_model* m, int32_t which, void** ptr, int64_t* n);
int tfr_set_stream(tfr_model* m, void* hip_stream);   /* NULL = the model's own stream; drains the stream in use first */
/* the same without draining: for a caller that alternates between two streams and orders them itself with events (the
 * row-sharded step prepares batch s+1 on a side stream while step s runs on the main one) */
int tfr_switch_stream(tfr_model* m, void* hip_stream);
int tfr_get_stream(tfr_model* m, void** hip_stream);
/* last step's device scalars {loss, reg, sum_g} without a host copy */
int tfr_scalars_devptr(tfr_model* m, void** ptr);

/* ---- row-sharded building blocks: the same kernels, split so the host can put
 *      an RCCL all-to-all between them.  The handle holds ONE rank's shard (user_num /
 *      item_num = local row counts); bias_global is replicated.  All pointers are device
 *      pointers; calls are asynchronous on the model's stream and NONE of them needs a host
 *      round trip: batch sizes that depend on the data stay on the device, the three exchanges
 *      use fixed-capacity buffers laid out [world][slot_cap] with rows of tfr_shard_row_stride()
 *      floats (dim features, the bias, padding to 16 bytes), so they are equal-split all-to-alls.
 *      Semantics per rank and step (ops.py:143-149 applied to the rows this rank owns):
 *        0. integer routing of the GLOBAL batch (identical on every rank): the samples whose user
 *           row this rank owns (owner = id / ceil(rows / world)), the distinct item ids among them
 *           grouped by owner into request slots (unused slots -1):          tfr_shard_route
 *        1. owners answer row requests:                                      tfr_shard_gather
 *        2. forward + backward on the rank's samples (user rows local, item rows addressed by
 *           slot); user rows are updated in place, item-row gradients (one per slot, already
 *           reduced over this rank's samples) are emitted in the exchange layout:
 *                                                                            tfr_shard_forward_reduce
 *        3. owners add the gradient rows received from all ranks (in rank order) and apply the
 *           optimiser to their item rows:                                    tfr_shard_apply_items
 *        4. with the all-reduced {loss, reg, sum_g}: bias_global update, beta powers,
 *           global_step:                                                     tfr_shard_finish_step
 *      More local samples than sample_cap, or more distinct items for one owner than slot_cap,
 *      void the step like an out-of-range id (TFR_ERR_OOB at the next synchronising call). */
int32_t tfr_shard_row_stride(tfr_model* m);
int tfr_shard_route(tfr_model* m, const int32_t* d_user, const int32_t* d_item, const float* d_rate, int64_t batch_global,
                    int32_t rank, int32_t world, int64_t user_num_global, int64_t item_num_global,
                    int32_t sample_cap, int32_t slot_cap, int32_t* d_req /* out [world * slot_cap] */);
/* the same routing with the global batch given as rows d_ids[0..batch_global) of this rank's copy of the rating store
 * (tfr_upload_triples / tfr_set_triples_dev with GLOBAL user / item ids): the ShuffleIterator gather (dataio.py:115-117)
 * happens inside the routing kernels and only the samples this rank owns leave the store. */
int tfr_shard_route_ids(tfr_model* m, const int64_t* d_ids, int64_t batch_global, int32_t rank, int32_t world,
                        int64_t user_num_global, int64_t item_num_global, int32_t sample_cap, int32_t slot_cap, int32_t* d_req);
/* pre-split batches (SURVEY 8e's other variant): every rank brings only ITS OWN batch rows d_ids[0..batch) of its store copy.
 * tfr_shard_bucket_ids groups their 16-byte records {user, item, rate bits, position} by the owner of the user row into
 * d_send = [world][pair_cap] records (batch order inside a group, unused slots have user = -1; overflow raises the error
 * flag); after one equal-split all-to-all of that buffer tfr_shard_route_recs routes what arrived (n = world * pair_cap
 * records, all owned by this rank) exactly as tfr_shard_route routes a global batch.  No rank ever looks at another rank's
 * samples it does not own: the id draw and the ownership test cost B per rank, not world * B. */
int tfr_shard_bucket_ids(tfr_model* m, const int64_t* d_ids, int64_t batch, int32_t world, int64_t user_num_global,
                         int32_t pair_cap, void* d_send);
int tfr_shard_route_recs(tfr_model* m, const void* d_recs, int64_t n, int32_t rank, int32_t world, int64_t user_num_global,
                         int64_t item_num_global, int32_t sample_cap, int32_t slot_cap, int32_t* d_req);
/* the routed batch, for the caller's bookkeeping and for tests: mine[sample_cap] global batch positions (-1 unused),
 * u_local[sample_cap], slot[sample_cap], counts = {local samples, distinct items, distinct items per owner [world]} */
int tfr_shard_routed_devptrs(tfr_model* m, void** mine, void** u_local, void** slot, void** counts);
int tfr_shard_gather(tfr_model* m, const int32_t* d_req_recv, int64_t n, float* d_rows_out /* [n, stride] */);
int tfr_shard_forward_reduce(tfr_model* m, const float* d_item_rows /* [world * slot_cap, stride] */,
                             float* d_logits /* [sample_cap], may be NULL */, float* d_item_grad /* out, same layout */,
                             float* d_scalars4 /* out: loss, reg, sum_g, - */);
/* the same step in two halves, so that a caller can run the user half BESIDE the gradient exchange (it reads the fetched rows and
 * the local user rows only): tfr_shard_forward_items = sort + forward + item-side reduce into d_item_grad + the local scalars;
 * tfr_shard_reduce_users = user-side reduce + optimiser on the local user rows.  forward_reduce == forward_items; reduce_users. */
int tfr_shard_forward_items(tfr_model* m, const float* d_item_rows, float* d_logits, float* d_item_grad, float* d_scalars4);
int tfr_shard_reduce_users(tfr_model* m, const float* d_item_rows);
int tfr_shard_apply_items(tfr_model* m, const int32_t* d_req_recv, const float* d_grad_recv, int64_t n);
/* Preparing batch s+1 while step s runs.  The model keeps TWO routed-batch sets; tfr_shard_select says which one the route_* calls
 * fill and the forward / reduce / apply calls consume.  tfr_shard_presort does the index work of a step ahead of time on the
 * selected set (it depends on the routed batch and on the requests received, never on a table): the routed samples sorted by
 * local user row and by request slot and - with d_req_recv [n], the requests this rank received as an owner - those requests
 * padded and sorted by item row; the step calls then skip their own sorts.  A caller puts bucket -> exchange -> route_recs ->
 * exchange -> presort of batch s+1 on a side stream (tfr_set_stream) beside step s; every buffer these calls write belongs to
 * the selected set or to the sort scratch, which the step calls then do not touch. */
int tfr_shard_select(tfr_model* m, int32_t which /* 0 or 1 */);
int tfr_shard_presort(tfr_model* m, const int32_t* d_req_recv /* may be NULL */, int64_t n);
int tfr_shard_finish_step(tfr_model* m, const float* d_scalars4 /* global sums */);

/* ---- data-parallel building blocks (replicated tables, small enough that every GPU holds
 *      them): each rank reduces ITS batch to dense gradient buffers, the host all-reduces one
 *      flat buffer (RCCL), every rank applies the same dense update - one `minimize`
 *      (ops.py:143-149) on the union of the ranks' batches.  Needs dense optimiser semantics
 *      (Adam tf1, which is what tf.train.AdamOptimizer computes, or SGD).
 *      d_flat: tfr_dp_flat_size() floats, device memory, zero before the first step:
 *      [user_features grads | item_features grads | user_bias | item_bias | loss, reg, sum_g, 0].
 *      d_store_ids != NULL: the batch is gathered from the resident store (ids index it). */
int64_t tfr_dp_flat_size(tfr_model* m);
int tfr_dp_local_grads(tfr_model* m, const int32_t* d_user, const int32_t* d_item, const float* d_rate,
                       int64_t batch, const int64_t* d_store_ids, float* d_flat);
/* Optional look-ahead for the call above: the store ids (device, same batch size) the NEXT
 * tfr_dp_local_grads will be given.  Their tile sort then rides in this step's launch instead of
 * heading the next step (the host knows the id stream: dataio.py:113-117 draws it).  One-shot. */
int tfr_dp_hint_next(tfr_model* m, const int64_t* d_next_store_ids);
int tfr_dp_apply(tfr_model* m, float* d_flat);
int tfr_staged_ids_devptr(tfr_model* m, void** ptr, int64_t* n);

/* ---- index work of the backward, exposed for bit-exact checks: stable sort of batch
 *      positions by row id (what tf.unique + unsorted_segment_sum's batch-order walk reduce
 *      to).  side 0 = user ids, 1 = item ids.  Host pointers; outputs [batch]. */
int tfr_sort_segments(tfr_model* m, int32_t side, const int32_t* ids, int64_t batch,
                      int32_t* sorted_ids_out, int32_t* sorted_pos_out);

/* ---- FM second-order forward on CSR rows (BASELINE config 5): forward.py:21-22
 *      y(x) = mu + x.W + 0.5 * (||x V||^2 - sum_j x_j^2 ||V_j||^2); design matrix fm.py:61-93.
 *      In the reference the model (mu, W, V) comes from the external libFM binary
 *      (fm.py:154-155, fm_mangaki.py:39-45); here it is uploaded or initialised on the device.
 *      CSR uses scipy.sparse's layout: indptr int64 [n_rows+1], indices int32, data f32. */
typedef struct tfr_fm tfr_fm;
/* opts (NULL = classification / SGD defaults): loss, optimizer (SGD or Adam - always the lazy,
 * touched-rows form), lr, reg, device.  The model is a regular model underneath: V = feature rows,
 * W = their bias column, mu = bias_global. */
int tfr_fm_create(tfr_fm** out, int64_t n_features, int32_t dim, const tfr_opts* opts);
int tfr_fm_destroy(tfr_fm* m);
int tfr_fm_set(tfr_fm* m, float mu, const float* W, const float* V);          /* host pointers */
int tfr_fm_get(tfr_fm* m, float* mu, float* W, float* V);                      /* any may be NULL */
int tfr_fm_init(tfr_fm* m, uint64_t seed, float stddev);                       /* random W, V on device */
int tfr_fm_forward(tfr_fm* m, const int64_t* indptr, const int32_t* indices, const float* data,
                   int64_t n_rows, float* out);                                /* host CSR, synchronous */
int tfr_fm_forward_dev(tfr_fm* m, const int64_t* d_indptr, const int32_t* d_indices,
                       const float* d_data, int64_t n_rows, float* d_out);     /* device CSR, async */
/* One minibatch of FM training.  NOT a restatement of reference code: the
 * reference trains this model in the external libFM binary by MCMC (fm.py:104-110,154-155).
 * Per non-zero (row r, feature j, x), s_r = x V, g_r = d loss / d y_r:
 *   dV_j += g_r x (s_r - x V_j) + reg V_j;  dW_j += g_r x + reg W_j;  dmu += g_r
 * applied with SGD or lazy Adam; deterministic (sorted segmented reduce, no atomics).
 * pred_out = predictions before the update, loss_out = data loss; either may be NULL. */
int tfr_fm_train_step(tfr_fm* m, const int64_t* indptr, const int32_t* indices, const float* data,
                      const float* y, int64_t n_rows, float* pred_out, float* loss_out);
int tfr_fm_train_step_dev(tfr_fm* m, const int64_t* d_indptr, const int32_t* d_indices,
                          const float* d_data, const float* d_y, int64_t n_rows, int64_t nnz,
                          float* d_pred /* may be NULL */);
int tfr_fm_sync(tfr_fm* m, float* last_kernel_ms);
const char* tfr_fm_last_error(void);

/* ---- ALS baseline (als3.py), float64 like the reference ----------------------
 *      MangakiALS3.fit (als3.py:20-35) = tfr_als_set (init_vars, als3.py:57-65, drawn by the host
 *      from np.random.rand) + tfr_als_load (bias = mean(y); per-user / per-work rating lists,
 *      als3.py:36-55) + tfr_als_sweep (fit_user / fit_work for every user then every work,
 *      als3.py:67-108); predict = tfr_als_predict at explicit pairs (als3.py:110-113).
 *      nb_components <= 32.  Host pointers; calls synchronise. */
typedef struct tfr_als tfr_als;
int tfr_als_create(tfr_als** out, int64_t nb_users, int64_t nb_works, int32_t nb_components,
                   double lambda_, int32_t device);
int tfr_als_destroy(tfr_als* m);
int tfr_als_set(tfr_als* m, const double* U, const double* V, const double* W_user, const double* W_work);
int tfr_als_get(tfr_als* m, double* U, double* V, double* W_user, double* W_work, double* bias);
int tfr_als_set_bias(tfr_als* m, double bias);
int tfr_als_load(tfr_als* m, const int64_t* user_ids, const int64_t* work_ids, const double* y, int64_t n);
int tfr_als_sweep(tfr_als* m, int32_t n_iterations, float* elapsed_ms /* may be NULL */);
int tfr_als_predict(tfr_als* m, const int64_t* user_ids, const int64_t* work_ids, int64_t n, double* out);
const char* tfr_als_last_error(void);

/* ---- per-kernel timing with HIP events on the model's stream (bench.py roofline) -------- */
enum {
    TFR_K_FORWARD = 0,        /* gather-dot forward (+ fused loss/grad when training)       */
    TFR_K_SORT = 1,           /* key sorts of both id columns                               */
    TFR_K_REDUCE_ITEM = 2,    /* deterministic segmented reduce, item side                  */
    TFR_K_REDUCE_USER = 3,    /* deterministic segmented reduce (+ fused lazy Adam), user   */
    TFR_K_APPLY = 4,          /* Adam / SGD apply kernels                                   */
    TFR_K_FINALIZE = 5,       /* scalar reduction + bias_global update                      */
    TFR_K_GATHER = 6,         /* resident-store triple gather                               */
    TFR_K_DRAW = 7,           /* MT19937 id draw (timed on its own side stream)             */
    TFR_K_COUNT = 8
};
int tfr_profile(tfr_model* m, int32_t enable);                 /* enable resets the counters */
/* "slot=kernel<template args>;..." - the kernels one training step of this model launches at this batch
 * size, spelled as rocprofv3 prints them (what the slots above time) */
int tfr_kernel_plan(tfr_model* m, int64_t batch, char* buf, int64_t buflen);
int tfr_profile_read(tfr_model* m, int32_t kernel, double* total_ms, int64_t* launches);

/* LDS budget guard: static + dynamic LDS bytes per workgroup that the dispatcher requests for a shape, computed
 * on the host exactly as the launchers compute it (no device needed).  kernel: 0 k_tile_step, 1 k_seg_reduce with
 * the forward inside, 2 k_front, 3 k_mt_draw, 4 k_seg_reduce.  A gfx950 CU has 160 KB. */
int tfr_lds_bytes(int32_t kernel, int32_t dim, int64_t batch, int64_t user_num, int64_t item_num,
                  int64_t* static_bytes, int64_t* dynamic_bytes);

/* ---- misc ------------------------------------------------------------------------------ */
int tfr_sync(tfr_model* m);            /* drains the stream; reports deferred TFR_ERR_OOB   */
const char* tfr_last_error(void);
int tfr_version(void);
int tfr_device_count(void);
/* Measurement yardstick (bench.py): GB/s (read + write bytes) of the library's own float4 copy kernel (one 16-byte element per
 * thread, streaming loads and stores - the fastest of the forms tools/probes/copy_bw.hip compares) over `bytes` of device
 * memory, best and mean of `reps` launches timed by HIP events.  Replaces no reference call. */
int tfr_device_copy_rate(int32_t device, int64_t bytes, int32_t reps, double* best_gbs, double* mean_gbs);

#ifdef __cplusplus
}
#endif
#endif /* TFRECOMM_H */
