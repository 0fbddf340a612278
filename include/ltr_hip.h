/*
 * ltr_hip.h -- C ABI of the MI355X (gfx950) ranking-loss / ranking-metric hot path.
 *
 * Drop-in boundary for rjagerman/pytorchltr's per-batch loss + evaluation path.  The
 * reference has no FFI on this path (it is pure PyTorch); its boundary is the Python
 * surface listed next to each entry point below (file:line relative to the upstream
 * tree, v0.2.1).  Each function here computes what that Python symbol computes, on
 * padded (B, L) row-major device buffers, without ever materialising the (B, L, L[, 2])
 * pair tensors the reference builds (utils/tensor_operations.py:94-119).
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. the PyTorch caching
 *     allocator); buffers are contiguous, row-major; nothing is allocated or freed here;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously on it,
 *     no host synchronisation, no globals, re-entrant, callable from any host thread;
 *   - return value: 0 = OK; < 0 = argument error (LTR_ERR_*); > 0 = a hipError_t from the
 *     launch.  ltr_error_string() renders either.
 *   - scores are fp32.  Labels (`rel`) may be int64 (what the reference's collate_fn emits,
 *     datasets/svmrank/svmrank.py:149-152), int32 or fp32, selected by `rel_dtype`; they
 *     are narrowed to fp32 in registers (exact for |y| < 2^24).  `n` is int64 (B).
 *   - n[b] is clamped to [0, L] exactly as the reference's masks act
 *     (`arange >= n`, utils/tensor_operations.py:25; `n_grid <= range_grid`,
 *     loss/pairwise_additive.py:75-81).
 *   - ranking ties: masked score descending, then document index ascending
 *     (deterministic replacement of the reference's global-RNG tie-break,
 *     utils/tensor_operations.py:43-45).
 */
#ifndef LTR_HIP_H
#define LTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Test and measurement hooks (ltr_debug_*): part of the default build -- the GPU test tier forces time-outs and fall-back
 * protocols through them and bench.py measures its launch ceilings with two of them -- and LEFT OUT of the library's
 * exports by a build with -DLTR_NO_DEBUG_HOOKS (LTR_NO_DEBUG_HOOKS=1 python -m pytorchltr_amd.build): the production build.
 */
#ifdef LTR_NO_DEBUG_HOOKS
#define LTR_DEBUG_HOOK __attribute__((visibility("hidden")))
#else
#define LTR_DEBUG_HOOK
#endif

#define LTR_VERSION 114 /* 0.1.14 */

/* Loss kinds: one per class exported by pytorchltr/loss/__init__.py:1-7. */
enum ltr_loss_kind {
    LTR_HINGE = 0,     /* PairwiseHingeLoss     loss/pairwise_additive.py:93-113  */
    LTR_DCG_HINGE = 1, /* PairwiseDCGHingeLoss  loss/pairwise_additive.py:116-133 */
    LTR_LOGISTIC = 2,  /* PairwiseLogisticLoss  loss/pairwise_additive.py:136-163 */
    LTR_ARP1 = 3,      /* LambdaARPLoss1        loss/pairwise_lambda.py:95-117    */
    LTR_ARP2 = 4,      /* LambdaARPLoss2        loss/pairwise_lambda.py:120-140   */
    LTR_NDCG1 = 5,     /* LambdaNDCGLoss1       loss/pairwise_lambda.py:143-173   */
    LTR_NDCG2 = 6      /* LambdaNDCGLoss2       loss/pairwise_lambda.py:176-218   */
};

enum ltr_label_dtype {
    LTR_LABEL_I64 = 0,
    LTR_LABEL_F32 = 1,
    LTR_LABEL_I32 = 2
};

#define LTR_OK 0
#define LTR_ERR_NULL (-1)        /* a required pointer is NULL                     */
#define LTR_ERR_SHAPE (-2)       /* B < 0, L <= 0, F <= 0 or k < 0                 */
#define LTR_ERR_KIND (-3)        /* unknown loss kind / label dtype                */
#define LTR_ERR_LIST_TOO_LONG (-4) /* L > ltr_max_list_len()                       */
#define LTR_ERR_WORKSPACE (-5)   /* workspace NULL or smaller than ltr_*_workspace_bytes */
#define LTR_ERR_CONFIG (-6)      /* invalid explicit launch configuration          */
#define LTR_ERR_TIMEOUT (-7)     /* a multi-workgroup kernel gave up waiting (see ltr_device_status) */

int ltr_version(void);
const char *ltr_error_string(int code);
/* Largest list_len one workgroup's LDS holds (all kinds, fp32 entry points). */
int ltr_max_list_len(void);
/* The same for the fp64 entry point (ltr_pairwise_loss_f64 keeps four double arrays in LDS). */
int ltr_max_list_len_f64(void);

/*
 * Sticky device status.  The kernels whose workgroups wait for each other inside one launch (the
 * cluster kernel behind ltr_linear_partials_f32 / ltr_linear_pairwise_f32 for long lists on small
 * batches) bound every wait; a wait that gives up poisons the query's outputs with NaN AND stores
 * LTR_ERR_TIMEOUT in a pinned, device-mapped status word of the process.  The ltr_linear_* entry
 * points return that code on the next call (a host read, no synchronisation); this function returns
 * the current word (0 = OK) and, with clear != 0, resets it.  Synchronise the stream first when the
 * answer must cover work that is still in flight.  The reference has no counterpart: its ops cannot
 * fail at run time.
 */
int ltr_device_status(int clear);
/* Tests only: != 0 makes every in-launch wait of the multi-workgroup kernels give up at once. */
LTR_DEBUG_HOOK void ltr_debug_force_timeout(int on);
/* Tests only: switches of the cluster kernel (long lists on small batches; LTR_CLUSTER_MODE=<bits> in the environment
 * does the same for a whole process).  Bit 0: treat every query's workgroups as spread over several XCDs, i.e. take the
 * write-through protocol the kernel falls back to when the placement check of a launch fails.  Bit 1: the hinge kinds by
 * the pair pass even where the labels are integer grades 0 .. 4 (which the kernel resolves by ranks). */
LTR_DEBUG_HOOK void ltr_debug_cluster_mode(int bits);
/* Tests only: != 0 lets the parts kernel (below) take every shape it CAN take instead of the shapes where it was
 * measured to pay (LTR_PARTS_ALL=1 in the environment does the same for a whole process); returns the old value. */
LTR_DEBUG_HOOK int ltr_debug_parts_all(int on);
/*
 * Exchange areas.  The fused-scorer kernels that spread a query over several workgroups (the parts kernel
 * behind ltr_linear_partials_f32 / ltr_linear_pairwise_f32 / ltr_linear_step_f32 for long lists and wide rows)
 * hand scores and gradient slices to each other through device memory as tagged 8-byte granules, next to a few
 * self-resetting counters.  That memory is NOT part of the caller's workspace: the library owns it -- one area
 * per (device, stream) for eager launches, a private one for every call made under stream capture (allocated
 * with the thread's capture mode relaxed; the allocation is not part of the captured work) --, zeroes it when
 * it is allocated and never hands it to anybody else, so the kernels never read uninitialised memory.  Areas
 * grow on demand and stay allocated; ltr_exchange_release() frees them all and may only be called when no
 * launch that used one is in flight and no captured graph containing one will be replayed again.  After a
 * launch that gave up (LTR_ERR_TIMEOUT) the areas are zeroed again before their next use, once the status has
 * been cleared with ltr_device_status(1).  The reference has no counterpart (one process, no device code).
 */
int ltr_exchange_release(void);
/* Tests only: the granule tag the NEXT launch on `stream`'s exchange area uses (tags run 1 .. 2^32 - 1 and
 * start over; the launch that uses the last one zeroes the granule buffers on its way out). */
LTR_DEBUG_HOOK int ltr_debug_set_exchange_tag(void *stream, unsigned tag);
/* Measurement aid (bench.py `roofline.launch_ceiling`): the launch geometry of the register-tile kernel --
 * one 512-thread workgroup per query, 16-byte buffer loads over the query's n[b] * F real floats, all in
 * flight at once -- with no computation behind the loads: every wave stores one dword to out
 * (B * 8 floats).  Shapes with L * F / 4 <= 19 * 512 vectors per query. */
LTR_DEBUG_HOOK int ltr_debug_stream_probe_f32(const float *X, const int64_t *n, int B, int L, int F, float *out, void *stream);
/* Measurement aid (bench.py `roofline` of the lazy step): the NEXT launch of the 512-thread register-tile kernel made by the
 * calling thread records the HIP events `start` right before and `stop` right behind the kernel (hipExtLaunchKernelGGL): the
 * kernel's own duration on its stream, as a profiler's kernel trace sees it. */
LTR_DEBUG_HOOK int ltr_debug_kernel_events(void *start /* hipEvent_t */, void *stop /* hipEvent_t */);
/* Tests / measurements only: which kernel layout ltr_mlp_pairwise_f32 takes where both apply.
 * 0 = automatic (the 4-wave tile kernel of csrc/ltr_mlp2.inc for batches of at least two queries per
 * CU-slot and for lists over 128 documents, else the 8-wave kernel of csrc/ltr_mlp.inc), 1 = the
 * 8-wave kernel wherever it applies, 2 = the tile kernel wherever it applies.
 * LTR_MLP_LAYOUT in the environment sets the initial value. */
LTR_DEBUG_HOOK void ltr_debug_mlp_layout(int layout);

/*
 * Seven pairwise losses, forward + analytic gradient in ONE pass.
 * Replaces _PairwiseAdditiveLoss.forward (loss/pairwise_additive.py:51-90) and
 * LambdaLoss.forward (loss/pairwise_lambda.py:50-92) together with the autograd graph
 * the reference replays for backward.
 *   loss[b]      = per-query loss (what forward() returns), fp32 (B)
 *   dscores[b,j] = d loss[b] / d scores[b,j], fp32 (B,L); may be NULL (forward only).
 *                  Entries j >= n[b] are written as 0.
 */
int ltr_pairwise_loss_f32(int kind, float sigma, const float *scores, const void *rel,
                          int rel_dtype, const int64_t *n, int B, int L, float *loss,
                          float *dscores, void *stream);

/* Same, with an explicit launch shape (tuning/tests): `owners` threads each own `dpt`
 * documents per chunk, the pair loop is split `msplit` ways; block = owners*msplit
 * threads.  owners % 64 == 0, dpt in {1,2,4}, owners*msplit <= 1024.  dpt == 0 selects the
 * symmetric pair pass (every unordered pair evaluated once; L <= 1024): owners*msplit threads. */
int ltr_pairwise_loss_f32_cfg(int kind, float sigma, const float *scores, const void *rel,
                              int rel_dtype, const int64_t *n, int B, int L, float *loss,
                              float *dscores, int owners, int dpt, int msplit, void *stream);

/* Same result as ltr_pairwise_loss_f32 for batches where one workgroup per query cannot balance
 * the machine (lists longer than 256 on a batch of at most ~2 queries per CU, e.g. 256 x 1000):
 * up to 8 workgroups share a query's pair work (for the NDCG kinds after one ranking pre-pass per
 * query), a finish kernel adds their parts in a fixed order (deterministic).  workspace: ltr_pairwise_loss_workspace_bytes(kind, B, L) bytes (0 = the plain path
 * is taken and workspace may be NULL).  Replaces the same reference symbols as
 * ltr_pairwise_loss_f32. */
size_t ltr_pairwise_loss_workspace_bytes(int kind, int B, int L);
int ltr_pairwise_loss_ws_f32(int kind, float sigma, const float *scores, const void *rel,
                             int rel_dtype, const int64_t *n, int B, int L, float *loss,
                             float *dscores, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Backward of the reference's per-query output w.r.t. scores: out[b,j] = grad_out[b] *
 * dscores[b,j] (what autograd computes for `loss.mean().backward()`, with grad_out = 1/B).
 * `out` may alias `dscores`.
 */
int ltr_scale_rows_f32(const float *dscores, const float *grad_out, int B, int L, float *out,
                       void *stream);
/* The same with ONE upstream gradient for every query, read from device memory: out = grad[0] *
 * dscores -- what `loss.mean().backward()` / `loss.sum().backward()` hand to the loss (autograd
 * passes an expanded, stride-0 tensor; this skips materialising it). */
int ltr_scale_rows_uniform_f32(const float *dscores, const float *grad_scalar, int B, int L,
                               float *out, void *stream);

/*
 * Double-precision instantiation of the two entry points above.  The reference computes in the
 * dtype of `scores` (fp64 in -> fp64 out), which `torch.autograd.gradcheck` users rely on.
 * Straightforward kernels, not tuned (fp64 VALU is half rate, no f64 transcendentals).
 */
int ltr_pairwise_loss_f64(int kind, double sigma, const double *scores, const void *rel,
                          int rel_dtype, const int64_t *n, int B, int L, double *loss,
                          double *dscores, void *stream);
int ltr_scale_rows_f64(const double *dscores, const double *grad_out, int B, int L, double *out,
                       void *stream);

/* rank_by_score, utils/tensor_operations.py:48-64 (mask_padded_values :6-26 +
 * tiebreak_argsort :29-45).  ranking[b,r] = index of the document at rank r (int64). */
int ltr_rank_by_score_f32(const float *scores, const int64_t *n, int B, int L,
                          int64_t *ranking, void *stream);

/*
 * dcg (evaluation/dcg.py:41-99) and, with normalize != 0, ndcg (evaluation/dcg.py:8-38).
 *   k > 0 : out is (B), metric@min(k,L);   k == 0 : out is (B,L), the metric at every rank.
 *   use_exp != 0 : gains 2^y - 1, else y.
 * As in the reference, labels of padded documents are NOT masked here.
 */
int ltr_dcg_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n, int B,
                int L, int k, int use_exp, int normalize, float *out, void *stream);

/* arp, evaluation/arp.py:7-42.  out is (B). */
int ltr_arp_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n, int B,
                int L, float *out, void *stream);

/*
 * The three ranking entry points above with the reference's RANDOM tie-break
 * (tiebreak_argsort, utils/tensor_operations.py:29-45: one permutation p = randperm(L) shared by
 * all rows decides the order of equal scores).  `tie` (L) int32 is a permutation of 0..L-1 drawn by
 * the caller: among documents with equal (masked) score the one with the SMALLER tie[j] ranks first.
 * tie == NULL is the deterministic index order of the plain entry points.  Rows without ties give
 * the same result for every `tie`.
 */
int ltr_rank_by_score_tie_f32(const float *scores, const int64_t *n, const int32_t *tie, int B, int L,
                              int64_t *ranking, void *stream);
int ltr_dcg_tie_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n,
                    const int32_t *tie, int B, int L, int k, int use_exp, int normalize, float *out,
                    void *stream);
int ltr_arp_tie_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n,
                    const int32_t *tie, int B, int L, float *out, void *stream);

/* The same three with the tie words made IN the kernel from a seed -- no permutation is drawn or shipped:
 * list position j gets the 31-bit word  hash19(seed, j) << 12 | j  (ltr_tie_hash_word; distinct for every
 * j < 4096), and of two documents with equal masked score the smaller word ranks first: one pseudo-random
 * permutation of the tied positions per call, shared by all rows, which is what tiebreak_argsort's
 * `p = randperm(L)` is for (utils/tensor_operations.py:43-45).  `seed_dev` (device int64[1]) overrides
 * `seed` when not NULL (a device generator's draw, no host round trip).  L <= 4096. */
int ltr_rank_by_score_seed_f32(const float *scores, const int64_t *n, uint64_t seed, const int64_t *seed_dev,
                               int B, int L, int64_t *ranking, void *stream);
int ltr_dcg_seed_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n, uint64_t seed,
                     const int64_t *seed_dev, int B, int L, int k, int use_exp, int normalize, float *out,
                     void *stream);
int ltr_arp_seed_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n, uint64_t seed,
                     const int64_t *seed_dev, int B, int L, float *out, void *stream);
uint32_t ltr_tie_hash_word(uint64_t seed, uint32_t position);     /* host helper: the word itself */

/*
 * Listwise softmax cross-entropy (ListNet top-one; named by the project brief, ABSENT from the
 * reference: pytorchltr/loss/__init__.py:1-7 exports no such class -- parity unpinned, the
 * specification is this header):
 *   P_y = softmax(y[b, :n]),  P_s = softmax(s[b, :n]),  loss[b] = -sum_j P_y(j) * ln P_s(j)
 *   dscores[b, j] = P_s(j) - P_y(j) for j < n[b], 0 after;  n[b] == 0 gives loss 0.
 */
int ltr_listwise_softmax_f32(const float *scores, const void *rel, int rel_dtype, const int64_t *n,
                             int B, int L, float *loss, float *dscores, void *stream);

/* mask_padded_values, utils/tensor_operations.py:6-26: out[b,j] = j >= n[b] ? mask_value
 * : xs[b,j].  `out` may alias `xs` (mutate=True). */
int ltr_mask_padded_values_f32(const float *xs, const int64_t *n, int B, int L,
                               float mask_value, float *out, void *stream);

/* batch_pairs, utils/tensor_operations.py:94-119: out[b,i,j,0] = x[b,i], out[b,i,j,1] =
 * x[b,j]; out is (B,L,L,2).  elem_bytes is 4 (fp32/int32) or 8 (int64/fp64): the copy is
 * dtype-agnostic.  Only for callers that want the materialised tensor -- the losses above
 * never build it. */
int ltr_batch_pairs(const void *x, int elem_bytes, int B, int L, void *out, void *stream);

/*
 * Plackett-Luce sampling keys (rank_by_plackettluce, utils/tensor_operations.py:67-91):
 *   keys[b,j] = log_softmax(masked scores)[b,j] - log(-log(u[b,j]))  for j < n[b], -inf after;
 * the sampled ranking is ltr_rank_by_score_f32(keys, n) (descending keys == ascending
 * log(-log u) - log p).  u is a caller-supplied uniform(0,1) tensor (B, L).
 */
int ltr_plackettluce_keys_f32(const float *scores, const int64_t *n, const float *u, int B, int L,
                              float *keys, void *stream);

/*
 * Position-based-model click simulator (simulate_pbm, click_simulation/pbm.py:12-63).
 * rankings (B,L) int64, ys (B,L) int64 labels, relevance_probs (n_probs) fp32 click probability
 * per label, u (B,L) uniform(0,1) indexed by RANK, cutoff < 0 = none, eta = position-bias
 * severity.  clicks (B,L) int64 in {0,1} and propensities (B,L) fp32, both in DOCUMENT order.
 */
int ltr_pbm_clicks(const int64_t *rankings, const int64_t *ys, const int64_t *n,
                   const float *relevance_probs, int n_probs, const float *u, int B, int L,
                   int cutoff, float eta, int64_t *clicks, float *propensities, void *stream);

/*
 * Device-side collate (the step immediately before the path): the dense branch of
 * SVMRankDataset.collate_fn, datasets/svmrank/svmrank.py:126-207.  Ragged storage
 *   xs (N, F) fp32, ys (N) int64, offsets (Q+1) int64 -- query q owns rows offsets[q]:offsets[q+1]
 * and a batch of query indices qidx (B) become the zero-padded batch
 *   out_x (B, L, F), out_y (B, L), out_n (B) = min(n_q, L).
 * sel (B, L) int64 or NULL: document indices *within the query* to gather for the first
 * min(n_q, L) slots (how a ListSampler truncates an over-long query, list_sampler.py:5-61);
 * NULL means the first min(n_q, L) documents.  L is chosen by the caller (the reference uses
 * max_i min(max_list_size, n_i)).
 */
int ltr_collate_pad_f32(const float *xs, const int64_t *ys, const int64_t *offsets,
                        const int64_t *qidx, const int64_t *sel, int Q, int B, int L, int F,
                        float *out_x, int64_t *out_y, int64_t *out_n, void *stream);
/* The sparse branch of the same collate (svmrank.py:162-176,197-202).  The split is CSR in device memory:
 * indptr (N + 1) int64 over ALL documents of the split, indices (nnz) int32 feature ids, values (nnz) fp32;
 * ys / offsets / qidx / sel / outputs as above.  The batch comes out DENSE and padded -- what the loss
 * kernels consume and what the reference's sparse batch is after .to_dense(); duplicate (row, column)
 * entries add up (torch coalesce()).  F * 16 bytes of LDS must fit a workgroup (F <= 10236). */
int ltr_collate_pad_csr_f32(const int64_t *indptr, const int32_t *indices, const float *values,
                            const int64_t *ys, const int64_t *offsets, const int64_t *qidx, const int64_t *sel,
                            int Q, int B, int L, int F, float *out_x, int64_t *out_y, int64_t *out_n, void *stream);

/*
 * Linear scorer fused with the loss: the caller of the path in every reference workflow,
 *   loss_fn(torch.nn.Linear(F,1)(xs), ys, n)  ... .backward()
 * (examples/01-basic-usage.py:70-75, tests/test_integration.py:42-52).
 *   X (B,L,F) fp32, W (F), bias (1, device)  ->  scores (B,L) [optional output], loss (B),
 *   dW (F) and db (1) = d( sum_b grad_out[b]*loss[b] ) / d{W, bias}.
 * grad_out (B) may be NULL, meaning 1/B for every query (`.mean().backward()`).
 * Rows l >= n[b] are never read (their gradient is 0) unless scores_out is requested; when
 * the (L x F) tile fits one workgroup's registers the features cross HBM exactly once.
 * `workspace` needs ltr_linear_workspace_bytes(B,L,F) bytes (per-query partials); it may be uninitialised
 * memory: everything in it is written before it is read (the kernels' cross-workgroup state lives in the
 * library's exchange areas, see ltr_exchange_release).
 */
size_t ltr_linear_workspace_bytes(int B, int L, int F);
/* Which kernel ltr_linear_partials_f32 / ltr_linear_pairwise_f32 take for this shape when no score
 * output is requested and X is 16-byte aligned: the register tile (features cross HBM once, one
 * workgroup per query), the cluster kernel (features once, a query spread over several workgroups:
 * long lists on small batches, rank-free kinds) or the general kernel (features twice). */
enum { LTR_PLAN_NONE = 0, LTR_PLAN_REGISTER_TILE = 1, LTR_PLAN_CLUSTER = 2, LTR_PLAN_GENERAL = 3, LTR_PLAN_PARTS = 4 };
int ltr_linear_fused_plan(int kind, int B, int L, int F);
int ltr_linear_pairwise_f32(int kind, float sigma, const float *X, const float *W,
                            const float *bias, const void *rel, int rel_dtype,
                            const int64_t *n, const float *grad_out, int B, int L, int F,
                            float *loss, float *scores_out, float *dW, float *db,
                            void *workspace, size_t workspace_bytes, void *stream);

/* The same step split at the autograd boundary (forward / backward of a fused
 * Linear+loss module): partials is (B, PF) row-major with PF = (F + 4) & ~3 floats per query --
 * partials[b, f] = d loss[b] / dW_f for f < F, partials[b, F] = d loss[b] / d bias, zero padding
 * after -- so that a query's row is one contiguous, 16-byte aligned store; then
 * dW_f = sum_b grad_out[b] * partials[b, f], db likewise (grad_out NULL = 1/B).
 * The `partials` buffer must be ltr_linear_workspace_bytes(B,L,F) bytes (the B * PF matrix,
 * rounded up, plus a reserved tail and, for some shapes, kernel scratch -- and from 4096 queries on the
 * scratch rows of the reduction, which is two launches whose work scales with B there: the reduce entry
 * points below WRITE behind the rows of such a batch, any L gives enough room).  Sums in a fixed order:
 * bit-identical run to run at every B. */
int ltr_linear_partials_f32(int kind, float sigma, const float *X, const float *W,
                            const float *bias, const void *rel, int rel_dtype,
                            const int64_t *n, int B, int L, int F, float *loss,
                            float *scores_out, float *partials /* (B, PF) */, void *stream);
int ltr_linear_reduce_f32(const float *partials, const float *grad_out, int B, int F, float *dW,
                          float *db, void *stream);
/* The same with ONE upstream gradient for every query, read from device memory: dW_f = scale[0] * sum_b partials[b, f]
 * -- what autograd hands the backward of `.mean()` / `.sum()` (an expanded scalar): no (B,) copy of it is made. */
int ltr_linear_reduce_bcast_f32(const float *partials, const float *scale /* device scalar */, int B, int F, float *dW,
                                float *db, void *stream);
/* Same reduction, additionally writing loss_sum[0] = sum_b loss[b] (the scalar a training loop
 * logs / all-reduces) in the same launch.  loss_sum may be NULL. */
int ltr_linear_reduce_loss_f32(const float *partials, const float *grad_out, const float *loss,
                               int B, int F, float *dW, float *db, float *loss_sum, void *stream);
/* The same with accumulate != 0: dW, db and loss_sum are ADDED to (gradient accumulation over
 * micro-batches before one all-reduce / optimizer step, as autograd's AccumulateGrad does). */
int ltr_linear_reduce_accum_f32(const float *partials, const float *grad_out, const float *loss,
                                int B, int F, float *dW, float *db, float *loss_sum, int accumulate,
                                void *stream);

/* --- the whole training-step slice in ONE call, and its gradient exchange -------------------------
 * ltr_linear_step_f32 = ltr_linear_partials_f32 + ltr_linear_reduce_accum_f32 behind one entry point:
 * what `loss_fn(Linear(F,1)(xs), ys, n)` -> `.backward()` costs a host thread per step is one call
 * instead of two (examples/01-basic-usage.py:66-75).  `bucket` is the step's flattened gradient
 * bucket, F + 2 floats [dW (F) | db | loss_sum]: the unit a data-parallel job all-reduces
 * (SURVEY.md section 8(e)); accumulate != 0 adds to it (gradient accumulation).
 *
 * With an overlap handle the call also runs the step's gradient all-reduce WITHOUT stalling the step
 * (one process per GPU, RCCL over xGMI; pytorchltr_amd.distributed.RcclOverlap builds the handle):
 *   1. `stream` waits for the all-reduce this slot's bucket was last given to (no host block),
 *   2. the step's kernels are launched on `stream`,
 *   3. an event is recorded behind them and the slot handed to the handle's helper thread, which makes
 *      the handle's side stream wait for that event and enqueues
 *      allreduce_fn(bucket, bucket, F + 2, ncclFloat32, ncclSum, comm, side_stream) -- it runs under the
 *      NEXT step's kernels (the caller rotates `depth` buckets / slots), and the host cost of the
 *      cross-stream dependency (6.5 us per wait on a just-recorded event on ROCm 7.2) is not on the
 *      thread that launches the steps.
 * depth = 0 makes an IN-STREAM handle: no side stream, no helper; the all-reduce is enqueued on `stream`
 * right behind the step's kernels (one ncclAllReduce call of host cost, the collective's latency on the
 * stream).
 * allreduce_fn has ncclAllReduce's signature; the library does not link RCCL, the caller passes the
 * function and the communicator (so the CPU-only build and the tests need no RCCL).
 * ltr_overlap_wait makes a stream wait for a slot's pending all-reduce (before an optimiser reads the
 * bucket), ltr_overlap_flush blocks the host until every pending all-reduce is done. */
typedef int (*ltr_allreduce_fn)(const void *sendbuf, void *recvbuf, size_t count, int datatype, int op,
                                void *comm, void *stream);
int ltr_overlap_create(ltr_allreduce_fn allreduce_fn, void *comm, int depth, void **handle);
int ltr_overlap_destroy(void *handle);
int ltr_overlap_wait(void *handle, int slot, void *stream);
int ltr_overlap_flush(void *handle);
int ltr_linear_step_f32(int kind, float sigma, const float *X, const float *W, const float *bias,
                        const void *rel, int rel_dtype, const int64_t *n, const float *grad_out,
                        int B, int L, int F, float *loss, float *bucket /* F + 2 */, int accumulate,
                        void *workspace, size_t workspace_bytes, void *overlap /* or NULL */, int slot,
                        void *stream);
/* One synchronous-SGD step of the reference's training loop in one call -- replaces
 *   loss = loss_fn(model(xs), ys, n).mean(); optimizer.zero_grad(); loss.backward(); optimizer.step()
 * with model = torch.nn.Linear(F, 1) and torch.optim.SGD(lr) (examples/01-basic-usage.py:66-75).  The step's
 * kernels write the bucket [dW (F) | db | loss_sum] (upstream gradient grad_out, NULL = 1 / B; data parallel:
 * 1 / (B * N)); with an IN-STREAM overlap handle (ltr_overlap_create(..., depth = 0)) the bucket is all-reduced
 * on `stream` right behind them; then W -= lr * dW and bias -= lr * db IN PLACE.  Step i + 1 scores with the
 * weights step i's all-reduced gradient produced: the collective is on the critical path, as in every
 * synchronous data-parallel SGD.  overlap handles with depth > 0 are refused (LTR_ERR_CONFIG).  At one rank
 * (overlap NULL) the update rides in the reduction kernel: two launches per step. */
int ltr_linear_sgd_step_f32(int kind, float sigma, const float *X, float *W, float *bias, const void *rel,
                            int rel_dtype, const int64_t *n, const float *grad_out, int B, int L, int F,
                            float lr, float *loss, float *bucket /* F + 2 */, void *workspace,
                            size_t workspace_bytes, void *overlap /* or NULL */, void *stream);
/* The same step with the update applied LAZILY (examples/01-basic-usage.py:66-75, one batch per call as the DataLoader hands
 * them over): the call computes this batch's per-query gradient rows into `workspace` and its losses into `loss`, and applies
 * the update of the PREVIOUS call's batch first -- pending_B (> 0) = the number of queries of that batch, whose rows and losses
 * `workspace` / `loss` still hold (the same buffers, the same F and lr) -- INSIDE this launch: its first workgroups reduce a
 * column each in front of their tile burst (the arithmetic of the reduction launch, bit for bit), update W / bias in place,
 * write the previous step's bucket [dW | db | loss_sum] and hand the new weights to every workgroup before the dot products.
 * The reduction launch and one kernel boundary per step are gone; W, bias, buckets and losses are bit-identical to
 * ltr_linear_sgd_step_f32 (grad_out = NULL).  The LAST batch's update is applied by ltr_linear_sgd_flush_f32 (also whenever
 * the weights are to be read between steps).  pending_B = 0: nothing pending (the first step).  `workspace` belongs to the lazy
 * entry points between a step and the flush (on the shapes it takes along the step keeps its rows column-group major); X must be
 * 16-byte aligned when F % 4 == 0 (LTR_ERR_CONFIG otherwise).  Shapes the register-tile
 * kernel does not take and streams under capture flush first and run the plain launch. */
int ltr_linear_sgd_lazy_step_f32(int kind, float sigma, const float *X, float *W, float *bias, const void *rel,
                                 int rel_dtype, const int64_t *n, int B, int L, int F, float lr, float *loss,
                                 float *bucket /* F + 2 */, void *workspace, size_t workspace_bytes, int pending_B,
                                 void *stream);
/* W -= lr * dW, bias -= lr * db of the batch whose rows `workspace` holds (pending_B queries of the same kind / L / F as the
 * lazy step that wrote them -- the step keeps its rows in a layout of its own on the shapes it takes along, which follows from
 * (kind, pending_B, L, F); 0: nothing to do), bucket = [dW | db | loss_sum]: the reduction launch of ltr_linear_sgd_step_f32. */
int ltr_linear_sgd_flush_f32(int kind, float *W, float *bias, int pending_B, int L, int F, float lr, const float *loss,
                             float *bucket /* F + 2 */, const void *workspace, void *stream);
/* The lazy step, DATA PARALLEL (one process per GPU, the queries of a global batch sharded over the ranks; the reference's loop
 * body examples/01-basic-usage.py:66-75 is single-process -- this is its synchronous data-parallel form): the same ONE launch per
 * step at every number of ranks.  `mailbox` is a connected ltr_mailbox_* handle created with count_max >= F + 2 (NULL, or a
 * one-rank mailbox: exactly ltr_linear_sgd_lazy_step_f32); pending_scale the weight of every pending query's gradient row --
 * 1 / (queries of the pending GLOBAL batch) makes the update the gradient of the global mean (<= 0: 1 / pending_B); with
 * pending_scale_dev query b's weight is READ from pending_scale_dev[b * pending_scale_stride] by the launch that applies the update
 * (the upstream gradient as autograd hands it over -- stride 0: `.sum()`'s expanded scalar, 1: `.mean()`'s vector of 1 / B or any
 * per-query weights; pytorchltr_amd.optim.SGD).  The reducer
 * workgroups in front of the launch sum this rank's rows of the pending batch, post {tag, sum} of their columns to every
 * peer's mailbox (its lazy half: the plain ltr_mailbox_allreduce calls are not disturbed), add what arrives IN RANK ORDER, and
 * publish the new weights to the launch; ONE workgroup (the losses' reducer) writes W / bias once every column has arrived -- all ranks
 * hold bit-identical weights after every step, and a peer that never shows up (ltr_mailbox_set_timeout_ms) leaves W / bias
 * untouched AS A WHOLE and raises LTR_ERR_TIMEOUT in the device status.  bucket = [dW | db | loss_sum] summed over the ranks.
 * Every rank makes the same sequence of calls with B > 0; eager launches only (a stream under capture: LTR_ERR_CONFIG when an
 * update is pending).  Shapes the launch does not take along flush first (below).  One process per GPU: a launch's reducers wait
 * for the peers' reducers INSIDE the kernel, so the ranks' launches must be resident at the same time -- ranks that share one
 * device (functional tests) must fit it together, B + (F + 4) / 4 + 1 workgroups each at four per CU. */
int ltr_linear_sgd_lazy_step_dp_f32(int kind, float sigma, const float *X, float *W, float *bias, const void *rel,
                                    int rel_dtype, const int64_t *n, int B, int L, int F, float lr, float *loss,
                                    float *bucket /* F + 2 */, void *workspace, size_t workspace_bytes, int pending_B,
                                    float pending_scale, const float *pending_scale_dev /* or NULL */,
                                    int pending_scale_stride, void *mailbox /* or NULL */, void *stream);
/* ltr_linear_sgd_flush_f32, data parallel: the pending batch's update with its all-reduce in ONE launch (the reducers on their
 * own; rows by query -- shapes off the register tile --: the reduction launch and the
 * plain mailbox all-reduce with the update riding in it). */
int ltr_linear_sgd_flush_dp_f32(int kind, float *W, float *bias, int pending_B, int L, int F, float lr, float pending_scale,
                                const float *pending_scale_dev /* or NULL */, int pending_scale_stride, const float *loss,
                                float *bucket /* F + 2 */, const void *workspace, void *mailbox /* or NULL */, void *stream);
/* The reduction alone: bucket = [dW | db | loss_sum] of the batch whose rows a lazy step left in `workspace` (its layout follows
 * from (kind, pending_B, L, F) and from whether the step had a bias), the weights untouched -- what reading `weight.grad` before
 * the optimizer step costs in the drop-in path (torch.Tensor.grad of examples/01-basic-usage.py:73). */
int ltr_linear_lazy_rows_reduce_f32(int kind, int pending_B, int L, int F, float pending_scale,
                                    const float *pending_scale_dev /* or NULL */, int pending_scale_stride, const float *loss,
                                    float *bucket /* F + 2 */, const void *workspace, int has_bias, void *stream);
/* Mailbox all-reduce: the < 3 KB gradient bucket of a data-parallel step summed over the ranks of ONE node by a
 * single small kernel per rank instead of a collective library (one process per GPU; no counterpart in the
 * reference, which is single-process).  Every rank owns a mailbox of 8-byte {tag, value} granules in fine-grained
 * device memory, mapped into its peers with hipIpcGetMemHandle / hipIpcOpenMemHandle; an all-reduce stores the
 * rank's granules into every peer's mailbox (over xGMI) and adds what arrives in its own IN RANK ORDER, so all
 * ranks obtain bit-identical sums.  Set-up: every rank calls ltr_mailbox_create (its IPC handle comes back in 64
 * bytes), the callers gather the handles (any transport), every rank calls ltr_mailbox_connect with all of them
 * in rank order.  ltr_mailbox_allreduce has ncclAllReduce's signature (ncclFloat32 / ncclSum only; comm = the
 * handle; count <= count_max), so it can be given to ltr_overlap_create(..., depth = 0); ltr_linear_sgd_step_f32
 * then applies the weight update inside the same kernel.  A rank that has not shown up after the time budget of one
 * all-reduce (120 s; LTR_MAILBOX_TIMEOUT_MS in the environment) turns into a NaN bucket and LTR_ERR_TIMEOUT in the
 * device status, not a hang -- and the WEIGHTS ARE NOT UPDATED by that step (they stay what they were: check
 * ltr_device_status before trusting W after a step that may have timed out).  ltr_mailbox_create fails with
 * LTR_ERR_CONFIG when fine-grained device memory is not available (no silent fall-back to memory whose stores a
 * peer might not see): callers then use a collective library.  world <= 16. */
int ltr_mailbox_create(int rank, int world, int count_max, void **handle, void *ipc_handle_out /* 64 bytes */);
int ltr_mailbox_connect(void *handle, const void *all_ipc_handles /* world x 64 bytes, rank order */);
int ltr_mailbox_destroy(void *handle);
/* The time budget of the mailbox's polls for the whole process (milliseconds, > 0; default 120 000 or LTR_MAILBOX_TIMEOUT_MS);
 * returns the previous value.  MailboxOverlap's set-up self-check runs under a short budget and restores it. */
long long ltr_mailbox_set_timeout_ms(long long timeout_ms);
int ltr_mailbox_allreduce(const void *sendbuf, void *recvbuf, size_t count, int datatype, int op, void *comm,
                          void *stream);
/* Tests only: timeout_ms > 0 sets the time budget of every later mailbox all-reduce of the process; tag >= 0 makes
 * `tag` the number of all-reduces this mailbox has done (the wrap 0xFFFFFFFF -> 2: every rank sets the same). */
LTR_DEBUG_HOOK int ltr_debug_mailbox_state(void *handle, long long timeout_ms, long long tag);
/* Tests only: an ltr_allreduce_fn that adds `comm` -- a device pointer to `count` floats, "the other rank's
 * bucket" -- to the buffer on `stream` (ncclFloat32 / ncclSum only). */
LTR_DEBUG_HOOK int ltr_debug_fake_allreduce(const void *sendbuf, void *recvbuf, size_t count, int datatype, int op, void *comm,
                             void *stream);

/* --- fused ReLU-MLP scorer + loss + backward --------------------
 * Replaces the user-side composition `loss_fn(model(xs), ys, n)` + `.backward()` with `model` the
 * feed-forward network of the reference's guide (docs/source/getting-started.rst:40-50:
 * Linear(F,H1) / ReLU / Linear(H1,H2) / ReLU / Linear(H2,1); training loop :70-101) by one MFMA
 * kernel plus a small deterministic reduction.
 *   X (B,L,F) fp32; W1 (H1,F), b1 (H1), W2 (H2,H1), b2 (H2), W3 (1,H2), b3 (1): torch.nn.Linear
 *   layout; grad_out (B) weights of the per-query losses (NULL = 1/B, i.e. `.mean()`).
 *   loss (B); scores_out (B,L) or NULL (written for documents < n[b] only);
 *   grads: ltr_mlp_param_count(F,H1,H2) floats = [dW1 | db1 | dW2 | db2 | dW3 | db3], the gradient
 *   of sum_b grad_out[b] * loss[b];  loss_sum[0] = sum_b loss[b] or NULL.
 * Shape limits of this kernel: L <= ltr_mlp_max_list_len(F) = 256 for F <= 144, 128 for wider rows
 * (LTR_ERR_LIST_TOO_LONG), F % 4 == 0, F <= 224, H1 <= 64, H2 <= 16 (LTR_ERR_SHAPE).
 * workspace: ltr_mlp_workspace_bytes(B,F,H1,H2) bytes. */
int ltr_mlp_max_list_len(int F);
size_t ltr_mlp_param_count(int F, int H1, int H2);
size_t ltr_mlp_workspace_bytes(int B, int F, int H1, int H2);
int ltr_mlp_pairwise_f32(int kind, float sigma, const float *X, const float *W1, const float *b1,
                         const float *W2, const float *b2, const float *W3, const float *b3,
                         const void *rel, int rel_dtype, const int64_t *n, const float *grad_out,
                         int B, int L, int F, int H1, int H2, float *loss, float *scores_out,
                         float *grads, float *loss_sum, void *workspace, size_t workspace_bytes,
                         void *stream);

/* Measurement aid (bench.py `extra.mlp_scorer_fused.launch_ceiling`): the launch geometry of the fused MLP training
 * step -- the tile kernel's grid and workgroups, the same 32-row fills requested, written to the LDS image and read
 * back, the same barriers and MFMA streams (80 forward + 88 backward MFMAs per fill and wave), the same assignment of queries to
 * workgroups -- with no labels, per-document layer-2/3 work, parking, pair pass or partial vectors: every wave stores one dword to
 * out (2 * #CUs * 4 floats at most).  F = 136 and L <= 128 only (the named batch); the numbers mean nothing. */
LTR_DEBUG_HOOK int ltr_debug_mlp_probe_f32(const float *X, const float *W1, const float *b1, const float *W2, const float *b2,
                            const float *W3, const float *b3, const int64_t *n, int B, int L, int F, int H1, int H2,
                            float *out, void *stream);

/* The same network, forward only: scores_out (B,L) = model(X) for documents < n[b], 0 for the
 * padded ones -- the `model(xs)` of the guide's evaluation loop (docs/source/getting-started.rst:
 * 117-127) without the rocBLAS round trips.  Same shape limits as ltr_mlp_pairwise_f32; no
 * workspace. */
int ltr_mlp_scores_f32(const float *X, const float *W1, const float *b1, const float *W2,
                       const float *b2, const float *W3, const float *b3, const int64_t *n, int B,
                       int L, int F, int H1, int H2, float *scores_out, void *stream);

/* --- the Linear(F, 1) scorer on its own ---------------------------------------------------
 * For the UNFUSED drop-in composition `loss_fn(model(xs), ys, n)` with `model` =
 * torch.nn.Linear(F, 1) (examples/01-basic-usage.py:45, tests/test_integration.py:30): on ROCm that
 * layer is a one-column rocBLAS GEMM far below HBM speed; these two kernels stream X once each.
 *   scores (B,L) = X.W + bias; documents >= n[b] score 0 and are not read (n may be NULL: all rows).
 *   dW_db (F+1) = [sum_r g[r] X[r,:] | sum_r g[r]] for the upstream gradient g (B,L) of the scores;
 *   rows with g == 0 or >= n[b] are not read.  F <= 1024 (scores).  workspace:
 *   ltr_linear_grad_workspace_bytes(B,L,F) bytes.  Deterministic (fixed-order partial sums). */
int ltr_linear_scores_f32(const float *X, const float *W, const float *bias, const int64_t *n, int B,
                          int L, int F, float *scores, void *stream);
size_t ltr_linear_grad_workspace_bytes(int B, int L, int F);
int ltr_linear_grad_f32(const float *X, const float *g, const int64_t *n, int B, int L, int F,
                        float *dW_db, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LTR_HIP_H */
