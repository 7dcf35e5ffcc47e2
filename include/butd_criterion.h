/*
 * butd_criterion.h -- C ABI of the fused pieces of the training criterion (gfx950).
 *
 * The reference's criterion (models/losses.py) is a few hundred tiny elementwise launches per step on
 * (7 prefixes x B scenes x 132 target slots) tensors.  These entry points evaluate its terms -- for ALL
 * prefixes at once, on the dense `match` of include/butd_lsap.h -- in one launch each, forward value and
 * gradient together (every term enters the total loss linearly, so the backward pass is a scaling):
 *
 *   butd_match_cost        HungarianMatcher cost tensor          losses.py:285-312
 *   butd_box_loss(+_bwd)   loss_bbox + loss_giou                 losses.py:392-418, 27-91
 *   butd_soft_token_ce     loss_labels_st                        losses.py:355-390
 *   butd_contrastive_rows / butd_contrastive_cols   loss_contrastive_align   losses.py:420-489
 *   butd_seed_objectness   compute_points_obj_cls_loss_hard_topk losses.py:161-223
 *
 * Layouts: P prefixes, B scenes, Q queries, G target slots, C classes, L tokens; everything row-major fp32
 * unless stated; `match` (P,B,G) int32, -1 = slot is not a target.  Device pointers, asynchronous launches
 * on `stream` (hipGraph-capturable); return 0 or a hipError_t.
 */
#ifndef BUTD_CRITERION_H
#define BUTD_CRITERION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *butd_stream_t;

/* cost[p][b][g][q] = w_bbox * L1(tgt[b][g], pred[p][b][q]) + w_class * class_cost[p][b][g][q]
 *                  + w_giou * (-GIoU(tgt[b][g], pred[p][b][q]))          (losses.py:301-312; boxes are
 * centre+size, sizes clamped at 1e-6 as box_cxcyczwhd_to_xyzxyz does).  class_cost may be NULL when
 * w_class == 0.  Rows of slots with valid[b][g] == 0 are written as zeros. */
int butd_match_cost(int P, int B, int Q, int G, const float *pred_boxes, const float *tgt_boxes,
                    const unsigned char *valid, const float *class_cost, float w_bbox, float w_class,
                    float w_giou, float *cost, butd_stream_t stream);

/* For every matched (p,b,g): src = pred_boxes[p][b][match], tgt = tgt_boxes[b][g];
 * sums[p][0] = sum |d centre| + 0.2 |d size|, sums[p][1] = sum (1 - GIoU)   (not yet divided by num_boxes);
 * grad[p][b][g][0..5] = d(l1 term)/d src, [6..11] = d(1 - GIoU)/d src (zeros for unmatched slots). */
int butd_box_loss(int P, int B, int Q, int G, const float *pred_boxes, const float *tgt_boxes,
                  const int *match, float *sums, float *grad, butd_stream_t stream);

/* grad_pred[p][b][q][:] = w[p][0] * grad[..][0..5] + w[p][1] * grad[..][6..11] at q = match[p][b][g], zero
 * elsewhere (grad_pred is written completely). */
int butd_box_loss_bwd(int P, int B, int Q, int G, const int *match, const float *grad, const float *w,
                      float *grad_pred, butd_stream_t stream);

/* Soft-token cross entropy (losses.py:355-390): row (p,b,q) of logits (P,B,Q,C) against target_sim =
 * positive_map[b][g][:C] if q = match[p][b][g] else one-hot(C-1); row_loss (P,B,Q) = weight * sum_c
 * (t log(t + 1e-6) - t log_softmax(x)_c), weight = 1 for matched rows, eos_coef otherwise;
 * dlogits (P,B,Q,C) = d row_loss / d logits.  positive_map (B,G,ldpm) with ldpm >= C. */
int butd_soft_token_ce(int P, int B, int Q, int G, int C, const float *logits, const int *match,
                       const float *positive_map, int ldpm, float eos_coef, float *row_loss,
                       float *dlogits, butd_stream_t stream);

/* Contrastive alignment (losses.py:420-489) on logits (P,B,Q,L) = proj_queries . proj_tokens^T / T.
 * positive[p][b][q][l] = positive_map[b][g][l] > 0 if q = match[p][b][g], else l in {last[b],
 * (last[b]-1) mod L}; last (B) int32 = attention_mask.sum(1) - 1.
 * _rows: "which tokens should each query match": row_loss (P,B,Q), dlogits (P,B,Q,L) WRITTEN, and
 *        owner (P,B,Q) int32 = matched slot of the query or -1 (input of _cols).
 * _cols: "which queries should each token match": col_loss (P,B,L), dlogits ACCUMULATED.
 * Both already carry the 1/2 of losses.py:488 and their weights (eos_coef for unmatched queries resp. for
 * tokens other than last[b]). */
int butd_contrastive_rows(int P, int B, int Q, int G, int L, const float *logits, const int *match,
                          const float *positive_map, int ldpm, const int *last, float eos_coef,
                          float *row_loss, float *dlogits, int *owner, butd_stream_t stream);
int butd_contrastive_cols(int P, int B, int Q, int G, int L, const float *logits, const int *owner,
                          const float *positive_map, int ldpm, const int *last, float eos_coef,
                          float *col_loss, float *dlogits, butd_stream_t stream);

/* Seed objectness (losses.py:161-223).  seed_xyz (B,K,3), seed_inds (B,K) int32 into point_instance_label
 * (B,N) int64 (< 0 = background; background seeds count as members of slot G-1, losses.py:175), gt_center /
 * gt_size (B,G,3), box_mask (B,G) fp32 (> 0 = real box), logits (B,K).  For every real box the `topk` seeds
 * with the smallest size-normalised distance among its member seeds are positives (non-members compete at
 * distance 100, ties to the lower seed index -- torch.topk leaves that order unspecified), background seeds are
 * never positive.  elem_loss (B,K) = sigmoid focal loss (alpha 0.25, gamma 2) / K, dlogits (B,K) = its
 * derivative; label (B,K) u8 is scratch + output.  topk <= 32. */
int butd_seed_objectness(int B, int K, int G, int N, int topk, const float *seed_xyz, const int *seed_inds,
                         const int64_t *point_instance_label, const float *gt_center, const float *gt_size,
                         const float *box_mask, const float *logits, unsigned char *label, float *elem_loss,
                         float *dlogits, butd_stream_t stream);

/* The weighting of compute_hungarian_loss (losses.py:592-617) in one launch: the (P,) per-prefix vectors of the four
 * terms (any of loss_ce / loss_giou / loss_align may be NULL) are summed and combined,
 *   out5 = [ w_gen * generation + w_sum * (sum ce + w_bbox * sum bbox + sum giou + sum align), sum ce, sum bbox,
 *            sum giou, sum align ];
 * the loss is NaN when any of the nstatus assignment status words (include/butd_lsap.h) is nonzero (scipy would have
 * raised there; inside a graph replay nothing can).  generation: device scalar or NULL.  P <= 64. */
int butd_loss_combine(int P, const float *loss_ce, const float *loss_bbox, const float *loss_giou,
                      const float *loss_align, const float *generation, const int *status_words, int nstatus,
                      float w_gen, float w_sum, float w_bbox, float *out5, butd_stream_t stream);

/* Its gradient for an upstream device scalar g: d term[i] = g * weight (NULL outputs are skipped); all zeros when any
 * status word is nonzero -- the stock expression torch.where(bad, nan, loss) sends no gradient into an invalid match, and
 * inside a captured step nobody can look at the loss before the optimizer runs. */
int butd_loss_combine_bwd(int P, const float *g, const int *status_words, int nstatus, float w_gen, float w_sum,
                          float w_bbox, float *d_ce, float *d_bbox, float *d_giou, float *d_align, float *d_generation,
                          butd_stream_t stream);

/* The whole tail of compute_hungarian_loss (losses.py:546-617) in ONE launch (round 5): per prefix p
 *   ce = sum ce_rows[p] / nb,  bbox = box_sums[p][0] / nb,  giou = box_sums[p][1] / nb,
 *   align = (sum align_rows[p] + sum align_cols[p]) / nb                  -> per_prefix (P,4)
 * (ce_rows (P, n_ce): butd_soft_token_ce's rows; box_sums (P,2): butd_box_loss; align_rows / _cols: butd_contrastive_rows /
 * _cols; nb = *num_boxes, a device scalar), generation = sum gen_elem / gen_div (butd_seed_objectness's elem_loss), and
 * butd_loss_combine's expression:  out6 = [loss, sum ce, sum bbox, sum giou, sum align, generation].  ce_rows, the align
 * pair and gen_elem may be NULL.  One workgroup, fixed order: bit-reproducible.  P <= 64. */
int butd_criterion_reduce(int P, const float *ce_rows, long n_ce, const float *box_sums, const float *align_rows,
                          long n_align_rows, const float *align_cols, long n_align_cols, const float *gen_elem,
                          long n_gen, float gen_div, const float *num_boxes, const int *status_words, int nstatus,
                          float w_gen, float w_sum, float w_bbox, float *per_prefix, float *out6,
                          butd_stream_t stream);

/* Its gradient for an upstream device scalar g, in one launch: with c = g w_sum / nb (0 when any status word is nonzero)
 *   d_logits = c dx_ce (n_ce floats),  d_align = c dx_align (n_align floats),  d_seed = (g w_gen / gen_div) dx_gen,
 *   box_w (P,2) = [c w_bbox, c]  -- the weights butd_box_loss_bwd takes.
 * dx_*: the derivatives of the row sums the forward kernels saved; NULL inputs are skipped. */
int butd_criterion_scale(int P, const float *g, const float *num_boxes, const int *status_words, int nstatus,
                         float w_gen, float w_sum, float w_bbox, float gen_div, const float *dx_ce, float *d_logits,
                         long n_ce, const float *dx_align, float *d_align, long n_align, const float *dx_gen,
                         float *d_seed, long n_gen, float *box_w, butd_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif
