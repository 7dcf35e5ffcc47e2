/*
 * butd_lsap.h -- C ABI of the batched Hungarian matching step of the criterion (gfx950).
 *
 * Replaces the host round trip of HungarianMatcher.forward (models/losses.py:306-320): the reference moves
 * the (B, Q, sum(n_b)) cost tensor to the host (`.cpu()`, one device sync per decoder prefix, 7 per step)
 * and calls scipy.optimize.linear_sum_assignment (scipy 1.7.3, environment.yml:92;
 * scipy/optimize/rectangular_lsap/rectangular_lsap.cpp: Crouse's rectangular shortest-augmenting-path
 * variant of Jonker-Volgenant) once per scene.  Here every (prefix, scene) problem is solved on the device
 * by one wavefront running the same algorithm in the same arithmetic (costs widened to double, duals in
 * double, rows visited in ascending order, the same tie rule on the column scan), so for a cost matrix
 * with a unique optimum -- and for the tie patterns the scan order decides -- the assignment is the one
 * scipy returns.  No host synchronisation: the call is an asynchronous launch on `stream`
 * (hipGraph-capturable).
 */
#ifndef BUTD_LSAP_H
#define BUTD_LSAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *butd_stream_t;

#define BUTD_LSAP_MAX_QUERIES 1024
#define BUTD_LSAP_MAX_TARGETS 1024

/* `count` independent problems.  cost (count, ng, nq) fp32: cost[p][g][q] = cost of giving target slot g
 * to query q -- the transpose of the reference's C[b] (queries x targets), i.e. the orientation scipy
 * works on internally when there are fewer targets than queries.  valid (count, ng) u8: target slots
 * that take part (the reference compacts them with box_label_mask, losses.py:558-565; ascending slot
 * order = the compacted order).  match (count, ng) i32: the query assigned to slot g, -1 for slots that are
 * not valid.  status (count) i32, may be NULL: 0 = solved, 1 = rejected as scipy rejects it (a NaN or
 * -inf entry in a valid row, more valid targets than queries, or no finite assignment): match is -1
 * for the whole problem.  Requires nq <= BUTD_LSAP_MAX_QUERIES, ng <= BUTD_LSAP_MAX_TARGETS. */
int butd_hungarian_match(int count, int nq, int ng, const float *cost, const unsigned char *valid,
                         int *match, int *status, butd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
