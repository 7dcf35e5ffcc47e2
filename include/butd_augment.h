/*
 * butd_augment.h -- C ABI of the device-side scene augmentation (gfx950): SURVEY.md section 8(f)-4.
 *
 * The reference augments every training sample on DataLoader workers in numpy
 * (src/joint_det_dataset.py:358-403 `_augment`, :595-607 detected boxes, :497-522 target boxes) and ships
 * 50 000 x 6 floats per scene to the GPU afterwards.  With the clouds resident in HBM the same arithmetic is
 * three small launches: the host only draws the per-scene parameters (a few dozen numbers).
 *
 * Arithmetic follows the reference's numpy dtypes step by step: clouds are float32 arrays that numpy updates
 * in place with float64 ARRAY operands, i.e. those steps are computed in double and ROUNDED TO FLOAT before the
 * next one, while the scale -- a Python float, a weak scalar for numpy -- multiplies in float32; boxes stay in
 * double until the final cast.
 */
#ifndef BUTD_AUGMENT_H
#define BUTD_AUGMENT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *butd_stream_t;

/* What `_augment` draws for one scene (joint_det_dataset.py:362-396).  rz / rx / ry are the row-major 3x3
 * matrices of rot_z / rot_x / rot_y (:930-966) for theta_z / theta_x / theta_y, built on the host in double. */
typedef struct {
  double rz[9], rx[9], ry[9];
  double shift[3];
  double scale;
  int32_t flip_yz, flip_xz; /* x -> -x, y -> -y (points: BEFORE the rotations, :365-372) */
} butd_scene_augment;

/* pc_in / pc_out (B, N, 3 + C) fp32: xyz, then C extra channels of which the first three are the
 * mean-subtracted colour when has_color != 0 (joint_det_dataset.py:414-415); other channels are copied.
 * params (B) on the device.  noise (B, N, 3) double = np.random.rand(N, 3) * 5e-3 of :386, color_gain
 * (B, N, 3) double = 0.98 + 0.04 * np.random.random((N, 3)) of :401; either may be NULL: the kernel then
 * draws them from a counter hash of (seed, scene, point, axis) (statistically, not bitwise, numpy's).
 * mean_rgb: the three doubles of :68.  pc_out may alias pc_in. */
int butd_augment_points(int B, int N, int C, int has_color, const float *pc_in,
                        const butd_scene_augment *params, const double *noise, const double *color_gain,
                        double mean_r, double mean_g, double mean_b, uint64_t seed, float *pc_out,
                        butd_stream_t stream);

/* Detected boxes (joint_det_dataset.py:595-607): boxes (B, D, 6) centre + size -> 8 corners -> rot z, x, y
 * -> flips (AFTER the rotations here, as the reference has it) -> shift -> scale -> axis-aligned hull ->
 * centre + size, for every slot including padding. */
int butd_augment_boxes(int B, int D, const float *boxes_in, const butd_scene_augment *params,
                       float *boxes_out, butd_stream_t stream);

/* Target boxes from the (augmented) cloud (joint_det_dataset.py:497-522 + visual_data_handlers.py:245-258):
 * for every instance id t in [0, G) the axis-aligned hull of the points with instance[b][n] == t, as
 * centre + size, times jitter (B, G, 6) double (0.95 + 0.1 * random, :516) when given; slots without points:
 * centre 1000, size 0, mask 0 (:518-520).  pc (B, N, ldp) fp32 (xyz first), instance (B, N) int64,
 * scratch: 6 * B * G uint32 of workspace. */
int butd_instance_boxes(int B, int N, int ldp, int G, const float *pc, const int64_t *instance,
                        const double *jitter, uint32_t *scratch, float *center_size, float *mask,
                        butd_stream_t stream);

/* The same target boxes for scenes that are RESIDENT in HBM with their objects as point lists (the reference keeps the
 * pickled Scan objects in host memory, joint_det_dataset.py:96-99, and every __getitem__ walks
 * scan.three_d_objects[tid]['points'], :507-512): obj_points = every object's point indices of every scene, one after
 * the other; obj_ptr[s * ptr_stride + k] = where object k of scene s starts (k = 0 .. number of objects).  For sample b
 * (scene scene[b]) and slot t < G with target_ids[b][t] >= 0: point_instance_label[b][p] = t for the object's points
 * (point_instance_label (B, N) int64, pre-filled with -1 by the caller, may be NULL; a point of several target objects
 * keeps the LAST slot, :507-508) and the hull of all the object's points in pc (B, N, ldp) as centre + size times
 * jitter; empty / absent slots: centre 1000, size 0, mask 0.  scratch: 6 * B * G uint32. */
int butd_object_boxes(int B, int N, int ldp, int G, const int *scene, const long long *obj_ptr, long long ptr_stride,
                      const int *obj_points, const int *target_ids, const float *pc, const double *jitter,
                      long long *point_instance_label, uint32_t *scratch, float *center_size, float *mask,
                      butd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
