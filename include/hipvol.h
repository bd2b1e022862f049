/*
 * hipvol.h — C ABI of libpyslam_hipvol.so, the MI355X-native (gfx950) volumetric fusion library.
 *
 * This is the drop-in boundary for pySLAM's dense-mapping hot path.  Each entry point states the
 * reference interface it replaces (paths relative to the pySLAM tree):
 *
 *   VOXEL_GRID mode  == the `volumetric` pybind11 module's VoxelBlockGrid
 *                       (cpp/volumetric/volumetric_grid_module.h:732-935, voxel_block_grid.h:61-234)
 *   TSDF mode        == open3d.pipelines.integration.ScalableTSDFVolume as pySLAM drives it
 *                       (pyslam/dense/volumetric_integrator_tsdf.py:104-108, 215-223, 239-267)
 *
 * Conventions
 *   - plain C types only; all pointers are borrowed for the duration of the call;
 *   - every function returns HV_OK (0) or a negative hv_status; hv_last_error() gives the message
 *     of the last failure on the calling thread (mirrors the std::runtime_error text the pybind
 *     module would raise, e.g. "points must be a contiguous Nx3 array");
 *   - `loc` says where the *input/output arrays* live: HV_HOST (the library stages them through
 *     HBM itself) or HV_DEVICE (already resident in HBM on the volume's device; zero-copy);
 *   - a volume is owned by one host thread at a time (pySLAM runs all volume access on the single
 *     thread of its VolumetricIntegratorProcess, volumetric_integrator_base.py:789-967);
 *   - all GPU work is enqueued on the volume's HIP stream; calls that return data to the host
 *     synchronise that stream, the integrate calls do not.
 *   - there is NO CPU fallback: hv_create fails if no gfx950 device is usable.
 */
#ifndef PYSLAM_HIPVOL_H
#define PYSLAM_HIPVOL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hv_volume hv_volume;

typedef enum hv_status {
    HV_OK = 0,
    HV_ERR_INVALID = -1,   /* bad argument (message mirrors the pybind module's) */
    HV_ERR_DEVICE = -2,    /* HIP runtime failure / no gfx950 device */
    HV_ERR_CAPACITY = -3,  /* block pool, hash table or scratch capacity exceeded */
    HV_ERR_MODE = -4       /* call not valid for this volume's mode */
} hv_status;

/* Values match pySLAM's VolumetricIntegratorType (pyslam/dense/volumetric_integrator_types.py:8-21). */
typedef enum hv_mode {
    HV_MODE_VOXEL_GRID = 0,
    HV_MODE_VOXEL_SEMANTIC_GRID = 1,               /* voting payload */
    HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID = 2, /* log-probability payload */
    HV_MODE_TSDF = 3,
    /* the two "*2" payload variants the reference's module also binds (volumetric_grid_module.h:1014-1032,
     * voxel_block_semantic_grid.h:120-123); pySLAM's VolumetricIntegratorType has no entry for them (4 is GAUSSIAN_SPLATTING) */
    HV_MODE_VOXEL_SEMANTIC_GRID2 = 11,               /* separate object / class counters, voxel_data_semantic2.h:46-196 */
    HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID2 = 12  /* marginal object / class label maps, voxel_data_semantic2.h:256-787 */
} hv_mode;
typedef enum hv_loc { HV_HOST = 0, HV_DEVICE = 1 } hv_loc;
typedef enum hv_color_dtype { HV_COLOR_NONE = 0, HV_COLOR_U8 = 1, HV_COLOR_F32 = 2 } hv_color_dtype;
typedef enum hv_depth_dtype { HV_DEPTH_F32 = 0, HV_DEPTH_U16 = 1 } hv_depth_dtype;

typedef struct hv_config {
    int32_t mode;          /* hv_mode */
    int32_t device;        /* HIP device ordinal */
    /* VOXEL_GRID: VoxelBlockGrid(voxel_size: float, block_size=8) — voxel_block_grid.hpp:4-9.
     * TSDF: ScalableTSDFVolume(voxel_length, sdf_trunc, RGB8, volume_unit_resolution=16,
     *       depth_sampling_stride=4) — volumetric_integrator_tsdf.py:104-108. */
    double voxel_size;     /* metres (narrowed to float in VOXEL_GRID mode, as pybind does) */
    double sdf_trunc;      /* TSDF only */
    int32_t block_size;    /* voxels per block side: 8 (VOXEL_GRID) / 16 (TSDF unit resolution) */
    int32_t depth_sampling_stride; /* TSDF only (Open3D default 4) */
    int64_t max_blocks;    /* block/unit pool capacity (HBM = max_blocks * bytes_per_block) */
    int64_t max_points;    /* largest point batch / H*W accepted by one integrate call */
} hv_config;

/* Camera: pinhole intrinsics {fx, fy, cx, cy} as doubles and T_cw (world->camera) as a row-major
 * 4x4 double — exactly what pySLAM hands over (keyframe.pose(), volumetric_integrator_base.py:116). */

const char *hv_last_error(void);
int hv_device_count(void);                       /* usable gfx950 devices, or negative hv_status */
void hv_default_config(int32_t mode, hv_config *cfg);

int hv_create(const hv_config *cfg, hv_volume **out);
void hv_destroy(hv_volume *v);
int hv_reset(hv_volume *v);                      /* VoxelBlockGrid.clear()/reset(), ScalableTSDFVolume.reset() */
int hv_synchronize(hv_volume *v);
int hv_set_stream(hv_volume *v, void *hip_stream); /* adopt a caller-owned hipStream_t (e.g. a torch stream) */
void *hv_get_stream(hv_volume *v);

/* ---- common introspection (volumetric_grid_module.h:761-812) ---------------------------------- */
int hv_num_blocks(hv_volume *v, int64_t *n);     /* num_blocks() / number of TSDF volume units */
int hv_block_size(hv_volume *v, int32_t *bs);    /* get_block_size() */
/* Grow the block pool + hash to new_max_blocks, keeping the contents (no-op when not larger).  The library also
 * doubles the pool by itself whenever a data-returning call finds it more than half full and the larger pool fits
 * into free HBM (HV_AUTO_GROW=0 disables this); the reference's containers grow without bound, so must the map. */
int hv_reserve_blocks(hv_volume *v, int64_t new_max_blocks);
int hv_max_blocks(hv_volume *v, int64_t *n);
int hv_bytes_per_block(hv_volume *v, int64_t *bytes);
int hv_dropped_points(hv_volume *v, int64_t *n); /* points/pixels rejected because their block key fell
                                                    outside the +/-2^20 packed range (never for sane maps) */

/* ---- VOXEL_GRID mode ---------------------------------------------------------------------------
 * hv_integrate_points == VoxelBlockGrid.integrate(points f32 [N,3], colors u8|f32 [N,3] | None)
 *   (volumetric_grid_module.h:743-749 -> integrate_raw, voxel_block_grid.hpp:115-136).
 *   Result is bit-identical to the reference's sequential (point-index-order) accumulation. */
int hv_integrate_points(hv_volume *v, const float *points, int64_t n, const void *colors,
                        int32_t color_dtype, int32_t loc);
/* The binding's float64 overload (volumetric_grid_module.h:738-741 -> integrate_raw<double, ...>): voxel keys come from the
 * doubles (floor(x * (double)inv_voxel_size)), the voxel sums take static_cast<float>(x) (voxel_data.h:54-56). */
int hv_integrate_points_f64(hv_volume *v, const double *points, int64_t n, const void *colors,
                            int32_t color_dtype, int32_t loc);

/* Fused L3 prep + integrate for one posed RGB-D frame: depth2pointcloud (pyslam/utilities/depth.py:
 * 45-85) + world transform (volumetric_integrator_voxel_grid.py:262-281) + integrate, without the
 * host round trip.  rgb is H*W*3 u8 (already RGB), depth H*W f32 metres (or u16 * depth_scale^-1).
 * Pixels with depth <= min_depth or >= max_depth are skipped (depth.py:64). */
int hv_integrate_rgbd_points(hv_volume *v, const void *depth, int32_t depth_dtype, double depth_scale,
                             const uint8_t *rgb, int32_t height, int32_t width, const double *intr,
                             const double *T_cw, double min_depth, double max_depth, int32_t loc);

/* Batched replay of F posed frames (rebuild(), offline reconstruction): depth F*H*W, rgb F*H*W*3, T_cw F*16 (host).
 * Bit-identical to F hv_integrate_rgbd_points calls; one sort + reduce per chunk of max_points / (H*W) frames. */
int hv_integrate_rgbd_points_batch(hv_volume *v, const void *depth, int32_t depth_dtype, double depth_scale,
                                   const uint8_t *rgb, int32_t n_frames, int32_t height, int32_t width, const double *intr,
                                   const double *T_cw, double min_depth, double max_depth, int32_t loc);

/* cv2.remap(src, map_x, map_y, INTER_LINEAR | INTER_NEAREST, BORDER_CONSTANT 0) as pySLAM's undistortion
 * uses it (volumetric_integrator_base.py:1017-1043).  src_kind 0: uint8 (channels interleaved), 1: float32,
 * 2: int32 (nearest only); maps float32 [H,W]; all arrays live at `loc`.  OpenCV semantics restated, unpinned. */
int hv_remap(hv_volume *v, const void *src, int32_t src_kind, int32_t channels, int32_t height, int32_t width,
             const float *map_x, const float *map_y, int32_t linear, void *dst, int32_t loc);

/* filter_shadow_points(depth, delta_depth=None, delta_x=2, delta_y=2, fill_value=-1)
 * (pyslam/utilities/depth.py:103-146): MAD-thresholded depth-discontinuity filter, bit-identical to
 * the numpy reference for float32 depth.  Works on any volume (uses its stream and scratch).  Host images (loc =
 * HV_HOST) are complete on return; with HV_DEVICE the launches are queued on the volume's stream (hv_get_stream) and the
 * call returns without waiting, like the integrate entry points. */
int hv_filter_shadow_points(hv_volume *v, const float *depth, int32_t height, int32_t width, int32_t delta_x,
                            int32_t delta_y, float fill_value, float *out, int32_t loc);
/* The same filter on DEVICE images, queued on the caller's HIP stream (`stream` = a hipStream_t) instead of the volume's, with
 * scratch of its own: the filter reads nothing of the volume, so a front can run it on its upload side - keyframe k + 1's depth is
 * filtered while keyframe k is fused (pyslam_amd/dense/device_pipeline.py).  The caller orders `out` against its consumers (an
 * event on `stream`).  Same reference: pyslam/utilities/depth.py filter_shadow_points, as called at
 * pyslam/dense/volumetric_integrator_voxel_grid.py:235 / volumetric_integrator_voxel_semantic_grid.py:334. */
int hv_filter_shadow_points_on_stream(hv_volume *v, const float *depth, int32_t height, int32_t width, int32_t delta_x,
                                      int32_t delta_y, float fill_value, float *out, void *stream);

/* get_voxels(min_count, min_confidence) (voxel_block_grid.hpp:717-819): rows = sum / count.
 * Pass points == NULL to query *n only.  At most `cap` rows are written; *n is the full count. */
int hv_get_voxels(hv_volume *v, int32_t min_count, float min_confidence, float *points,
                  float *colors, int64_t cap, int64_t *n, int32_t loc);
/* get_voxels_in_bb (voxel_block_grid.hpp:822-1016); bbox = {min x,y,z, max x,y,z}. */
int hv_get_voxels_in_bb(hv_volume *v, const double *bbox, int32_t min_count, float min_confidence,
                        float *points, float *colors, int64_t cap, int64_t *n, int32_t loc);
/* get_voxels_in_camera_frustrum (voxel_block_grid.hpp:1019-1195); CameraFrustrum(fx,fy,cx,cy f32,
 * width, height, T_cw, depth_max, depth_min) (camera_frustrum.h:37-130). */
int hv_get_voxels_in_frustum(hv_volume *v, const float *intr_f32, int32_t width, int32_t height,
                             const double *T_cw, float depth_max, float depth_min, int32_t min_count,
                             float min_confidence, float *points, float *colors, int64_t cap,
                             int64_t *n, int32_t loc);
/* carve(camera_frustrum, depth_image f32 HxW, depth_threshold) (voxel_grid_carving.h:47-79); VOXEL_GRID and both
 * semantic modes. */
int hv_carve(hv_volume *v, const float *intr_f32, int32_t width, int32_t height, const double *T_cw,
             float depth_max, float depth_min, const float *depth, float depth_threshold, int32_t loc);
int hv_remove_low_count_voxels(hv_volume *v, int32_t min_count); /* voxel_block_grid.hpp:625-646 */
int hv_size(hv_volume *v, int64_t *n);           /* size()/get_total_voxel_count(): voxels with count>0 */

/* Parity/debug export: all blocks sorted by (x,y,z) key.  keys [B,3] i32; hashes [B] u64 =
 * BlockKeyHash (voxel_hashing.h:106-113); counts [B,bs^3] i32; sums [B,bs^3,6] f32 (position_sum,
 * color_sum); voxel order inside a block = lx + ly*bs + lz*bs^2 (voxel_block.h:67-70).  Host
 * pointers; any may be NULL.  *n_blocks receives B. */
int hv_dump_blocks(hv_volume *v, int32_t *keys, uint64_t *hashes, int32_t *counts, float *sums,
                   int64_t *n_blocks);
/* K2 parity probe: key arithmetic of voxel_hashing.h:69-75,139-161 for N f32 points (host arrays). */
int hv_keys_from_points(hv_volume *v, const float *points, int64_t n, int32_t *voxel_keys,
                        int32_t *block_keys, int32_t *local_keys, uint64_t *block_hashes);

/* ---- semantic block grids ------------------------------------------------------------------------
 * HV_MODE_VOXEL_SEMANTIC_GRID               == volumetric.VoxelBlockSemanticGrid (voting payload,
 *                                              VoxelSemanticData, voxel_data_semantic.h:106-202)
 * HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID == volumetric.VoxelBlockSemanticProbabilisticGrid (log-probability
 *                                              payload, VoxelSemanticDataProbabilistic, voxel_data_semantic.h:249-672)
 * HV_MODE_VOXEL_SEMANTIC_GRID2               == volumetric.VoxelBlockSemanticGrid2 (VoxelSemanticData2: one confidence counter
 *                                              for the object id and one for the class id, voxel_data_semantic2.h:46-196)
 * HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID2 == volumetric.VoxelBlockSemanticProbabilisticGrid2 (VoxelSemanticDataProbabilistic2:
 *                                              one log-probability map per object id and one per class id, half of every
 *                                              observation's log-probability to each, voxel_data_semantic2.h:256-787)
 * (bindings: cpp/volumetric/volumetric_grid_module.h:939-1033; class: voxel_block_semantic_grid.h:57-123).  Every entry point of
 * this section takes the four modes; the reference itself documents the two *2 payloads as the inferior variants
 * (voxel_data_semantic.h:52-100) and nothing under pyslam/dense instantiates them.
 *
 * hv_integrate_points_semantic == .integrate(points f32|f64 [N,3], colors u8|f32, class_ids i32 [N] | None,
 *   instance_ids i32 [N] | None, depths f32 [N] | None)
 *   (volumetric_grid_module.h:131-467 -> integrate_raw -> update_voxel_direct, voxel_block_grid.hpp:524-614).
 *   point_dtype: 0 float32, 1 float64 (keys follow get_voxel_key_inv<Tpos,Tpos>).  Labels, confidence
 *   counters / log-probabilities, counts and float64 position sums are bit-identical to the reference's
 *   sequential (point-index) order.  The probabilistic payload's per-voxel label map (std::map in the reference) holds
 *   6 pairs in the voxel record and chains 10-pair nodes from a per-volume pool beyond them; a pair is dropped, and
 *   counted (hv_label_overflows), only past 254 pairs in one voxel or when the node pool is exhausted. */
int hv_integrate_points_semantic(hv_volume *v, const void *points, int32_t point_dtype, int64_t n, const void *colors,
                                 int32_t color_dtype, const int32_t *class_ids, const int32_t *instance_ids,
                                 const float *depths, int32_t loc);
/* Fused L3 prep + integrate for one posed RGB-D frame with label images (the per-keyframe body of
 * VolumetricIntegratorVoxelSemanticGrid, pyslam/dense/volumetric_integrator_voxel_semantic_grid.py:402-461):
 * depth2pointcloud(depth, rgb, ..., semantic_image, object_ids_image) + world transform + float32 casts +
 * integrate(points, colors, class_ids, instance_ids, depths = camera z when use_depths).  depth f32 metres HxW, rgb
 * u8 HxWx3, label images i32 HxW or NULL. */
int hv_integrate_rgbd_semantic(hv_volume *v, const float *depth, const uint8_t *rgb, const int32_t *class_ids_image,
                               const int32_t *object_ids_image, int32_t height, int32_t width, const double *intr,
                               const double *T_cw, double min_depth, double max_depth, int32_t use_depths, int32_t loc);
/* get_voxels(min_count, min_confidence) for semantic voxels: rows with count >= min_count and
 * confidence >= min_confidence; points f64 [M,3], colors f32 [M,3], class_ids/object_ids i32 [M],
 * confidences f32 [M] (host).  points == NULL: size query. */
int hv_get_voxels_semantic(hv_volume *v, int32_t min_count, float min_confidence, double *points, float *colors,
                           int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap, int64_t *n);
/* get_voxels_in_bb / get_voxels_in_camera_frustrum for semantic voxels (voxel_block_grid.hpp:822-1016, 1019-1195; the
 * label / confidence outputs are what IncludeSemantics=true adds).  Same output conventions as hv_get_voxels_semantic. */
int hv_get_voxels_semantic_in_bb(hv_volume *v, const double *bbox, int32_t min_count, float min_confidence, double *points,
                                 float *colors, int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap,
                                 int64_t *n);
int hv_get_voxels_semantic_in_frustum(hv_volume *v, const float *intr_f32, int32_t width, int32_t height, const double *T_cw,
                                      float depth_max, float depth_min, int32_t min_count, float min_confidence, double *points,
                                      float *colors, int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap,
                                      int64_t *n);
/* set_depth_threshold() (voxel_block_semantic_grid.hpp:24-30): voting: observations with depth >= threshold do
 * not vote (voxel_data_semantic.h:168-198); probabilistic: beyond it the evidence decays (:337-352).  Defaults
 * 10 m / 5 m.  The reference keeps these as process-wide statics; here they are per volume. */
int hv_set_depth_threshold(hv_volume *v, float depth_threshold);
/* set_depth_decay_rate() (voxel_block_semantic_grid.hpp:32-37), probabilistic payload; default 0.07 1/m. */
int hv_set_depth_decay_rate(hv_volume *v, float depth_decay_rate);
/* Label observations the probabilistic payload could not store (a voxel's map past 254 pairs, or the overflow-node pool exhausted):
 * 0 in every test and bench run; never silent. */
int hv_label_overflows(hv_volume *v, int64_t *n);
/* Overflow nodes of the probabilistic label maps handed out so far (of the pool's 4 per block; HV_PROB_NODE_CAP overrides).  A voxel
 * that is reset (carve, remove_low_*, remove_segment) keeps its chain and grows its next map into it, so the count is bounded by the
 * longest map every voxel ever held - it does not grow with the number of carve / re-observe cycles. */
int hv_prob_nodes_used(hv_volume *v, int64_t *n);
/* assign_object_ids_to_instance_ids(camera_frustrum, class_ids_image i32 HxW, semantic_instances_image i32 HxW,
 * depth_image f32 HxW | NULL, depth_threshold, do_carving, min_vote_ratio, min_votes)
 * (voxel_semantic_data_association.h:70-373).  Returns the instance -> object map sorted by instance id in
 * map_inst / map_obj (host, at most cap entries; *n_map = full size).  Mutates the volume (carving, object ids of
 * voxels that had none) exactly once per call: there is no size-query mode, pass cap >= the number of distinct
 * instance ids in the image.  New object ids come from a process-wide counter (hv_peek/set_next_object_id). */
int hv_assign_object_ids_to_instance_ids(hv_volume *v, const float *intr_f32, int32_t width, int32_t height, const double *T_cw,
                                         float depth_max, float depth_min, const int32_t *class_ids_image,
                                         const int32_t *instance_ids_image, const float *depth_image, float depth_threshold,
                                         int32_t do_carving, float min_vote_ratio, int32_t min_votes, int32_t *map_inst,
                                         int32_t *map_obj, int64_t cap, int64_t *n_map, int32_t loc);
/* The same association in stages, none of which waits for the GPU unless it hands data to the host (pySLAM's per-keyframe flow -
 * assign -> remap -> integrate, volumetric_integrator_voxel_semantic_grid.py:326-461 - then runs without a host round trip):
 *   hv_assoc_vote      per-voxel votes + the image's instance ids -> (instance << 32 | object, votes) pairs in device memory
 *   hv_assoc_decide    the reference's winner / min_votes / min_vote_ratio rules on the pairs (voxel_semantic_data_association.h:
 *                      268-361), new object ids from the process-wide counter (kept in device memory), deferred set_object_id
 *   hv_assoc_map_fetch the instance -> object map, sorted by instance id (synchronises; reports capacity overflows of the stages)
 *   hv_remap_instance_ids_last  remap_instance_ids (image_utils.h:69-163) with that map, read where it lies
 * Multi-GPU (block ownership, hv_set_owner): every GPU votes with its own voxels, hv_assoc_pairs_fetch -> all-gather of the pair
 * lists -> hv_assoc_pairs_set(concatenation) on every GPU -> hv_assoc_decide: identical maps and object ids everywhere. */
int hv_assoc_vote(hv_volume *v, const float *intr_f32, int32_t width, int32_t height, const double *T_cw, float depth_max,
                  float depth_min, const int32_t *class_ids_image, const int32_t *instance_ids_image, const float *depth_image,
                  float depth_threshold, int32_t do_carving, int32_t loc);
int hv_assoc_pairs_fetch(hv_volume *v, uint64_t *pair_keys, int32_t *pair_counts, int64_t cap, int64_t *n_pairs);
int hv_assoc_pairs_set(hv_volume *v, const uint64_t *pair_keys, const int32_t *pair_counts, int64_t n_pairs);
/* Multi-GPU exchange with the pair lists staying in device memory (no host round trip per keyframe): d_msg / d_msgs are DEVICE
 * buffers of the caller (torch tensors), int64 words: one GPU's message = [n, keys[cap], votes[cap]] (1 + 2 cap words, cap <= 4096);
 * hv_assoc_pairs_import takes `world` messages back to back - the output of an all-gather - and adds equal pairs up.  Both are
 * queued on the volume's stream and return at once. */
int hv_assoc_pairs_export(hv_volume *v, int64_t *d_msg, int64_t cap);
int hv_assoc_pairs_import(hv_volume *v, const int64_t *d_msgs, int32_t world, int64_t cap);
int hv_assoc_decide(hv_volume *v, float min_vote_ratio, int32_t min_votes);
int hv_assoc_map_fetch(hv_volume *v, int32_t *map_inst, int32_t *map_obj, int64_t cap, int64_t *n_map);
int hv_remap_instance_ids_last(hv_volume *v, const int32_t *instance_ids, int32_t height, int32_t width, int32_t *out, int32_t loc);
/* One semantic keyframe in one call, on DEVICE images, queued on the volume's stream (nothing waits): the per-keyframe body of
 * VolumetricIntegratorVoxelSemanticGrid (pyslam/dense/volumetric_integrator_voxel_semantic_grid.py:326-461) - filter_shadow_points
 * (:334, when filter_shadow_points != 0) -> assign_object_ids_to_instance_ids + remap_instance_ids (:340-372, when use_instance_ids
 * != 0 and instance_ids_image != NULL; a NULL class image gives the reference's empty map: every object id -1) or carve (:373-380,
 * when do_carving != 0) -> depth2pointcloud + world transform + integrate (:402-461).  The stages are the entry points above, in that
 * order; the filtered depth and the object-id image live in scratch of the volume.  frustum_*: the CameraFrustrum of the
 * association / carve (float32 intrinsics, camera_frustrum.h:37-130); intr: the float64 intrinsics of depth2pointcloud. */
int hv_semantic_fuse_keyframe(hv_volume *v, const float *depth, const uint8_t *rgb, const int32_t *class_ids_image,
                              const int32_t *instance_ids_image, int32_t height, int32_t width, const float *frustum_intr_f32,
                              float frustum_depth_max, float frustum_depth_min, const double *intr, const double *T_cw,
                              int32_t filter_shadow_points, int32_t use_instance_ids, float assoc_depth_threshold, int32_t do_carving,
                              float min_vote_ratio, int32_t min_votes, double min_depth, double max_depth, int32_t use_depths);
int32_t hv_peek_next_object_id(void); /* VoxelSemanticSharedData::next_object_id, voxel_semantic_shared_data.h:26-34 */
void hv_set_next_object_id(int32_t id);
/* volumetric.remap_instance_ids(instance_ids i32 HxW, map) (binding image_utils_module.h:49-94 over image_utils.h:69-163): ids
 * absent from the map become -1; an EMPTY map returns the image unchanged (the binding's early return, :52-58). */
int hv_remap_instance_ids(hv_volume *v, const int32_t *instance_ids, int32_t height, int32_t width, const int32_t *map_inst,
                          const int32_t *map_obj, int64_t n_map, int32_t *out, int32_t loc);
/* get_object_segments(min_count, min_confidence) (voxel_block_semantic_grid.hpp:217-267): voxels with
 * count > min_count (strict, as the reference), confidence >= min_confidence and object id >= 0, grouped by object
 * id.  _compute runs the query and caches the result on the host; _fetch copies it out: points f64 [R,3] and
 * colors f32 [R,3] grouped by ascending object id, row_object_ids i32 [R]; per object: object_ids i32 [O,3] =
 * {object_id, class_id, n_points}, confidences f32 [O,2] = {min, max}, obbs f64 [O,10] = {center xyz, quaternion
 * wxyz, size xyz} = OrientedBoundingBox3D::compute_from_points(PCA) (bounding_boxes_3d.cpp:373-553).  Any pointer
 * may be NULL. */
int hv_object_segments_compute(hv_volume *v, int32_t min_count, float min_confidence, int64_t *n_rows, int64_t *n_objects);
int hv_object_segments_fetch(hv_volume *v, double *points, float *colors, int32_t *row_object_ids, int32_t *object_ids,
                             float *confidences, double *obbs);
int hv_compute_obb_pca(const double *points, int64_t n, double *obb);
/* merge_segments / remove_segment / remove_low_confidence_segments (voxel_block_semantic_grid.hpp:119-196) and
 * remove_low_confidence_voxels (voxel_block_grid.hpp:650-676; a no-op on non-semantic volumes). */
int hv_merge_segments(hv_volume *v, int32_t instance_id1, int32_t instance_id2);
int hv_remove_segment(hv_volume *v, int32_t object_id);
int hv_remove_low_confidence_segments(hv_volume *v, int32_t min_confidence);
int hv_remove_low_confidence_voxels(hv_volume *v, float min_confidence);
/* Parity/debug export, key-sorted: keys [B,3]; ints [B,bs^3,4] {count, object_id, class_id, confidence_counter};
 * pos_sums [B,bs^3,3] f64; col_sums [B,bs^3,3] f32.  The second form adds conf [B,bs^3] f32 and, for the
 * probabilistic payload, label_counts [B,bs^3] (the size of each voxel's label map), labels [B,bs^3,max_labels,2]
 * {object, class} and log_probs [B,bs^3,max_labels]: each map's first max_labels pairs in insertion order. */
int hv_dump_blocks_semantic(hv_volume *v, int32_t *keys, int32_t *ints, double *pos_sums, float *col_sums,
                            int64_t *n_blocks);
int hv_dump_blocks_semantic2(hv_volume *v, int32_t *keys, int32_t *ints, float *conf, double *pos_sums, float *col_sums,
                             int32_t *label_counts, int32_t *labels, float *log_probs, int32_t max_labels,
                             int64_t *n_blocks);
/* The marginal confidences of the two "*2" payloads, [B,bs^3] f32 each in the dumps' order: get_object_confidence() /
 * get_class_confidence() (voxel_data_semantic2.h:60-76, 528-560); -1 everywhere for the other two payloads, which have none.
 * For HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID2 the labels of hv_dump_blocks_semantic2 are the entries of the voxel's two maps,
 * {id, which map (0 object, 1 class)}, in insertion order. */
int hv_dump_marginals_semantic(hv_volume *v, float *object_confidences, float *class_confidences, int64_t *n_blocks);

/* ---- TSDF mode ---------------------------------------------------------------------------------
 * hv_tsdf_integrate == RGBDImage.create_from_color_and_depth(color, depth, depth_scale,
 *   depth_trunc, convert_rgb_to_intensity=False) + volume.integrate(rgbd, intrinsic, T_cw)
 *   (volumetric_integrator_tsdf.py:215-223). */
int hv_tsdf_integrate(hv_volume *v, const void *depth, int32_t depth_dtype, const uint8_t *rgb,
                      int32_t height, int32_t width, const double *intr, const double *T_cw,
                      double depth_scale, double depth_trunc, int32_t loc);

/* Batched replay of F posed frames resident in HBM (rebuild(), volumetric_integrator_base.py:
 * 1242-1318): identical results to F successive hv_tsdf_integrate calls.  depth: F*H*W, rgb:
 * F*H*W*3, T_cw: F*16 (host array). */
int hv_tsdf_integrate_batch(hv_volume *v, const void *depth, int32_t depth_dtype, const uint8_t *rgb,
                            int32_t n_frames, int32_t height, int32_t width, const double *intr,
                            const double *T_cw, double depth_scale, double depth_trunc, int32_t loc);

/* The same for HOST-resident frames given one pointer per frame - what pySLAM's worker holds after draining its queue of
 * INTEGRATE tasks (volumetric_integrator_base.py:101-137: every keyframe carries its own pageable numpy arrays).  The frames
 * are copied to page-locked slots by worker threads and cross PCIe on a copy stream while the previous batch is swept; the
 * caller's memory has been read completely when the call returns.  depth_frames[f]: H*W of depth_dtype, rgb_frames[f]:
 * H*W*3 uint8, T_cw: F*16. */
int hv_tsdf_integrate_frames(hv_volume *v, const void *const *depth_frames, int32_t depth_dtype, const uint8_t *const *rgb_frames,
                             int32_t n_frames, int32_t height, int32_t width, const double *intr, const double *T_cw,
                             double depth_scale, double depth_trunc);

/* Undistort / rectify on the device.  The reference remaps every keyframe on the host before it is fused
 * (estimate_depth_if_needed_and_rectify, volumetric_integrator_base.py:1017-1043: cv2.remap colour INTER_LINEAR, depth INTER_NEAREST,
 * maps from cv2.initUndistortRectifyMap, :758-786).  With maps set (float32 [H,W], at `loc`; copied), every frame handed to
 * hv_tsdf_integrate / _batch / _frames goes through them first - one launch per batch, beside the batch's touch + pack launch, the
 * depth in its own storage type (float32 or uint16: 5 bytes per pixel for a TUM-style keyframe end to end) - and the caller passes
 * the RECTIFIED intrinsics.  map_x == NULL clears them.  OpenCV's remap semantics restated (hv_remap): unpinned. */
int hv_tsdf_set_rectify_maps(hv_volume *v, const float *map_x, const float *map_y, int32_t height, int32_t width, int32_t loc);

/* Channel order of the colour frames handed to hv_tsdf_integrate*: 0 = R, G, B (Open3D's RGBDImage, the default), 1 = B, G, R -
 * pySLAM's keyframe.img as OpenCV loads it; the reference converts every keyframe on the host with cv2.cvtColor
 * (volumetric_integrator_base.py:1054), here the pack kernel swaps the bytes of the record it writes anyway. */
int hv_tsdf_set_color_order(hv_volume *v, int32_t bgr);

/* Page-lock caller memory for the H2D DMA of hv_tsdf_integrate_frames / hv_integrate_*(HV_HOST): pySLAM's front hands keyframes
 * over in a shared-memory ring (volumetric_integrator_base.py:401-410 carries them pickled through a Manager queue); once the
 * ring is registered, frames that lie inside it are DMA'd in place - no staging copy - and the call returns when the DMA has
 * read them.  This holds for EVERY entry point that takes HV_HOST arrays: a page-locked source (registered here, or any other
 * pinned allocation) is read by the DMA engine when the stream gets to the copy, so those calls wait for their copies before they
 * return (hv_core.hip: hv_h2d); the kernels behind the copies stay asynchronous.  Process-wide (any volume of the process sees the
 * range); unregister before the memory is unmapped. */
int hv_host_register(void *ptr, int64_t bytes);
int hv_host_unregister(void *ptr);

/* Multi-GPU image-tile sharding (north-star form): this volume fuses only voxels whose
 * projection lands in pixel tile [u0,u1) x [v0,v1); units that cannot project into the tile are
 * allocated (so all GPUs agree on the unit set) but not swept.  All zeros = whole image (default). */
int hv_tsdf_set_tile(hv_volume *v, int32_t u0, int32_t v0, int32_t u1, int32_t v1);

/* The same for the VOXEL_GRID mode: owner(block) = hash(block key) mod world_size; a point whose block another GPU owns is
 * skipped (not counted as dropped).  The GPUs' voxel sets are disjoint and their union is the single-GPU grid bit for bit
 * (cpp/volumetric/voxel_block_grid.hpp:371-456 already treats blocks as independent).  hv_block_owner evaluates the ownership
 * function on the host for block keys [n,3] (-1 for keys outside the supported range). */
int hv_set_owner(hv_volume *v, int32_t rank, int32_t world_size);
int hv_block_owner(const int32_t *block_keys, int64_t n, int32_t world_size, int32_t *owner);

/* Multi-GPU unit-ownership sharding (SURVEY §8e "zero reduce" form): every GPU sees every frame but
 * claims, stores and fuses only the units with owner(unit index) == rank (a fixed hash of the
 * index modulo world_size).  Per-frame work and HBM footprint divide by world_size, results are
 * bit-identical to a single GPU, and no collective is needed while fusing.  (1 GPU: rank 0 of 1.) */
int hv_tsdf_set_owner(hv_volume *v, int32_t rank, int32_t world_size);

/* extract_triangle_mesh() (volumetric_integrator_tsdf.py:239,260).  vertices/vertex_colors f64
 * [V,3] (colours in [0,1]); triangles i32 [T,3].  NULL arrays = size query.  The destination arrays may be host memory
 * (pySLAM's viewer / PLY writer) or device memory of the volume's GPU (a GPU consumer: the 272 MB device-to-host copy of a
 * 32 k-unit mesh is the whole wall time of an output tick); the same holds for hv_tsdf_extract_points / _point_normals. */
int hv_tsdf_extract_mesh(hv_volume *v, double *vertices, double *vertex_colors, int64_t cap_vertices,
                         int32_t *triangles, int64_t cap_triangles, int64_t *n_vertices,
                         int64_t *n_triangles);
/* The same mesh with float32 vertices / vertex_colors: the float64 values of hv_tsdf_extract_mesh rounded once on the device
 * (numpy's astype(float32) of its arrays, bit for bit).  pySLAM's viewer and dense-map consumers take float32
 * (Parameters.kDenseMappingDtypeVertices / kDenseMappingDtypeColors, pyslam/config_parameters.py:290-291; the viewer casts,
 * pyslam/viz/viewer3D.py:1335-1342): a third fewer bytes across PCIe per output tick and half the vertex bytes written.
 * Opt-in: Open3D's arrays (and the reference's VolumetricIntegrationMesh, volumetric_integrator_base.py:209-214) are float64. */
int hv_tsdf_extract_mesh_f32(hv_volume *v, float *vertices, float *vertex_colors, int64_t cap_vertices,
                             int32_t *triangles, int64_t cap_triangles, int64_t *n_vertices,
                             int64_t *n_triangles);
/* extract_point_cloud() (volumetric_integrator_tsdf.py:246,267): points/colors f64 [N,3]. */
int hv_tsdf_extract_points(hv_volume *v, double *points, double *colors, int64_t cap, int64_t *n);
/* ... with float32 points / colors (the float64 rows rounded once, as hv_tsdf_extract_mesh_f32). */
int hv_tsdf_extract_points_f32(hv_volume *v, float *points, float *colors, int64_t cap, int64_t *n);
/* The normals Open3D's extract_point_cloud() attaches to those points (ScalableTSDFVolume::GetNormalAt: central differences of
 * the trilinearly interpolated tsdf at +/- 0.99 voxel, normalised); o3d.io.write_point_cloud stores them in dense_map.ply
 * (volumetric_integrator_tsdf.py:246-247).  normals f64 [N,3] in the order of hv_tsdf_extract_points; NULL to query *n. */
int hv_tsdf_extract_point_normals(hv_volume *v, double *normals, int64_t cap, int64_t *n);

/* Parity/debug export, units sorted by (x,y,z) index: keys [U,3] i32; tsdf, weight [U,R^3] f32;
 * color [U,R^3,3] f64 = running-mean RGB on the 0..255 scale; voxel order = Open3D's IndexOf
 * x*R^2 + y*R + z.  Host pointers; any may be NULL. */
int hv_tsdf_dump(hv_volume *v, int32_t *keys, float *tsdf, float *weight, double *color,
                 int64_t *n_units);
/* Unit indices touched by the most recent hv_tsdf_integrate, sorted; keys may be NULL. */
int hv_tsdf_touched(hv_volume *v, int32_t *keys, int64_t cap, int64_t *n);

/* Multi-GPU merge support: export/import additive numerators of the listed units.
 * keys [K,3] i32 host; payload [K, R^3, 5] f32 = {sum_tsdf_w, weight, sum_r, sum_g, sum_b}; units
 * absent from the volume export zeros; import REPLACES the unit's state from merged numerators
 * (allocating the unit if needed).  payload lives at `loc`. */
int hv_tsdf_export_numerators(hv_volume *v, const int32_t *keys, int64_t k, float *payload, int32_t loc);
int hv_tsdf_import_numerators(hv_volume *v, const int32_t *keys, int64_t k, const float *payload, int32_t loc);
/* All allocated unit keys (unsorted) for the key all-gather; keys may be NULL. */
int hv_tsdf_unit_keys(hv_volume *v, int32_t *keys, int64_t cap, int64_t *n);

/* ---- halo merge of image-tile-sharded volumes (SURVEY 8b `hv_merge_halo`, 8e; north-star "RCCL all-reduce of
 * overlapping-block TSDF/weight").  pySLAM has no multi-GPU path (no NCCL/MPI/torch.distributed call anywhere in the
 * reference): these are new.  With hv_tsdf_set_tile each GPU fuses the voxels that project into its image tile, so units
 * on tile borders (and revisits from other viewpoints) hold PARTIAL running means on several GPUs.  A merge:
 *   1. hv_tsdf_dirty_keys      every rank: the units it stamped since its last merge (sorted, 12 B per key)
 *   2. (caller) all-gather of the key lists
 *   3. hv_merge_halo_plan_held every rank, same input -> same output: the keys that some rank updated since its last merge
 *                              AND that two ranks or more hold (second all-gather: hv_tsdf_unit_keys), in sorted order, and
 *                              what THIS rank does with each: 1 = keeps it (the lowest holding rank), 2 = zeroes its copy (a
 *                              no-op where the rank has none).  (hv_merge_halo_plan: the same from the dirty lists alone -
 *                              keys listed by >= 2 ranks, lowest listing rank keeps, EVERY other rank zeroes: a unit that is
 *                              updated by one rank per window is never consolidated by it.)
 *   4. hv_merge_halo_pack      additive numerators {sum w*tsdf, w, sum r, sum g, sum b} of the shared units only, dense
 *                              [K, 16^3, 5] f32 in plan order (zeros where the rank does not hold a unit)
 *   5. (caller) all-reduce(sum) of that buffer - message size = shared units x 81 920 B, not the whole volume
 *   6. hv_merge_halo_unpack    keeper: state := reduced numerators; other holders: unit zeroed (they go on fusing deltas)
 *   7. hv_tsdf_mark_merged
 * Afterwards the sum over ranks of every unit's numerators is still the single-GPU total, and every unit that went through
 * the merge is complete on exactly one rank (with the _held plan: every unit two ranks hold).  The library does the device work; the caller owns the transport (pyslam_amd/distributed.py:
 * torch.distributed, backend nccl = RCCL over xGMI). */
int hv_tsdf_dirty_keys(hv_volume *v, int32_t *keys /* [cap,3] host, may be NULL */, int64_t cap, int64_t *n);
int hv_tsdf_mark_merged(hv_volume *v);
/* Host-only.  gathered_keys = the ranks' lists back to back ([sum counts, 3]); shared_keys/action may be NULL to query *n_shared. */
int hv_merge_halo_plan(const int32_t *gathered_keys, const int64_t *counts, int32_t world_size, int32_t rank,
                       int32_t *shared_keys, uint8_t *action, int64_t cap, int64_t *n_shared);
int hv_merge_halo_plan_held(const int32_t *dirty_keys, const int64_t *dirty_counts, const int32_t *held_keys,
                            const int64_t *held_counts, int32_t world_size, int32_t rank, int32_t *shared_keys, uint8_t *action,
                            int64_t cap, int64_t *n_shared);
int hv_merge_halo_pack(hv_volume *v, const int32_t *shared_keys, int64_t k, float *payload, int32_t loc);
int hv_merge_halo_unpack(hv_volume *v, const int32_t *shared_keys, int64_t k, const float *payload, const uint8_t *action,
                         int32_t loc);
/* The same merge with the key lists and the plan staying in DEVICE memory (round 6: the transport is RCCL, which gathers device
 * buffers; rounds 3-5 took the lists through the host twice per merge).  Keys are packed 64-bit words (the library's block key) in
 * int64 device buffers of the caller (torch tensors):
 *   1. hv_merge_halo_lists_device   this rank's dirty list and held list into d_dirty_keys / d_held_keys (NULL buffers: *n_held = an
 *                                   upper bound of both lengths, to size them); the two lengths come back to the host (they size the
 *                                   all-gather)
 *   2. (caller) all-gather of the counts and of the padded lists -> [world][stride] device buffers
 *   3. hv_merge_halo_plan_device    radix sort of (key, rank, held) + two small kernels: the plan of hv_merge_halo_plan_held, in
 *                                   packed-key order, left IN THE VOLUME; *n_shared comes back to the host (it sizes the payload).
 *                                   all_dirty_kept != 0: one rank takes every dirty unit through the path as its own keeper.
 *   4. hv_merge_halo_pack_planned / 6. _unpack_planned   units [first, first + count) of the stored plan, device payload, queued on the
 *                                   volume's stream (the caller orders its all-reduce against that stream)
 *   hv_merge_halo_plan_fetch        the stored plan as host arrays (inspection, tests). */
int hv_merge_halo_lists_device(hv_volume *v, int64_t *d_dirty_keys, int64_t dirty_cap, int64_t *d_held_keys, int64_t held_cap,
                               int64_t *n_dirty, int64_t *n_held);
int hv_merge_halo_plan_device(hv_volume *v, const int64_t *d_dirty_all, const int64_t *dirty_counts /* host [world] */, int64_t dirty_stride,
                              const int64_t *d_held_all, const int64_t *held_counts /* host [world] */, int64_t held_stride,
                              int32_t world_size, int32_t rank, int32_t all_dirty_kept, int64_t *n_shared);
int hv_merge_halo_plan_fetch(hv_volume *v, int32_t *shared_keys, uint8_t *action, int64_t cap, int64_t *n_shared);
int hv_merge_halo_pack_planned(hv_volume *v, int64_t first, int64_t count, float *d_payload);
int hv_merge_halo_unpack_planned(hv_volume *v, int64_t first, int64_t count, const float *d_payload);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------------
 * When enabled, the dominant kernel of each integrate call is bracketed by HIP events on the
 * volume's stream; hv_profile_read synchronises and returns the summed device time. */
int hv_profile_enable(hv_volume *v, int32_t on);
int hv_profile_read(hv_volume *v, double *kernel_ms_total, int64_t *kernel_launches,
                    int64_t *units_processed);
/* The same measurement launch by launch: durations (ms) of the bracketed launches since hv_profile_enable / the last
 * read, in issue order, into launch_ms[0 .. min(*n, cap)); *n = launches recorded.  Does not reset (hv_profile_read does). */
int hv_profile_read_launches(hv_volume *v, float *launch_ms, int64_t cap, int64_t *n);

#ifdef __cplusplus
}
#endif
#endif /* PYSLAM_HIPVOL_H */
