// TEST INFRASTRUCTURE ONLY — never linked into the product (pyslam_amd/).
//
// extern "C" shim around the *unmodified* reference sources under
// /root/reference/cpp/volumetric, compiled where they lie by oracle/Makefile into
// oracle/_ref/libref_volumetric.so (git-ignored).  It exists so that tests/ and bench.py's
// cpu_baseline leg can (a) validate the C restatement in oracle/voxel_oracle.c against the real
// reference and (b) time the real reference on the host cores.
//
// Build mode: no TBB_FOUND (no TBB headers in this image) -> every `#ifdef TBB_FOUND` in the
// reference takes its sequential branch (voxel_block_grid.hpp:221-287, 457-461, 640-646, 785-817),
// i.e. deterministic point-index-order accumulation.
//
// What is wrapped (reference file:line):
//   VoxelBlockGridT<VoxelData>             cpp/volumetric/voxel_block_grid.h:61-234
//   integrate_raw<float,float|uint8_t>      cpp/volumetric/voxel_block_grid.hpp:115-136
//   get_voxels / get_voxels_in_bb / _in_camera_frustrum   .hpp:717-819, 822-1016, 1019-1195
//   carve                                  cpp/volumetric/voxel_grid_carving.h:47-79
//   remove_low_count_voxels, clear, size…  .hpp:625-646, 1543-1572
//   CameraFrustrum::contains               cpp/volumetric/camera_frustrum.cpp:175-196
//   get_voxel_key_inv/get_block_key/get_local_voxel_key/BlockKeyHash  voxel_hashing.h:51-161
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

#include "camera_frustrum.h"
#include "voxel_block_grid.h"
#include "voxel_hashing.h"

namespace {

using volumetric::BlockKey;
using volumetric::CameraFrustrum;
using volumetric::VoxelBlockGrid;

// Exposes the protected block map for a key-sorted dump.
class DumpableGrid : public VoxelBlockGrid {
  public:
    using VoxelBlockGrid::VoxelBlockGrid;
    const auto &blocks() const { return blocks_; }
    float inv_voxel_size() const { return inv_voxel_size_; }
};

CameraFrustrum make_frustum(const float *intr, int width, int height, const double *T_cw_rowmajor,
                            float depth_max, float depth_min) {
    Eigen::Matrix4d T;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            T(r, c) = T_cw_rowmajor[r * 4 + c];
    return CameraFrustrum(intr[0], intr[1], intr[2], intr[3], width, height, T, depth_max,
                          depth_min);
}

template <typename VG> int64_t copy_out(const VG &vg, float *pts, float *cols, int64_t cap) {
    const int64_t n = static_cast<int64_t>(vg.points.size());
    if (pts != nullptr && cols != nullptr) {
        const int64_t m = std::min(n, cap);
        for (int64_t i = 0; i < m; ++i) {
            for (int k = 0; k < 3; ++k) {
                pts[i * 3 + k] = vg.points[i][k];
                cols[i * 3 + k] = vg.colors[i][k];
            }
        }
    }
    return n;
}

} // namespace

extern "C" {

void *ref_grid_create(float voxel_size, int block_size) {
    return new DumpableGrid(voxel_size, block_size);
}
void ref_grid_destroy(void *g) { delete static_cast<DumpableGrid *>(g); }

// color_kind: 0 none, 1 uint8, 2 float32
void ref_grid_integrate(void *gv, const float *pts, int64_t n, const void *cols, int color_kind) {
    auto *g = static_cast<DumpableGrid *>(gv);
    if (color_kind == 0) {
        g->integrate_raw<float>(pts, static_cast<size_t>(n));
    } else if (color_kind == 1) {
        g->integrate_raw<float, uint8_t>(pts, static_cast<size_t>(n),
                                         static_cast<const uint8_t *>(cols));
    } else {
        g->integrate_raw<float, float>(pts, static_cast<size_t>(n),
                                       static_cast<const float *>(cols));
    }
}

// the binding's py::array_t<double> overload (volumetric_grid_module.h:738-741): integrate_raw<double, ...>
void ref_grid_integrate_f64(void *gv, const double *pts, int64_t n, const void *cols, int color_kind) {
    auto *g = static_cast<DumpableGrid *>(gv);
    if (color_kind == 0) {
        g->integrate_raw<double>(pts, static_cast<size_t>(n));
    } else if (color_kind == 1) {
        g->integrate_raw<double, uint8_t>(pts, static_cast<size_t>(n),
                                          static_cast<const uint8_t *>(cols));
    } else {
        g->integrate_raw<double, float>(pts, static_cast<size_t>(n),
                                        static_cast<const float *>(cols));
    }
}

int64_t ref_grid_num_blocks(void *g) {
    return static_cast<int64_t>(static_cast<DumpableGrid *>(g)->num_blocks());
}
int64_t ref_grid_size(void *g) {
    return static_cast<int64_t>(static_cast<DumpableGrid *>(g)->size());
}
int64_t ref_grid_total_voxel_count(void *g) {
    return static_cast<int64_t>(static_cast<DumpableGrid *>(g)->get_total_voxel_count());
}
int ref_grid_block_size(void *g) { return static_cast<DumpableGrid *>(g)->get_block_size(); }
int ref_grid_empty(void *g) { return static_cast<DumpableGrid *>(g)->empty() ? 1 : 0; }
void ref_grid_clear(void *g) { static_cast<DumpableGrid *>(g)->clear(); }
void ref_grid_remove_low_count(void *g, int min_count) {
    static_cast<DumpableGrid *>(g)->remove_low_count_voxels(min_count);
}

// Dumps all blocks sorted by (x,y,z) block key.  keys: B*3 int32; hashes: B uint64
// (BlockKeyHash); counts: B*bs^3 int32; sums: B*bs^3*6 float (pos_sum xyz, color_sum rgb), voxel
// order inside a block = the reference's flat index lx + ly*bs + lz*bs^2 (voxel_block.h:67-70).
// Any output pointer may be null; returns B.
int64_t ref_grid_dump(void *gv, int32_t *keys, uint64_t *hashes, int32_t *counts, float *sums) {
    auto *g = static_cast<DumpableGrid *>(gv);
    std::vector<const std::pair<const BlockKey, DumpableGrid::Block> *> order;
    order.reserve(g->blocks().size());
    for (const auto &kv : g->blocks())
        order.push_back(&kv);
    std::sort(order.begin(), order.end(), [](auto *a, auto *b) {
        const auto &ka = a->first;
        const auto &kb = b->first;
        if (ka.x != kb.x) return ka.x < kb.x;
        if (ka.y != kb.y) return ka.y < kb.y;
        return ka.z < kb.z;
    });
    const int bs = g->get_block_size();
    const size_t nv = size_t(bs) * bs * bs;
    volumetric::BlockKeyHash hasher;
    for (size_t b = 0; b < order.size(); ++b) {
        const auto &key = order[b]->first;
        const auto &blk = order[b]->second;
        if (keys) {
            keys[b * 3 + 0] = key.x;
            keys[b * 3 + 1] = key.y;
            keys[b * 3 + 2] = key.z;
        }
        if (hashes) hashes[b] = static_cast<uint64_t>(hasher(key));
        for (size_t i = 0; i < nv; ++i) {
            const auto &v = blk.data[i];
            if (counts) counts[b * nv + i] = v.count;
            if (sums) {
                float *s = sums + (b * nv + i) * 6;
                s[0] = v.position_sum[0];
                s[1] = v.position_sum[1];
                s[2] = v.position_sum[2];
                s[3] = v.color_sum[0];
                s[4] = v.color_sum[1];
                s[5] = v.color_sum[2];
            }
        }
    }
    return static_cast<int64_t>(order.size());
}

// Returns the number of voxels; fills up to cap rows when pts/cols are non-null.  Row order is the
// reference's unordered_map iteration order (compare as sets).
int64_t ref_grid_get_voxels(void *gv, int min_count, float min_confidence, float *pts, float *cols,
                            int64_t cap) {
    auto *g = static_cast<DumpableGrid *>(gv);
    const auto vg = g->get_voxels(min_count, min_confidence);
    return copy_out(vg, pts, cols, cap);
}

int64_t ref_grid_get_voxels_in_bb(void *gv, const double *bb /*min xyz, max xyz*/, int min_count,
                                  float min_confidence, float *pts, float *cols, int64_t cap) {
    auto *g = static_cast<DumpableGrid *>(gv);
    volumetric::BoundingBox3D bbox(bb[0], bb[1], bb[2], bb[3], bb[4], bb[5]);
    const auto vg = g->get_voxels_in_bb<false>(bbox, min_count, min_confidence);
    return copy_out(vg, pts, cols, cap);
}

int64_t ref_grid_get_voxels_in_frustum(void *gv, const float *intr, int width, int height,
                                       const double *T_cw, float depth_max, float depth_min,
                                       int min_count, float min_confidence, float *pts, float *cols,
                                       int64_t cap) {
    auto *g = static_cast<DumpableGrid *>(gv);
    const CameraFrustrum fr = make_frustum(intr, width, height, T_cw, depth_max, depth_min);
    const auto vg = g->get_voxels_in_camera_frustrum<false>(fr, min_count, min_confidence);
    return copy_out(vg, pts, cols, cap);
}

void ref_grid_carve(void *gv, const float *intr, int width, int height, const double *T_cw,
                    float depth_max, float depth_min, const float *depth, float depth_threshold) {
    auto *g = static_cast<DumpableGrid *>(gv);
    const CameraFrustrum fr = make_frustum(intr, width, height, T_cw, depth_max, depth_min);
    cv::Mat depth_mat(height, width, CV_32F, const_cast<float *>(depth));
    g->carve(fr, depth_mat, depth_threshold);
}

// CameraFrustrum::contains<float>; out = {u, v, depth}.  Returns 1 if inside.
int ref_frustum_contains(const float *intr, int width, int height, const double *T_cw,
                         float depth_max, float depth_min, const float *p_w, float *out) {
    const CameraFrustrum fr = make_frustum(intr, width, height, T_cw, depth_max, depth_min);
    const auto res = fr.contains<float>(p_w[0], p_w[1], p_w[2]);
    out[0] = res.second.u;
    out[1] = res.second.v;
    out[2] = res.second.depth;
    return res.first ? 1 : 0;
}

// Frustum AABB (min xyz, max xyz) as the reference computes it (camera_frustrum.cpp:209-264).
void ref_frustum_bbox(const float *intr, int width, int height, const double *T_cw, float depth_max,
                      float depth_min, double *bb) {
    const CameraFrustrum fr = make_frustum(intr, width, height, T_cw, depth_max, depth_min);
    const auto &b = fr.get_bbox();
    bb[0] = b.min_x; bb[1] = b.min_y; bb[2] = b.min_z;
    bb[3] = b.max_x; bb[4] = b.max_y; bb[5] = b.max_z;
}

// The rest of CameraFrustrum's bound surface (camera_frustrum_module.h:50-130) for one frustum: corners [8,3] (near / far per image
// corner, camera_frustrum.cpp:209-245), obb {center xyz, quaternion wxyz, size xyz} (:266-301), K [9], R_cw [9], t_cw [3],
// orientation_cw wxyz [4]; then for n float64 points (the binding's Eigen::Vector3d overloads): is_in_bbox, is_in_obb, contains and
// its ImagePoint {u, v, depth}.  quat != nullptr builds the frustum through the (orientation, translation) constructor instead.
void ref_frustum_surface(const float *intr, int width, int height, const double *T_cw, const double *quat_wxyz, const double *trans,
                         float depth_max, float depth_min, double *corners24, double *obb10, double *K9, double *R9, double *t3,
                         double *q4, const double *pts, int64_t n, uint8_t *in_bbox, uint8_t *in_obb, uint8_t *inside, float *uvd) {
    CameraFrustrum fr = quat_wxyz != nullptr
                            ? CameraFrustrum(intr[0], intr[1], intr[2], intr[3], width, height,
                                             Eigen::Quaterniond(quat_wxyz[0], quat_wxyz[1], quat_wxyz[2], quat_wxyz[3]),
                                             Eigen::Vector3d(trans[0], trans[1], trans[2]), depth_max, depth_min)
                            : make_frustum(intr, width, height, T_cw, depth_max, depth_min);
    const auto &cs = fr.get_corners();
    for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 3; ++k) corners24[i * 3 + k] = cs[i][k];
    const auto &b = fr.get_obb();
    obb10[0] = b.center.x(); obb10[1] = b.center.y(); obb10[2] = b.center.z();
    obb10[3] = b.orientation.w(); obb10[4] = b.orientation.x(); obb10[5] = b.orientation.y(); obb10[6] = b.orientation.z();
    obb10[7] = b.size.x(); obb10[8] = b.size.y(); obb10[9] = b.size.z();
    const Eigen::Matrix3d K = fr.get_K(), R = fr.get_R_cw();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { K9[r * 3 + c] = K(r, c); R9[r * 3 + c] = R(r, c); }
    const Eigen::Vector3d t = fr.get_t_cw();
    const Eigen::Quaterniond q = fr.get_orientation_cw();
    for (int k = 0; k < 3; ++k) t3[k] = t[k];
    q4[0] = q.w(); q4[1] = q.x(); q4[2] = q.y(); q4[3] = q.z();
    for (int64_t i = 0; i < n; ++i) {
        const Eigen::Vector3d p(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]);
        in_bbox[i] = fr.is_in_bbox(p) ? 1 : 0;
        in_obb[i] = fr.is_in_obb(p) ? 1 : 0;
        const auto res = fr.contains(p);
        inside[i] = res.first ? 1 : 0;
        uvd[i * 3] = res.second.u; uvd[i * 3 + 1] = res.second.v; uvd[i * 3 + 2] = res.second.depth;
    }
}

// Key arithmetic straight from voxel_hashing.h for n float32 points.
void ref_keys(float voxel_size, int block_size, const float *pts, int64_t n, int32_t *voxel_keys,
              int32_t *block_keys, int32_t *local_keys, uint64_t *block_hashes) {
    const float inv = 1.0f / voxel_size; // voxel_block_grid.hpp:6
    volumetric::BlockKeyHash hasher;
    for (int64_t i = 0; i < n; ++i) {
        const auto vk = volumetric::get_voxel_key_inv<float, float>(pts[i * 3], pts[i * 3 + 1],
                                                                    pts[i * 3 + 2], inv);
        const auto bk = volumetric::get_block_key(vk, static_cast<size_t>(block_size));
        const auto lk = volumetric::get_local_voxel_key(vk, bk, static_cast<size_t>(block_size));
        voxel_keys[i * 3 + 0] = vk.x; voxel_keys[i * 3 + 1] = vk.y; voxel_keys[i * 3 + 2] = vk.z;
        block_keys[i * 3 + 0] = bk.x; block_keys[i * 3 + 1] = bk.y; block_keys[i * 3 + 2] = bk.z;
        local_keys[i * 3 + 0] = lk.x; local_keys[i * 3 + 1] = lk.y; local_keys[i * 3 + 2] = lk.z;
        block_hashes[i] = static_cast<uint64_t>(hasher(bk));
    }
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// Semantic voting payload: VoxelBlockGridT<VoxelSemanticData> (voxel_data_semantic.h:106-202), the
// container part of VoxelBlockSemanticGrid (segment / object-association methods are not wrapped).
// ------------------------------------------------------------------------------------------------
#include "voxel_data_semantic.h"

namespace {
using SemGridBase = volumetric::VoxelBlockGridT<volumetric::VoxelSemanticData>;
class DumpableSemGrid : public SemGridBase {
  public:
    using SemGridBase::SemGridBase;
    const auto &blocks() const { return blocks_; }
};

template <typename Tpos, typename Tcolor>
void sem_integrate(DumpableSemGrid *g, const Tpos *pts, size_t n, const Tcolor *cols, const int *cls, const int *inst,
                   const float *depths) {
    if (inst != nullptr && depths != nullptr) {
        g->integrate_raw<Tpos, Tcolor, int, int, float>(pts, n, cols, cls, inst, depths);
    } else if (inst != nullptr) {
        g->integrate_raw<Tpos, Tcolor, int, int>(pts, n, cols, cls, inst);
    } else if (depths != nullptr) {
        g->integrate_raw<Tpos, Tcolor, std::nullptr_t, int, float>(pts, n, cols, cls, nullptr, depths);
    } else {
        g->integrate_raw<Tpos, Tcolor, std::nullptr_t, int>(pts, n, cols, cls);
    }
}
} // namespace

extern "C" {

void *ref_sgrid_create(float voxel_size, int block_size) { return new DumpableSemGrid(voxel_size, block_size); }
void ref_sgrid_destroy(void *g) { delete static_cast<DumpableSemGrid *>(g); }
void ref_sgrid_clear(void *g) { static_cast<DumpableSemGrid *>(g)->clear(); }
int64_t ref_sgrid_num_blocks(void *g) { return (int64_t) static_cast<DumpableSemGrid *>(g)->num_blocks(); }
void ref_sgrid_set_depth_threshold(float thr) { volumetric::VoxelSemanticData::kDepthThreshold = thr; }

// pos_kind: 0 float32, 1 float64; color_kind: 1 uint8, 2 float32 (colours are required with semantics)
void ref_sgrid_integrate(void *gv, const void *pts, int pos_kind, int64_t n, const void *cols, int color_kind,
                         const int32_t *class_ids, const int32_t *instance_ids, const float *depths) {
    auto *g = static_cast<DumpableSemGrid *>(gv);
    const size_t nn = static_cast<size_t>(n);
    if (pos_kind == 0 && color_kind == 1)
        sem_integrate<float, uint8_t>(g, (const float *)pts, nn, (const uint8_t *)cols, class_ids, instance_ids, depths);
    else if (pos_kind == 0)
        sem_integrate<float, float>(g, (const float *)pts, nn, (const float *)cols, class_ids, instance_ids, depths);
    else if (color_kind == 1)
        sem_integrate<double, uint8_t>(g, (const double *)pts, nn, (const uint8_t *)cols, class_ids, instance_ids, depths);
    else
        sem_integrate<double, float>(g, (const double *)pts, nn, (const float *)cols, class_ids, instance_ids, depths);
}

// key-sorted dump: keys [B,3]; ints [B,bs^3,4] = {count, object_id, class_id, confidence_counter};
// pos_sums [B,bs^3,3] f64; col_sums [B,bs^3,3] f32
int64_t ref_sgrid_dump(void *gv, int32_t *keys, int32_t *ints, double *pos_sums, float *col_sums) {
    auto *g = static_cast<DumpableSemGrid *>(gv);
    std::vector<const std::pair<const BlockKey, DumpableSemGrid::Block> *> order;
    for (const auto &kv : g->blocks()) order.push_back(&kv);
    std::sort(order.begin(), order.end(), [](auto *a, auto *b) {
        const auto &ka = a->first;
        const auto &kb = b->first;
        if (ka.x != kb.x) return ka.x < kb.x;
        if (ka.y != kb.y) return ka.y < kb.y;
        return ka.z < kb.z;
    });
    const int bs = g->get_block_size();
    const size_t nv = size_t(bs) * bs * bs;
    for (size_t b = 0; b < order.size(); ++b) {
        const auto &key = order[b]->first;
        const auto &blk = order[b]->second;
        if (keys) { keys[b * 3] = key.x; keys[b * 3 + 1] = key.y; keys[b * 3 + 2] = key.z; }
        for (size_t i = 0; i < nv; ++i) {
            const auto &v = blk.data[i];
            if (ints) {
                int32_t *d = ints + (b * nv + i) * 4;
                d[0] = v.count; d[1] = v.get_object_id(); d[2] = v.get_class_id(); d[3] = v.get_confidence_counter();
            }
            if (pos_sums) for (int k = 0; k < 3; ++k) pos_sums[(b * nv + i) * 3 + k] = v.position_sum[k];
            if (col_sums) for (int k = 0; k < 3; ++k) col_sums[(b * nv + i) * 3 + k] = v.color_sum[k];
        }
    }
    return (int64_t)order.size();
}

int64_t ref_sgrid_get_voxels(void *gv, int min_count, float min_confidence, double *pts, float *cols, int32_t *class_ids,
                             int32_t *object_ids, float *confidences, int64_t cap) {
    auto *g = static_cast<DumpableSemGrid *>(gv);
    const auto vg = g->get_voxels(min_count, min_confidence);
    const int64_t n = (int64_t)vg.points.size();
    if (pts != nullptr) {
        const int64_t m = std::min(n, cap);
        for (int64_t i = 0; i < m; ++i) {
            for (int k = 0; k < 3; ++k) { pts[i * 3 + k] = vg.points[i][k]; cols[i * 3 + k] = vg.colors[i][k]; }
            class_ids[i] = vg.class_ids[i];
            object_ids[i] = vg.object_ids[i];
            confidences[i] = vg.confidences[i];
        }
    }
    return n;
}

} // extern "C"
