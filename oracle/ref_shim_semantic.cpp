// TEST INFRASTRUCTURE ONLY — never linked into the product (pyslam_amd/).
//
// extern "C" shim around the *unmodified* reference semantic block grids, compiled where they lie
// under /root/reference/cpp/volumetric by oracle/Makefile into oracle/_ref/libref_volumetric.so:
//
//   VoxelBlockSemanticGrid              = VoxelBlockSemanticGridT<VoxelSemanticData>               (kind 0, voting)
//   VoxelBlockSemanticProbabilisticGrid = VoxelBlockSemanticGridT<VoxelSemanticDataProbabilistic>  (kind 1, log-prob)
//   VoxelBlockSemanticGrid2              = VoxelBlockSemanticGridT<VoxelSemanticData2>              (kind 2, two counters)
//   VoxelBlockSemanticProbabilisticGrid2 = VoxelBlockSemanticGridT<VoxelSemanticDataProbabilistic2> (kind 3, marginal maps)
//   (cpp/volumetric/voxel_block_semantic_grid.h:57-123, voxel_data_semantic.h:106-202, 249-672, voxel_data_semantic2.h:46-196, 256-787)
//
// Wrapped beyond the container part: assign_object_ids_to_instance_ids
// (voxel_semantic_data_association.h:70-373), remap_instance_ids (image_utils.h:69-163), carve
// (voxel_grid_carving.h:47-79), get_object_segments + OrientedBoundingBox3D::compute_from_points
// (voxel_block_semantic_grid.hpp:217-267, bounding_boxes_3d.cpp:373-553), merge_segments /
// remove_segment / remove_low_confidence_segments / get_ids (voxel_block_semantic_grid.hpp:119-213),
// set_depth_threshold / set_depth_decay_rate (:24-37).
// Built without TBB_FOUND: the reference's sequential branches.
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "camera_frustrum.h"
#include "image_utils.h"
#include "voxel_block_semantic_grid.h"
#include "voxel_hashing.h"

namespace {

using volumetric::BlockKey;
using volumetric::CameraFrustrum;

template <typename Base> class Dumpable : public Base {
  public:
    using Base::Base;
    const auto &blocks() const { return this->blocks_; }
};
using VoteGrid = Dumpable<volumetric::VoxelBlockSemanticGrid>;
using ProbGrid = Dumpable<volumetric::VoxelBlockSemanticProbabilisticGrid>;
using Vote2Grid = Dumpable<volumetric::VoxelBlockSemanticGrid2>;
using Prob2Grid = Dumpable<volumetric::VoxelBlockSemanticProbabilisticGrid2>;

struct Handle {
    int kind;
    VoteGrid *vote = nullptr;
    ProbGrid *prob = nullptr;
    Vote2Grid *vote2 = nullptr;
    Prob2Grid *prob2 = nullptr;
};

CameraFrustrum make_frustum(const float *intr, int width, int height, const double *T_cw_rowmajor, float depth_max,
                            float depth_min) {
    Eigen::Matrix4d T;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) T(r, c) = T_cw_rowmajor[r * 4 + c];
    return CameraFrustrum(intr[0], intr[1], intr[2], intr[3], width, height, T, depth_max, depth_min);
}

template <typename G, typename Tpos, typename Tcolor>
void integrate_t(G *g, const Tpos *pts, size_t n, const Tcolor *cols, const int *cls, const int *inst, const float *depths) {
    if (cls == nullptr) {
        g->template integrate_raw<Tpos, Tcolor>(pts, n, cols);
    } else if (inst != nullptr && depths != nullptr) {
        g->template integrate_raw<Tpos, Tcolor, int, int, float>(pts, n, cols, cls, inst, depths);
    } else if (inst != nullptr) {
        g->template integrate_raw<Tpos, Tcolor, int, int>(pts, n, cols, cls, inst);
    } else if (depths != nullptr) {
        g->template integrate_raw<Tpos, Tcolor, std::nullptr_t, int, float>(pts, n, cols, cls, nullptr, depths);
    } else {
        g->template integrate_raw<Tpos, Tcolor, std::nullptr_t, int>(pts, n, cols, cls);
    }
}

template <typename G>
void integrate_g(G *g, const void *pts, int pos_kind, size_t n, const void *cols, int color_kind, const int *cls, const int *inst,
                 const float *depths) {
    if (pos_kind == 0 && color_kind == 1)
        integrate_t<G, float, uint8_t>(g, (const float *)pts, n, (const uint8_t *)cols, cls, inst, depths);
    else if (pos_kind == 0)
        integrate_t<G, float, float>(g, (const float *)pts, n, (const float *)cols, cls, inst, depths);
    else if (color_kind == 1)
        integrate_t<G, double, uint8_t>(g, (const double *)pts, n, (const uint8_t *)cols, cls, inst, depths);
    else
        integrate_t<G, double, float>(g, (const double *)pts, n, (const float *)cols, cls, inst, depths);
}

template <typename G> auto sorted_blocks(const G *g) {
    using Entry = std::remove_reference_t<decltype(*g->blocks().begin())>;
    std::vector<const Entry *> order;
    for (const auto &kv : g->blocks()) order.push_back(&kv);
    std::sort(order.begin(), order.end(), [](auto *a, auto *b) {
        const auto &ka = a->first;
        const auto &kb = b->first;
        if (ka.x != kb.x) return ka.x < kb.x;
        if (ka.y != kb.y) return ka.y < kb.y;
        return ka.z < kb.z;
    });
    return order;
}

// per-voxel observable state: ints {count, object_id, class_id, confidence_counter}, confidence, sums
template <typename G>
int64_t dump_g(const G *g, int32_t *keys, int32_t *ints, float *conf, double *pos_sums, float *col_sums) {
    const auto order = sorted_blocks(g);
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
            if (conf) conf[b * nv + i] = v.get_confidence();
            if (pos_sums) for (int k = 0; k < 3; ++k) pos_sums[(b * nv + i) * 3 + k] = v.position_sum[k];
            if (col_sums) for (int k = 0; k < 3; ++k) col_sums[(b * nv + i) * 3 + k] = v.color_sum[k];
        }
    }
    return (int64_t)order.size();
}

template <typename G> int64_t marginals_g(const G *g, float *obj_conf, float *cls_conf) {
    const auto order = sorted_blocks(g);
    const int bs = g->get_block_size();
    const size_t nv = size_t(bs) * bs * bs;
    for (size_t b = 0; b < order.size(); ++b)
        for (size_t i = 0; i < nv; ++i) {
            const auto &v = order[b]->second.data[i];
            if constexpr (requires { v.get_object_confidence(); }) {
                obj_conf[b * nv + i] = v.get_object_confidence();
                cls_conf[b * nv + i] = v.get_class_confidence();
            } else {
                obj_conf[b * nv + i] = cls_conf[b * nv + i] = -1.0f;
            }
        }
    return (int64_t)order.size();
}

template <typename G>
int64_t get_voxels_g(const G *g, int min_count, float min_confidence, double *pts, float *cols, int32_t *class_ids,
                     int32_t *object_ids, float *confidences, int64_t cap) {
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

template <typename G>
int64_t assign_g(G *g, const float *intr, int width, int height, const double *T_cw, float depth_max, float depth_min,
                 const int32_t *class_img, const int32_t *inst_img, const float *depth, float depth_threshold, int do_carving,
                 float min_vote_ratio, int min_votes, int32_t *map_inst, int32_t *map_obj, int64_t cap) {
    const CameraFrustrum fr = make_frustum(intr, width, height, T_cw, depth_max, depth_min);
    cv::Mat cm(height, width, CV_32S, const_cast<int32_t *>(class_img));
    cv::Mat im(height, width, CV_32S, const_cast<int32_t *>(inst_img));
    cv::Mat dm;
    if (depth != nullptr) dm = cv::Mat(height, width, CV_32F, const_cast<float *>(depth));
    const auto m = g->assign_object_ids_to_instance_ids(fr, cm, im, dm, depth_threshold, do_carving != 0, min_vote_ratio, min_votes);
    std::map<int, int> sorted(m.begin(), m.end());
    int64_t i = 0;
    for (const auto &[inst, obj] : sorted) {
        if (i < cap && map_inst != nullptr) { map_inst[i] = inst; map_obj[i] = obj; }
        ++i;
    }
    return i;
}

template <typename G>
void carve_g(G *g, const float *intr, int width, int height, const double *T_cw, float depth_max, float depth_min,
             const float *depth, float depth_threshold) {
    const CameraFrustrum fr = make_frustum(intr, width, height, T_cw, depth_max, depth_min);
    cv::Mat depth_mat(height, width, CV_32F, const_cast<float *>(depth));
    g->carve(fr, depth_mat, depth_threshold);
}

// objects sorted by object id; per object: ids {object_id, class_id, n_points}, conf {min, max},
// obb {center xyz, quaternion wxyz, size xyz} (10 doubles); points/colors concatenated in object order
// (rows inside one object in the reference's iteration order: compare as sets).
template <typename G>
int64_t segments_g(const G *g, int min_count, float min_confidence, int32_t *ids, float *conf, double *obb, double *pts,
                   float *cols, int64_t cap_objects, int64_t cap_points, int64_t *n_points) {
    const auto group = g->get_object_segments(min_count, min_confidence);
    std::vector<std::shared_ptr<volumetric::ObjectData>> objs(group->object_vector.begin(), group->object_vector.end());
    std::sort(objs.begin(), objs.end(), [](const auto &a, const auto &b) { return a->object_id < b->object_id; });
    int64_t np = 0;
    for (size_t o = 0; o < objs.size(); ++o) {
        const auto &d = *objs[o];
        if (ids != nullptr && (int64_t)o < cap_objects) {
            ids[o * 3] = d.object_id; ids[o * 3 + 1] = d.class_id; ids[o * 3 + 2] = (int32_t)d.points.size();
            conf[o * 2] = d.confidence_min; conf[o * 2 + 1] = d.confidence_max;
            const auto &b = d.oriented_bounding_box;
            double *q = obb + o * 10;
            q[0] = b.center.x(); q[1] = b.center.y(); q[2] = b.center.z();
            q[3] = b.orientation.w(); q[4] = b.orientation.x(); q[5] = b.orientation.y(); q[6] = b.orientation.z();
            q[7] = b.size.x(); q[8] = b.size.y(); q[9] = b.size.z();
        }
        for (size_t i = 0; i < d.points.size(); ++i, ++np) {
            if (pts != nullptr && np < cap_points) {
                for (int k = 0; k < 3; ++k) { pts[np * 3 + k] = d.points[i][k]; cols[np * 3 + k] = d.colors[i][k]; }
            }
        }
    }
    if (n_points) *n_points = np;
    return (int64_t)objs.size();
}

template <typename G> int64_t get_ids_g(const G *g, int32_t *class_ids, int32_t *object_ids, int64_t cap) {
    const auto p = g->get_ids();
    const int64_t n = (int64_t)p.first.size();
    if (class_ids != nullptr)
        for (int64_t i = 0; i < std::min(n, cap); ++i) { class_ids[i] = p.first[i]; object_ids[i] = p.second[i]; }
    return n;
}

} // namespace

#define H(h) static_cast<Handle *>(h)
// f(grid) on the grid the handle holds (every lambda below returns the same type for the four grids)
template <typename F> auto visit(void *h, F &&f) {
    Handle *x = H(h);
    switch (x->kind) {
    case 0: return f(x->vote);
    case 1: return f(x->prob);
    case 2: return f(x->vote2);
    default: return f(x->prob2);
    }
}

extern "C" {

// kind 0: VoxelBlockSemanticGrid (voting); kind 1: VoxelBlockSemanticProbabilisticGrid; kind 2: VoxelBlockSemanticGrid2 (separate
// object / class counters, voxel_data_semantic2.h:46-196); kind 3: VoxelBlockSemanticProbabilisticGrid2 (marginal label maps, :256-787)
void *ref_sem2_create(int kind, double voxel_size, int block_size) {
    auto *h = new Handle{kind};
    if (kind == 0) h->vote = new VoteGrid(voxel_size, block_size);
    else if (kind == 1) h->prob = new ProbGrid(voxel_size, block_size);
    else if (kind == 2) h->vote2 = new Vote2Grid(voxel_size, block_size);
    else h->prob2 = new Prob2Grid(voxel_size, block_size);
    return h;
}
void ref_sem2_destroy(void *h) {
    delete H(h)->vote;
    delete H(h)->prob;
    delete H(h)->vote2;
    delete H(h)->prob2;
    delete H(h);
}
void ref_sem2_clear(void *h) { visit(h, [](auto *g) { g->clear(); }); }
int64_t ref_sem2_num_blocks(void *h) { return visit(h, [](auto *g) { return (int64_t)g->num_blocks(); }); }
// the thresholds are static members of the payload types (process-wide), exactly as in the reference
void ref_sem2_set_depth_threshold(void *h, float t) { visit(h, [&](auto *g) { g->set_depth_threshold(t); }); }
void ref_sem2_set_depth_decay_rate(void *h, float r) { visit(h, [&](auto *g) { g->set_depth_decay_rate(r); }); }
int32_t ref_sem2_peek_next_object_id() { return volumetric::VoxelSemanticSharedData::next_object_id.load(); }
void ref_sem2_set_next_object_id(int32_t v) { volumetric::VoxelSemanticSharedData::next_object_id.store(v); }

void ref_sem2_integrate(void *h, const void *pts, int pos_kind, int64_t n, const void *cols, int color_kind,
                        const int32_t *class_ids, const int32_t *instance_ids, const float *depths) {
    visit(h, [&](auto *g) { integrate_g(g, pts, pos_kind, (size_t)n, cols, color_kind, class_ids, instance_ids, depths); });
}
int64_t ref_sem2_dump(void *h, int32_t *keys, int32_t *ints, float *conf, double *pos_sums, float *col_sums) {
    return visit(h, [&](auto *g) { return dump_g(g, keys, ints, conf, pos_sums, col_sums); });
}
// the marginal confidences of the two *2 payloads (get_object_confidence / get_class_confidence, voxel_data_semantic2.h:60-76, 528-560),
// blocks and voxels in ref_sem2_dump's order; -1 for the payloads that have none
int64_t ref_sem2_dump_marginals(void *h, float *obj_conf, float *cls_conf) {
    return visit(h, [&](auto *g) { return marginals_g(g, obj_conf, cls_conf); });
}
int64_t ref_sem2_get_voxels(void *h, int min_count, float min_confidence, double *pts, float *cols, int32_t *class_ids,
                            int32_t *object_ids, float *confidences, int64_t cap) {
    return visit(h, [&](auto *g) { return get_voxels_g(g, min_count, min_confidence, pts, cols, class_ids, object_ids, confidences, cap); });
}
int64_t ref_sem2_assign_object_ids(void *h, const float *intr, int width, int height, const double *T_cw, float depth_max,
                                   float depth_min, const int32_t *class_img, const int32_t *inst_img, const float *depth,
                                   float depth_threshold, int do_carving, float min_vote_ratio, int min_votes, int32_t *map_inst,
                                   int32_t *map_obj, int64_t cap) {
    return visit(h, [&](auto *g) {
        return assign_g(g, intr, width, height, T_cw, depth_max, depth_min, class_img, inst_img, depth, depth_threshold, do_carving,
                        min_vote_ratio, min_votes, map_inst, map_obj, cap);
    });
}
void ref_sem2_carve(void *h, const float *intr, int width, int height, const double *T_cw, float depth_max, float depth_min,
                    const float *depth, float depth_threshold) {
    visit(h, [&](auto *g) { carve_g(g, intr, width, height, T_cw, depth_max, depth_min, depth, depth_threshold); });
}
int64_t ref_sem2_get_object_segments(void *h, int min_count, float min_confidence, int32_t *ids, float *conf, double *obb,
                                     double *pts, float *cols, int64_t cap_objects, int64_t cap_points, int64_t *n_points) {
    return visit(h, [&](auto *g) { return segments_g(g, min_count, min_confidence, ids, conf, obb, pts, cols, cap_objects, cap_points, n_points); });
}
void ref_sem2_merge_segments(void *h, int id1, int id2) { visit(h, [&](auto *g) { g->merge_segments(id1, id2); }); }
void ref_sem2_remove_segment(void *h, int id) { visit(h, [&](auto *g) { g->remove_segment(id); }); }
void ref_sem2_remove_low_confidence_segments(void *h, int min_confidence) {
    visit(h, [&](auto *g) { g->remove_low_confidence_segments(min_confidence); });
}
int64_t ref_sem2_get_ids(void *h, int32_t *class_ids, int32_t *object_ids, int64_t cap) {
    return visit(h, [&](auto *g) { return get_ids_g(g, class_ids, object_ids, cap); });
}

// volumetric.remap_instance_ids as bound (image_utils_module.h:49-94): remap_instance_ids<MapInstanceIdToObjectId, int32_t>, image_utils.h:69-163
void ref_remap_instance_ids(const int32_t *inst_img, int height, int width, const int32_t *map_inst, const int32_t *map_obj,
                            int64_t n_map, int32_t *out) {
    cv::Mat im(height, width, CV_32S, const_cast<int32_t *>(inst_img));
    std::unordered_map<int, int> m;
    for (int64_t i = 0; i < n_map; ++i) m[map_inst[i]] = map_obj[i];
    if (m.empty()) { // what Python sees: the binding's lambda returns the image itself for an empty map (image_utils_module.h:52-58)
        for (int i = 0; i < height; ++i) std::memcpy(out + (size_t)i * width, im.ptr<int32_t>(i), sizeof(int32_t) * width);
        return;
    }
    const cv::Mat r = volumetric::remap_instance_ids<std::unordered_map<int, int>, int32_t>(im, m);
    for (int i = 0; i < height; ++i) std::memcpy(out + (size_t)i * width, r.ptr<int32_t>(i), sizeof(int32_t) * width);
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// The non-default direct voxel hash: VoxelGrid = VoxelGridT<VoxelData> (cpp/volumetric/voxel_grid.h:83-245).
// float32 points + float32 colours take the reference's SIMD path (voxel_grid_simd.hpp), uint8 colours or
// float64 points the scalar one (voxel_grid.hpp:136-175).
// ------------------------------------------------------------------------------------------------
#include "voxel_grid.h"

extern "C" {

void *ref_vgrid_create(double voxel_size) { return new volumetric::VoxelGrid(voxel_size); }
void ref_vgrid_destroy(void *g) { delete static_cast<volumetric::VoxelGrid *>(g); }
int64_t ref_vgrid_size(void *g) { return (int64_t) static_cast<volumetric::VoxelGrid *>(g)->size(); }
// color_kind: 0 none, 1 uint8, 2 float32
void ref_vgrid_integrate(void *gv, const float *pts, int64_t n, const void *cols, int color_kind) {
    auto *g = static_cast<volumetric::VoxelGrid *>(gv);
    if (color_kind == 0) g->integrate_raw<float>(pts, (size_t)n);
    else if (color_kind == 1) g->integrate_raw<float, uint8_t>(pts, (size_t)n, static_cast<const uint8_t *>(cols));
    else g->integrate_raw<float, float>(pts, (size_t)n, static_cast<const float *>(cols));
}
int64_t ref_vgrid_get_voxels(void *gv, int min_count, float min_confidence, float *pts, float *cols, int64_t cap) {
    const auto vg = static_cast<volumetric::VoxelGrid *>(gv)->get_voxels(min_count, min_confidence);
    const int64_t n = (int64_t)vg.points.size();
    if (pts != nullptr)
        for (int64_t i = 0; i < std::min(n, cap); ++i)
            for (int k = 0; k < 3; ++k) { pts[i * 3 + k] = vg.points[i][k]; cols[i * 3 + k] = vg.colors[i][k]; }
    return n;
}

} // extern "C"

// ---- spatial queries with semantics, integrate_segment, get_class_segments ------------------------------------
namespace {
template <typename VG>
int64_t copy_sem(const VG &vg, double *pts, float *cols, int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap) {
    const int64_t n = (int64_t)vg.points.size();
    if (pts != nullptr) {
        for (int64_t i = 0; i < std::min(n, cap); ++i) {
            for (int k = 0; k < 3; ++k) { pts[i * 3 + k] = vg.points[i][k]; cols[i * 3 + k] = vg.colors[i][k]; }
            class_ids[i] = vg.class_ids[i];
            object_ids[i] = vg.object_ids[i];
            confidences[i] = vg.confidences[i];
        }
    }
    return n;
}
template <typename G>
int64_t class_segments_g(const G *g, int min_count, float min_confidence, int32_t *ids, float *conf, int64_t cap) {
    const auto group = g->get_class_segments(min_count, min_confidence);
    std::vector<std::shared_ptr<volumetric::ClassData>> cs(group->class_vector.begin(), group->class_vector.end());
    std::sort(cs.begin(), cs.end(), [](const auto &a, const auto &b) { return a->class_id < b->class_id; });
    for (size_t i = 0; i < cs.size() && (int64_t)i < cap; ++i) {
        if (ids) { ids[i * 2] = cs[i]->class_id; ids[i * 2 + 1] = (int32_t)cs[i]->points.size(); }
        if (conf) { conf[i * 2] = cs[i]->confidence_min; conf[i * 2 + 1] = cs[i]->confidence_max; }
    }
    return (int64_t)cs.size();
}
} // namespace

extern "C" {

int64_t ref_sem2_get_voxels_in_bb(void *h, const double *bb, int min_count, float min_confidence, double *pts, float *cols,
                                  int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap) {
    volumetric::BoundingBox3D bbox(bb[0], bb[1], bb[2], bb[3], bb[4], bb[5]);
    return visit(h, [&](auto *g) { return copy_sem(g->template get_voxels_in_bb<true>(bbox, min_count, min_confidence), pts, cols, class_ids, object_ids, confidences, cap); });
}
int64_t ref_sem2_get_voxels_in_frustum(void *h, const float *intr, int width, int height, const double *T_cw, float depth_max,
                                       float depth_min, int min_count, float min_confidence, double *pts, float *cols,
                                       int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap) {
    const CameraFrustrum fr = make_frustum(intr, width, height, T_cw, depth_max, depth_min);
    return visit(h, [&](auto *g) { return copy_sem(g->template get_voxels_in_camera_frustrum<true>(fr, min_count, min_confidence), pts, cols, class_ids, object_ids, confidences, cap); });
}
void ref_sem2_integrate_segment(void *h, const double *pts, int64_t n, const float *cols, int object_id, int class_id) {
    visit(h, [&](auto *g) { g->template integrate_segment_raw<double, float, int, int>(pts, (size_t)n, cols, class_id, object_id); });
}
int64_t ref_sem2_get_class_segments(void *h, int min_count, float min_confidence, int32_t *ids, float *conf, int64_t cap) {
    return visit(h, [&](auto *g) { return class_segments_g(g, min_count, min_confidence, ids, conf, cap); });
}

} // extern "C"

// ---- 3D bounding boxes (cpp/volumetric/bounding_boxes_3d.h/.cpp, bindings bounding_boxes_module.h:49-160): thin C exports
// of the reference's own classes, so that the Python mirrors in pyslam_amd can be compared with the compiled code.
// aabb = {min xyz, max xyz} (6 doubles); obb = {center xyz, quaternion wxyz, size xyz} (10 doubles).
namespace {
volumetric::BoundingBox3D mk_aabb(const double *b) { return volumetric::BoundingBox3D(b[0], b[1], b[2], b[3], b[4], b[5]); }
volumetric::OrientedBoundingBox3D mk_obb(const double *o) {
    return volumetric::OrientedBoundingBox3D(Eigen::Vector3d(o[0], o[1], o[2]), Eigen::Quaterniond(o[3], o[4], o[5], o[6]),
                                             Eigen::Vector3d(o[7], o[8], o[9]));
}
void put_obb(const volumetric::OrientedBoundingBox3D &b, double *o) {
    o[0] = b.center.x(); o[1] = b.center.y(); o[2] = b.center.z();
    o[3] = b.orientation.w(); o[4] = b.orientation.x(); o[5] = b.orientation.y(); o[6] = b.orientation.z();
    o[7] = b.size.x(); o[8] = b.size.y(); o[9] = b.size.z();
}
} // namespace

extern "C" {
void ref_aabb3_scalars(const double *b, double *out9) { // center 3, size 3, volume, surface area, diagonal
    const auto a = mk_aabb(b);
    const auto c = a.get_center(), s = a.get_size();
    out9[0] = c.x(); out9[1] = c.y(); out9[2] = c.z(); out9[3] = s.x(); out9[4] = s.y(); out9[5] = s.z();
    out9[6] = a.get_volume(); out9[7] = a.get_surface_area(); out9[8] = a.get_diagonal_length();
}
void ref_aabb3_contains(const double *b, const double *pts, int64_t n, uint8_t *out) {
    const auto a = mk_aabb(b);
    for (int64_t i = 0; i < n; ++i) out[i] = a.contains<double>(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]) ? 1 : 0;
}
int ref_aabb3_intersects(const double *a, const double *b) { return mk_aabb(a).intersects(mk_aabb(b)) ? 1 : 0; }
void ref_aabb3_from_points(const double *pts, int64_t n, double *out6) {
    std::vector<Eigen::Vector3d> v((size_t)n);
    for (int64_t i = 0; i < n; ++i) v[i] = Eigen::Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    const auto a = volumetric::BoundingBox3D::compute_from_points<double>(v);
    out6[0] = a.min_x; out6[1] = a.min_y; out6[2] = a.min_z; out6[3] = a.max_x; out6[4] = a.max_y; out6[5] = a.max_z;
}
void ref_obb3_scalars(const double *o, double *out3, double *M16, double *Minv16, double *corners24) {
    const auto b = mk_obb(o);
    out3[0] = b.get_volume(); out3[1] = b.get_surface_area(); out3[2] = b.get_diagonal_length();
    const Eigen::Matrix4d M = b.get_matrix(), Mi = b.get_inverse_matrix();
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { M16[r * 4 + c] = M(r, c); Minv16[r * 4 + c] = Mi(r, c); }
    const auto cs = b.get_corners();
    for (int i = 0; i < 8; ++i) { corners24[3 * i] = cs[i].x(); corners24[3 * i + 1] = cs[i].y(); corners24[3 * i + 2] = cs[i].z(); }
}
void ref_obb3_contains(const double *o, const double *pts, int64_t n, uint8_t *out) {
    const auto b = mk_obb(o);
    for (int64_t i = 0; i < n; ++i) out[i] = b.contains<double>(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]) ? 1 : 0;
}
int ref_obb3_intersects_obb(const double *a, const double *b) { return mk_obb(a).intersects(mk_obb(b)) ? 1 : 0; }
int ref_obb3_intersects_aabb(const double *a, const double *b) { return mk_obb(a).intersects(mk_aabb(b)) ? 1 : 0; }
void ref_obb3_from_points(const double *pts, int64_t n, double *out10) {
    std::vector<Eigen::Vector3d> v((size_t)n);
    for (int64_t i = 0; i < n; ++i) v[i] = Eigen::Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    put_obb(volumetric::OrientedBoundingBox3D::compute_from_points<double>(v, volumetric::OBBComputationMethod::PCA), out10);
}
}

// ---- 2D bounding boxes (cpp/volumetric/bounding_boxes_2d.h/.cpp, bindings bounding_boxes_module.h:165-258): the same thin exports.
// aabb = {min xy, max xy} (4 doubles); obb = {center xy, angle, size xy} (5 doubles).
#include "bounding_boxes_2d.h"
namespace {
volumetric::BoundingBox2D mk_aabb2(const double *b) { return volumetric::BoundingBox2D(b[0], b[1], b[2], b[3]); }
volumetric::OrientedBoundingBox2D mk_obb2(const double *o) {
    return volumetric::OrientedBoundingBox2D(Eigen::Vector2d(o[0], o[1]), o[2], Eigen::Vector2d(o[3], o[4]));
}
} // namespace
extern "C" {
void ref_aabb2_scalars(const double *b, double *out7) { // center 2, size 2, area, perimeter, diagonal
    const auto a = mk_aabb2(b);
    const auto c = a.get_center(), s = a.get_size();
    out7[0] = c.x(); out7[1] = c.y(); out7[2] = s.x(); out7[3] = s.y();
    out7[4] = a.get_area(); out7[5] = a.get_perimeter(); out7[6] = a.get_diagonal_length();
}
void ref_aabb2_contains(const double *b, const double *pts, int64_t n, uint8_t *out) {
    const auto a = mk_aabb2(b);
    for (int64_t i = 0; i < n; ++i) out[i] = a.contains<double>(pts[2 * i], pts[2 * i + 1]) ? 1 : 0;
}
int ref_aabb2_intersects(const double *a, const double *b) { return mk_aabb2(a).intersects(mk_aabb2(b)) ? 1 : 0; }
void ref_aabb2_from_points(const double *pts, int64_t n, double *out4) {
    std::vector<Eigen::Vector2d> v((size_t)n);
    for (int64_t i = 0; i < n; ++i) v[i] = Eigen::Vector2d(pts[2 * i], pts[2 * i + 1]);
    const auto a = volumetric::BoundingBox2D::compute_from_points<double>(v);
    out4[0] = a.min_x; out4[1] = a.min_y; out4[2] = a.max_x; out4[3] = a.max_y;
}
void ref_obb2_scalars(const double *o, double *out4, double *corners8) { // volume, area, perimeter, diagonal; corners [4,2]
    const auto b = mk_obb2(o);
    out4[0] = b.get_volume(); out4[1] = b.get_area(); out4[2] = b.get_perimeter(); out4[3] = b.get_diagonal_length();
    const auto cs = b.get_corners();
    for (int i = 0; i < 4; ++i) { corners8[2 * i] = cs[i].x(); corners8[2 * i + 1] = cs[i].y(); }
}
void ref_obb2_contains(const double *o, const double *pts, int64_t n, uint8_t *out) {
    const auto b = mk_obb2(o);
    for (int64_t i = 0; i < n; ++i) out[i] = b.contains<double>(pts[2 * i], pts[2 * i + 1]) ? 1 : 0;
}
int ref_obb2_intersects_obb(const double *a, const double *b) { return mk_obb2(a).intersects(mk_obb2(b)) ? 1 : 0; }
int ref_obb2_intersects_aabb(const double *a, const double *b) { return mk_obb2(a).intersects(mk_aabb2(b)) ? 1 : 0; }
void ref_obb2_from_points(const double *pts, int64_t n, double *out5) {
    std::vector<Eigen::Vector2d> v((size_t)n);
    for (int64_t i = 0; i < n; ++i) v[i] = Eigen::Vector2d(pts[2 * i], pts[2 * i + 1]);
    const auto b = volumetric::OrientedBoundingBox2D::compute_from_points<double>(v, volumetric::OBBComputationMethod::PCA);
    out5[0] = b.center.x(); out5[1] = b.center.y(); out5[2] = b.angle_rad; out5[3] = b.size.x(); out5[4] = b.size.y();
}
}
