// Shared by the VOXEL_GRID and the semantic payloads: grid parameters, and the bbox / camera-frustum
// query description with its host-side setup (reference: camera_frustrum.cpp:175-264,
// voxel_block_grid.hpp:827-835, 1339-1349).
#pragma once
#include <cmath>
#include <cstring>

#include "hv_common.h"

struct HvGridParams {
    float inv_voxel_size; // 1.0f / voxel_size  (voxel_block_grid.hpp:6)
    int32_t bs;           // block_size
    int32_t nvox;         // bs^3
    int32_t local_bits;
    int32_t bs_shift;     // log2(bs) when bs is a power of two (the default 8 and 16), else -1
    int32_t owner_rank, owner_world; // multi-GPU block ownership (hv_set_owner): this GPU fuses block b iff hv_owner_of(b) % world == rank
};

// Block-ownership sharding of the grid modes (SURVEY 8e "zero reduce" form): a point whose block another GPU owns is skipped
// here - it is neither fused nor counted as dropped.
__host__ __device__ inline bool hv_block_is_foreign(const HvGridParams &G, uint64_t packed_block_key) {
    return G.owner_world > 1 && hv_owner_of(packed_block_key, G.owner_world) != G.owner_rank;
}

// floor_div, voxel_hashing.h:139-142
__host__ __device__ inline int32_t hv_floor_div(int32_t a, int32_t b) {
    const int64_t aa = a, bb = b;
    return (int32_t)((aa >= 0) ? (aa / bb) : ((aa - bb + 1) / bb));
}

// ---- scans over all allocated voxels -----------------------------------------------------------
struct HvQuery {
    int32_t kind; // 0: all, 1: bbox, 2: frustum, 3: carve
    int32_t min_count;
    int32_t vmin[3], vmax[3], bmin[3], bmax[3];
    double bb[6];
    // frustum (CameraFrustrum, camera_frustrum.h:108-121)
    float fx, fy, cx, cy, depth_max, depth_min;
    int32_t width, height;
    double R[9], t[3];
    float carve_threshold;
};

// CameraFrustrum::contains<T>, camera_frustrum.cpp:175-196 (the point is cast to double first)
__device__ __forceinline__ bool hv_frustum_contains_d(const HvQuery &Q, double p0, double p1, double p2, float *uvd) {
    double pc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) pc[r] = (Q.R[r * 3 + 0] * p0 + Q.R[r * 3 + 1] * p1 + Q.R[r * 3 + 2] * p2) + Q.t[r];
    const float depth = (float)pc[2];
    if (!(depth >= Q.depth_min && depth <= Q.depth_max)) return false;
    const float u = (float)((double)Q.fx * (pc[0] / pc[2]) + (double)Q.cx);
    const float v = (float)((double)Q.fy * (pc[1] / pc[2]) + (double)Q.cy);
    uvd[0] = u;
    uvd[1] = v;
    uvd[2] = depth;
    return u >= 0.0f && u < (float)Q.width && v >= 0.0f && v < (float)Q.height;
}
__device__ __forceinline__ bool hv_frustum_contains(const HvQuery &Q, float xw, float yw, float zw, float *uvd) {
    return hv_frustum_contains_d(Q, (double)xw, (double)yw, (double)zw, uvd);
}

static inline void fill_frustum_query(HvQuery &Q, const hv_volume *v, const float *intr, int width, int height,
                               const double *T_cw, float depth_max, float depth_min) {
    Q.fx = intr[0];
    Q.fy = intr[1];
    Q.cx = intr[2];
    Q.cy = intr[3];
    Q.width = width;
    Q.height = height;
    Q.depth_max = depth_max;
    Q.depth_min = depth_min;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Q.R[r * 3 + c] = T_cw[r * 4 + c];
        Q.t[r] = T_cw[r * 4 + 3];
    }
    // compute_frustum_corners_world_ + compute_bbox_, camera_frustrum.cpp:209-264
    double Rwc[9], twc[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rwc[r * 3 + c] = Q.R[c * 3 + r];
    for (int r = 0; r < 3; ++r) twc[r] = -(Rwc[r * 3 + 0] * Q.t[0] + Rwc[r * 3 + 1] * Q.t[1] + Rwc[r * 3 + 2] * Q.t[2]);
    const double cu[4] = {0.0, (double)width, (double)width, 0.0};
    const double cv[4] = {0.0, 0.0, (double)height, (double)height};
    for (int k = 0; k < 3; ++k) {
        Q.bb[k] = 1.7976931348623157e308;
        Q.bb[3 + k] = -1.7976931348623157e308;
    }
    for (int i = 0; i < 4; ++i) {
        const double xn = (cu[i] - (double)Q.cx) / (double)Q.fx;
        const double yn = (cv[i] - (double)Q.cy) / (double)Q.fy;
        const double ds[2] = {(double)depth_min, (double)depth_max};
        for (int j = 0; j < 2; ++j) {
            const double pc[3] = {xn * ds[j], yn * ds[j], ds[j]};
            for (int r = 0; r < 3; ++r) {
                const double w = (Rwc[r * 3 + 0] * pc[0] + Rwc[r * 3 + 1] * pc[1] + Rwc[r * 3 + 2] * pc[2]) + twc[r];
                if (w < Q.bb[r]) Q.bb[r] = w;
                if (w > Q.bb[3 + r]) Q.bb[3 + r] = w;
            }
        }
    }
    (void)v;
}

// bbox -> voxel/block key range, voxel_block_grid.hpp:827-835: get_voxel_key_inv<double,double>
// with the float inv_voxel_size_ promoted to double.
static inline void fill_key_range(HvQuery &Q, const HvGridParams &G) {
    for (int k = 0; k < 3; ++k) {
        Q.vmin[k] = (int32_t)std::floor(Q.bb[k] * (double)G.inv_voxel_size);
        Q.vmax[k] = (int32_t)std::floor(Q.bb[3 + k] * (double)G.inv_voxel_size);
        Q.bmin[k] = hv_floor_div(Q.vmin[k], G.bs);
        Q.bmax[k] = hv_floor_div(Q.vmax[k], G.bs);
    }
}


// semantic-payload implementations behind the mode-independent entry points (hv_semantic_ops.hip)
int hv_sem_carve(hv_volume *v, const HvQuery &Q, const float *d_depth, int64_t n_blocks);
int hv_sem_segment_op(hv_volume *v, int op, int32_t a, int32_t b, float fa);
int hv_sem_size(hv_volume *v, int64_t *n);
int hv_unproject_frame(hv_volume *v, const void *depth, int32_t depth_dtype, double depth_scale, const uint8_t *rgb,
                       int32_t height, int32_t width, const double *intr, const double *T_cw, double min_depth,
                       double max_depth, int32_t loc, const void **d_depth_out); // hv_voxel_grid.hip
