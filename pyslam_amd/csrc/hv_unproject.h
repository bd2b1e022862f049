// depth2pointcloud + world transform of one pixel, shared by the VOXEL_GRID and the semantic grids' frame entry points.
#pragma once
#include "hv_common.h"

// depth2pointcloud (pyslam/utilities/depth.py:45-85) + world transform
// (volumetric_integrator_voxel_grid.py:262-281), f64 arithmetic in a fixed order, rounded to f32.
struct HvUnprojectParams {
    double cx, cy, inv_fx, inv_fy;
    double Rwc[9], twc[3];
    float min_depth, max_depth, depth_scale_f;
    int32_t H, W, depth_is_u16;
};

#ifdef __HIPCC__
__device__ __forceinline__ bool hv_unproject_point(const HvUnprojectParams &U, const void *__restrict__ depth_raw, int64_t i, float pt[3]) {
    float d = U.depth_is_u16 ? (float)((const uint16_t *)depth_raw)[i] : ((const float *)depth_raw)[i];
    if (U.depth_scale_f != 1.0f) d = d / U.depth_scale_f;
    if (!((d > U.min_depth) && (d < U.max_depth))) return false;
    const int row = (int)((uint32_t)i / (uint32_t)U.W), col_px = (int)((uint32_t)i - (uint32_t)row * (uint32_t)U.W); // i < 2^31 (max_points)
    const double z = (double)d;
    const double x = ((double)col_px - U.cx) * z * U.inv_fx;
    const double y = ((double)row - U.cy) * z * U.inv_fy;
#pragma unroll
    for (int r = 0; r < 3; ++r) pt[r] = (float)(((U.Rwc[r * 3 + 0] * x + U.Rwc[r * 3 + 1] * y) + U.Rwc[r * 3 + 2] * z) + U.twc[r]);
    return true;
}
__device__ __forceinline__ bool hv_unproject_pixel(const HvUnprojectParams &U, const void *__restrict__ depth_raw,
                                                   const uint8_t *__restrict__ rgb, int64_t i, float pt[3], float col[3]) {
    if (!hv_unproject_point(U, depth_raw, i, pt)) return false;
    const uint8_t *c = rgb + i * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) col[k] = (float)((double)c[k] / 255.0);
    return true;
}
#endif // __HIPCC__

static inline HvUnprojectParams unproject_params(int32_t depth_dtype, double depth_scale, int32_t height, int32_t width, const double *intr,
                                          const double *T_cw, double min_depth, double max_depth) {
    HvUnprojectParams U;
    U.cx = intr[2];
    U.cy = intr[3];
    U.inv_fx = 1.0 / intr[0]; // depth.py:67-68
    U.inv_fy = 1.0 / intr[1];
    // inv_T, pyslam/utilities/geometry.py:98-104
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) U.Rwc[r * 3 + c] = T_cw[c * 4 + r];
    for (int r = 0; r < 3; ++r)
        U.twc[r] = -((U.Rwc[r * 3 + 0] * T_cw[3] + U.Rwc[r * 3 + 1] * T_cw[7]) + U.Rwc[r * 3 + 2] * T_cw[11]);
    U.min_depth = (float)min_depth;
    U.max_depth = (float)max_depth;
    U.depth_scale_f = (float)depth_scale;
    U.H = height;
    U.W = width;
    U.depth_is_u16 = depth_dtype == HV_DEPTH_U16;
    return U;
}

