// libpyslam_hipvol.so — VOXEL_GRID fusion (pySLAM cpp/volumetric VoxelBlockGrid semantics) for gfx950.
//
// integrate(points, colors)  (reference: voxel_block_grid.hpp:115-136, 292-462, 524-614)
//   k_vg_keys     1 thread/point: f32 key arithmetic of voxel_hashing.h:69-75,139-161 (no FMA
//                 contraction), block claimed in the hash, sort key = (slot << local_bits) | local idx
//   radix sort    rocPRIM LSD pairs sort (stable) of (key, point index) — groups points per voxel
//                 while keeping point-index order inside every voxel
//   k_vg_reduce   the first thread of every voxel run folds its points into the voxel record *in
//                 point-index order*, exactly the order of the reference's sequential branch, so
//                 count, position_sum and color_sum are bit-identical to the reference — no float
//                 atomics, one 32-byte read-modify-write per touched voxel.
// The reference groups by block with per-thread hash maps and merges them (TBB); on a GPU a
// device-wide stable sort is the natural group-by and also removes all payload contention.
//
// Latency-bound, not HBM-bound: a 640x480 frame moves ~20 MB (points + touched records).
#include <algorithm>
#include <array>
#include <cmath>
#include <numeric>

#include "hv_common.h"
#include "hv_query.h"
#include "hv_bins.h"
#include "hv_unproject.h"
#include <rocprim/device/device_radix_sort.hpp>

static constexpr uint32_t HV_SORT_SENTINEL = 0xFFFFFFFFu;

struct HvPointKey {
    int32_t v[3], b[3], l[3];
};

// get_voxel_key_inv<Tp,float> + get_block_key + get_local_voxel_key.  Tp = float: floorf(x * inv) in float; Tp = double (the
// binding's py::array_t<double> overload, volumetric_grid_module.h:738-741): the float inverse voxel size is promoted and the
// product and floor are double - a float64 point near a cell border keeps its own cell instead of its float32 neighbour's.
template <typename Tp>
__device__ __forceinline__ HvPointKey hv_point_key(Tp x, Tp y, Tp z, const HvGridParams &G) {
    HvPointKey k;
    const Tp inv = (Tp)G.inv_voxel_size;
    const Tp f[3] = {floor(x * inv), floor(y * inv), floor(z * inv)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        k.v[a] = (int32_t)f[a];
        // floor_div (voxel_hashing.h:139-142) without the 64-bit division (3 per point: most of the key kernel's time): an
        // arithmetic shift for power-of-two blocks, else floor(v * (1 / bs) + 2^-10) in double - exact for |v| < 2^31 and
        // bs <= 16: a non-integer v / bs is at least 1/16 away from an integer, the product's error is below 2^-20
        k.b[a] = G.bs_shift >= 0 ? (k.v[a] >> G.bs_shift) : (int32_t)floor(fma((double)k.v[a], 1.0 / (double)G.bs, 0x1p-10));
        k.l[a] = k.v[a] - k.b[a] * G.bs; // in [0, bs): no overflow (|b * bs| <= |v| + bs)
    }
    return k;
}
// the point is finite and its voxel index fits the 32-bit key
template <typename Tp>
__device__ __forceinline__ bool hv_point_keyable(Tp x, Tp y, Tp z, const HvGridParams &G) {
    const Tp inv = (Tp)G.inv_voxel_size, lim = (Tp)1.0e9;
    return isfinite(x) && isfinite(y) && isfinite(z) && fabs(x * inv) < lim && fabs(y * inv) < lim && fabs(z * inv) < lim;
}

// pts64 != nullptr: the caller's points are float64 - keys from the doubles, and the float32 narrowing the voxel sums take
// (position_sum += static_cast<float>(x), voxel_data.h:54-56) is written to pts for the reduce.
__global__ __launch_bounds__(256) void k_vg_keys(HvTable table, float *__restrict__ pts, int64_t n,
                                                  HvGridParams G, uint32_t *__restrict__ keys_out,
                                                  uint32_t *__restrict__ vals_out,
                                                  const uint32_t *__restrict__ valid_mask_keys, const double *__restrict__ pts64) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vals_out[i] = (uint32_t)i;
    if (valid_mask_keys != nullptr && valid_mask_keys[i] == HV_SORT_SENTINEL) { // pixel rejected by the unprojection
        keys_out[i] = HV_SORT_SENTINEL;
        return;
    }
    uint32_t key = HV_SORT_SENTINEL;
    bool keyable;
    HvPointKey k;
    if (pts64 != nullptr) {
        const double x = pts64[i * 3 + 0], y = pts64[i * 3 + 1], z = pts64[i * 3 + 2];
        pts[i * 3 + 0] = (float)x;
        pts[i * 3 + 1] = (float)y;
        pts[i * 3 + 2] = (float)z;
        keyable = hv_point_keyable(x, y, z, G);
        if (keyable) k = hv_point_key(x, y, z, G);
    } else {
        const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        keyable = hv_point_keyable(x, y, z, G);
        if (keyable) k = hv_point_key(x, y, z, G);
    }
    bool foreign = false;
    if (keyable) {
        if (hv_key_in_range(k.b[0], k.b[1], k.b[2])) {
            const unsigned long long bkey = hv_pack_key(k.b[0], k.b[1], k.b[2]);
            foreign = hv_block_is_foreign(G, bkey);
            const int32_t slot = foreign ? -1 : hv_table_insert(table, bkey);
            if (slot >= 0) {
                const uint32_t lidx = (uint32_t)(k.l[0] + k.l[1] * G.bs + k.l[2] * G.bs * G.bs); // voxel_block.h:67-70
                key = ((uint32_t)slot << G.local_bits) | lidx;
            }
        }
    }
    if (key == HV_SORT_SENTINEL && !foreign) atomicAdd(&table.counters[HV_CNT_DROPPED], 1);
    keys_out[i] = key;
}

// update_voxel_direct, voxel_block_grid.hpp:524-614 for VoxelData, folded over one voxel's run.
template <int COLOR_KIND>
__global__ __launch_bounds__(256) void k_vg_reduce(HvTable table, HvVoxel *__restrict__ pool,
                                                    const uint32_t *__restrict__ keys,
                                                    const uint32_t *__restrict__ vals, int64_t n,
                                                    HvGridParams G, const float *__restrict__ pts,
                                                    const void *__restrict__ cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = keys[i];
    if (key == HV_SORT_SENTINEL) return;
    if (i > 0 && keys[i - 1] == key) return; // not the head of its run
    const int32_t slot = (int32_t)(key >> G.local_bits);
    const int32_t idx = table.vals[slot];
    if (idx < 0) return;
    HvVoxel *vx = pool + (int64_t)idx * G.nvox + (key & ((1u << G.local_bits) - 1u));
    HvVoxel acc = *vx;
    const float inv_255 = 1.0f / 255.0f; // voxel_data.h:82
    int64_t j = i;
    do {
        const int64_t p = vals[j];
        acc.pos[0] += pts[p * 3 + 0];
        acc.pos[1] += pts[p * 3 + 1];
        acc.pos[2] += pts[p * 3 + 2];
        if (COLOR_KIND == HV_COLOR_U8) {
            const uint8_t *c = (const uint8_t *)cols + p * 3;
            acc.col[0] += (float)c[0] * inv_255;
            acc.col[1] += (float)c[1] * inv_255;
            acc.col[2] += (float)c[2] * inv_255;
        } else if (COLOR_KIND == HV_COLOR_F32) {
            const float *c = (const float *)cols + p * 3;
            acc.col[0] += c[0];
            acc.col[1] += c[1];
            acc.col[2] += c[2];
        }
        acc.count = acc.count == 0 ? 1 : acc.count + 1;
        ++j;
    } while (j < n && keys[j] == key);
    *vx = acc;
}

__global__ __launch_bounds__(256) void k_vg_unproject(const void *__restrict__ depth_raw,
                                                       const uint8_t *__restrict__ rgb, HvUnprojectParams U,
                                                       float *__restrict__ pts_out, float *__restrict__ cols_out,
                                                       uint32_t *__restrict__ valid_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)U.H * U.W) return;
    float pt[3], col[3];
    const bool valid = hv_unproject_pixel(U, depth_raw, rgb, i, pt, col);
    valid_out[i] = valid ? 0u : HV_SORT_SENTINEL;
    if (!valid) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pts_out[i * 3 + k] = pt[k];
        cols_out[i * 3 + k] = col[k];
    }
}

// NS1 (the north star's "MFMA only for the 3xN point-transform GEMM"), measured instead of argued (HV_VG_UNPROJECT=mfma, radix path
// only; tools/ns1_mfma.py, profiles/r05/ns1_mfma.txt): the same unprojection with the rigid transform of a wave's 64 points as four
// v_mfma_f64_16x16x4_f64 (D = [R | t] (16x4, rows 0..2 used) x [x; y; z; 1] (4x16 points)).  The operands have to be gathered across
// lanes (B[k][j] = coordinate k of point j lives in lane j's registers) and the results - rows 0..2 land in lanes 0..15 - scattered
// back: 48 ds_bpermute per wave around 4 MFMAs, against 18 scalar-f64 VALU operations per lane.  The matrix core also FUSES the
// four multiply-adds of a row (one rounding per step instead of two), so its points are not the numpy-order points the parity
// contract names (section 2: ((r0 x + r1 y) + r2 z) + t, every product and sum rounded): NOT the default.
__global__ __launch_bounds__(256) void k_vg_unproject_mfma(const void *__restrict__ depth_raw, const uint8_t *__restrict__ rgb,
                                                            HvUnprojectParams U, float *__restrict__ pts_out,
                                                            float *__restrict__ cols_out, uint32_t *__restrict__ valid_out) {
    typedef double hv_d4 __attribute__((ext_vector_type(4)));
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = hv_lane_id();
    const int64_t npx = (int64_t)U.H * U.W;
    bool valid = false;
    double x = 0.0, y = 0.0, z = 0.0;
    if (i < npx) {
        float d = U.depth_is_u16 ? (float)((const uint16_t *)depth_raw)[i] : ((const float *)depth_raw)[i];
        if (U.depth_scale_f != 1.0f) d = d / U.depth_scale_f;
        valid = (d > U.min_depth) && (d < U.max_depth);
        const int row = (int)((uint32_t)i / (uint32_t)U.W), col_px = (int)((uint32_t)i - (uint32_t)row * (uint32_t)U.W);
        z = (double)d;
        x = ((double)col_px - U.cx) * z * U.inv_fx;
        y = ((double)row - U.cy) * z * U.inv_fy;
    }
    // A[i][k]: lane l holds i = l & 15, k = l >> 4
    const int ai = lane & 15, ak = lane >> 4;
    const double a = ai < 3 ? (ak < 3 ? U.Rwc[ai * 3 + ak] : U.twc[ai]) : 0.0;
    double wx = 0.0, wy = 0.0, wz = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // B[k][j]: lane l holds k = l >> 4, j = l & 15 = coordinate k of point 16 g + j
        const int src = 16 * g + (lane & 15);
        const double bx = __shfl(x, src), by = __shfl(y, src), bz = __shfl(z, src);
        const double b = ak == 0 ? bx : ak == 1 ? by : ak == 2 ? bz : 1.0;
        const hv_d4 c = {0.0, 0.0, 0.0, 0.0};
        const hv_d4 dres = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        // D[i][j] sits in lane (i & 3) * 16 + j, register i >> 2 (tools/mfma_f64_layout.hip, found by one-hot operands): rows 0..2 of
        // point 16 g + j are register 0 of lanes j, 16 + j, 32 + j
        const double rx = __shfl(dres[0], lane & 15), ry = __shfl(dres[0], 16 + (lane & 15)), rz = __shfl(dres[0], 32 + (lane & 15));
        if ((lane >> 4) == g) {
            wx = rx;
            wy = ry;
            wz = rz;
        }
    }
    if (i >= npx) return;
    valid_out[i] = valid ? 0u : HV_SORT_SENTINEL;
    if (!valid) return;
    pts_out[i * 3 + 0] = (float)wx;
    pts_out[i * 3 + 1] = (float)wy;
    pts_out[i * 3 + 2] = (float)wz;
    const uint8_t *cpx = rgb + i * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) cols_out[i * 3 + k] = (float)((double)cpx[k] / 255.0);
}

static void hv_launch_unproject(hipStream_t s, int64_t npx, const void *d_depth, const uint8_t *d_rgb, const HvUnprojectParams &U, float *pts,
                                float *cols, uint32_t *valid) {
    const bool mfma = getenv("HV_VG_UNPROJECT") && strcmp(getenv("HV_VG_UNPROJECT"), "mfma") == 0;
    if (mfma)
        hipLaunchKernelGGL(k_vg_unproject_mfma, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, d_depth, d_rgb, U, pts, cols, valid);
    else
        hipLaunchKernelGGL(k_vg_unproject, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, d_depth, d_rgb, U, pts, cols, valid);
}

// ================================================================================================
// Per-call bin path (production for single frames; hv_bins.h).  The device-wide radix sort above is 18 small launches and 115 of
// the 140 us a 640x480 frame takes (profiles/r01): a frame's points only have to be grouped per BLOCK (a few thousand bins of a
// few dozen to a few hundred points) and ordered inside a block by (voxel, point index) - SURVEY 7.2 K4.
//   k_vgb_bin      1 thread / point: (the pixel's unprojection,) key arithmetic + block claim as k_vg_keys; the point's entry
//                  (local voxel index << 20 | point index) goes straight into its block's bin, its packed record (position +
//                  colour, 16 or 32 bytes) into the record array in point order; a block's first point of the call appends the
//                  block to the touched list (hv_bins_push)
//   k_vgb_fold     1 wave (or workgroup) / touched block: bin -> LDS, the 32-bit entries into ascending order (= by voxel, then by
//                  point index), the first lane of every voxel run folds its points in point order into the voxel record - exactly
//                  the order of the reference's sequential branch, so the result is bit-identical to the radix path and to the
//                  reference.  Bins beyond the LDS capacity (a block that catches > 4096 points of one frame: coarse voxels / very
//                  close surfaces) are folded in point-index windows, still exact.
// 2 launches (rounds 3-5: 4 - count, offsets over every allocated block, scatter, fold).  Needs point indices < 2^20 and
// local_bits <= 12; larger inputs take the radix path.
// ================================================================================================
// The records the bin pass leaves for the fold, in point order: ONE 16-byte load per point (f32 colours: two) where the fold used to
// gather three position and three colour scalars.  8-bit colours stay 8-bit; the fold turns them into the float the reference adds
// through a 256-entry table in LDS: c * (1/255) in float for the binding's uint8 overload (voxel_data.h:82), (float)(c / 255.0) in
// double for a frame's pixels (depth.py:76 divides the uint8 image by 255.0, voxel_grid.py:270-281 casts to float32).
enum { HV_REC_NONE = 0, HV_REC_U8_MUL = 1, HV_REC_U8_DIV = 2, HV_REC_F32 = 3 };
template <int REC>
__device__ __forceinline__ void hv_rec_fetch(const void *__restrict__ rec, int64_t p, const float *lut, float out[6]) {
    if (REC == HV_REC_F32) {
        const float4 a = ((const float4 *)rec)[2 * p], b = ((const float4 *)rec)[2 * p + 1];
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y;
    } else {
        const float4 a = ((const float4 *)rec)[p];
        out[0] = a.x; out[1] = a.y; out[2] = a.z;
        if (REC != HV_REC_NONE) {
            const uint32_t c = __float_as_uint(a.w);
            out[3] = lut[c & 255u]; out[4] = lut[(c >> 8) & 255u]; out[5] = lut[(c >> 16) & 255u];
        }
    }
}
template <int REC>
__device__ __forceinline__ void hv_rec_lut(float *lut) { // by all 256 threads of the workgroup
    if (REC == HV_REC_U8_MUL) lut[threadIdx.x] = (float)threadIdx.x * (1.0f / 255.0f);
    if (REC == HV_REC_U8_DIV) lut[threadIdx.x] = (float)((double)threadIdx.x / 255.0);
    __syncthreads();
}

// FUSED: the points do not exist yet - the thread unprojects its pixel (k_vg_unproject's arithmetic) and goes on with the key: one
// launch and one pass over the points less per RGB-D frame.
template <bool FUSED, int REC>
__global__ __launch_bounds__(HV_BIN_THREADS) void k_vgb_bin(HvTable table, HvBins B, const float *__restrict__ pts, const double *__restrict__ pts64,
                                                  const void *__restrict__ cols, int64_t n, HvGridParams G,
                                                  const uint32_t *__restrict__ valid_mask_keys, HvUnprojectParams U,
                                                  const void *__restrict__ depth_raw, const uint8_t *__restrict__ rgb, void *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool has = false;
    unsigned long long bkey = 0ull;
    uint32_t lidx = 0;
    if (i < n) {
        bool masked;
        float x = 0.f, y = 0.f, z = 0.f;
        double xd = 0.0, yd = 0.0, zd = 0.0; // pts64 != nullptr (never with FUSED): float64 points, see k_vg_keys
        const bool wide = !FUSED && pts64 != nullptr;
        if (FUSED) {
            float pt[3];
            masked = !hv_unproject_point(U, depth_raw, i, pt);
            x = pt[0]; y = pt[1]; z = pt[2];
        } else {
            masked = valid_mask_keys != nullptr && valid_mask_keys[i] == HV_SORT_SENTINEL; // pixel rejected by the unprojection
            if (!masked) {
                if (wide) {
                    xd = pts64[i * 3 + 0]; yd = pts64[i * 3 + 1]; zd = pts64[i * 3 + 2];
                    x = (float)xd; y = (float)yd; z = (float)zd; // the narrowing the voxel sums take (voxel_data.h:54-56)
                } else {
                    x = pts[i * 3 + 0]; y = pts[i * 3 + 1]; z = pts[i * 3 + 2];
                }
            }
        }
        if (!masked) {
            bool foreign = false;
            if (wide ? hv_point_keyable(xd, yd, zd, G) : hv_point_keyable(x, y, z, G)) {
                const HvPointKey k = wide ? hv_point_key(xd, yd, zd, G) : hv_point_key(x, y, z, G);
                if (hv_key_in_range(k.b[0], k.b[1], k.b[2])) {
                    bkey = hv_pack_key(k.b[0], k.b[1], k.b[2]);
                    foreign = hv_block_is_foreign(G, bkey);
                    has = !foreign;
                    lidx = (uint32_t)(k.l[0] + k.l[1] * G.bs + k.l[2] * G.bs * G.bs); // voxel_block.h:67-70
                }
            }
            if (!has && !foreign) atomicAdd(&table.counters[HV_CNT_DROPPED], 1);
            if (has) {
                if (REC == HV_REC_F32) {
                    const float *c = (const float *)cols + i * 3;
                    ((float4 *)rec)[2 * i] = make_float4(x, y, z, c[0]);
                    ((float4 *)rec)[2 * i + 1] = make_float4(c[1], c[2], 0.f, 0.f);
                } else {
                    uint32_t packed = 0u;
                    if (REC != HV_REC_NONE) {
                        const uint8_t *c = (FUSED ? rgb : (const uint8_t *)cols) + i * 3;
                        packed = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
                    }
                    ((float4 *)rec)[i] = make_float4(x, y, z, __uint_as_float(packed));
                }
            }
        }
    }
    hv_bins_push(table, B, has, bkey, lidx, (uint32_t)i);
}

// (Measured dead end, round 3: an ORDER-FREE single-launch form - one thread per pixel adding its point to the voxel record with
// seven hardware atomics, global_atomic_add / global_atomic_add_f32; keys and counts exact, sums within the contract's 1e-4 -
// runs at 89.9 us per 640x480 frame against 45.4 us for the four launches of the point-ordered bucket path: 2.1 M scattered
// float atomics per frame are slower than sorting 12 points per block in LDS.  Taken out again.)
// update_voxel_direct (voxel_block_grid.hpp:524-614) folded over the sorted entries s[0 .. m) of one block: the head of every voxel
// run, one per `stride` threads
template <int REC>
__device__ __forceinline__ void hv_vgb_fold_sorted(const uint32_t *s, int m, int first, int stride, HvVoxel *__restrict__ block,
                                                   const void *__restrict__ rec, const float *lut) {
    for (int e = first; e < m; e += stride) {
        const uint32_t lidx = s[e] >> HV_VGB_IDX_BITS;
        if (e > 0 && (s[e - 1] >> HV_VGB_IDX_BITS) == lidx) continue; // not the head of its voxel's run
        HvVoxel *vx = block + lidx;
        HvVoxel acc = *vx;
        int j = e;
        do {
            float r[6];
            hv_rec_fetch<REC>(rec, (int64_t)(s[j] & ((1u << HV_VGB_IDX_BITS) - 1u)), lut, r);
            acc.pos[0] += r[0];
            acc.pos[1] += r[1];
            acc.pos[2] += r[2];
            if (REC != HV_REC_NONE) {
                acc.col[0] += r[3];
                acc.col[1] += r[4];
                acc.col[2] += r[5];
            }
            acc.count = acc.count == 0 ? 1 : acc.count + 1;
            ++j;
        } while (j < m && (s[j] >> HV_VGB_IDX_BITS) == lidx);
        *vx = acc;
    }
}

__device__ __forceinline__ void hv_vgb_bitonic(uint32_t *s, int m2) { // ascending, m2 a power of two, all threads of the block
    for (int k = 2; k <= m2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < m2; t += blockDim.x) {
                const int x = t ^ j;
                if (x > t) {
                    const uint32_t a = s[t], b = s[x];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) {
                        s[t] = b;
                        s[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// One WAVE per touched slot (a frame's bins hold a few dozen points: a 256-thread workgroup with block barriers per bin spends its
// time in barriers).  Bins of up to HV_VGB_WCAP entries are sorted in the wave's LDS window at once, larger ones in point-index
// windows of HV_VGB_WCAP (correct for any size; when the previous frame had such bins the host launches the workgroup form
// k_vgb_fold instead).
static constexpr int HV_VGB_RANK = 256; // bins up to this size are rank-sorted (<= 4 entries per lane)
static constexpr int HV_VGB_STAGE = 128; // ... and up to this size their points are staged in the window's spare 3 KB (24 B each)
template <int REC>
__global__ __launch_bounds__(256) void k_vgb_fold_wave(HvTable table, HvVoxel *__restrict__ pool, HvBins B, HvGridParams G,
                                                        const void *__restrict__ rec, int64_t n_points, HvStatus *status, int32_t status_seq) {
    __shared__ uint32_t s_all[4][HV_VGB_WCAP];
    __shared__ float s_lut[256];
    hv_rec_lut<REC>(s_lut);
    const HvBinLists L = hv_bins_lists(B);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hv_bins_clear_next(B);                            // the next call's list lengths
        status->pad = table.counters[HV_CNT_OUT2];        // largest bin of this frame
        table.counters[HV_CNT_OUT2] = 0;
        hv_publish_status(table, status, status_seq);
    }
    const int wave = threadIdx.x >> 6, lane = hv_lane_id();
    uint32_t *s = s_all[wave];
    hv_bins_for_each(table, B, L, blockIdx.x * 4 + wave, gridDim.x * 4, [&](const HvBinHeader &h) {
        const int32_t slot = h.slot, nb = h.nb, idx = h.idx;
        const uint32_t head = h.head; // (a bin holds ~24 points: for most bins the header's 64 entries are all of them)
        hv_wave_lds_sync(); // the window of the previous bin is no longer read
        if (lane == 0) B.cnt[slot] = 0; // clean for the next call
        if (idx < 0) return;            // (the block did not get a pool slot: overflow, reported by the caller)
        HvVoxel *block = pool + (int64_t)idx * G.nvox;
        if (nb <= HV_VGB_RANK) {
            // small bin (the common case: a few dozen points): every lane ranks its <= 4 entries against the whole bin
            // (LDS broadcast reads, entries are distinct) and drops them at their rank - no passes, one synchronisation
            uint32_t mine[HV_VGB_RANK / HV_WAVE];
            int rank[HV_VGB_RANK / HV_WAVE];
#pragma unroll
            for (int q = 0; q < HV_VGB_RANK / HV_WAVE; ++q) {
                const int e = lane + q * HV_WAVE;
                mine[q] = e >= nb ? 0xFFFFFFFFu : q == 0 ? head : B.inl[(size_t)slot * HV_BIN_K0 + e]; // (HV_VGB_RANK <= HV_BIN_K0: inline entries only)
                rank[q] = 0;
                if (e < nb) s[e] = mine[q];
            }
            hv_wave_lds_sync();
            const bool staged = nb <= HV_VGB_STAGE;
            // the records of this lane's (still unsorted) entries and the lines of their voxels are requested NOW and travel while
            // the bin is ranked: one memory round trip less on the bin's dependent chain
            float4 r4[2][2];
            int32_t warm = 0;
            if (staged) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (lane + q * HV_WAVE < nb) {
                        const int64_t p = mine[q] & ((1u << HV_VGB_IDX_BITS) - 1u);
                        if (REC == HV_REC_F32) {
                            r4[q][0] = ((const float4 *)rec)[2 * p];
                            r4[q][1] = ((const float4 *)rec)[2 * p + 1];
                        } else {
                            r4[q][0] = ((const float4 *)rec)[p];
                        }
                        warm += block[mine[q] >> HV_VGB_IDX_BITS].count;
                    }
                }
            }
            for (int j = 0; j < nb; ++j) {
                const uint32_t o = s[j];
#pragma unroll
                for (int q = 0; q < HV_VGB_RANK / HV_WAVE; ++q) rank[q] += o < mine[q];
            }
            hv_wave_lds_sync();
#pragma unroll
            for (int q = 0; q < HV_VGB_RANK / HV_WAVE; ++q)
                if (lane + q * HV_WAVE < nb) s[rank[q]] = mine[q];
            if (staged) {
                // ... and land at their entry's rank in the rest of the wave's LDS window; the run heads then add from LDS instead
                // of chasing one global load per point
                float *stage = (float *)(s + HV_VGB_RANK);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (lane + q * HV_WAVE < nb) {
                        float *d = stage + rank[q] * 6;
                        d[0] = r4[q][0].x; d[1] = r4[q][0].y; d[2] = r4[q][0].z;
                        if (REC == HV_REC_F32) {
                            d[3] = r4[q][0].w; d[4] = r4[q][1].x; d[5] = r4[q][1].y;
                        } else if (REC != HV_REC_NONE) {
                            const uint32_t c = __float_as_uint(r4[q][0].w);
                            d[3] = s_lut[c & 255u]; d[4] = s_lut[(c >> 8) & 255u]; d[5] = s_lut[(c >> 16) & 255u];
                        }
                    }
                }
                asm volatile("" ::"v"(warm)); // (the voxel lines have arrived)
                hv_wave_lds_sync();
                for (int e = lane; e < nb; e += HV_WAVE) {
                    const uint32_t lidx = s[e] >> HV_VGB_IDX_BITS;
                    if (e > 0 && (s[e - 1] >> HV_VGB_IDX_BITS) == lidx) continue; // not the head of its voxel's run
                    HvVoxel *vx = block + lidx;
                    HvVoxel acc = *vx;
                    int j = e;
                    do { // update_voxel_direct, in point order
                        acc.pos[0] += stage[j * 6 + 0];
                        acc.pos[1] += stage[j * 6 + 1];
                        acc.pos[2] += stage[j * 6 + 2];
                        if (REC != HV_REC_NONE) {
                            acc.col[0] += stage[j * 6 + 3];
                            acc.col[1] += stage[j * 6 + 4];
                            acc.col[2] += stage[j * 6 + 5];
                        }
                        acc.count = acc.count == 0 ? 1 : acc.count + 1;
                        ++j;
                    } while (j < nb && (s[j] >> HV_VGB_IDX_BITS) == lidx);
                    *vx = acc;
                }
                return;
            }
            hv_wave_lds_sync();
            hv_vgb_fold_sorted<REC>(s, nb, lane, HV_WAVE, block, rec, s_lut);
            return;
        }
        if (nb <= HV_VGB_WCAP) {
            int m2 = HV_WAVE;
            while (m2 < nb) m2 <<= 1;
            for (int e = lane; e < m2; e += HV_WAVE) s[e] = e < nb ? hv_bins_entry(B, slot, e) : 0xFFFFFFFFu;
            hv_wave_lds_sync();
            hv_vgb_bitonic_wave(s, m2);
            hv_vgb_fold_sorted<REC>(s, nb, lane, HV_WAVE, block, rec, s_lut);
            return;
        }
        for (int64_t w = 0; w < n_points; w += HV_VGB_WCAP) { // point-index windows, ascending: a voxel's points stay in order
            hv_wave_lds_sync();
            int m = 0;
            for (int e0 = 0; e0 < nb; e0 += HV_WAVE) {
                const int e = e0 + lane;
                const uint32_t ent = e < nb ? hv_bins_entry(B, slot, e) : 0u;
                const int64_t p = ent & ((1u << HV_VGB_IDX_BITS) - 1u);
                const bool in = e < nb && p >= w && p < w + HV_VGB_WCAP;
                const unsigned long long bm = __ballot(in);
                const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
                if (in) s[m + __popcll(bm & lt)] = ent;
                m += __popcll(bm);
            }
            if (m == 0) continue;
            int m2 = HV_WAVE;
            while (m2 < m) m2 <<= 1;
            for (int e = m + lane; e < m2; e += HV_WAVE) s[e] = 0xFFFFFFFFu;
            hv_wave_lds_sync();
            hv_vgb_bitonic_wave(s, m2);
            hv_vgb_fold_sorted<REC>(s, m, lane, HV_WAVE, block, rec, s_lut);
        }
    });
}

template <int REC>
__global__ __launch_bounds__(256) void k_vgb_fold(HvTable table, HvVoxel *__restrict__ pool, HvBins B, HvGridParams G,
                                                   const void *__restrict__ rec, int64_t n_points, HvStatus *status, int32_t status_seq) {
    __shared__ uint32_t s[HV_VGB_CAP];
    __shared__ float s_lut[256];
    __shared__ int s_m;
    hv_rec_lut<REC>(s_lut);
    const HvBinLists L = hv_bins_lists(B);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hv_bins_clear_next(B);                            // the next call's list lengths
        status->pad = table.counters[HV_CNT_OUT2];        // largest bin of this frame
        table.counters[HV_CNT_OUT2] = 0;
        hv_publish_status(table, status, status_seq);
    }
    for (int t = blockIdx.x; t < L.total; t += gridDim.x) {
        const int32_t slot = hv_bins_touched(B, L, t);
        const int32_t nb = B.cnt[slot];
        const int32_t idx = table.vals[slot];
        __syncthreads(); // s[] of the previous iteration is no longer read (and every thread has read cnt[slot])
        if (threadIdx.x == 0) B.cnt[slot] = 0; // clean for the next call
        if (idx < 0) continue;                 // (the block did not get a pool slot: overflow, reported by the caller)
        HvVoxel *block = pool + (int64_t)idx * G.nvox;
        if (nb <= HV_VGB_CAP) {
            int m2 = 1;
            while (m2 < nb) m2 <<= 1;
            for (int e = threadIdx.x; e < m2; e += blockDim.x) s[e] = e < nb ? hv_bins_entry(B, slot, e) : 0xFFFFFFFFu;
            __syncthreads();
            hv_vgb_bitonic(s, m2);
            hv_vgb_fold_sorted<REC>(s, nb, threadIdx.x, blockDim.x, block, rec, s_lut);
            continue;
        }
        // a bin larger than the LDS window: fold the points of index window [w, w + CAP) at a time, windows ascending - a
        // voxel's points still arrive in point order (a window holds <= CAP entries: point indices are distinct)
        for (int64_t w = 0; w < n_points; w += HV_VGB_CAP) {
            __syncthreads();
            if (threadIdx.x == 0) s_m = 0;
            __syncthreads();
            for (int e = threadIdx.x; e < nb; e += blockDim.x) {
                const uint32_t ent = hv_bins_entry(B, slot, e);
                const int64_t p = ent & ((1u << HV_VGB_IDX_BITS) - 1u);
                if (p >= w && p < w + HV_VGB_CAP) s[atomicAdd(&s_m, 1)] = ent;
            }
            __syncthreads();
            const int m = s_m;
            if (m == 0) continue;
            int m2 = 1;
            while (m2 < m) m2 <<= 1;
            for (int e = m + threadIdx.x; e < m2; e += blockDim.x) s[e] = 0xFFFFFFFFu;
            __syncthreads();
            hv_vgb_bitonic(s, m2);
            hv_vgb_fold_sorted<REC>(s, m, threadIdx.x, blockDim.x, block, rec, s_lut);
        }
    }
}

// Shared predicate of get_voxels / get_voxels_in_bb / get_voxels_in_camera_frustrum / carve.
__device__ __forceinline__ bool hv_voxel_selected(const HvQuery &Q, const HvTable &table, const HvVoxel &vx,
                                                  int64_t b, int l, const HvGridParams &G, float *pos, float *uvd) {
    if (vx.count < Q.min_count) return false;
    const float c = (float)vx.count;
    pos[0] = vx.pos[0] / c; // get_position(), voxel_data.h:58-69
    pos[1] = vx.pos[1] / c;
    pos[2] = vx.pos[2] / c;
    if (Q.kind == 0) return true;
    int32_t bk[3];
    hv_unpack_key(table.block_keys[b], bk[0], bk[1], bk[2]);
    const int32_t lc[3] = {l % G.bs, (l / G.bs) % G.bs, l / (G.bs * G.bs)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (bk[a] < Q.bmin[a] || bk[a] > Q.bmax[a]) return false;
        const int32_t vk = bk[a] * G.bs + lc[a];
        if (vk < Q.vmin[a] || vk > Q.vmax[a]) return false;
    }
    if (Q.kind == 1) { // BoundingBox3D::contains<float>, bounding_boxes_3d.cpp:207-210
        return (double)pos[0] >= Q.bb[0] && (double)pos[0] <= Q.bb[3] && (double)pos[1] >= Q.bb[1] &&
               (double)pos[1] <= Q.bb[4] && (double)pos[2] >= Q.bb[2] && (double)pos[2] <= Q.bb[5];
    }
    return hv_frustum_contains(Q, pos[0], pos[1], pos[2], uvd);
}

__global__ __launch_bounds__(256) void k_vg_collect(HvTable table, const HvVoxel *__restrict__ pool,
                                                     int64_t n_blocks, HvGridParams G, HvQuery Q,
                                                     float *__restrict__ out_pts, float *__restrict__ out_cols,
                                                     int64_t cap) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool pred = false;
    float pos[3] = {0, 0, 0}, uvd[3];
    HvVoxel vx;
    vx.count = 0;
    if (gid < n_blocks * G.nvox) {
        vx = pool[gid];
        pred = hv_voxel_selected(Q, table, vx, gid / G.nvox, (int)(gid % G.nvox), G, pos, uvd);
    }
    const int32_t at = hv_wave_append(&table.counters[HV_CNT_OUT], pred);
    if (pred && at < cap && out_pts != nullptr) {
        const float c = (float)vx.count;
        out_pts[(int64_t)at * 3 + 0] = pos[0];
        out_pts[(int64_t)at * 3 + 1] = pos[1];
        out_pts[(int64_t)at * 3 + 2] = pos[2];
        out_cols[(int64_t)at * 3 + 0] = vx.col[0] / c; // get_color(), voxel_data.h:98-109
        out_cols[(int64_t)at * 3 + 1] = vx.col[1] / c;
        out_cols[(int64_t)at * 3 + 2] = vx.col[2] / c;
    }
}

// carve(), voxel_grid_carving.h:47-79: reset voxels seen in front of the measured depth.
__global__ __launch_bounds__(256) void k_vg_carve(HvTable table, HvVoxel *__restrict__ pool, int64_t n_blocks,
                                                   HvGridParams G, HvQuery Q, const float *__restrict__ depth) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_blocks * G.nvox) return;
    const HvVoxel vx = pool[gid];
    float pos[3], uvd[3];
    if (!hv_voxel_selected(Q, table, vx, gid / G.nvox, (int)(gid % G.nvox), G, pos, uvd)) return;
    const float image_depth = depth[(int64_t)(int)uvd[1] * Q.width + (int)uvd[0]];
    if (image_depth <= 0.0f || !isfinite(image_depth)) return;
    if (uvd[2] < image_depth - Q.carve_threshold) {
        HvVoxel zero;
        memset(&zero, 0, sizeof(zero));
        pool[gid] = zero; // VoxelData::reset(), voxel_data.h:128-132
    }
}

__global__ __launch_bounds__(256) void k_vg_remove_low_count(HvVoxel *__restrict__ pool, int64_t n_voxels, int min_count) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_voxels) return;
    if (pool[gid].count < min_count) {
        HvVoxel zero;
        memset(&zero, 0, sizeof(zero));
        pool[gid] = zero;
    }
}

__global__ __launch_bounds__(256) void k_vg_probe_keys(const float *__restrict__ pts, int64_t n, HvGridParams G,
                                                        int32_t *__restrict__ vk, int32_t *__restrict__ bk,
                                                        int32_t *__restrict__ lk, unsigned long long *__restrict__ hashes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const HvPointKey k = hv_point_key(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], G);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        vk[i * 3 + a] = k.v[a];
        bk[i * 3 + a] = k.b[a];
        lk[i * 3 + a] = k.l[a];
    }
    hashes[i] = hv_reference_hash(k.b[0], k.b[1], k.b[2]);
}


// ---- filter_shadow_points (pyslam/utilities/depth.py:103-146) -----------------------------------
// |d(r,c) - d(r-dy,c)| and |d(r,c) - d(r,c-dx)|; threshold = 3 * 1.4826 * median(positive deltas)
// (float32 arithmetic, as numpy evaluates it for a float32 image); both endpoints of a large jump
// are replaced by fill_value.  The global median is an exact radix SELECT on the deltas' bit patterns (order-preserving for
// positive floats): three histogram passes over the image (12 + 12 + 8 bits), each followed by a one-workgroup pick of the
// bin(s) that hold the two middle ranks; the deltas are recomputed in every pass (two subtractions) instead of being stored,
// and the threshold stays in device memory for the mask kernel - no sort, no host round trip.  (Round 2: the deltas were
// written out, radix-sorted (19 launches) and the middle elements copied to the host - 0.33 ms of a 1.2 ms semantic keyframe,
// profiles/r03/kernel_stats_semantic.csv.)
// state words: [0] number of positive deltas, [1] [2] remaining ranks of the two middle elements, [3] [4] their key prefixes,
// [5] the threshold (float bits)
enum { HV_SH_N = 0, HV_SH_RANK0 = 1, HV_SH_RANK1 = 2, HV_SH_PRE0 = 3, HV_SH_PRE1 = 4, HV_SH_THR = 5, HV_SH_TICKET = 8 /* + pass */, HV_SH_WORDS = 16 };
static constexpr int HV_SH_BINS01 = 4096, HV_SH_BINS2 = 256;
static constexpr int HV_SH_GRID = 256; // workgroups of a histogram pass
static constexpr int HV_SH_HIST_WORDS = HV_SH_BINS01 + 2 * HV_SH_BINS01 + 2 * HV_SH_BINS2;

// one workgroup: the bin holding each of the two ranks, the rank inside it.  (Round 5 tried the pick as the tail of the histogram
// launch - run by its last workgroup to finish, the histogram read back with L2-level loads -: three launches less per keyframe and
// 18 us MORE, 69.6 us for the three passes against 51.5: the 4096 bypassing loads of one workgroup cost more than a 5 us launch.)
template <int PASS>
__device__ __forceinline__ void hv_shadow_pick(const uint32_t *__restrict__ hist, uint32_t *__restrict__ state) {
    constexpr int BINS = PASS == 2 ? HV_SH_BINS2 : HV_SH_BINS01;
    constexpr int PER = BINS / 256;
    constexpr int BITS = PASS == 2 ? 8 : 12;
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t s_rank[2];
    const int t = threadIdx.x;
    for (int target = 0; target < 2; ++target) {
        const uint32_t *h = hist + (PASS == 0 ? 0 : target * BINS);
        uint32_t local[PER], sum = 0u;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            local[k] = h[t * PER + k];
            sum += local[k];
        }
        s_scan[t] = sum;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) { // Hillis-Steele inclusive scan
            const uint32_t addv = t >= off ? s_scan[t - off] : 0u;
            __syncthreads();
            s_scan[t] += addv;
            __syncthreads();
        }
        if (PASS == 0 && t == 0) {
            const uint32_t n = s_scan[255];
            if (target == 0) state[HV_SH_N] = n;
            s_rank[target] = n ? (target == 0 ? (n - 1u) / 2u : n / 2u) : 0u; // ranks of np.median's two middle elements
        }
        if (PASS > 0 && t == 0) s_rank[target] = state[HV_SH_RANK0 + target];
        __syncthreads();
        const uint32_t rank = s_rank[target];
        const uint32_t incl = s_scan[t];
        uint32_t before = incl - sum;
        if (rank >= before && rank < incl) { // exactly one thread when the rank exists
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                if (rank < before + local[k]) {
                    const uint32_t prefix = PASS == 0 ? 0u : state[HV_SH_PRE0 + target];
                    state[HV_SH_PRE0 + target] = (prefix << BITS) | (uint32_t)(t * PER + k);
                    state[HV_SH_RANK0 + target] = rank - before;
                    break;
                }
                before += local[k];
            }
        }
        __syncthreads();
    }
    if (PASS == 2 && t == 0) {
        float thr = __uint_as_float(0x7fc00000u); // np.median of an empty array -> nan -> nothing is masked
        if (state[HV_SH_N]) {
            const float a = __uint_as_float(state[HV_SH_PRE0]), b = __uint_as_float(state[HV_SH_PRE1]);
            const float mad = state[HV_SH_PRE0] == state[HV_SH_PRE1] ? a : (a + b) * 0.5f; // np.median -> np.mean of the two middle float32
            const float sigma = 1.4826f * mad;
            thr = 3.0f * sigma;
        }
        state[HV_SH_THR] = __float_as_uint(thr);
    }
}

template <int PASS>
__global__ __launch_bounds__(256) void k_shadow_hist(const float *__restrict__ depth, int H, int W, int dx, int dy,
                                                      uint32_t *__restrict__ state, uint32_t *__restrict__ hist) {
    constexpr int BINS = PASS == 2 ? HV_SH_BINS2 : HV_SH_BINS01;
    constexpr int NB = (PASS == 0 ? 1 : 2) * BINS;
    __shared__ uint32_t s_h[NB];
    for (int i = threadIdx.x; i < NB; i += 256) s_h[i] = 0u;
    __syncthreads();
    const uint32_t p0 = PASS > 0 ? state[HV_SH_PRE0] : 0u, p1 = PASS > 0 ? state[HV_SH_PRE1] : 0u;
    auto add = [&](float v) {
        if (!(v > 0.0f)) return; // NaN compares false, like numpy's `delta_values > 0`
        const uint32_t key = __float_as_uint(v);
        if (PASS == 0) {
            atomicAdd(&s_h[key >> 20], 1u);
        } else if (PASS == 1) {
            if ((key >> 20) == p0) atomicAdd(&s_h[(key >> 8) & 0xfffu], 1u);
            if ((key >> 20) == p1) atomicAdd(&s_h[BINS + ((key >> 8) & 0xfffu)], 1u);
        } else {
            if ((key >> 8) == p0) atomicAdd(&s_h[key & 0xffu], 1u);
            if ((key >> 8) == p1) atomicAdd(&s_h[BINS + (key & 0xffu)], 1u);
        }
    };
    const int npx = H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
        const int r = i / W, c = i - r * W;
        const float d = depth[i];
        if (dy > 0 && r >= dy) add(fabsf(d - depth[i - dy * W]));
        if (dx > 0 && c >= dx) add(fabsf(d - depth[i - dx]));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += 256)
        if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

template <int PASS>
__global__ __launch_bounds__(256) void k_shadow_pick(const uint32_t *__restrict__ hist, uint32_t *__restrict__ state) {
    hv_shadow_pick<PASS>(hist, state);
}

__global__ __launch_bounds__(256) void k_shadow_mask(const float *__restrict__ depth, int H, int W, int dx, int dy,
                                                      const uint32_t *__restrict__ state, float fill, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const float thr = __uint_as_float(state[HV_SH_THR]);
    const int r = (int)(i / W), c = (int)(i % W);
    const float d = depth[i];
    bool m = false;
    if (dy > 0) {
        if (r >= dy) m |= fabsf(d - depth[i - (int64_t)dy * W]) > thr;
        if (r + dy < H) m |= fabsf(depth[i + (int64_t)dy * W] - d) > thr;
    }
    if (dx > 0) {
        if (c >= dx) m |= fabsf(d - depth[i - dx]) > thr;
        if (c + dx < W) m |= fabsf(depth[i + dx] - d) > thr;
    }
    out[i] = m ? fill : d;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static HvGridParams grid_params(const hv_volume *v) {
    HvGridParams G;
    const float vs = (float)v->cfg.voxel_size; // pybind narrows the Python double to float first
    G.inv_voxel_size = 1.0f / vs;
    G.bs = v->cfg.block_size;
    G.nvox = G.bs * G.bs * G.bs;
    G.local_bits = v->local_bits;
    G.bs_shift = -1;
    for (int sh = 0; sh <= 4; ++sh)
        if ((1 << sh) == G.bs) G.bs_shift = sh;
    G.owner_rank = v->owner_rank;
    G.owner_world = v->owner_world;
    return G;
}

static int sort_bits(const hv_volume *v) {
    int slot_bits = 0;
    while ((1ull << slot_bits) < v->table_capacity) slot_bits++;
    return std::min(32, slot_bits + v->local_bits + 1);
}

static int ensure_sort_tmp(hv_volume *v, int64_t n) {
    size_t bytes = 0;
    HV_HIP(rocprim::radix_sort_pairs(nullptr, bytes, v->sort_keys_in, v->sort_keys_out, v->sort_vals_in,
                                     v->sort_vals_out, (size_t)n, 0, sort_bits(v), v->stream));
    return hv_ensure_buffer(v, &v->sort_tmp, &v->sort_tmp_bytes, bytes);
}

// keys -> group -> ordered reduce over device-resident points/colours.  Single frames (n < 2^20 points) take the bin path
// (2 launches), larger inputs - the batched replay - the device-wide radix sort; HV_VG_PATH=sort forces the latter (A/B, tests).
// `frame` != nullptr: the points do not exist yet - they are one posed RGB-D frame (device pointers + unprojection constants);
// the bin path unprojects inside its bin kernel, the radix path launches k_vg_unproject first.
struct HvFrameSource {
    HvUnprojectParams U;
    const void *d_depth;
    const uint8_t *d_rgb;
};

static int integrate_device_points(hv_volume *v, const float *d_pts, int64_t n, const void *d_cols,
                                   int color_kind, const uint32_t *d_valid, const HvFrameSource *frame = nullptr,
                                   const double *d_pts64 = nullptr) {
    const HvGridParams G = grid_params(v);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    bool checked = false;
    int rc = hv_capacity_gate(v, &checked); // may grow the pool: the sort-key width / bucket arrays follow the table
    if (rc != HV_OK) return rc;
    const char *force = getenv("HV_VG_PATH");
    const bool bins = hv_bins_usable(v, n, HV_VGB_IDX_BITS) && v->local_bits <= 32 - HV_VGB_IDX_BITS && !(force && strcmp(force, "sort") == 0);
    if (bins) {
        const int rec_kind = frame ? HV_REC_U8_DIV : color_kind == HV_COLOR_U8 ? HV_REC_U8_MUL : color_kind == HV_COLOR_F32 ? HV_REC_F32 : HV_REC_NONE;
        rc = hv_ensure_buffer(v, &v->bin_rec, &v->bin_rec_bytes, (size_t)32 * (size_t)std::min<int64_t>(v->cfg.max_points, 1ll << HV_VGB_IDX_BITS));
        if (rc != HV_OK) return rc;
        hv_profile_begin(v); // measurement hook: the two launches of one integrate call
        HvBins B{};
        for (int attempt = 0;; ++attempt) {
            rc = hv_bins_ensure(v);
            if (rc != HV_OK) return rc;
            B = hv_bins_begin(v, HV_VGB_IDX_BITS);
#define HV_LAUNCH_BIN(FUSED, REC)                                                                                      \
    hipLaunchKernelGGL((k_vgb_bin<FUSED, REC>), dim3((unsigned)((n + HV_BIN_THREADS - 1) / HV_BIN_THREADS)), dim3(HV_BIN_THREADS), 0, v->stream, v->table, B, d_pts, d_pts64, d_cols, n, G, \
                       d_valid, frame ? frame->U : HvUnprojectParams{}, frame ? frame->d_depth : (const void *)nullptr,  \
                       frame ? frame->d_rgb : (const uint8_t *)nullptr, v->bin_rec)
            if (frame) HV_LAUNCH_BIN(true, HV_REC_U8_DIV);
            else if (rec_kind == HV_REC_U8_MUL) HV_LAUNCH_BIN(false, HV_REC_U8_MUL);
            else if (rec_kind == HV_REC_F32) HV_LAUNCH_BIN(false, HV_REC_F32);
            else HV_LAUNCH_BIN(false, HV_REC_NONE);
#undef HV_LAUNCH_BIN
            if (!checked) break;
            rc = hv_claims_fit(v); // blocks that did not fit: grow and claim again (the bins restart from clean arrays)
            if (rc == HV_OK) break;
            v->bins_clean = false; // counts / lists of the aborted claim pass are void (and the table may have moved)
            if (rc != HV_RETRY_CLAIM || attempt >= 8) return rc == HV_RETRY_CLAIM ? HV_ERR_CAPACITY : rc;
        }
        v->bins.parity ^= 1;
        const int32_t seq = hv_next_status_seq(v);
        // one wave per bin, or - when the last finished frame had bins beyond a wave's LDS window (coarse voxels, very
        // close surfaces) - one workgroup per bin; both are exact for any bin size
        const bool big = v->h_status->pad > HV_VGB_WCAP;
        // wave form: persistent waves, one per wave slot of the machine (hv_bins_for_each); workgroup form: one workgroup per bin
        const unsigned big_grid = (unsigned)std::min<int64_t>(std::max<int64_t>(n / 32, 256), 16384);
#define HV_LAUNCH_FOLD(REC)                                                                                            \
    do {                                                                                                               \
        if (big)                                                                                                       \
            hipLaunchKernelGGL(k_vgb_fold<REC>, dim3(big_grid), dim3(256), 0, v->stream, v->table, (HvVoxel *)v->pool, B, G, \
                               (const void *)v->bin_rec, n, v->d_status, seq);                                          \
        else                                                                                                           \
            hipLaunchKernelGGL(k_vgb_fold_wave<REC>, dim3(hv_bins_fold_grid(v, k_vgb_fold_wave<REC>, 0)), dim3(256), 0, v->stream, v->table, \
                               (HvVoxel *)v->pool, B, G, (const void *)v->bin_rec, n, v->d_status, seq);                \
    } while (0)
        if (rec_kind == HV_REC_U8_DIV) HV_LAUNCH_FOLD(HV_REC_U8_DIV);
        else if (rec_kind == HV_REC_U8_MUL) HV_LAUNCH_FOLD(HV_REC_U8_MUL);
        else if (rec_kind == HV_REC_F32) HV_LAUNCH_FOLD(HV_REC_F32);
        else HV_LAUNCH_FOLD(HV_REC_NONE);
#undef HV_LAUNCH_FOLD
        hv_profile_end(v, n);
        HV_HIP(hipGetLastError());
        v->frame_counter += 1;
        return HV_OK;
    }
    if (frame) { // radix path on a frame: unproject first (validity flags go straight into the sort-key input buffer)
        hv_launch_unproject(v->stream, n, frame->d_depth, frame->d_rgb, frame->U, (float *)d_pts, (float *)d_cols, v->sort_keys_out);
        d_valid = v->sort_keys_out;
    }
    rc = ensure_sort_tmp(v, n);
    if (rc != HV_OK) return rc;
    hv_profile_begin(v); // measurement hook: keys + sort + ordered reduce of one integrate call
    for (int attempt = 0;; ++attempt) {
        hipLaunchKernelGGL(k_vg_keys, dim3(blocks), dim3(256), 0, v->stream, v->table, (float *)d_pts, n, G, v->sort_keys_in,
                           v->sort_vals_in, d_valid, d_pts64);
        if (!checked) break;
        rc = hv_claims_fit(v); // blocks that did not fit: grow, claim again (sort keys embed table slots)
        if (rc == HV_OK) break;
        if (rc != HV_RETRY_CLAIM || attempt >= 8) return rc == HV_RETRY_CLAIM ? HV_ERR_CAPACITY : rc;
        rc = ensure_sort_tmp(v, n);
        if (rc != HV_OK) return rc;
    }
    size_t tmp_bytes = v->sort_tmp_bytes;
    HV_HIP(rocprim::radix_sort_pairs(v->sort_tmp, tmp_bytes, v->sort_keys_in, v->sort_keys_out, v->sort_vals_in,
                                     v->sort_vals_out, (size_t)n, 0, sort_bits(v), v->stream));
    if (color_kind == HV_COLOR_U8) {
        hipLaunchKernelGGL(k_vg_reduce<HV_COLOR_U8>, dim3(blocks), dim3(256), 0, v->stream, v->table,
                           (HvVoxel *)v->pool, v->sort_keys_out, v->sort_vals_out, n, G, d_pts, d_cols);
    } else if (color_kind == HV_COLOR_F32) {
        hipLaunchKernelGGL(k_vg_reduce<HV_COLOR_F32>, dim3(blocks), dim3(256), 0, v->stream, v->table,
                           (HvVoxel *)v->pool, v->sort_keys_out, v->sort_vals_out, n, G, d_pts, d_cols);
    } else {
        hipLaunchKernelGGL(k_vg_reduce<HV_COLOR_NONE>, dim3(blocks), dim3(256), 0, v->stream, v->table,
                           (HvVoxel *)v->pool, v->sort_keys_out, v->sort_vals_out, n, G, d_pts, d_cols);
    }
    hv_launch_publish_status(v); // pool occupancy for the next call's hv_capacity_gate
    hv_profile_end(v, n);
    HV_HIP(hipGetLastError());
    v->frame_counter += 1;
    return HV_OK;
}

static int run_collect(hv_volume *v, const HvQuery &Q, float *points, float *colors, int64_t cap, int64_t *n,
                       int32_t loc) {
    HV_REQUIRE(n != nullptr, HV_ERR_INVALID, "get_voxels: null count pointer");
    HV_REQUIRE(v->cfg.mode == HV_MODE_VOXEL_GRID, HV_ERR_MODE, "get_voxels: volume is not in VOXEL_GRID mode");
    HV_HIP(hipSetDevice(v->device));
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    *n = 0;
    if (nb == 0) return HV_OK;
    const HvGridParams G = grid_params(v);
    const int64_t total = nb * G.nvox;
    const bool want = points != nullptr && colors != nullptr && cap > 0;
    float *d_pts = nullptr, *d_cols = nullptr;
    if (want) {
        if (loc == HV_DEVICE) {
            d_pts = points;
            d_cols = colors;
        } else {
            rc = hv_ensure_buffer(v, &v->out_a, &v->out_a_bytes, sizeof(float) * 3 * (size_t)cap);
            if (rc != HV_OK) return rc;
            rc = hv_ensure_buffer(v, &v->out_b, &v->out_b_bytes, sizeof(float) * 3 * (size_t)cap);
            if (rc != HV_OK) return rc;
            d_pts = (float *)v->out_a;
            d_cols = (float *)v->out_b;
        }
    }
    HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT], 0, sizeof(int32_t), v->stream));
    hipLaunchKernelGGL(k_vg_collect, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->stream, v->table,
                       (const HvVoxel *)v->pool, nb, G, Q, d_pts, d_cols, want ? cap : 0);
    HV_HIP(hipGetLastError());
    rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    *n = v->h_counters[HV_CNT_OUT];
    if (want && loc == HV_HOST) {
        const int64_t m = std::min(*n, cap);
        if (m > 0) {
            HV_HIP(hipMemcpyAsync(points, d_pts, sizeof(float) * 3 * m, hipMemcpyDeviceToHost, v->stream));
            HV_HIP(hipMemcpyAsync(colors, d_cols, sizeof(float) * 3 * m, hipMemcpyDeviceToHost, v->stream));
            HV_HIP(hipStreamSynchronize(v->stream));
        }
    }
    return HV_OK;
}

extern "C" {

int hv_integrate_points(hv_volume *v, const float *points, int64_t n, const void *colors, int32_t color_dtype,
                        int32_t loc) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_integrate_points: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_VOXEL_GRID, HV_ERR_MODE, "hv_integrate_points: volume is not in VOXEL_GRID mode");
    if (n == 0) return HV_OK; // integrate_raw: `if (num_points == 0) return;`
    HV_REQUIRE(points != nullptr && n > 0, HV_ERR_INVALID, "points must be a contiguous Nx3 array");
    HV_REQUIRE(color_dtype == HV_COLOR_NONE || color_dtype == HV_COLOR_U8 || color_dtype == HV_COLOR_F32,
               HV_ERR_INVALID, "Colors must be uint8 or float32");
    HV_REQUIRE(color_dtype == HV_COLOR_NONE || colors != nullptr, HV_ERR_INVALID,
               "points and colors must have the same size");
    HV_REQUIRE(n <= v->cfg.max_points, HV_ERR_CAPACITY, "hv_integrate_points: %lld points exceed max_points=%lld",
               (long long)n, (long long)v->cfg.max_points);
    HV_HIP(hipSetDevice(v->device));
    const void *d_pts = nullptr, *d_cols = nullptr;
    int rc = hv_stage_in(v, points, sizeof(float) * 3 * (size_t)n, loc, 0, &d_pts);
    if (rc != HV_OK) return rc;
    if (color_dtype != HV_COLOR_NONE) {
        rc = hv_stage_in(v, colors, (color_dtype == HV_COLOR_U8 ? 1 : 4) * 3 * (size_t)n, loc, 1, &d_cols);
        if (rc != HV_OK) return rc;
    }
    return integrate_device_points(v, (const float *)d_pts, n, d_cols, color_dtype, nullptr);
}

int hv_integrate_points_f64(hv_volume *v, const double *points, int64_t n, const void *colors, int32_t color_dtype,
                            int32_t loc) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_integrate_points_f64: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_VOXEL_GRID, HV_ERR_MODE, "hv_integrate_points_f64: volume is not in VOXEL_GRID mode");
    if (n == 0) return HV_OK; // integrate_raw: `if (num_points == 0) return;`
    HV_REQUIRE(points != nullptr && n > 0, HV_ERR_INVALID, "points must be a contiguous Nx3 array");
    HV_REQUIRE(color_dtype == HV_COLOR_NONE || color_dtype == HV_COLOR_U8 || color_dtype == HV_COLOR_F32,
               HV_ERR_INVALID, "Colors must be uint8 or float32");
    HV_REQUIRE(color_dtype == HV_COLOR_NONE || colors != nullptr, HV_ERR_INVALID,
               "points and colors must have the same size");
    HV_REQUIRE(n <= v->cfg.max_points, HV_ERR_CAPACITY, "hv_integrate_points_f64: %lld points exceed max_points=%lld",
               (long long)n, (long long)v->cfg.max_points);
    HV_HIP(hipSetDevice(v->device));
    const void *d_pts64 = nullptr, *d_cols = nullptr;
    int rc = hv_stage_in(v, points, sizeof(double) * 3 * (size_t)n, loc, 0, &d_pts64);
    if (rc != HV_OK) return rc;
    if (color_dtype != HV_COLOR_NONE) {
        rc = hv_stage_in(v, colors, (color_dtype == HV_COLOR_U8 ? 1 : 4) * 3 * (size_t)n, loc, 1, &d_cols);
        if (rc != HV_OK) return rc;
    }
    // keys from the doubles; the key kernel leaves the float32 narrowing the voxel sums take in scratch_points
    return integrate_device_points(v, v->scratch_points, n, d_cols, color_dtype, nullptr, nullptr, (const double *)d_pts64);
}

} // extern "C" (reopened below)

// Shared front half of the fused RGB-D entry points: stages the frame, runs k_vg_unproject into
// v->scratch_points / v->scratch_colors (float32) with the per-pixel validity flags in v->sort_keys_out.
// d_depth_out receives the device address of the (staged) depth image.
// k_vg_unproject of one frame already in HBM into slice [out_offset, out_offset + H*W) of the scratch arrays.
static int unproject_device(hv_volume *v, const void *d_depth, int32_t depth_dtype, double depth_scale, const uint8_t *d_rgb,
                            int32_t height, int32_t width, const double *intr, const double *T_cw, double min_depth,
                            double max_depth, int64_t out_offset) {
    const int64_t npx = (int64_t)height * width;
    const HvUnprojectParams U = unproject_params(depth_dtype, depth_scale, height, width, intr, T_cw, min_depth, max_depth);
    // the unprojection's validity flags go straight into the sort-key input buffer
    hv_launch_unproject(v->stream, npx, d_depth, d_rgb, U, v->scratch_points + 3 * out_offset, v->scratch_colors + 3 * out_offset,
                        v->sort_keys_out + out_offset);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

int hv_unproject_frame(hv_volume *v, const void *depth, int32_t depth_dtype, double depth_scale, const uint8_t *rgb,
                       int32_t height, int32_t width, const double *intr, const double *T_cw, double min_depth,
                       double max_depth, int32_t loc, const void **d_depth_out) {
    const int64_t npx = (int64_t)height * width;
    const void *d_depth = nullptr, *d_rgb = nullptr;
    int rc = hv_stage_in(v, depth, (size_t)npx * (depth_dtype == HV_DEPTH_U16 ? 2 : 4), loc, 0, &d_depth);
    if (rc != HV_OK) return rc;
    rc = hv_stage_in(v, rgb, (size_t)npx * 3, loc, 1, &d_rgb);
    if (rc != HV_OK) return rc;
    if (d_depth_out) *d_depth_out = d_depth;
    return unproject_device(v, d_depth, depth_dtype, depth_scale, (const uint8_t *)d_rgb, height, width, intr, T_cw, min_depth,
                            max_depth, 0);
}

// Batched replay (rebuild() / offline reconstruction): F posed frames are unprojected into one point array and fused
// with ONE sort + reduce per chunk of max_points / (H*W) frames.  A voxel's points are still folded in global point
// index order = frame order, then pixel order, so the result is bit-identical to F hv_integrate_rgbd_points calls;
// the device-wide radix sort (18 small launches: the per-frame cost of this path) is paid once per chunk.
extern "C" int hv_integrate_rgbd_points_batch(hv_volume *v, const void *depth, int32_t depth_dtype, double depth_scale,
                                              const uint8_t *rgb, int32_t n_frames, int32_t height, int32_t width,
                                              const double *intr, const double *T_cw, double min_depth, double max_depth,
                                              int32_t loc) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_integrate_rgbd_points_batch: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_VOXEL_GRID, HV_ERR_MODE, "hv_integrate_rgbd_points_batch: volume is not in VOXEL_GRID mode");
    HV_REQUIRE(depth != nullptr && rgb != nullptr && intr != nullptr && T_cw != nullptr && height > 0 && width > 0 && n_frames > 0,
               HV_ERR_INVALID, "hv_integrate_rgbd_points_batch: null or empty input");
    const int64_t npx = (int64_t)height * width;
    HV_REQUIRE(npx <= v->cfg.max_points, HV_ERR_CAPACITY, "hv_integrate_rgbd_points_batch: image exceeds max_points");
    HV_HIP(hipSetDevice(v->device));
    const size_t dsz = depth_dtype == HV_DEPTH_U16 ? 2 : 4;
    const void *d_depth = nullptr, *d_rgb = nullptr;
    int rc = hv_stage_in(v, depth, (size_t)npx * dsz * n_frames, loc, 0, &d_depth);
    if (rc != HV_OK) return rc;
    rc = hv_stage_in(v, rgb, (size_t)npx * 3 * n_frames, loc, 1, &d_rgb);
    if (rc != HV_OK) return rc;
    const int per_chunk = (int)std::max<int64_t>(1, v->cfg.max_points / npx);
    for (int f0 = 0; f0 < n_frames; f0 += per_chunk) {
        const int nf = std::min(per_chunk, n_frames - f0);
        for (int f = 0; f < nf; ++f) {
            rc = unproject_device(v, (const char *)d_depth + npx * dsz * (size_t)(f0 + f), depth_dtype, depth_scale,
                                  (const uint8_t *)d_rgb + npx * 3 * (size_t)(f0 + f), height, width, intr,
                                  T_cw + 16 * (size_t)(f0 + f), min_depth, max_depth, (int64_t)f * npx);
            if (rc != HV_OK) return rc;
        }
        rc = integrate_device_points(v, v->scratch_points, (int64_t)nf * npx, v->scratch_colors, HV_COLOR_F32, v->sort_keys_out);
        if (rc != HV_OK) return rc;
    }
    return HV_OK;
}

// Multi-GPU block ownership for the VOXEL_GRID mode (the "zero reduce" form: owner(block) = hash(block key) mod world,
// every GPU sees every point and fuses only the blocks it owns; the union of the GPUs' voxels is the single-GPU grid, bit for
// bit, and no collective runs while fusing).  hv_block_owner is the same function on the host (tests, planners).
extern "C" int hv_set_owner(hv_volume *v, int32_t rank, int32_t world_size) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_set_owner: null volume");
    HV_REQUIRE(v->cfg.mode != HV_MODE_TSDF, HV_ERR_MODE, "hv_set_owner: grid modes only (TSDF: hv_tsdf_set_owner)");
    HV_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, HV_ERR_INVALID, "hv_set_owner: bad rank/world");
    v->owner_rank = rank;
    v->owner_world = world_size;
    return HV_OK;
}

extern "C" int hv_block_owner(const int32_t *block_keys, int64_t n, int32_t world_size, int32_t *owner) {
    HV_REQUIRE((n == 0 || (block_keys != nullptr && owner != nullptr)) && world_size >= 1, HV_ERR_INVALID, "hv_block_owner: bad argument");
    for (int64_t i = 0; i < n; ++i) {
        const int32_t x = block_keys[i * 3], y = block_keys[i * 3 + 1], z = block_keys[i * 3 + 2];
        owner[i] = hv_key_in_range(x, y, z) ? hv_owner_of(hv_pack_key(x, y, z), world_size) : -1;
    }
    return HV_OK;
}

extern "C" int hv_integrate_rgbd_points(hv_volume *v, const void *depth, int32_t depth_dtype, double depth_scale,
                             const uint8_t *rgb, int32_t height, int32_t width, const double *intr,
                             const double *T_cw, double min_depth, double max_depth, int32_t loc) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_integrate_rgbd_points: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_VOXEL_GRID, HV_ERR_MODE, "hv_integrate_rgbd_points: volume is not in VOXEL_GRID mode");
    HV_REQUIRE(depth != nullptr && rgb != nullptr && intr != nullptr && T_cw != nullptr && height > 0 && width > 0,
               HV_ERR_INVALID, "hv_integrate_rgbd_points: null or empty input");
    const int64_t npx = (int64_t)height * width;
    HV_REQUIRE(npx <= v->cfg.max_points, HV_ERR_CAPACITY, "hv_integrate_rgbd_points: image exceeds max_points");
    HV_HIP(hipSetDevice(v->device));
    HvFrameSource frame;
    const void *d_depth = nullptr, *d_rgb = nullptr;
    int rc = hv_stage_in(v, depth, (size_t)npx * (depth_dtype == HV_DEPTH_U16 ? 2 : 4), loc, 0, &d_depth);
    if (rc != HV_OK) return rc;
    rc = hv_stage_in(v, rgb, (size_t)npx * 3, loc, 1, &d_rgb);
    if (rc != HV_OK) return rc;
    frame.U = unproject_params(depth_dtype, depth_scale, height, width, intr, T_cw, min_depth, max_depth);
    frame.d_depth = d_depth;
    frame.d_rgb = (const uint8_t *)d_rgb;
    return integrate_device_points(v, v->scratch_points, npx, v->scratch_colors, HV_COLOR_F32, nullptr, &frame);
}

extern "C" {

// the seven launches of the filter on `st`: head = [histograms][state] (zeroed here), dd -> d_out (device images)
static int shadow_filter_launch(const float *dd, int32_t height, int32_t width, int32_t delta_x, int32_t delta_y, float fill_value,
                                float *d_out, uint32_t *head, hipStream_t st) {
    const int64_t npx = (int64_t)height * width;
    uint32_t *hist0 = head, *hist1 = hist0 + HV_SH_BINS01, *hist2 = hist1 + 2 * HV_SH_BINS01;
    uint32_t *state = hist0 + HV_SH_HIST_WORDS;
    HV_HIP(hipMemsetAsync(hist0, 0, sizeof(uint32_t) * (HV_SH_HIST_WORDS + HV_SH_WORDS), st));
    // one workgroup per CU (1024 measured the same, round 6: the passes are bound by the LDS atomics of a few dozen hot bins)
    const unsigned hist_grid = (unsigned)std::min<int64_t>((npx + 255) / 256, HV_SH_GRID);
    hipLaunchKernelGGL(k_shadow_hist<0>, dim3(hist_grid), dim3(256), 0, st, dd, height, width, delta_x, delta_y, state, hist0);
    hipLaunchKernelGGL(k_shadow_pick<0>, dim3(1), dim3(256), 0, st, hist0, state);
    hipLaunchKernelGGL(k_shadow_hist<1>, dim3(hist_grid), dim3(256), 0, st, dd, height, width, delta_x, delta_y, state, hist1);
    hipLaunchKernelGGL(k_shadow_pick<1>, dim3(1), dim3(256), 0, st, hist1, state);
    hipLaunchKernelGGL(k_shadow_hist<2>, dim3(hist_grid), dim3(256), 0, st, dd, height, width, delta_x, delta_y, state, hist2);
    hipLaunchKernelGGL(k_shadow_pick<2>, dim3(1), dim3(256), 0, st, hist2, state);
    hipLaunchKernelGGL(k_shadow_mask, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, dd, height, width, delta_x, delta_y, state, fill_value, d_out);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

int hv_filter_shadow_points(hv_volume *v, const float *depth, int32_t height, int32_t width, int32_t delta_x,
                            int32_t delta_y, float fill_value, float *out, int32_t loc) {
    HV_REQUIRE(v != nullptr && depth != nullptr && out != nullptr, HV_ERR_INVALID, "hv_filter_shadow_points: null argument");
    HV_REQUIRE(height > 0 && width > 0 && delta_x >= 0 && delta_y >= 0 && delta_x < width && delta_y < height,
               HV_ERR_INVALID, "hv_filter_shadow_points: bad image size or deltas");
    HV_HIP(hipSetDevice(v->device));
    const int64_t npx = (int64_t)height * width;
    HV_REQUIRE(npx < (int64_t)1 << 30, HV_ERR_INVALID, "hv_filter_shadow_points: image too large");
    const void *d_depth = nullptr;
    int rc = hv_stage_in(v, depth, sizeof(float) * npx, loc, 0, &d_depth);
    if (rc != HV_OK) return rc;
    // scratch: [histograms][state][out npx]
    const size_t head = sizeof(uint32_t) * (HV_SH_HIST_WORDS + HV_SH_WORDS);
    rc = hv_ensure_buffer(v, &v->out_c, &v->out_c_bytes, head + sizeof(float) * npx);
    if (rc != HV_OK) return rc;
    float *d_out = loc == HV_DEVICE ? out : (float *)((char *)v->out_c + head);
    rc = shadow_filter_launch((const float *)d_depth, height, width, delta_x, delta_y, fill_value, d_out, (uint32_t *)v->out_c, v->stream);
    if (rc != HV_OK) return rc;
    // host images: copied back and complete on return; device images: queued on the volume's stream like every other launch
    if (loc == HV_HOST) {
        HV_HIP(hipMemcpyAsync(out, d_out, sizeof(float) * npx, hipMemcpyDeviceToHost, v->stream));
        HV_HIP(hipStreamSynchronize(v->stream));
    }
    return HV_OK;
}

int hv_filter_shadow_points_on_stream(hv_volume *v, const float *depth, int32_t height, int32_t width, int32_t delta_x,
                                      int32_t delta_y, float fill_value, float *out, void *stream) {
    HV_REQUIRE(v != nullptr && depth != nullptr && out != nullptr, HV_ERR_INVALID, "hv_filter_shadow_points_on_stream: null argument");
    HV_REQUIRE(height > 0 && width > 0 && delta_x >= 0 && delta_y >= 0 && delta_x < width && delta_y < height,
               HV_ERR_INVALID, "hv_filter_shadow_points_on_stream: bad image size or deltas");
    HV_REQUIRE((int64_t)height * width < (int64_t)1 << 30, HV_ERR_INVALID, "hv_filter_shadow_points_on_stream: image too large");
    HV_HIP(hipSetDevice(v->device));
    // the filter touches nothing of the volume: it may run beside the volume's own launches (a keyframe's depth is filtered on the
    // upload side while the previous keyframe is fused).  Histograms and state of its own, four sets used in turn.
    constexpr int SETS = 4;
    const size_t head = sizeof(uint32_t) * (HV_SH_HIST_WORDS + HV_SH_WORDS);
    if (v->shadow_ring == nullptr) HV_HIP(hipMalloc(&v->shadow_ring, head * SETS));
    uint32_t *scratch = (uint32_t *)((char *)v->shadow_ring + head * (size_t)v->shadow_ring_next);
    v->shadow_ring_next = (v->shadow_ring_next + 1) % SETS;
    return shadow_filter_launch(depth, height, width, delta_x, delta_y, fill_value, out, scratch, (hipStream_t)stream);
}

int hv_get_voxels(hv_volume *v, int32_t min_count, float min_confidence, float *points, float *colors,
                  int64_t cap, int64_t *n, int32_t loc) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_get_voxels: null volume");
    (void)min_confidence; // non-semantic voxels: count criterion only (voxel_block_grid.hpp:749-751)
    HvQuery Q;
    memset(&Q, 0, sizeof(Q));
    Q.kind = 0;
    Q.min_count = min_count;
    return run_collect(v, Q, points, colors, cap, n, loc);
}

int hv_get_voxels_in_bb(hv_volume *v, const double *bbox, int32_t min_count, float min_confidence, float *points,
                        float *colors, int64_t cap, int64_t *n, int32_t loc) {
    HV_REQUIRE(v != nullptr && bbox != nullptr, HV_ERR_INVALID, "hv_get_voxels_in_bb: null argument");
    (void)min_confidence;
    HvQuery Q;
    memset(&Q, 0, sizeof(Q));
    Q.kind = 1;
    Q.min_count = min_count;
    for (int k = 0; k < 6; ++k) Q.bb[k] = bbox[k];
    fill_key_range(Q, grid_params(v));
    return run_collect(v, Q, points, colors, cap, n, loc);
}

int hv_get_voxels_in_frustum(hv_volume *v, const float *intr_f32, int32_t width, int32_t height, const double *T_cw,
                             float depth_max, float depth_min, int32_t min_count, float min_confidence,
                             float *points, float *colors, int64_t cap, int64_t *n, int32_t loc) {
    HV_REQUIRE(v != nullptr && intr_f32 != nullptr && T_cw != nullptr, HV_ERR_INVALID,
               "hv_get_voxels_in_frustum: null argument");
    (void)min_confidence;
    HvQuery Q;
    memset(&Q, 0, sizeof(Q));
    Q.kind = 2;
    Q.min_count = min_count;
    fill_frustum_query(Q, v, intr_f32, width, height, T_cw, depth_max, depth_min);
    fill_key_range(Q, grid_params(v));
    return run_collect(v, Q, points, colors, cap, n, loc);
}

int hv_carve(hv_volume *v, const float *intr_f32, int32_t width, int32_t height, const double *T_cw, float depth_max,
             float depth_min, const float *depth, float depth_threshold, int32_t loc) {
    HV_REQUIRE(v != nullptr && intr_f32 != nullptr && T_cw != nullptr, HV_ERR_INVALID, "hv_carve: null argument");
    const bool semantic = hv_mode_is_semantic(v->cfg.mode);
    HV_REQUIRE(v->cfg.mode == HV_MODE_VOXEL_GRID || semantic, HV_ERR_MODE, "hv_carve: volume is not a voxel grid");
    if (depth == nullptr || width <= 0 || height <= 0) return HV_OK; // "Depth image is empty": reference returns
    HV_HIP(hipSetDevice(v->device));
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    if (nb == 0) return HV_OK;
    const void *d_depth = nullptr;
    rc = hv_stage_in(v, depth, sizeof(float) * (size_t)width * height, loc, 0, &d_depth);
    if (rc != HV_OK) return rc;
    HvQuery Q;
    memset(&Q, 0, sizeof(Q));
    Q.kind = 3;
    Q.min_count = 1; // iterate_voxels_in_camera_frustrum default, voxel_block_grid.h:161-163
    Q.carve_threshold = depth_threshold;
    fill_frustum_query(Q, v, intr_f32, width, height, T_cw, depth_max, depth_min);
    const HvGridParams G = grid_params(v);
    fill_key_range(Q, G);
    if (semantic) return hv_sem_carve(v, Q, (const float *)d_depth, nb);
    const int64_t total = nb * G.nvox;
    hipLaunchKernelGGL(k_vg_carve, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->stream, v->table,
                       (HvVoxel *)v->pool, nb, G, Q, (const float *)d_depth);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

int hv_remove_low_count_voxels(hv_volume *v, int32_t min_count) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_remove_low_count_voxels: null volume");
    if (hv_mode_is_semantic(v->cfg.mode))
        return hv_sem_segment_op(v, 3, min_count, 0, 0.f);
    HV_REQUIRE(v->cfg.mode == HV_MODE_VOXEL_GRID, HV_ERR_MODE, "hv_remove_low_count_voxels: not a voxel grid");
    HV_HIP(hipSetDevice(v->device));
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    if (nb == 0) return HV_OK;
    const int64_t total = nb * grid_params(v).nvox;
    hipLaunchKernelGGL(k_vg_remove_low_count, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->stream,
                       (HvVoxel *)v->pool, total, min_count);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

int hv_size(hv_volume *v, int64_t *n) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_size: null argument");
    if (hv_mode_is_semantic(v->cfg.mode)) return hv_sem_size(v, n);
    return hv_get_voxels(v, 1, 0.0f, nullptr, nullptr, 0, n, HV_HOST);
}

int hv_dump_blocks(hv_volume *v, int32_t *keys, uint64_t *hashes, int32_t *counts, float *sums, int64_t *n_blocks) {
    HV_REQUIRE(v != nullptr && n_blocks != nullptr, HV_ERR_INVALID, "hv_dump_blocks: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_VOXEL_GRID, HV_ERR_MODE, "hv_dump_blocks: not a VOXEL_GRID volume");
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    *n_blocks = nb;
    if (nb == 0 || (!keys && !hashes && !counts && !sums)) return HV_OK;
    const HvGridParams G = grid_params(v);
    std::vector<uint64_t> bkeys((size_t)nb);
    HV_HIP(hipMemcpy(bkeys.data(), v->table.block_keys, sizeof(uint64_t) * nb, hipMemcpyDeviceToHost));
    std::vector<std::array<int32_t, 3>> xyz((size_t)nb);
    for (int64_t i = 0; i < nb; ++i) hv_unpack_key(bkeys[i], xyz[i][0], xyz[i][1], xyz[i][2]);
    std::vector<int64_t> order((size_t)nb);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return xyz[a] < xyz[b]; });
    std::vector<HvVoxel> host;
    if (counts || sums) {
        host.resize((size_t)nb * G.nvox);
        HV_HIP(hipMemcpy(host.data(), v->pool, sizeof(HvVoxel) * host.size(), hipMemcpyDeviceToHost));
    }
    for (int64_t o = 0; o < nb; ++o) {
        const int64_t i = order[o];
        if (keys) memcpy(keys + o * 3, xyz[i].data(), 12);
        if (hashes) hashes[o] = hv_reference_hash(xyz[i][0], xyz[i][1], xyz[i][2]);
        if (counts || sums) {
            for (int l = 0; l < G.nvox; ++l) {
                const HvVoxel &vx = host[(size_t)i * G.nvox + l];
                if (counts) counts[o * G.nvox + l] = vx.count;
                if (sums) {
                    float *s = sums + (o * G.nvox + l) * 6;
                    s[0] = vx.pos[0]; s[1] = vx.pos[1]; s[2] = vx.pos[2];
                    s[3] = vx.col[0]; s[4] = vx.col[1]; s[5] = vx.col[2];
                }
            }
        }
    }
    return HV_OK;
}

int hv_keys_from_points(hv_volume *v, const float *points, int64_t n, int32_t *voxel_keys, int32_t *block_keys,
                        int32_t *local_keys, uint64_t *block_hashes) {
    HV_REQUIRE(v != nullptr && points != nullptr && voxel_keys && block_keys && local_keys && block_hashes,
               HV_ERR_INVALID, "hv_keys_from_points: null argument");
    if (n == 0) return HV_OK;
    HV_HIP(hipSetDevice(v->device));
    const void *d_pts = nullptr;
    int rc = hv_stage_in(v, points, sizeof(float) * 3 * (size_t)n, HV_HOST, 0, &d_pts);
    if (rc != HV_OK) return rc;
    rc = hv_ensure_buffer(v, &v->out_a, &v->out_a_bytes, sizeof(int32_t) * 9 * (size_t)n);
    if (rc != HV_OK) return rc;
    rc = hv_ensure_buffer(v, &v->out_b, &v->out_b_bytes, sizeof(uint64_t) * (size_t)n);
    if (rc != HV_OK) return rc;
    int32_t *d_vk = (int32_t *)v->out_a, *d_bk = d_vk + 3 * n, *d_lk = d_bk + 3 * n;
    HvGridParams G = grid_params(v);
    hipLaunchKernelGGL(k_vg_probe_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, v->stream,
                       (const float *)d_pts, n, G, d_vk, d_bk, d_lk, (unsigned long long *)v->out_b);
    HV_HIP(hipGetLastError());
    HV_HIP(hipMemcpyAsync(voxel_keys, d_vk, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipMemcpyAsync(block_keys, d_bk, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipMemcpyAsync(local_keys, d_lk, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipMemcpyAsync(block_hashes, v->out_b, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}

} // extern "C"
