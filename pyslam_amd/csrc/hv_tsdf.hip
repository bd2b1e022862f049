// libpyslam_hipvol.so — TSDF fusion kernels (Open3D ScalableTSDFVolume semantics) for gfx950.
//
// Online path, per frame (reference call site pyslam/dense/volumetric_integrator_tsdf.py:215-223):
//   k_tsdf_prep_touch   one launch, two block roles:
//       prep  blocks: depth -> float metres with depth_scale/depth_trunc applied
//                     (Image::ConvertDepthToFloatImage), RGB u8x3 packed: one 8-byte
//                     {depth, rgb} record per pixel (the sweep's per-voxel gather);
//       touch blocks: every `stride`-th pixel is back-projected in f64 and the volume units within
//                     +/- sdf_trunc are claimed in the block hash (ScalableTSDFVolume::Integrate
//                     front half), de-duplicated per wave; the first toucher of a unit this frame
//                     appends it to the touched list.
//   k_tsdf_integrate    one workgroup (4 waves) per touched unit; wave w owns z in [4w, 4w+4); lane
//                       (x, 4 y's).  Each z-slab of each plane is one contiguous 1 KiB dwordx4
//                       burst per wave.  Arithmetic follows UniformTSDFVolume::
//                       IntegrateWithDepthToCameraDistanceMultiplier operation by operation
//                       (compiled with -ffp-contract=off; IEEE div/sqrt) so tsdf and weight are
//                       bit-identical to the CPU restatement in oracle/tsdf_oracle.c.
//   HBM-bound: algorithmic bytes per touched unit = 4096 voxels * 20 B read (+ 20 B per updated
//   voxel written); no reuse of voxel data within a frame.  Frame gathers are served by L1/L2 (a
//   640x480 frame = 2.4 MB packed, resident in every XCD's 4 MiB L2).
// Multi-frame sweep (hv_tsdf_integrate_batch): k_tsdf_prep_touch_batch + k_tsdf_integrate_batch further
// down — unit slabs are reused in registers across up to 64 frames.
// Multi-GPU hooks: image-tile restriction (hv_tsdf_set_tile) and unit ownership (hv_tsdf_set_owner).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <numeric>

#include "hv_common.h"

static constexpr int R = 16;
static constexpr int RR = R * R;
static constexpr int RRR = R * R * R;
static constexpr int PLANE_BYTES = RRR * 4;
static constexpr int HV_TOUCH_FAN = 8; // lanes per depth sample in the online touch pass
// image-coherent ownership plan (k_tsdf_touch_plan ...): bins of the middle frame's image, column-major
static constexpr int HV_PLAN_NU = 64, HV_PLAN_NV = 16, HV_PLAN_BINS = HV_PLAN_NU * HV_PLAN_NV;
static constexpr uint32_t HV_REC_ONE = 1u << 24; // observation count byte of a batch frame record's colour word

// packed colour word {byte0 = R, byte1 = G, byte2 = B}: a B, G, R source swaps bytes 0 and 2 (one v_perm_b32)
__device__ __forceinline__ uint32_t hv_colour_order(uint32_t c, int bgr) {
    return bgr ? __builtin_amdgcn_perm(0u, c, 0x03000102u) : c;
}

// Image::CreateDepthToCameraDistanceMultiplierFloatImage, evaluated per gather instead of tabulated.
__device__ __forceinline__ float hv_multiplier(const HvFrameParams &P, int u, int v) {
    const float xx = ((float)u - P.cx) * P.ffl_inv_x;
    const float yy = ((float)v - P.cy) * P.ffl_inv_y;
    return sqrtf(xx * xx + yy * yy + 1.0f);
}

// The depth-to-camera-distance multiplier depends on the pixel and the intrinsics only: one table per (intrinsics,
// image size), rebuilt when they change, lets the multi-frame sweep replace ~20 VALU instructions (two conversions, the
// normalisation, a correctly rounded sqrt) per voxel visit by a 4-byte gather at the pixel index it already has.
__global__ __launch_bounds__(256) void k_tsdf_multiplier_table(HvFrameParams P, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.H * P.W) return;
    out[i] = hv_multiplier(P, i % P.W, i / P.W);
}

__device__ __forceinline__ float hv_convert_depth(const HvFrameParams &P, const void *depth_raw, int64_t i) {
    float p = P.depth_is_u16 ? (float)((const uint16_t *)depth_raw)[i] : ((const float *)depth_raw)[i];
    p = p / P.depth_scale_f;
    if ((double)p >= P.depth_trunc_d) p = 0.0f;
    return p;
}

// Conservative test: can any voxel centre of unit (ux,uy,uz) project into this GPU's image tile?
// (Only used to skip units when the frame is tile-sharded across GPUs; with the default whole-image
// tile every touched unit is kept, exactly as in ScalableTSDFVolume::Integrate.)
__device__ inline bool hv_unit_hits_tile(const HvFrameParams &P, int32_t ux, int32_t uy, int32_t uz) {
    if (P.tile_u0 <= 0 && P.tile_v0 <= 0 && P.tile_u1 >= P.W && P.tile_v1 >= P.H) return true;
    const float len = (float)P.unit_length;
    const float o[3] = {(float)((double)ux * P.unit_length), (float)((double)uy * P.unit_length),
                        (float)((double)uz * P.unit_length)};
    float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
    for (int c = 0; c < 8; ++c) {
        const float x = o[0] + ((c & 1) ? len : 0.0f), y = o[1] + ((c & 2) ? len : 0.0f), z = o[2] + ((c & 4) ? len : 0.0f);
        const float pz = P.ext[8] * x + P.ext[9] * y + P.ext[10] * z + P.ext[11];
        if (pz <= 1.0e-3f) return true; // straddles the camera plane: keep
        const float px = P.ext[0] * x + P.ext[1] * y + P.ext[2] * z + P.ext[3];
        const float py = P.ext[4] * x + P.ext[5] * y + P.ext[6] * z + P.ext[7];
        const float u = px * P.fx / pz + P.cx + 0.5f, v = py * P.fy / pz + P.cy + 0.5f;
        umin = fminf(umin, u); umax = fmaxf(umax, u);
        vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
    }
    // tiles on the image border extend outwards without bound: a touched unit that projects entirely outside the image (it
    // only has a sample's +/- sdf_trunc box in view) still belongs to exactly the ranks it is nearest to, so the union of the
    // ranks' units stays Open3D's set of opened units
    const bool u_ok = (P.tile_u0 <= 0 || umax + 2.0f >= (float)P.tile_u0) && (P.tile_u1 >= P.W || umin - 2.0f < (float)P.tile_u1);
    const bool v_ok = (P.tile_v0 <= 0 || vmax + 2.0f >= (float)P.tile_v0) && (P.tile_v1 >= P.H || vmin - 2.0f < (float)P.tile_v1);
    return u_ok && v_ok;
}

// ---- touch pass: PointCloud::CreateFromDepthImage(stride) + unit enumeration, all f64 -------------------------------
// One wave = one 8x8 patch of depth samples (32x32 pixels at stride 4), one lane = one sample: the f64 back-projection
// runs once per sample.  Neighbouring samples open the same few volume units, so the wave first reduces its samples'
// unit ranges to their bounding box, marks every unit some sample's range covers in a per-wave LDS bitmap of the box
// (ScalableTSDFVolume::Integrate opens exactly those), compacts the set bits and hands ONE lane per distinct unit to
// `visit(key, ux, uy, uz)`: all hash probes of a patch are in flight together and a unit is probed once per patch, not
// once per sample.  Boxes larger than HV_TOUCH_BOX_BITS units (a patch straddling a long depth discontinuity) take the
// per-sample loop with ballot de-duplication instead; P.touch_box_bits = 0 forces that path (tests).
static constexpr int HV_TOUCH_PATCH = 8;                   // samples per patch side
static constexpr int HV_TOUCH_BOX_BITS = 2048;             // units in the largest bitmap-enumerated box
static constexpr int HV_TOUCH_BOX_WORDS = HV_TOUCH_BOX_BITS / 32;
static_assert(HV_TOUCH_BOX_WORDS == HV_WAVE, "one bitmap word per lane");

struct HvTouchScratch { // per wave
    uint32_t bits[HV_TOUCH_BOX_WORDS];
    uint16_t list[HV_TOUCH_BOX_BITS];
};

__device__ __forceinline__ int32_t hv_wave_min_i32(int32_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = min(x, __shfl_xor(x, o));
    return x;
}
__device__ __forceinline__ int32_t hv_wave_max_i32(int32_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = max(x, __shfl_xor(x, o));
    return x;
}

__host__ __device__ inline int hv_touch_patches_1d(int extent, int stride) {
    return ((extent + stride - 1) / stride + HV_TOUCH_PATCH - 1) / HV_TOUCH_PATCH;
}
__host__ __device__ inline int hv_touch_patches(int W, int H, int stride) {
    return hv_touch_patches_1d(W, stride) * hv_touch_patches_1d(H, stride);
}
__host__ __device__ inline int hv_touch_patches(const HvFrameParams &P) { return hv_touch_patches(P.W, P.H, P.stride); }

template <typename Visit>
__device__ __forceinline__ void hv_touch_patch(const HvTable &table, const HvFrameParams &P, const void *depth_f, int patch,
                                               HvTouchScratch &scratch, Visit visit) {
    const int ns_w = (P.W + P.stride - 1) / P.stride;
    const int ns_h = (P.H + P.stride - 1) / P.stride;
    const int pw = hv_touch_patches_1d(P.W, P.stride);
    const int lane = hv_lane_id();
    const int sj = (patch % pw) * HV_TOUCH_PATCH + (lane & (HV_TOUCH_PATCH - 1));
    const int si = (patch / pw) * HV_TOUCH_PATCH + lane / HV_TOUCH_PATCH;
    int32_t lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1}; // empty range for lanes without a valid sample
    bool has = false;
    if (sj < ns_w && si < ns_h) {
        const int i = si * P.stride;
        const int j = sj * P.stride;
        float p = hv_convert_depth(P, depth_f, (int64_t)i * P.W + j);
        if (P.tiled && p > 0.0f) {
            // Tile-sharded volume: a sample far outside this GPU's image tile cannot open a unit that projects into the tile
            // (hv_unit_hits_tile would refuse every one of them) - leave before the double-precision back-projection.  The
            // sample opens units over an L-infinity box of +/- sdf_trunc per axis, so a corner / voxel centre of an opened unit lies
            // within rad = sqrt(3) (unit_length + sdf_trunc) of its point; for a point q that close, with camera
            // depth >= zn = p - rad > 0, |u_q - u_s| <= (rad / zn) (fx + |u_s - cx|) (same for v).  Border tiles extend
            // outwards without bound, as in hv_unit_hits_tile.
            const float rad = (float)((P.unit_length + P.sdf_trunc_d) * 1.7320508075688772) * 1.001f;
            const float zn = p - rad;
            if (zn > 0.05f) {
                const float k = rad / zn;
                const float mu = k * (P.fx + fabsf((float)j - P.cx)) + 4.0f, mv = k * (P.fy + fabsf((float)i - P.cy)) + 4.0f;
                const bool out_u = (P.tile_u0 > 0 && (float)j + mu < (float)P.tile_u0) || (P.tile_u1 < P.W && (float)j - mu >= (float)P.tile_u1);
                const bool out_v = (P.tile_v0 > 0 && (float)i + mv < (float)P.tile_v0) || (P.tile_v1 < P.H && (float)i - mv >= (float)P.tile_v1);
                if (out_u || out_v) p = 0.0f;
            }
        }
        if (p > 0.0f) {
            const double z = (double)p;
            const double x = ((double)j - P.cx_d) * z / P.fx_d;
            const double y = ((double)i - P.cy_d) * z / P.fy_d;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double pw_r = ((P.pose[r * 4 + 0] * x + P.pose[r * 4 + 1] * y) + P.pose[r * 4 + 2] * z) + P.pose[r * 4 + 3];
                lo[r] = (int32_t)floor((pw_r - P.sdf_trunc_d) / P.unit_length);
                hi[r] = (int32_t)floor((pw_r + P.sdf_trunc_d) / P.unit_length);
            }
            has = hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2];
        }
    }
    if (!__any(has)) return;
    // bounding box of the patch's unit ranges
    int32_t blo[3], bhi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        blo[r] = hv_wave_min_i32(has ? lo[r] : INT32_MAX);
        bhi[r] = hv_wave_max_i32(has ? hi[r] : INT32_MIN);
    }
    const int64_t d0 = (int64_t)bhi[0] - blo[0] + 1, d1 = (int64_t)bhi[1] - blo[1] + 1, d2 = (int64_t)bhi[2] - blo[2] + 1;
    const bool boxed = d0 <= P.touch_box_bits && d1 <= P.touch_box_bits && d2 <= P.touch_box_bits &&
                       d0 * d1 * d2 <= (int64_t)P.touch_box_bits;
    if (boxed) {
        const int e1 = (int)d1, e2 = (int)d2;
        scratch.bits[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (has) {
            for (int32_t x = lo[0]; x <= hi[0]; ++x)
                for (int32_t y = lo[1]; y <= hi[1]; ++y)
                    for (int32_t z = lo[2]; z <= hi[2]; ++z) {
                        const int c = ((x - blo[0]) * e1 + (y - blo[1])) * e2 + (z - blo[2]);
                        atomicOr(&scratch.bits[c >> 5], 1u << (c & 31));
                    }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // compact the set bits: lane l owns word l; its units go to list[prefix(l) ...]
        uint32_t word = scratch.bits[lane];
        const int cnt = __popc(word);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < HV_WAVE; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const int total = __shfl(incl, HV_WAVE - 1);
        int at = incl - cnt;
        while (word) {
            const int b = __ffs((int)word) - 1;
            scratch.list[at++] = (uint16_t)(lane * 32 + b);
            word &= word - 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int n = lane; n < total; n += HV_WAVE) {
            const int c = scratch.list[n];
            const int32_t ux = blo[0] + c / (e1 * e2);
            const int32_t uy = blo[1] + (c / e2) % e1;
            const int32_t uz = blo[2] + c % e2;
            if (hv_key_in_range(ux, uy, uz)) {
                const unsigned long long key = hv_pack_key(ux, uy, uz);
                // unit-ownership sharding: another GPU fuses (and stores) this unit
                if (!(P.owner_world > 1 && hv_owner_of(key, P.owner_world) != P.owner_rank)) visit(key, ux, uy, uz);
            } else {
                atomicAdd(&table.counters[HV_CNT_DROPPED], 1);
            }
        }
        // the next patch of this wave (none today) would reuse the scratch: keep the phases ordered
        __builtin_amdgcn_wave_barrier();
        return;
    }
    // general path: every lane walks its own sample's units; per step the wave's distinct keys are visited once
    const int64_t n0 = (int64_t)hi[0] - lo[0] + 1, n1 = (int64_t)hi[1] - lo[1] + 1, n2 = (int64_t)hi[2] - lo[2] + 1;
    const int64_t count = has ? n0 * n1 * n2 : 0;
    for (int64_t k = 0; __any(k < count); ++k) {
        unsigned long long key = HV_EMPTY_KEY;
        int32_t ux = 0, uy = 0, uz = 0;
        if (k < count) {
            ux = lo[0] + (int32_t)(k / (n1 * n2));
            uy = lo[1] + (int32_t)((k / n2) % n1);
            uz = lo[2] + (int32_t)(k % n2);
            if (hv_key_in_range(ux, uy, uz)) {
                key = hv_pack_key(ux, uy, uz);
                if (P.owner_world > 1 && hv_owner_of(key, P.owner_world) != P.owner_rank) key = HV_EMPTY_KEY;
            } else {
                atomicAdd(&table.counters[HV_CNT_DROPPED], 1);
            }
        }
        // wave-level de-duplication (ballot + shuffle, no memory traffic)
        bool leader = false;
        unsigned long long remaining = __ballot(key != HV_EMPTY_KEY);
        while (remaining) {
            const int first = __ffsll((long long)remaining) - 1;
            const unsigned long long fkey = __shfl(key, first);
            const unsigned long long same = __ballot(key == fkey);
            if (lane == first) leader = true;
            remaining &= ~same;
        }
        if (leader) visit(key, ux, uy, uz);
    }
}


__global__ __launch_bounds__(256) void k_tsdf_prep_touch(HvTable table, int32_t *__restrict__ stamp,
                                                          int32_t *__restrict__ list, int parity,
                                                          const void *__restrict__ depth_raw,
                                                          const uint8_t *__restrict__ rgb,
                                                          uint2 *__restrict__ frame_px, HvFrameParams P,
                                                          int n_touch_blocks) {
    const int64_t npx = (int64_t)P.H * P.W;
    if ((int)blockIdx.x >= n_touch_blocks) { // touch blocks (latency chains) are dispatched first, the streaming prep blocks fill in
        // prep role: one 8-byte {depth f32, rgb packed} record per pixel so that the per-voxel
        // gather of the integrate kernel is a single dwordx2 load
        const int64_t i = (int64_t)((int)blockIdx.x - n_touch_blocks) * blockDim.x + threadIdx.x;
        if (i >= npx) return;
        const uint8_t *c = rgb + i * 3;
        uint2 rec;
        rec.x = __float_as_uint(hv_convert_depth(P, depth_raw, i));
        rec.y = hv_colour_order((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16), P.bgr);
        frame_px[i] = rec;
        return;
    }
    // ---- touch role, online form: the launch is as long as its longest wave (a single frame has only ~300 patches:
    // nothing to hide a patch's chains behind), so the samples are fanned out instead - HV_TOUCH_FAN lanes per sample,
    // lane (sample, k0) handles the sample's units k0, k0 + FAN, ...: the (usually 8) hash inserts of one sample run in
    // parallel, every wave is ONE short chain, and 8 neighbouring samples are de-duplicated by ballot.  (Measured: 15 us
    // per frame against 28 us for the patch form, whose discontinuity patches walk several chains back to back; over
    // the 32 frames of a batch the patch form is the faster one: 83 vs 128 us.) ----
    const int ns_w = (P.W + P.stride - 1) / P.stride;
    const int ns_h = (P.H + P.stride - 1) / P.stride;
    const int tid = (int)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = tid / HV_TOUCH_FAN;
    const int k0 = tid % HV_TOUCH_FAN;
    int32_t lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1}; // empty range for lanes without a valid sample
    if (s < ns_w * ns_h) {
        const int i = (s / ns_w) * P.stride;
        const int j = (s % ns_w) * P.stride;
        const float p = hv_convert_depth(P, depth_raw, (int64_t)i * P.W + j);
        if (p > 0.0f) {
            const double z = (double)p;
            const double x = ((double)j - P.cx_d) * z / P.fx_d;
            const double y = ((double)i - P.cy_d) * z / P.fy_d;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double pw = ((P.pose[r * 4 + 0] * x + P.pose[r * 4 + 1] * y) + P.pose[r * 4 + 2] * z) + P.pose[r * 4 + 3];
                lo[r] = (int32_t)floor((pw - P.sdf_trunc_d) / P.unit_length);
                hi[r] = (int32_t)floor((pw + P.sdf_trunc_d) / P.unit_length);
            }
        }
    }
    const int64_t nx = (int64_t)hi[0] - lo[0] + 1, ny = (int64_t)hi[1] - lo[1] + 1, nz = (int64_t)hi[2] - lo[2] + 1;
    const int64_t count = (nx > 0 && ny > 0 && nz > 0) ? nx * ny * nz : 0;
    const int lane = hv_lane_id();
    for (int64_t k = k0; __any(k < count); k += HV_TOUCH_FAN) {
        unsigned long long key = HV_EMPTY_KEY;
        int32_t ux = 0, uy = 0, uz = 0;
        if (k < count) {
            ux = lo[0] + (int32_t)(k / (ny * nz));
            uy = lo[1] + (int32_t)((k / nz) % ny);
            uz = lo[2] + (int32_t)(k % nz);
            if (hv_key_in_range(ux, uy, uz)) {
                key = hv_pack_key(ux, uy, uz);
                // unit-ownership sharding: another GPU fuses (and stores) this unit
                if (P.owner_world > 1 && hv_owner_of(key, P.owner_world) != P.owner_rank) key = HV_EMPTY_KEY;
            } else {
                atomicAdd(&table.counters[HV_CNT_DROPPED], 1);
            }
        }
        // wave-level de-duplication: neighbouring samples hit the same units; only one lane per distinct key goes to
        // the hash (ballot + shuffle, no memory traffic)
        bool leader = false;
        unsigned long long remaining = __ballot(key != HV_EMPTY_KEY);
        while (remaining) {
            const int first = __ffsll((long long)remaining) - 1;
            const unsigned long long fkey = __shfl(key, first);
            const unsigned long long same = __ballot(key == fkey);
            if (lane == first) leader = true;
            remaining &= ~same;
        }
        // image-tile sharding: a unit none of whose voxels can project into this GPU's tile is neither allocated nor
        // stamped here (whole-image tile: always true) - so "stamped since the last merge" == "may hold updates"
        if (leader && hv_unit_hits_tile(P, ux, uy, uz)) {
            const int32_t slot = hv_table_insert(table, key);
            if (slot >= 0) {
                // L1-bypassing pre-check: most units were already stamped by another wave this frame
                if (__hip_atomic_load(&stamp[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.frame_id) {
                    const int32_t old = atomicExch(&stamp[slot], P.frame_id);
                    if (old != P.frame_id) {
                        const int32_t at = atomicAdd(&table.counters[HV_CNT_TOUCH(parity)], 1);
                        if (at < table.max_blocks) list[at] = slot;
                    }
                }
            }
        }
    }
}

// One voxel update: UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier inner body.
__device__ __forceinline__ bool hv_tsdf_update(const HvFrameParams &P, const uint2 *__restrict__ frame_px,
                                               float pc0, float pc1, float pc2, float &tsdf, uint32_t &w,
                                               uint32_t &sr, uint32_t &sg, uint32_t &sb) {
    if (pc2 <= 0.0f) return false;
    const float u_f = pc0 * P.fx / pc2 + P.cx + 0.5f;
    const float v_f = pc1 * P.fy / pc2 + P.cy + 0.5f;
    if (!(u_f >= 0.0001f && u_f < P.safe_width_f && v_f >= 0.0001f && v_f < P.safe_height_f)) return false;
    const int u = (int)u_f;
    const int v = (int)v_f;
    // multi-GPU image-tile sharding: a voxel is fused by the GPU that owns the pixel it projects to
    if (P.tiled && (u < P.tile_u0 || u >= P.tile_u1 || v < P.tile_v0 || v >= P.tile_v1)) return false;
    const uint2 rec = frame_px[(int64_t)v * P.W + u];
    const float d = __uint_as_float(rec.x);
    if (d <= 0.0f) return false;
    const float sdf = (d - pc2) * hv_multiplier(P, u, v);
    if (!(sdf > -P.sdf_trunc_f)) return false;
    float t = sdf * P.sdf_trunc_inv_f;
    if (t > 1.0f) t = 1.0f;
    const uint32_t c = rec.y;
    const float wf = (float)w;
    tsdf = (tsdf * wf + t) / (wf + 1.0f);
    w += 1u;
    sr += c & 255u;
    sg += (c >> 8) & 255u;
    sb += (c >> 16) & 255u;
    return true;
}

// Two-phase form of the same update: hv_tsdf_eval decides whether the voxel is updated and with
// what (needs only the frame), hv_tsdf_apply folds it into the voxel state.  Splitting them lets the
// kernel fetch voxel planes only for lanes that really update something.
// a0 / b and a1 / b, both correctly rounded (bit-identical to the IEEE divisions the reference performs), sharing one
// refined reciprocal: v_rcp_f32 + one Newton step, then the quotient / residual / correction chain the compiler itself
// emits for an f32 division, minus v_div_scale / v_div_fixup, which are no-ops while the operands stay clear of the
// overflow / denormal bands.  Verified exhaustively-at-random on gfx950: 0 mismatches in 1.4e11 divisions with
// operands in 2^-60 .. 2^60 (tools/divtest.hip); callers guarantee b >= 2^-20 and |a| < 2^60.
__device__ __forceinline__ void hv_div2(float a0, float a1, float b, float &q0, float &q1) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r, 1.0f);
    r = fmaf(e, r, r);
    float q = a0 * r;
    float rem = fmaf(-b, q, a0);
    q = fmaf(rem, r, q);
    rem = fmaf(-b, q, a0);
    q0 = fmaf(rem, r, q);
    q = a1 * r;
    rem = fmaf(-b, q, a1);
    q = fmaf(rem, r, q);
    rem = fmaf(-b, q, a1);
    q1 = fmaf(rem, r, q);
}

__device__ __forceinline__ bool hv_tsdf_eval(const HvFrameParams &P, const uint2 *__restrict__ frame_px, float pc0,
                                             float pc1, float pc2, float &t, uint32_t &rgb) {
    if (pc2 <= 0.0f) return false;
    float q0, q1;
    if (pc2 >= 0x1p-20f) { // always, for voxels a camera can resolve
        hv_div2(pc0 * P.fx, pc1 * P.fy, pc2, q0, q1);
    } else {
        q0 = pc0 * P.fx / pc2;
        q1 = pc1 * P.fy / pc2;
    }
    const float u_f = q0 + P.cx + 0.5f;
    const float v_f = q1 + P.cy + 0.5f;
    if (!(u_f >= 0.0001f && u_f < P.safe_width_f && v_f >= 0.0001f && v_f < P.safe_height_f)) return false;
    const int u = (int)u_f;
    const int v = (int)v_f;
    if (P.tiled && (u < P.tile_u0 || u >= P.tile_u1 || v < P.tile_v0 || v >= P.tile_v1)) return false;
    const uint2 rec = frame_px[(int64_t)v * P.W + u];
    const float d = __uint_as_float(rec.x);
    if (d <= 0.0f) return false;
    const float sdf = (d - pc2) * hv_multiplier(P, u, v);
    if (!(sdf > -P.sdf_trunc_f)) return false;
    t = sdf * P.sdf_trunc_inv_f;
    if (t > 1.0f) t = 1.0f;
    rgb = rec.y;
    return true;
}

// a / b correctly rounded for operands clear of the overflow / denormal bands (same chain as hv_div2).
__device__ __forceinline__ float hv_div1(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r, 1.0f);
    r = fmaf(e, r, r);
    float q = a * r;
    float rem = fmaf(-b, q, a);
    q = fmaf(rem, r, q);
    rem = fmaf(-b, q, a);
    return fmaf(rem, r, q);
}

__device__ __forceinline__ void hv_tsdf_apply(bool valid, float t, uint32_t c, float &tsdf, uint32_t &w, uint32_t &sr,
                                              uint32_t &sg, uint32_t &sb) {
    if (!valid) return;
    const float wf = (float)w;
    // |tsdf * wf + t| <= 2^24 + 1 and 1 <= wf + 1 <= 2^24 for w < 2^24: inside hv_div1's verified band; beyond
    // (a voxel observed 16.7 M times) fall back to the plain division
    const float num = tsdf * wf + t;
    tsdf = (w < (1u << 24)) ? hv_div1(num, wf + 1.0f) : num / (wf + 1.0f);
    w += 1u;
    sr += c & 255u;
    sg += (c >> 8) & 255u;
    sb += (c >> 16) & 255u;
}

// ---- Predicated ("fast") forms for the multi-frame sweep -------------------------------------------------------------
// Same arithmetic, no divergent control flow: every lane runs the whole chain and a single predicate selects the
// result, so the compiler can interleave the ZH voxels of a lane (ZH gathers in flight) and does not spend VALU slots on
// re-materialising phi values.  The two rare regimes the short division chains do not cover are picked out by
// wave-uniform tests in the caller, which then runs the EXACT forms: a voxel column that comes within 1 mm of the camera
// plane (hv_div2 wants pc2 >= 2^-20 when pc2 > 0) and voxels observed more than 2^24 - 64 times (integer weights).

// sqrtf(x), correctly rounded, for x >= 2^-96 (here: x >= 1): v_sqrt_f32 (1 ulp) + the compiler's own neighbour test,
// minus the denormal pre-scaling and the zero / infinity class fix-up it has to add for arbitrary operands.
__device__ __forceinline__ float hv_sqrt_ge1(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u);
    const float su = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = fmaf(-sd, s, x);
    const float vs = fmaf(-su, s, x);
    float r = (vp <= 0.0f) ? sd : s;
    r = (vs > 0.0f) ? su : r;
    return r;
}

// EXACT = false: operands inside hv_div2's band (the caller's wave-uniform test), whole-image frames.
// EXACT = true: any operands (IEEE division where pc2 < 2^-20) and the image-tile test of the tile-sharded mode.
// MT: take the multiplier from the per-pixel table instead of computing it.
// REC12: frame_px points at 12-byte {depth, colour, multiplier} records (the fold form's batch layout) instead of 8-byte
// {depth, colour} records beside the multiplier table.
template <bool EXACT, bool MT, bool REC12 = false>
__device__ __forceinline__ bool hv_tsdf_eval_fast(const HvFrameParams &P, const uint2 *__restrict__ frame_px,
                                                  const float *__restrict__ mult, float pc0, float pc1, float pc2,
                                                  float &t, uint32_t &rgb) {
    const float a0 = pc0 * P.fx, a1 = pc1 * P.fy;
    float q0, q1;
    hv_div2(a0, a1, pc2, q0, q1); // pc2 <= 0: inf / NaN / a mirrored pixel, rejected by the pc2 > 0 term below
    if (EXACT) {
        const bool tiny = !(pc2 >= 0x1p-20f);
        const float e0 = a0 / pc2, e1 = a1 / pc2;
        q0 = tiny ? e0 : q0;
        q1 = tiny ? e1 : q1;
    }
    const float u_f = q0 + P.cx + 0.5f;
    const float v_f = q1 + P.cy + 0.5f;
    // u_f in [0.0001, safe_width) as ONE unsigned compare: for non-negative floats the bit patterns order like the
    // values, and a negative / NaN operand has a pattern above every finite positive one
    const uint32_t lo = __float_as_uint(0.0001f);
    const bool in_u = (__float_as_uint(u_f) - lo) < (__float_as_uint(P.safe_width_f) - lo);
    const bool in_v = (__float_as_uint(v_f) - lo) < (__float_as_uint(P.safe_height_f) - lo);
    bool ok = (int)(pc2 > 0.0f) & (int)in_u & (int)in_v;
    const int u = (int)u_f; // saturating conversions: garbage lanes stay defined
    const int v = (int)v_f;
    if (EXACT && P.tiled) {
        const bool in_tile_u = (int)(u >= P.tile_u0) & (int)(u < P.tile_u1);
        const bool in_tile_v = (int)(v >= P.tile_v0) & (int)(v < P.tile_v1);
        ok = (int)ok & (int)in_tile_u & (int)in_tile_v;
    }
    const uint32_t off = ok ? (uint32_t)v * (uint32_t)P.W + (uint32_t)u : 0u;
    uint2 rec;
    float m;
    if (REC12) {
        const uint32_t *r3 = (const uint32_t *)frame_px + (size_t)off * 3;
        rec = make_uint2(r3[0], r3[1]);
        m = __uint_as_float(r3[2]);
    } else {
        rec = frame_px[off];
    }
    const float d = __uint_as_float(rec.x);
    if (REC12) {
    } else if (MT) {
        m = mult[off];
    } else {
        const float xx = ((float)u - P.cx) * P.ffl_inv_x;
        const float yy = ((float)v - P.cy) * P.ffl_inv_y;
        m = hv_sqrt_ge1(xx * xx + yy * yy + 1.0f);
    }
    const float sdf = (d - pc2) * m;
    ok = (int)ok & (int)(d > 0.0f) & (int)(sdf > -P.sdf_trunc_f);
    t = fminf(sdf * P.sdf_trunc_inv_f, 1.0f); // == `if (t > 1) t = 1` for the non-NaN t of an accepted voxel
    rgb = rec.y;
    return ok;
}

// Running mean with the weight carried as a float (exact below 2^24): no int->float conversion and one add less per
// update; the caller converts back once per unit.
__device__ __forceinline__ void hv_tsdf_apply_fast(bool ok, float t, uint32_t rgb, float &tsdf, float &wf, uint32_t &sr,
                                                   uint32_t &sg, uint32_t &sb) {
    const float wf1 = wf + 1.0f;
    const float nt = hv_div1(tsdf * wf + t, wf1);
    tsdf = ok ? nt : tsdf;
    wf = ok ? wf1 : wf;
    const uint32_t c = ok ? rgb : 0u;
    sr += c & 255u;
    sg += (c >> 8) & 255u;
    sb += (c >> 16) & 255u;
}

// ZB z-slabs of one lane (ZB x 4 voxels).  Phase 1 evaluates every voxel (ZB*4 independent 8-byte
// gathers in flight, no voxel-plane traffic); phase 2 read-modify-writes only the 16-byte pieces
// that hold an updated voxel.  pc[][] is advanced by ZB z-steps.
// EVAL 0: branching evaluation (hv_tsdf_eval); 1: predicated (all ZB*4 gathers really in flight together); 2: predicated
// with the per-pixel multiplier table.
template <int ZB, int EVAL>
__device__ __forceinline__ void hv_tsdf_slabs(const HvFrameParams &P, const uint2 *__restrict__ frame_px,
                                              const float *__restrict__ mult, char *__restrict__ unit, int wordb,
                                              float (&pc)[4][3], float inc0, float inc1, float inc2) {
    float tv[ZB][4];
    uint32_t cv[ZB][4];
    unsigned mask = 0;
    if (EVAL == 0) {
#pragma unroll
        for (int zz = 0; zz < ZB; ++zz) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                tv[zz][c] = 0.f;
                cv[zz][c] = 0u;
                if (hv_tsdf_eval(P, frame_px, pc[c][0], pc[c][1], pc[c][2], tv[zz][c], cv[zz][c])) mask |= 1u << (zz * 4 + c);
                pc[c][0] += inc0;
                pc[c][1] += inc1;
                pc[c][2] += inc2;
            }
        }
    } else {
        // wave-uniform choice of the division form: columns that come within 1 mm of the camera plane on these ZB steps
        // (and tile-sharded frames) take the EXACT form
        bool near_plane = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float e = pc[c][2] + (float)ZB * inc2;
            near_plane |= (int)(fminf(pc[c][2], e) < 0x1p-10f) & (int)(fmaxf(pc[c][2], e) > -0x1p-10f);
        }
        if (P.tiled || __any(near_plane)) {
#pragma unroll
            for (int zz = 0; zz < ZB; ++zz) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (hv_tsdf_eval_fast<true, EVAL == 2>(P, frame_px, mult, pc[c][0], pc[c][1], pc[c][2], tv[zz][c], cv[zz][c]))
                        mask |= 1u << (zz * 4 + c);
                    pc[c][0] += inc0;
                    pc[c][1] += inc1;
                    pc[c][2] += inc2;
                }
            }
        } else {
#pragma unroll
            for (int zz = 0; zz < ZB; ++zz) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (hv_tsdf_eval_fast<false, EVAL == 2>(P, frame_px, mult, pc[c][0], pc[c][1], pc[c][2], tv[zz][c], cv[zz][c]))
                        mask |= 1u << (zz * 4 + c);
                    pc[c][0] += inc0;
                    pc[c][1] += inc1;
                    pc[c][2] += inc2;
                }
            }
        }
    }
    float4 vt[ZB];
    uint4 vw[ZB], vr[ZB], vg[ZB], vb[ZB];
#pragma unroll
    for (int zz = 0; zz < ZB; ++zz) {
        if ((mask >> (zz * 4)) & 15u) {
            const int q = (wordb + zz * RR) >> 2;
            vt[zz] = ((const float4 *)(unit + 0 * PLANE_BYTES))[q];
            vw[zz] = ((const uint4 *)(unit + 1 * PLANE_BYTES))[q];
            vr[zz] = ((const uint4 *)(unit + 2 * PLANE_BYTES))[q];
            vg[zz] = ((const uint4 *)(unit + 3 * PLANE_BYTES))[q];
            vb[zz] = ((const uint4 *)(unit + 4 * PLANE_BYTES))[q];
        }
    }
#pragma unroll
    for (int zz = 0; zz < ZB; ++zz) {
        const unsigned m = (mask >> (zz * 4)) & 15u;
        if (m) {
            hv_tsdf_apply(m & 1u, tv[zz][0], cv[zz][0], vt[zz].x, vw[zz].x, vr[zz].x, vg[zz].x, vb[zz].x);
            hv_tsdf_apply(m & 2u, tv[zz][1], cv[zz][1], vt[zz].y, vw[zz].y, vr[zz].y, vg[zz].y, vb[zz].y);
            hv_tsdf_apply(m & 4u, tv[zz][2], cv[zz][2], vt[zz].z, vw[zz].z, vr[zz].z, vg[zz].z, vb[zz].z);
            hv_tsdf_apply(m & 8u, tv[zz][3], cv[zz][3], vt[zz].w, vw[zz].w, vr[zz].w, vg[zz].w, vb[zz].w);
            const int q = (wordb + zz * RR) >> 2;
            ((float4 *)(unit + 0 * PLANE_BYTES))[q] = vt[zz];
            ((uint4 *)(unit + 1 * PLANE_BYTES))[q] = vw[zz];
            ((uint4 *)(unit + 2 * PLANE_BYTES))[q] = vr[zz];
            ((uint4 *)(unit + 3 * PLANE_BYTES))[q] = vg[zz];
            ((uint4 *)(unit + 4 * PLANE_BYTES))[q] = vb[zz];
        }
    }
}

// VARIANT 0: production (evaluate first - predicated, all gathers of a lane in flight -, then fetch only the pieces that
// are updated).  9: the same with the branching evaluation; 8: with the multiplier table of the sweep.  1: no voxel-plane
// traffic (math + gathers only); 2: plane traffic only (no projection / gather / update) - 1 and 2 exist for the roofline
// ablation in profiles/ (env HV_TSDF_DEBUG_VARIANT); they do not produce a valid volume.
template <int VARIANT>
__global__ __launch_bounds__(256) void k_tsdf_integrate(HvTable table, const int32_t *__restrict__ list,
                                                         int parity, char *__restrict__ pool,
                                                         const uint2 *__restrict__ frame_px, HvFrameParams P,
                                                         const float *__restrict__ mult, HvStatus *status, int32_t status_seq) {
    int n_touched = table.counters[HV_CNT_TOUCH(parity)];
    if (n_touched > table.max_blocks) n_touched = table.max_blocks;
    // the next frame's touch pass appends to the other parity's counter: zero it here (stream order); the pool occupancy
    // this frame's touch pass left goes to the host-visible status word (hv_capacity_gate reads it before the next call)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        table.counters[HV_CNT_TOUCH(parity ^ 1)] = 0;
        hv_publish_status(table, status, status_seq);
    }
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int x = lane >> 2;
    const int y0 = (lane & 3) << 2;
    const int z0 = wave * 4;
    const float inc0 = P.ext_scaled_col2[0], inc1 = P.ext_scaled_col2[1], inc2 = P.ext_scaled_col2[2];

    for (int t = blockIdx.x; t < n_touched; t += gridDim.x) {
        const int32_t slot = list[t];
        const int32_t idx = table.vals[slot];
        if (idx < 0) continue;
        int32_t ux, uy, uz;
        hv_unpack_key(table.keys[slot], ux, uy, uz);
        const double o0 = (double)ux * P.unit_length;
        const double o1 = (double)uy * P.unit_length;
        const double o2 = (double)uz * P.unit_length;
        const float p0 = (float)((double)(P.half_voxel_length_f + P.voxel_length_f * (float)x) + o0);
        const float p2 = (float)((double)P.half_voxel_length_f + o2);
        float pc[4][3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float p1 = (float)((double)(P.half_voxel_length_f + P.voxel_length_f * (float)(y0 + c)) + o1);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                pc[c][r] = ((P.ext[r * 4 + 0] * p0 + P.ext[r * 4 + 1] * p1) + P.ext[r * 4 + 2] * p2) + P.ext[r * 4 + 3];
            }
        }
        // the reference advances pt_camera by repeated float additions along z: replay them so the
        // wave that starts at z0 holds bit-identical values
        for (int s = 0; s < z0; ++s) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                pc[c][0] += inc0;
                pc[c][1] += inc1;
                pc[c][2] += inc2;
            }
        }
        char *unit = pool + (int64_t)idx * (PLANE_BYTES * HV_TSDF_PLANES);
        if (VARIANT == 0 || VARIANT >= 8) {
            // 2 z-slabs evaluated per batch: (107 VGPRs, 4 waves/SIMD) measured 7269 frames/s vs 6342 for 4
            // (180 VGPRs, 2 waves/SIMD) on the headline config.  Evaluation form: predicated without the table
            // (8.5 k frames/s; 7.9 k with the table: this kernel waits on memory, an extra gather costs more than the
            // VALU work it saves; 6.2-7.2 k branching)
            constexpr int ZB = 2;
            constexpr int EVAL = VARIANT == 9 ? 0 : (VARIANT == 8 ? 2 : 1);
#pragma unroll
            for (int zb = 0; zb < 4; zb += ZB) {
                hv_tsdf_slabs<ZB, EVAL>(P, frame_px, mult, unit, (z0 + zb) * RR + x * R + y0, pc, inc0, inc1, inc2);
            }
            continue;
        }
        // ablations: VARIANT 1 keeps the math and the gathers but no plane traffic, 2 the plane traffic only
        const int word0 = z0 * RR + x * R + y0;
        float4 vt[4];
        uint4 vw[4], vr[4], vg[4], vb[4];
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) {
            if (VARIANT == 2) {
                const int q = (word0 + zz * RR) >> 2;
                vt[zz] = ((const float4 *)(unit + 0 * PLANE_BYTES))[q];
                vw[zz] = ((const uint4 *)(unit + 1 * PLANE_BYTES))[q];
                vr[zz] = ((const uint4 *)(unit + 2 * PLANE_BYTES))[q];
                vg[zz] = ((const uint4 *)(unit + 3 * PLANE_BYTES))[q];
                vb[zz] = ((const uint4 *)(unit + 4 * PLANE_BYTES))[q];
            } else {
                vt[zz] = make_float4(0.f, 0.f, 0.f, 0.f);
                vw[zz] = vr[zz] = vg[zz] = vb[zz] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) {
            bool any = false;
            if (VARIANT == 1) {
                any |= hv_tsdf_update(P, frame_px, pc[0][0], pc[0][1], pc[0][2], vt[zz].x, vw[zz].x, vr[zz].x, vg[zz].x, vb[zz].x);
                any |= hv_tsdf_update(P, frame_px, pc[1][0], pc[1][1], pc[1][2], vt[zz].y, vw[zz].y, vr[zz].y, vg[zz].y, vb[zz].y);
                any |= hv_tsdf_update(P, frame_px, pc[2][0], pc[2][1], pc[2][2], vt[zz].z, vw[zz].z, vr[zz].z, vg[zz].z, vb[zz].z);
                any |= hv_tsdf_update(P, frame_px, pc[3][0], pc[3][1], pc[3][2], vt[zz].w, vw[zz].w, vr[zz].w, vg[zz].w, vb[zz].w);
                // keep the math alive without plane traffic
                if (any && vt[zz].x == 123.456f) ((float *)unit)[0] = vt[zz].y + (float)(vr[zz].x + vg[zz].y + vb[zz].z + vw[zz].w);
            } else {
                vw[zz].x += 1u;
                const int q = (word0 + zz * RR) >> 2;
                ((float4 *)(unit + 0 * PLANE_BYTES))[q] = vt[zz];
                ((uint4 *)(unit + 1 * PLANE_BYTES))[q] = vw[zz];
                ((uint4 *)(unit + 2 * PLANE_BYTES))[q] = vr[zz];
                ((uint4 *)(unit + 3 * PLANE_BYTES))[q] = vg[zz];
                ((uint4 *)(unit + 4 * PLANE_BYTES))[q] = vb[zz];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                pc[c][0] += inc0;
                pc[c][1] += inc1;
                pc[c][2] += inc2;
            }
        }
    }
}

// Pack role of the multi-frame paths: pixels [i0, i0 + 4) of frame f -> frame records (4 pixels per thread: one 16-byte depth load -
// 8 for uint16 -, three dwords of RGB, two / three 16-byte record stores).  Shared by k_tsdf_prep_touch_batch and k_tsdf_fused.
__device__ __forceinline__ void hv_pack_px4(const HvFrameParams &P, const int f, const int64_t i0, const void *depth_f,
                                            const uint8_t *rgb_f, uint2 *__restrict__ frame_px, const float *__restrict__ mult12,
                                            const int4 *__restrict__ pack_box) {
    const int64_t npx = (int64_t)P.H * P.W;
    if (i0 >= npx) return;
    if (pack_box != nullptr && (P.W & 3) == 0) {
        // image-coherent ownership (k_tsdf_plan_assign): this GPU's units project into pack_box[f] = {u0, v0, u1, v1} of frame
        // f and nowhere else, so only those pixels' records are ever gathered (the box is conservative and already padded)
        const int4 bb = pack_box[f];
        const int u = (int)(i0 % P.W), v = (int)(i0 / P.W);
        if (u + 3 < bb.x || u >= bb.z || v < bb.y || v >= bb.w) return;
    }
    if (P.tiled && (P.W & 3) == 0) {
        // tile-sharded volume: only voxels that project into this GPU's tile gather a record (the sweep's image-range test
        // uses the tile's bounds), so only the tile's columns and rows are packed (+ 4 pixels: a garbage lane may read
        // outside, its value is never used)
        const int u = (int)(i0 % P.W), v = (int)(i0 / P.W);
        if (u + 3 < P.tile_u0 - 4 || u >= P.tile_u1 + 4 || v < P.tile_v0 - 4 || v >= P.tile_v1 + 4) return;
    }
    uint2 *dst = frame_px + (int64_t)f * npx + i0;
    // mult12 != nullptr: 12-byte records {depth, colour, multiplier} (the fold form of the sweep gathers a voxel's pixel
    // with ONE load; the multiplier comes from the per-pixel table, which is built before this launch)
    uint32_t *dst12 = (uint32_t *)frame_px + ((int64_t)f * npx + i0) * 3;
    if (i0 + 4 <= npx && (npx & 3) == 0) {
        const uint32_t *c4 = (const uint32_t *)(rgb_f + i0 * 3); // i0 % 4 == 0 -> 12-byte multiple: dword aligned
        const uint32_t w0 = c4[0], w1 = c4[1], w2 = c4[2];
        // byte 3 of a batch record's colour word is 1: the fold form of the sweep adds accepted records' words into packed
        // accumulators and that byte counts the observations (every other consumer masks the colour bytes out)
        const uint32_t col[4] = {hv_colour_order(w0 & 0xffffffu, P.bgr) | HV_REC_ONE,
                                 hv_colour_order((w0 >> 24) | ((w1 & 0xffffu) << 8), P.bgr) | HV_REC_ONE,
                                 hv_colour_order((w1 >> 16) | ((w2 & 0xffu) << 16), P.bgr) | HV_REC_ONE,
                                 hv_colour_order(w2 >> 8, P.bgr) | HV_REC_ONE};
        float d[4];
        if (P.depth_is_u16) {
            const uint2 raw = *(const uint2 *)((const uint16_t *)depth_f + i0);
            d[0] = (float)(raw.x & 0xffffu); d[1] = (float)(raw.x >> 16);
            d[2] = (float)(raw.y & 0xffffu); d[3] = (float)(raw.y >> 16);
        } else {
            const float4 raw = *(const float4 *)((const float *)depth_f + i0);
            d[0] = raw.x; d[1] = raw.y; d[2] = raw.z; d[3] = raw.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { // hv_convert_depth
            d[k] = d[k] / P.depth_scale_f;
            if ((double)d[k] >= P.depth_trunc_d) d[k] = 0.0f;
        }
        if (mult12 != nullptr) {
            const float4 m4 = *(const float4 *)(mult12 + i0);
            ((uint4 *)dst12)[0] = make_uint4(__float_as_uint(d[0]), col[0], __float_as_uint(m4.x), __float_as_uint(d[1]));
            ((uint4 *)dst12)[1] = make_uint4(col[1], __float_as_uint(m4.y), __float_as_uint(d[2]), col[2]);
            ((uint4 *)dst12)[2] = make_uint4(__float_as_uint(m4.z), __float_as_uint(d[3]), col[3], __float_as_uint(m4.w));
        } else {
            ((uint4 *)dst)[0] = make_uint4(__float_as_uint(d[0]), col[0], __float_as_uint(d[1]), col[1]);
            ((uint4 *)dst)[1] = make_uint4(__float_as_uint(d[2]), col[2], __float_as_uint(d[3]), col[3]);
        }
    } else {
        for (int64_t i = i0; i < npx && i < i0 + 4; ++i) {
            const uint8_t *c = rgb_f + i * 3;
            uint2 rec;
            rec.x = __float_as_uint(hv_convert_depth(P, depth_f, i));
            rec.y = hv_colour_order((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16), P.bgr) | HV_REC_ONE;
            if (mult12 != nullptr) {
                uint32_t *r3 = (uint32_t *)frame_px + ((int64_t)f * npx + i) * 3;
                r3[0] = rec.x;
                r3[1] = rec.y;
                r3[2] = __float_as_uint(mult12[i]);
            } else {
                frame_px[(int64_t)f * npx + i] = rec;
            }
        }
    }
}

// Touch role of the multi-frame paths: one wave = one 8x8 patch of frame f's depth samples; every distinct unit the patch opens gets
// bit f of its frame mask and - at its first touch in the batch - its place in the batch's union list.  Shared by
// k_tsdf_prep_touch_batch and k_tsdf_fused.
__device__ __forceinline__ void hv_touch_batch_patch(const HvTable &table, const HvFrameParams &P, const void *depth_f, const int patch,
                                                     HvTouchScratch &scratch, const int f, unsigned long long *__restrict__ frame_mask,
                                                     int32_t *__restrict__ stamp, int32_t *__restrict__ list, const int batch_stamp,
                                                     const int parity) {
    const unsigned long long fbit = 1ull << f;
    hv_touch_patch(table, P, depth_f, patch, scratch,
                   [&](unsigned long long key, int32_t ux, int32_t uy, int32_t uz) {
                       // image-tile sharding: units that cannot project into this GPU's tile are not allocated here
                       if (!hv_unit_hits_tile(P, ux, uy, uz)) return;
                       const int32_t slot = hv_table_insert(table, key);
                       if (slot < 0) return;
                       // both L1-bypassing pre-checks in flight together
                       const unsigned long long seen = __hip_atomic_load(&frame_mask[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                       const int32_t stamped = __hip_atomic_load(&stamp[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                       // frame bit (skip the atomic when another wave of this frame already set it)
                       if (!(seen & fbit)) atomicOr(&frame_mask[slot], fbit);
                       if (stamped != batch_stamp) {
                           if (list == nullptr) {
                               // the union list is built afterwards from the allocated units (k_tsdf_batch_list): an append here
                               // is one returning atomic on a single counter per first touch - thousands per batch, serialised
                               __hip_atomic_store(&stamp[slot], batch_stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                           } else {
                               const int32_t old = atomicExch(&stamp[slot], batch_stamp);
                               if (old != batch_stamp) {
                                   const int32_t at = atomicAdd(&table.counters[HV_CNT_TOUCH(parity)], 1);
                                   // (an L2-level store: the launch's last touch workgroup may read the list back - hv_list_by_work_tail)
                                   if (at < table.max_blocks) __hip_atomic_store(&list[at], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                               }
                           }
                       }
                   });
}

// ================================================================================================
// Multi-frame sweep (hv_tsdf_integrate_batch; the rebuild()/offline-replay use case,
// volumetric_integrator_base.py:1242-1318).  B <= 64 posed frames are resident in HBM:
//   k_tsdf_prep_touch_batch  one launch for all B frames: packs every frame and ORs bit f into the 64-bit
//                            frame mask of each unit frame f touches (one wave per 8x8 sample patch,
//                            hv_touch_patch); the first toucher of a unit in the batch appends it to the
//                            union list
//   k_tsdf_integrate_batch_col  per union unit: a lane's voxels are loaded ONCE, then for every frame bit in
//                            ascending (= chronological) order the voxels are evaluated and updated in
//                            registers, then stored once.
// Identical results to B successive hv_tsdf_integrate calls: a unit is updated by frame f iff frame
// f touched it, and its frames are applied in order.  Plane traffic per frame drops by ~B x (the
// union of 32 consecutive frames' units is ~1.6x one frame's); what remains is the per-voxel math
// and the 8-byte frame gathers.
// ================================================================================================
// Longest-processing-time-first order for the sweep, WITHOUT a launch of its own (round 5): the last touch workgroup of the touch + pack
// launch to finish counting-sorts the batch's union list by decreasing work (set bits of the unit's frame mask) while the launch's pack
// workgroups are still streaming.  As a kernel of its own (k_tsdf_list_by_work) the sort sat in the dependency chain between two
// sweeps and cost the step what the better order saved the sweep (profiles/r05/pipeline_experiments.md).  `lds`: >= 16 384 bytes of
// the workgroup's LDS (the touch role's scratch, free by now) holds one byte of work per list entry; longer lists are copied as they
// are.  List entries and masks were written with L2-level atomics by every workgroup: read back the same way.
__device__ __forceinline__ void hv_list_by_work_tail(const HvTable &table, const int32_t *list, const unsigned long long *frame_mask,
                                                     int32_t *__restrict__ sorted, const int parity, uint8_t *lds) {
    __shared__ int32_t s_bin[65];
    constexpr int CAP = 16384;
    int n = __hip_atomic_load(&table.counters[HV_CNT_TOUCH(parity)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n > table.max_blocks) n = table.max_blocks;
    const int nt = (int)blockDim.x;
    if (n > CAP) {
        for (int i = threadIdx.x; i < n; i += nt) sorted[i] = __hip_atomic_load(&list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    for (int i = threadIdx.x; i < 65; i += nt) s_bin[i] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 8 * nt) { // eight list entries, then their eight masks, in flight per thread
        int32_t slot[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * nt + (int)threadIdx.x;
            slot[k] = i < n ? __hip_atomic_load(&list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        }
        unsigned long long m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = slot[k] >= 0 ? __hip_atomic_load(&frame_mask[slot[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * nt + (int)threadIdx.x;
            if (i < n) {
                const int b = 64 - __popcll(m[k]); // bin 0 = all 64 frames: the longest tasks first
                lds[i] = (uint8_t)b;
                atomicAdd(&s_bin[b], 1);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int at = 0;
        for (int b = 0; b < 65; ++b) {
            const int c = s_bin[b];
            s_bin[b] = at;
            at += c;
        }
    }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 8 * nt) {
        int32_t slot[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * nt + (int)threadIdx.x;
            slot[k] = i < n ? __hip_atomic_load(&list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * nt + (int)threadIdx.x;
            if (i < n) sorted[atomicAdd(&s_bin[lds[i]], 1)] = slot[k];
        }
    }
}

__global__ __launch_bounds__(256) void k_tsdf_prep_touch_batch(HvTable table, int32_t *__restrict__ stamp,
                                                                unsigned long long *__restrict__ frame_mask,
                                                                int32_t *__restrict__ list, int batch_stamp,
                                                                const char *__restrict__ depth_raw, int64_t depth_stride,
                                                                const uint8_t *__restrict__ rgb,
                                                                uint2 *__restrict__ frame_px,
                                                                const HvFrameParams *__restrict__ Ps, int n_prep_blocks,
                                                                int n_touch_blocks, int n_frames, int parity,
                                                                const float *__restrict__ mult12,
                                                                const int4 *__restrict__ pack_box, uint32_t *__restrict__ plan_hist,
                                                                HvStatus *status, int32_t status_seq, int32_t *__restrict__ sorted,
                                                                uint32_t *__restrict__ touch_ticket) {
    if (plan_hist != nullptr && blockIdx.x == 0) {
        // image-coherent ownership: this is the last launch of the batch's plan chain - k_tsdf_plan_assign has read the histogram
        // (clean it for this scratch set's next batch) and made its claims (publish the pool's occupancy: hv_capacity_gate)
        for (int i = threadIdx.x; i < HV_PLAN_BINS; i += blockDim.x) plan_hist[i] = 0u;
        if (threadIdx.x == 0) hv_publish_status(table, status, status_seq);
    }
    // block order: the touch blocks of ALL frames first, then the pack blocks.  A touch wave is one chain of dependent
    // memory round trips (depth -> hash probe -> mask / stamp -> atomics; a patch on a long depth discontinuity walks
    // several such chains), the pack blocks are pure streaming: dispatched last they fill the machine while the touch
    // chains drain, instead of the launch ending on the chains of the last frame.
    int f, bx;
    const bool touch_role = (int)blockIdx.x < n_touch_blocks * n_frames;
    if (touch_role) {
        f = (int)blockIdx.x / n_touch_blocks;
        bx = (int)blockIdx.x % n_touch_blocks;
    } else {
        const int b = (int)blockIdx.x - n_touch_blocks * n_frames;
        f = b / n_prep_blocks;
        bx = b % n_prep_blocks;
    }
    const HvFrameParams &P = Ps[f];
    const int64_t npx = (int64_t)P.H * P.W;
    const void *depth_f = depth_raw + (int64_t)f * depth_stride;
    const uint8_t *rgb_f = rgb + (int64_t)f * npx * 3;
    if (!touch_role) {
        // pack role: 4 pixels per thread - every wave access is a contiguous burst (prep blocks cover 1024 pixels each)
        hv_pack_px4(P, f, ((int64_t)bx * blockDim.x + threadIdx.x) * 4, depth_f, rgb_f, frame_px, mult12, pack_box);
        return;
    }
    // ---- touch role: one wave per 8x8 sample patch (hv_touch_patch) ----
    __shared__ HvTouchScratch scratch[4];
    const int patch = bx * (int)(blockDim.x / HV_WAVE) + (int)(threadIdx.x / HV_WAVE); // (blocks of 256 or - HV_TSDF_AUX_W64 - 64 threads)
    if (patch < hv_touch_patches(P))
        hv_touch_batch_patch(table, P, depth_f, patch, scratch[threadIdx.x / HV_WAVE], f, frame_mask, stamp, list, batch_stamp, parity);
    if (sorted == nullptr) return;
    // the launch's LAST touch workgroup sorts the union list by decreasing work (hv_list_by_work_tail); every workgroup's list entries
    // and mask bits are L2-level atomics, acknowledged (vmcnt = 0) before its ticket is taken
    __shared__ uint32_t s_last;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(touch_ticket, 1u) == (uint32_t)(n_touch_blocks * n_frames) - 1u ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) *touch_ticket = 0u;
    static_assert(sizeof(scratch) >= 16384, "the sort keeps one byte of work per list entry in the touch role's scratch");
    hv_list_by_work_tail(table, list, frame_mask, sorted, parity, (uint8_t *)scratch);
}

// Union list of a batch from the allocated units: unit b belongs to it iff its slot carries the batch's stamp.  One thread per
// unit, one atomic per wave (ballot + prefix), list in pool order.
__global__ __launch_bounds__(256) void k_tsdf_batch_list(HvTable table, const int32_t *__restrict__ stamp, int batch_stamp,
                                                          int32_t *__restrict__ list) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t n_blocks = table.counters[HV_CNT_BLOCKS];
    if (n_blocks > table.max_blocks) n_blocks = table.max_blocks;
    int32_t slot = -1;
    if (b < n_blocks) {
        slot = hv_table_find(table, table.block_keys[b]);
        if (slot >= 0 && stamp[slot] != batch_stamp) slot = -1;
    }
    const int32_t at = hv_wave_append(&table.counters[HV_CNT_TOUCH0], slot >= 0);
    if (slot >= 0) list[at] = slot;
}

// ================================================================================================
// Image-coherent unit ownership for N GPUs, decided once per batch with NO communication (hv_tsdf_set_sharding(COHERENT)).
//
// Under hash ownership (hv_owner_of) a GPU's units are scattered over the image: every GPU packs every pixel of every frame
// (187 MB per 32-frame batch at 640x480) and gathers from all of them, so the per-batch replicated work does not shrink with
// N and bounds the scaling curve (profiles/r03/rank8_timeline.txt).  Here every GPU - they all see every frame - first
// enumerates the batch's units into a small replicated BATCH TABLE (key -> 64-bit frame mask; no pool claims), then all of
// them evaluate the same deterministic plan on it:
//   work(unit) = popcount(frame mask)            (voxel visits ~ frames that see the unit)
//   bin(unit)  = cell of a PLAN_NU x PLAN_NV grid over the image of the batch's MIDDLE frame that the unit's centre projects
//                into, numbered column-major (units not in front of that camera: a hash of the key)
//   owner(bin) = rank whose share [r, r + 1) * total / N of the prefix sum of work over the bins holds the bin's midpoint
// -> vertical strips of the reference image with ragged edges, equal WORK per GPU, and each GPU's units stay together in
// every frame of the batch.  A GPU then claims only its own units in its hash (frame masks, union list: what the sweep reads),
// accumulates the pixel box its units can project into per frame, and packs the records of THOSE pixels only.
// A (unit, frame) pair is fused by exactly one GPU; ownership moves with the camera from batch to batch, so a unit's additive
// numerators may live on several GPUs - exactly the state hv_merge_halo_* / gather_to_root() already consolidate (tile form).
// Integer atomics only: every GPU computes bit-identical histograms, hence identical plans.
// ================================================================================================

struct HvPlan { // one per scratch set
    unsigned long long *bt_keys;  // [cap] batch table: packed unit key or HV_EMPTY_KEY (self-cleaning: k_tsdf_plan_assign empties it)
    unsigned long long *bt_masks; // [cap] frames of the batch that touch the unit
    uint32_t *hist;               // [HV_PLAN_BINS] work per bin
    int4 *box;                    // [64] per frame {u0, v0, u1, v1}: pixels this GPU's units can project into
    uint32_t cap_mask;            // cap - 1
};

__device__ __forceinline__ int hv_plan_bin(unsigned long long key, const HvFrameParams &Pm) {
    int32_t ux, uy, uz;
    hv_unpack_key(key, ux, uy, uz);
    const float h = 0.5f * (float)Pm.unit_length;
    const float p0 = (float)((double)ux * Pm.unit_length) + h, p1 = (float)((double)uy * Pm.unit_length) + h,
                p2 = (float)((double)uz * Pm.unit_length) + h;
    const float pc0 = ((Pm.ext[0] * p0 + Pm.ext[1] * p1) + Pm.ext[2] * p2) + Pm.ext[3];
    const float pc1 = ((Pm.ext[4] * p0 + Pm.ext[5] * p1) + Pm.ext[6] * p2) + Pm.ext[7];
    const float pc2 = ((Pm.ext[8] * p0 + Pm.ext[9] * p1) + Pm.ext[10] * p2) + Pm.ext[11];
    if (!(pc2 > 0.1f)) return (int)(hv_slot_hash(key ^ 0x5bd1e995ull) % (uint32_t)HV_PLAN_BINS);
    const float u = pc0 * Pm.fx / pc2 + Pm.cx, v = pc1 * Pm.fy / pc2 + Pm.cy;
    int ub = (int)floorf(u * ((float)HV_PLAN_NU / (float)Pm.W)), vb = (int)floorf(v * ((float)HV_PLAN_NV / (float)Pm.H));
    ub = min(max(ub, 0), HV_PLAN_NU - 1);
    vb = min(max(vb, 0), HV_PLAN_NV - 1);
    return ub * HV_PLAN_NV + vb;
}

// Touch role of ALL frames into the batch table (no ownership test: the frame constants carry owner_world = 1, no pool claim).
// The frame bit an atomicOr newly sets is a new (unit, frame) pair - the plan's measure of work; it is counted in the
// workgroup's LDS histogram, flushed once per workgroup.  (Measured alone on the GPU, 32 frames of 640x480: patch enumeration
// without any global access 33 us, with the batch table 46 us; the same pairs counted with global atomics on the 1024 bins:
// 116 us; a wave taking its patch through 2 / 4 / 8 consecutive frames and merging them in an LDS table first: 68 / 93 / 180 us -
// the launch is bound by the patch arithmetic and its occupancy, not by the table's atomics.)
__global__ __launch_bounds__(256) void k_tsdf_touch_plan(HvTable table, HvPlan plan, const char *__restrict__ depth_raw,
                                                          int64_t depth_stride, const HvFrameParams *__restrict__ Ps,
                                                          int n_touch_blocks, int n_frames, int parity) {
    __shared__ HvTouchScratch scratch[4];
    __shared__ uint32_t s_hist[HV_PLAN_BINS];
    const int wave = (int)(threadIdx.x / HV_WAVE);
    for (int i = threadIdx.x; i < HV_PLAN_BINS; i += blockDim.x) s_hist[i] = 0u;
    if (blockIdx.x == 0) {
        // this scratch set's last batch is swept (the chain waited for it): its union list restarts, its boxes start empty
        if (threadIdx.x < 64) plan.box[threadIdx.x] = make_int4(INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN);
        if (threadIdx.x == 0) table.counters[HV_CNT_TOUCH(parity)] = 0;
    }
    __syncthreads();
    const int f = (int)blockIdx.x / n_touch_blocks, bx = (int)blockIdx.x % n_touch_blocks;
    const int patch = bx * 4 + wave;
    const HvFrameParams &Pm = Ps[n_frames / 2];
    if (patch < hv_touch_patches(Ps[f])) {
        const unsigned long long fbit = 1ull << f;
        hv_touch_patch(table, Ps[f], depth_raw + (int64_t)f * depth_stride, patch, scratch[wave],
                       [&](unsigned long long key, int32_t, int32_t, int32_t) {
                           uint32_t s = hv_slot_hash(key) & plan.cap_mask;
                           for (uint32_t probe = 0; probe <= plan.cap_mask; ++probe) {
                               unsigned long long k = __hip_atomic_load(&plan.bt_keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                               if (k == HV_EMPTY_KEY) {
                                   k = atomicCAS(&plan.bt_keys[s], HV_EMPTY_KEY, key);
                                   if (k == HV_EMPTY_KEY) k = key;
                               }
                               if (k == key) {
                                   const unsigned long long seen = __hip_atomic_load(&plan.bt_masks[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                   if (!(seen & fbit) && !(atomicOr(&plan.bt_masks[s], fbit) & fbit)) atomicAdd(&s_hist[hv_plan_bin(key, Pm)], 1u);
                                   return;
                               }
                               s = (s + 1) & plan.cap_mask;
                           }
                           atomicAdd(&table.counters[HV_CNT_OVERFLOW], 1);
                       });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HV_PLAN_BINS; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&plan.hist[i], s_hist[i]);
}

// Every workgroup derives the bin -> owner table from the histogram (1024 bins: an LDS scan), then handles its slots of the
// batch table: empties them, and for the units this GPU owns claims the unit in the volume's hash, publishes its frame mask,
// stamps it, appends it to the union list - and adds the pixels the unit can project into to the box of every frame that
// touches it (the workgroup's owned units are few: a wave takes one of them at a time, lane = frame).
__global__ __launch_bounds__(256) void k_tsdf_plan_assign(HvTable table, HvPlan plan, int32_t *__restrict__ stamp,
                                                           unsigned long long *__restrict__ frame_mask, int32_t *__restrict__ list,
                                                           int batch_stamp, const HvFrameParams *__restrict__ Ps, int n_frames,
                                                           int parity, int rank, int world) {
    __shared__ uint32_t s_owner[HV_PLAN_BINS];
    __shared__ uint32_t s_part[256];
    __shared__ int s_box[64][4];
    __shared__ unsigned long long s_mine_key[256], s_mine_mask[256];
    __shared__ int s_n_mine;
    static_assert(HV_PLAN_BINS == 4 * 256, "four bins per thread");
    const int tid = threadIdx.x;
    // this workgroup's slots first: most workgroups find nothing in theirs and leave before the scan
    const uint32_t s = blockIdx.x * blockDim.x + tid;
    unsigned long long key = HV_EMPTY_KEY, mask = 0ull;
    if (s <= plan.cap_mask) {
        key = plan.bt_keys[s];
        if (key != HV_EMPTY_KEY) {
            mask = plan.bt_masks[s];
            plan.bt_keys[s] = HV_EMPTY_KEY; // the table is empty again when this launch ends
            plan.bt_masks[s] = 0ull;
        }
    }
    if (!__syncthreads_or(key != HV_EMPTY_KEY)) return;
    uint32_t w[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w[k] = plan.hist[tid * 4 + k];
        sum += w[k];
    }
    s_part[tid] = sum;
    if (tid < 64) {
        s_box[tid][0] = INT32_MAX; s_box[tid][1] = INT32_MAX; s_box[tid][2] = INT32_MIN; s_box[tid][3] = INT32_MIN;
    }
    if (tid == 0) s_n_mine = 0;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) { // inclusive scan of the per-thread sums
        const uint32_t add = tid >= o ? s_part[tid - o] : 0u;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    const uint32_t total = s_part[255];
    uint32_t run = s_part[tid] - sum; // exclusive
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // owner of the bin: where the midpoint of its work interval falls (2 * midpoint against 2 * total * r / world, in 64 bits)
        const unsigned long long mid2 = 2ull * run + w[k];
        const uint32_t owner = total == 0u ? 0u : (uint32_t)min((unsigned long long)(world - 1), mid2 * (unsigned long long)world / (2ull * total));
        s_owner[tid * 4 + k] = owner;
        run += w[k];
    }
    __syncthreads();
    const bool mine = key != HV_EMPTY_KEY && mask != 0ull && (int)s_owner[hv_plan_bin(key, Ps[n_frames / 2])] == rank;
    int32_t slot = -1;
    if (mine) {
        slot = hv_table_insert(table, key);
        if (slot >= 0) {
            frame_mask[slot] = mask; // (plain store: the sweep only reads the masks of the units its list names)
            stamp[slot] = batch_stamp;
            const int at = atomicAdd(&s_n_mine, 1);
            s_mine_key[at] = key;
            s_mine_mask[at] = mask;
        }
    }
    const int32_t at = hv_wave_append(&table.counters[HV_CNT_TOUCH(parity)], slot >= 0);
    if (slot >= 0 && at < table.max_blocks) list[at] = slot;
    __syncthreads();
    const int n_mine = s_n_mine, wave = tid / HV_WAVE, f = hv_lane_id();
    for (int m = wave; m < n_mine; m += 4) {
        if (f >= n_frames || !((s_mine_mask[m] >> f) & 1ull)) continue;
        const HvFrameParams &P = Ps[f];
        int32_t ux, uy, uz;
        hv_unpack_key(s_mine_key[m], ux, uy, uz);
        const float L = (float)P.unit_length;
        const float o0 = (float)((double)ux * P.unit_length), o1 = (float)((double)uy * P.unit_length), o2 = (float)((double)uz * P.unit_length);
        float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
        bool behind = false;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float p0 = o0 + ((c & 1) ? L : 0.f), p1 = o1 + ((c & 2) ? L : 0.f), p2 = o2 + ((c & 4) ? L : 0.f);
            const float pc0 = ((P.ext[0] * p0 + P.ext[1] * p1) + P.ext[2] * p2) + P.ext[3];
            const float pc1 = ((P.ext[4] * p0 + P.ext[5] * p1) + P.ext[6] * p2) + P.ext[7];
            const float pc2 = ((P.ext[8] * p0 + P.ext[9] * p1) + P.ext[10] * p2) + P.ext[11];
            if (!(pc2 > 0.02f)) behind = true;
            const float u = pc0 * P.fx / pc2 + P.cx, v = pc1 * P.fy / pc2 + P.cy;
            umin = fminf(umin, u); umax = fmaxf(umax, u);
            vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
        }
        int b0 = 0, b1 = 0, b2 = P.W, b3 = P.H; // a unit that reaches behind the camera plane may project anywhere
        if (!behind) {
            // voxel centres lie inside the unit's box, their projections inside the box of the corner projections; +/- 3 pixels
            // cover the float rounding of either side and the + 0.5 of the reference's pixel choice
            b0 = (int)fmaxf(fminf(floorf(umin) - 3.f, (float)P.W), 0.f);
            b1 = (int)fmaxf(fminf(floorf(vmin) - 3.f, (float)P.H), 0.f);
            b2 = (int)fmaxf(fminf(ceilf(umax) + 4.f, (float)P.W), 0.f);
            b3 = (int)fmaxf(fminf(ceilf(vmax) + 4.f, (float)P.H), 0.f);
        }
        if (b2 > b0 && b3 > b1) {
            atomicMin(&s_box[f][0], b0); atomicMin(&s_box[f][1], b1);
            atomicMax(&s_box[f][2], b2); atomicMax(&s_box[f][3], b3);
        }
    }
    __syncthreads();
    if (tid < 64 && s_box[tid][2] > s_box[tid][0]) {
        atomicMin(&plan.box[tid].x, s_box[tid][0]); atomicMin(&plan.box[tid].y, s_box[tid][1]);
        atomicMax(&plan.box[tid].z, s_box[tid][2]); atomicMax(&plan.box[tid].w, s_box[tid][3]);
    }
}

// Column mapping: lane -> one (x, y) column of the unit, wave -> 64 consecutive columns (4 x-values: word index
// z*256 + cg*64 + lane, so every plane access of a wave is one contiguous 256-byte dword burst) and ZH consecutive z.
// Compared with the online kernel's slab mapping (lane -> 4 y's of one x) a lane projects ONE column per frame instead of
// four and replays at most 16 - ZH steps of the reference's z-walk: fewer VALU instructions per voxel (the sweep is
// VALU-bound).  A unit = 4 column groups x (16 / ZH) z ranges = 64 / ZH wave tasks, SPLIT workgroups per unit: a unit
// seen by all frames of the batch is otherwise one long work item, which bounds the sweep when a GPU owns few units
// (multi-GPU ownership sharding).  Work items are independent: the unit's frame mask is only read here and cleared
// afterwards by k_tsdf_batch_finish.
// 5 waves / SIMD for the production configuration (96 VGPRs, 20 B of scratch in the rare-regime code): 28.5 k vs 27.7 k
// frames/s at 4 (100 VGPRs).
template <int ZH, int SPLIT, bool MT, int WPE = (ZH == 4 && MT) ? 5 : 1>
__global__ __launch_bounds__(64 * (64 / ZH) / SPLIT, WPE) void k_tsdf_integrate_batch_col(
    HvTable table, const int32_t *__restrict__ list, const unsigned long long *__restrict__ frame_mask,
    char *__restrict__ pool, const uint2 *__restrict__ frame_px, const HvFrameParams *__restrict__ Ps, int parity,
    int general, const float *__restrict__ mult) {
    constexpr int G = ZH < 4 ? ZH : 4; // voxels of a lane evaluated together (G gathers in flight), then folded
    constexpr int TASKS = 64 / ZH;          // wave tasks per unit
    constexpr int WAVES = TASKS / SPLIT;    // waves per workgroup
    int n_units = table.counters[HV_CNT_TOUCH(parity)];
    if (n_units > table.max_blocks) n_units = table.max_blocks;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: keeps the z-walk replay loop scalar
    const int lane = threadIdx.x & 63;
    for (int item = blockIdx.x; item < n_units * SPLIT; item += gridDim.x) {
        const int t = item / SPLIT;
        const int task = (item % SPLIT) * WAVES + wave;
        const int cg = task & 3;            // column group: x in [4 cg, 4 cg + 4)
        const int z0 = (task >> 2) * ZH;
        const int x = cg * 4 + (lane >> 4);
        const int y = lane & 15;
        const int32_t slot = list[t];
        const int32_t idx = table.vals[slot];
        unsigned long long mask = frame_mask[slot];
        if (idx < 0 || mask == 0ull) continue;
        int32_t ux, uy, uz;
        hv_unpack_key(table.keys[slot], ux, uy, uz);
        char *unit = pool + (int64_t)idx * (PLANE_BYTES * HV_TSDF_PLANES);
        const int wordb = z0 * RR + cg * 64 + lane;
        float vt[ZH];
        uint32_t vw[ZH], vr[ZH], vg[ZH], vb[ZH];
#pragma unroll
        for (int zz = 0; zz < ZH; ++zz) {
            const int q = wordb + zz * RR;
            vt[zz] = ((const float *)(unit + 0 * PLANE_BYTES))[q];
            vw[zz] = ((const uint32_t *)(unit + 1 * PLANE_BYTES))[q];
            vr[zz] = ((const uint32_t *)(unit + 2 * PLANE_BYTES))[q];
            vg[zz] = ((const uint32_t *)(unit + 3 * PLANE_BYTES))[q];
            vb[zz] = ((const uint32_t *)(unit + 4 * PLANE_BYTES))[q];
        }
        // the voxel centre of (x, y, z = 0) does not depend on the frame (voxel / unit length are the volume's)
        float p0, p1, p2;
        bool tiled;
        {
            const HvFrameParams &P = Ps[__ffsll((long long)mask) - 1];
            const double o0 = (double)ux * P.unit_length;
            const double o1 = (double)uy * P.unit_length;
            const double o2 = (double)uz * P.unit_length;
            p0 = (float)((double)(P.half_voxel_length_f + P.voxel_length_f * (float)x) + o0);
            p1 = (float)((double)(P.half_voxel_length_f + P.voxel_length_f * (float)y) + o1);
            p2 = (float)((double)P.half_voxel_length_f + o2);
            tiled = P.tiled != 0; // image-tile sharding is a property of the volume: the same for every frame
        }
        bool heavy = false; // a voxel near 2^24 observations: float weights would stop being exact
#pragma unroll
        for (int zz = 0; zz < ZH; ++zz) heavy |= vw[zz] >= (1u << 24) - 64u;
        unsigned dirty = 0;
        if (general || __any(heavy)) {
            while (mask) {
                const int f = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const HvFrameParams &P = Ps[f];
                const uint2 *px = frame_px + (int64_t)f * P.H * P.W;
                const float inc0 = P.ext_scaled_col2[0], inc1 = P.ext_scaled_col2[1], inc2 = P.ext_scaled_col2[2];
                float pc0 = ((P.ext[0] * p0 + P.ext[1] * p1) + P.ext[2] * p2) + P.ext[3];
                float pc1 = ((P.ext[4] * p0 + P.ext[5] * p1) + P.ext[6] * p2) + P.ext[7];
                float pc2 = ((P.ext[8] * p0 + P.ext[9] * p1) + P.ext[10] * p2) + P.ext[11];
                for (int s = 0; s < z0; ++s) { // the reference's repeated float additions along z, replayed
                    pc0 += inc0;
                    pc1 += inc1;
                    pc2 += inc2;
                }
#pragma unroll
                for (int zz = 0; zz < ZH; ++zz) {
                    float tv;
                    uint32_t cv;
                    const bool ok = hv_tsdf_eval_fast<true, MT>(P, px, mult, pc0, pc1, pc2, tv, cv);
                    pc0 += inc0;
                    pc1 += inc1;
                    pc2 += inc2;
                    hv_tsdf_apply(ok, tv, cv, vt[zz], vw[zz], vr[zz], vg[zz], vb[zz]);
                    if (ok) dirty |= 1u << zz;
                }
            }
        } else {
            float wf[ZH];
#pragma unroll
            for (int zz = 0; zz < ZH; ++zz) wf[zz] = (float)vw[zz];
            while (mask) {
                const int f = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const HvFrameParams &P = Ps[f];
                const uint2 *px = frame_px + (int64_t)f * P.H * P.W;
                const float inc0 = P.ext_scaled_col2[0], inc1 = P.ext_scaled_col2[1], inc2 = P.ext_scaled_col2[2];
                float pc0 = ((P.ext[0] * p0 + P.ext[1] * p1) + P.ext[2] * p2) + P.ext[3];
                float pc1 = ((P.ext[4] * p0 + P.ext[5] * p1) + P.ext[6] * p2) + P.ext[7];
                float pc2 = ((P.ext[8] * p0 + P.ext[9] * p1) + P.ext[10] * p2) + P.ext[11];
                for (int s = 0; s < z0; ++s) {
                    pc0 += inc0;
                    pc1 += inc1;
                    pc2 += inc2;
                }
                // does this column come within 1 mm of the camera plane on its ZH steps?  (pc2 is monotone along z up to
                // rounding that is orders of magnitude below the margin)
                const float pc2_end = pc2 + (float)ZH * inc2;
                const bool near_plane = (int)(fminf(pc2, pc2_end) < 0x1p-10f) & (int)(fmaxf(pc2, pc2_end) > -0x1p-10f);
                if (tiled || __any(near_plane)) {
#pragma unroll
                    for (int zz = 0; zz < ZH; ++zz) {
                        float tv;
                        uint32_t cv;
                        const bool ok = hv_tsdf_eval_fast<true, MT>(P, px, mult, pc0, pc1, pc2, tv, cv);
                        pc0 += inc0;
                        pc1 += inc1;
                        pc2 += inc2;
                        hv_tsdf_apply_fast(ok, tv, cv, vt[zz], wf[zz], vr[zz], vg[zz], vb[zz]);
                    }
                } else {
                    // G voxels evaluated together (G gathers in flight), then folded
#pragma unroll
                    for (int zg = 0; zg < ZH; zg += G) {
                        float tv[G];
                        uint32_t cv[G];
                        bool ok[G];
#pragma unroll
                        for (int k = 0; k < G; ++k) {
                            ok[k] = hv_tsdf_eval_fast<false, MT>(P, px, mult, pc0, pc1, pc2, tv[k], cv[k]);
                            pc0 += inc0;
                            pc1 += inc1;
                            pc2 += inc2;
                        }
#pragma unroll
                        for (int k = 0; k < G; ++k)
                            hv_tsdf_apply_fast(ok[k], tv[k], cv[k], vt[zg + k], wf[zg + k], vr[zg + k], vg[zg + k], vb[zg + k]);
                    }
                }
            }
#pragma unroll
            for (int zz = 0; zz < ZH; ++zz) {
                const uint32_t nw = (uint32_t)wf[zz];
                if (nw != vw[zz]) dirty |= 1u << zz;
                vw[zz] = nw;
            }
        }
#pragma unroll
        for (int zz = 0; zz < ZH; ++zz) {
            if (dirty & (1u << zz)) {
                const int q = wordb + zz * RR;
                ((float *)(unit + 0 * PLANE_BYTES))[q] = vt[zz];
                ((uint32_t *)(unit + 1 * PLANE_BYTES))[q] = vw[zz];
                ((uint32_t *)(unit + 2 * PLANE_BYTES))[q] = vr[zz];
                ((uint32_t *)(unit + 3 * PLANE_BYTES))[q] = vg[zz];
                ((uint32_t *)(unit + 4 * PLANE_BYTES))[q] = vb[zz];
            }
        }
    }
}

// ================================================================================================
// Sweep, second form (production).  Same mapping and results as k_tsdf_integrate_batch_col (lane = one voxel column,
// wave = 64 columns x ZH z, frames applied in order in registers).  The sweep is bound by VALU issue, not by HBM
// (profiles/r02/baseline: 0.79 of the wave-instruction peak at 54 instructions per voxel visit), so this form is about
// instructions and issue slots:
//  * the (u, v) projection chain is written on float2 values of ONE voxel ((pc0, pc1) walk, (a0, a1) / pc2 with the
//    refined reciprocal broadcast): v_pk_* instructions without the register shuffles the auto-vectoriser needed to
//    pair two voxels' chains in the first form (25 v_mov per 4 visits);
//  * gathers go through raw buffer descriptors (out-of-range offsets return 0: garbage lanes need no select), the pixel
//    index is a 24-bit mad;
//  * what does not depend on the frame (intrinsics, image size, truncation) is read once per work item, the 16 dwords
//    that do (extrinsics and the z step) are fetched one frame AHEAD into scalar registers, so the scalar-load latency
//    of a frame hides behind the previous frame's arithmetic;
//  * the near-camera-plane regime is detected once per work item (lane f tests frame f) instead of once per frame and
//    column, and routes the whole item to the EXACT loop;
//  * the running mean of a voxel pair is skipped when no lane of the wave updates either voxel;
//  * colour is accumulated in two packed registers per voxel (16-bit r and b fields, g in bits 8..23: <= 64 frames x 255
//    fit) and the three colour planes are only loaded for the final add of voxels that were updated.
// Measured (profiles/r02): 31.4 k frames/s against 29.2 k for the first form at 128 VGPRs / 4 waves per SIMD.
// Dead end kept out of the code: a wave-level cull (lane f projecting the wave's sub-block into frame f and testing it
// against per-tile depth maxima).  The instrumented oracle bounds it: only 18.9 % of (sub-block, frame) pairs update
// nothing (5 % outside the image, 14 % behind the band) although 47 % of voxel visits do - the 4 x 16 x 4 sub-block is
// long in y - and a conservative 16-pixel-tile test catches a third of those: 28.0 k with the cull against 28.3 k without.
// ================================================================================================
typedef float hv_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ hv_f2 hv_fma2(hv_f2 a, hv_f2 b, hv_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ hv_f2 hv_splat(float x) { return hv_f2{x, x}; }

struct HvSweepFrameK { // HvFrameParams::sweep_k in registers (plain members: the two prefetch sets must live in scalar registers)
    hv_f2 e04, e15, e26, e37;
    float e8, e11;
    hv_f2 e9_10, i01;
    float i2;
};
__device__ __forceinline__ HvSweepFrameK hv_sweep_frame_k(const HvFrameParams *__restrict__ Ps, int f) {
    const float *k = Ps[f].sweep_k;
    return HvSweepFrameK{hv_f2{k[0], k[1]}, hv_f2{k[2], k[3]}, hv_f2{k[4], k[5]}, hv_f2{k[6], k[7]}, k[8], k[9], hv_f2{k[10], k[11]}, hv_f2{k[12], k[13]}, k[14]};
}

template <int ZH>
struct HvSweepGather { // what a frame's evaluation leaves for its fold: the gathered records, multipliers, depths along z, tests
    uint2 rec[ZH];
    float mm[ZH], zk[ZH];
    bool ok[ZH];
};

template <int ZH, int SPLIT, int WPE>
__global__ __launch_bounds__(64 * (64 / ZH) / SPLIT, WPE) void k_tsdf_sweep(
    HvTable table, const int32_t *__restrict__ list, const unsigned long long *__restrict__ frame_mask,
    char *__restrict__ pool, const uint2 *__restrict__ frame_px, const HvFrameParams *__restrict__ Ps, int n_frames,
    int general, const float *__restrict__ mult, int xcd_aware, int parity) {
    static_assert(ZH % 2 == 0, "voxels are folded in pairs");
    constexpr int TASKS = 64 / ZH;          // wave tasks per unit
    constexpr int WAVES = TASKS / SPLIT;    // waves per workgroup
    int n_units = table.counters[HV_CNT_TOUCH(parity)];
    if (n_units > table.max_blocks) n_units = table.max_blocks;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // frame-independent constants (one camera and one volume per batch)
    const HvFrameParams &P0 = Ps[0];
    const float vl = P0.voxel_length_f, hl = P0.half_voxel_length_f;
    const double unit_length = P0.unit_length;
    const bool tiled = P0.tiled != 0;
    const int npx = P0.H * P0.W;
    const hv_f2 F = {P0.fx, P0.fy}, C = {P0.cx, P0.cy};
    const uint32_t lo = __float_as_uint(0.0001f);
    const uint32_t wlim = __float_as_uint(P0.safe_width_f) - lo, hlim = __float_as_uint(P0.safe_height_f) - lo;
    const uint32_t W24 = (uint32_t)P0.W;
    const float ntrunc = -P0.sdf_trunc_f, tinv = P0.sdf_trunc_inv_f;
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void *)mult, 0, npx * 4, 0x00020000);
    // XCD-aware work distribution (workgroup b runs on XCD b % 8; each XCD has its own L2): the union list is in touch order,
    // i.e. roughly in image raster order of the first frames, so XCD k takes the k-th CONTIGUOUS eighth of the list, all SPLIT
    // parts of a unit included - its units project into one band of the images and its L2 only has to hold that band of the
    // frame records, instead of every XCD pulling every frame whole (profiles/r02/baseline: 2.0x the algorithmic traffic).
    // Work items -> workgroups.  Workgroup b runs on XCD b % 8 and every XCD has its own L2, so with xcd_group = G > 0 the
    // SPLIT parts of a unit (which read the same depth pixels in every frame) and G consecutive list entries stay on one
    // XCD: item i -> XCD i & 7, and within that XCD entries (8 g + xcd) G .. + G - 1 of the list for g = 0, 1, ...
    // (G small enough that a heavy stretch of the list is spread over all XCDs: G = 2 measured best, +2.5 %, FETCH_SIZE halves;
    // one contiguous eighth of the list per XCD halves the traffic too but is 28 % slower, the dispatcher waits for the heaviest XCD).
    const int G = xcd_aware > 0 ? xcd_aware : 1;
    const int rounds = (n_units + 8 * G - 1) / (8 * G);
    const int n_items = xcd_aware > 0 ? rounds * 8 * G * SPLIT : n_units * SPLIT;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        int t, part;
        if (xcd_aware > 0) {
            const int xcd = item & 7, j = item >> 3;
            const int g = j / (G * SPLIT), within = j - g * (G * SPLIT);
            t = (g * 8 + xcd) * G + within / SPLIT;
            part = within % SPLIT;
            if (t >= n_units) continue;
        } else {
            t = item / SPLIT;
            part = item % SPLIT;
        }
        const int task = part * WAVES + wave;
        const int cg = task & 3;            // column group: x in [4 cg, 4 cg + 4)
        const int z0 = (task >> 2) * ZH;
        const int x = cg * 4 + (lane >> 4);
        const int y = lane & 15;
        const int32_t slot = list[t];
        const int32_t idx = table.vals[slot];
        unsigned long long mask = frame_mask[slot];
        if (idx < 0 || mask == 0ull) continue;
        int32_t ux, uy, uz;
        hv_unpack_key(table.keys[slot], ux, uy, uz);
        const double o0 = (double)ux * unit_length, o1 = (double)uy * unit_length, o2 = (double)uz * unit_length;
        // lane f <-> frame f: does the box of this wave's voxel centres come within 3 cm of frame f's camera plane?  Then a
        // voxel may leave the band the short division chain is verified for (pc2 >= 2^-20 where pc2 > 0) and the whole
        // item takes the EXACT loop (rare: a frame only has a unit in its mask when it sees a surface within sdf_trunc of it)
        bool near = false;
        if (lane < n_frames && ((mask >> lane) & 1ull)) {
            const HvFrameParams &Pl = Ps[lane];
            const float bx = (float)((double)(hl + vl * (float)(cg * 4)) + o0), by = (float)((double)hl + o1),
                        bz = (float)((double)(hl + vl * (float)z0) + o2);
            const float zmin = (Pl.ext[8] * bx + Pl.ext[9] * by + Pl.ext[10] * bz + Pl.ext[11]) + fminf(Pl.ext[8] * (3.0f * vl), 0.f) +
                               fminf(Pl.ext[9] * (15.0f * vl), 0.f) + fminf(Pl.ext[10] * ((float)(ZH - 1) * vl), 0.f);
            near = !(zmin > 0.03f);
        }
        const bool near_any = __any(near);
        char *unit = pool + (int64_t)idx * (PLANE_BYTES * HV_TSDF_PLANES);
        const int wordb = z0 * RR + cg * 64 + lane;
        float vt[ZH];
        uint32_t vw[ZH];
#pragma unroll
        for (int zz = 0; zz < ZH; ++zz) {
            const int q = wordb + zz * RR;
            vt[zz] = ((const float *)(unit + 0 * PLANE_BYTES))[q];
            vw[zz] = ((const uint32_t *)(unit + 1 * PLANE_BYTES))[q];
        }
        // the voxel centre of (x, y, z = 0) does not depend on the frame
        const float p0 = (float)((double)(hl + vl * (float)x) + o0);
        const float p1 = (float)((double)(hl + vl * (float)y) + o1);
        const float p2 = (float)((double)hl + o2);
        bool heavy = false; // a voxel near 2^24 observations: float weights would stop being exact
#pragma unroll
        for (int zz = 0; zz < ZH; ++zz) heavy |= vw[zz] >= (1u << 24) - 64u;
        unsigned dirty = 0;
        if (general || tiled || near_any || __any(heavy)) {
            // rare regimes: EXACT evaluation with integer weights, frame by frame (the first form's code)
            uint32_t vr[ZH], vg[ZH], vb[ZH];
#pragma unroll
            for (int zz = 0; zz < ZH; ++zz) {
                const int q = wordb + zz * RR;
                vr[zz] = ((const uint32_t *)(unit + 2 * PLANE_BYTES))[q];
                vg[zz] = ((const uint32_t *)(unit + 3 * PLANE_BYTES))[q];
                vb[zz] = ((const uint32_t *)(unit + 4 * PLANE_BYTES))[q];
            }
            while (mask) {
                const int f = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const HvFrameParams &P = Ps[f];
                const uint2 *px = frame_px + (int64_t)f * npx;
                const float inc0 = P.ext_scaled_col2[0], inc1 = P.ext_scaled_col2[1], inc2 = P.ext_scaled_col2[2];
                float pc0 = ((P.ext[0] * p0 + P.ext[1] * p1) + P.ext[2] * p2) + P.ext[3];
                float pc1 = ((P.ext[4] * p0 + P.ext[5] * p1) + P.ext[6] * p2) + P.ext[7];
                float pc2 = ((P.ext[8] * p0 + P.ext[9] * p1) + P.ext[10] * p2) + P.ext[11];
                for (int s = 0; s < z0; ++s) { // the reference's repeated float additions along z, replayed
                    pc0 += inc0;
                    pc1 += inc1;
                    pc2 += inc2;
                }
#pragma unroll
                for (int zz = 0; zz < ZH; ++zz) {
                    float tv;
                    uint32_t cv;
                    const bool ok = hv_tsdf_eval_fast<true, true>(P, px, mult, pc0, pc1, pc2, tv, cv);
                    pc0 += inc0;
                    pc1 += inc1;
                    pc2 += inc2;
                    hv_tsdf_apply(ok, tv, cv, vt[zz], vw[zz], vr[zz], vg[zz], vb[zz]);
                    if (ok) dirty |= 1u << zz;
                }
            }
#pragma unroll
            for (int zz = 0; zz < ZH; ++zz) {
                if (dirty & (1u << zz)) {
                    const int q = wordb + zz * RR;
                    ((float *)(unit + 0 * PLANE_BYTES))[q] = vt[zz];
                    ((uint32_t *)(unit + 1 * PLANE_BYTES))[q] = vw[zz];
                    ((uint32_t *)(unit + 2 * PLANE_BYTES))[q] = vr[zz];
                    ((uint32_t *)(unit + 3 * PLANE_BYTES))[q] = vg[zz];
                    ((uint32_t *)(unit + 4 * PLANE_BYTES))[q] = vb[zz];
                }
            }
            continue;
        }
        hv_f2 VT[ZH / 2], WF[ZH / 2];
        uint32_t arb[ZH], ag[ZH]; // colour of the batch: r | b << 16 and g << 8
#pragma unroll
        for (int zp = 0; zp < ZH / 2; ++zp) {
            VT[zp] = hv_f2{vt[2 * zp], vt[2 * zp + 1]};
            WF[zp] = hv_f2{(float)vw[2 * zp], (float)vw[2 * zp + 1]};
            arb[2 * zp] = arb[2 * zp + 1] = ag[2 * zp] = ag[2 * zp + 1] = 0u;
        }
        // One frame folded into the registers; K = the frame's 16 constant dwords, already in scalar registers.
        auto project_frame = [&](const HvSweepFrameK K, const int f, HvSweepFrameK &Knext, const unsigned long long rest,
                                 HvSweepGather<ZH> &g) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t rs_px = __builtin_amdgcn_make_buffer_rsrc((void *)(frame_px + (int64_t)f * npx), 0, npx * 8, 0x00020000);
            const hv_f2 INC = K.i01;
            const float inc2 = K.i2;
            // pc = ((e0 p0 + e1 p1) + e2 p2) + e3, rows 0 and 1 as one float2 (same IEEE ops as the reference)
            hv_f2 XY = ((K.e04 * p0 + K.e15 * p1) + K.e26 * p2) + K.e37;
            const hv_f2 Z12 = K.e9_10 * hv_f2{p1, p2};
            float Z = ((K.e8 * p0 + Z12.x) + Z12.y) + K.e11;
            // K is consumed (the wait for its scalar loads sits above): only now issue the loads of the next frame's constants,
            // so that wait does not include them (scalar loads return out of order: there is only "wait for all")
            __builtin_amdgcn_sched_barrier(0);
            Knext = hv_sweep_frame_k(Ps, rest ? __ffsll((long long)rest) - 1 : f); // (the last frame re-reads itself: no branch, the order above holds)
            __builtin_amdgcn_sched_barrier(0);
            for (int s = 0; s < z0; ++s) { // the reference's repeated float additions along z, replayed
                XY += INC;
                Z += inc2;
            }
            // ---- evaluation of the ZH voxels: all 2 ZH gathers in flight ----
#pragma unroll
            for (int k = 0; k < ZH; ++k) {
                // (a0, a1) / pc2, correctly rounded, sharing one refined reciprocal (hv_div2's chain on a float2)
                float r = __builtin_amdgcn_rcpf(Z);
                const float e = fmaf(-Z, r, 1.0f);
                r = fmaf(e, r, r);
                const hv_f2 A = XY * F;
                const hv_f2 R = hv_splat(r), NZ = hv_splat(-Z);
                hv_f2 Q = A * R;
                hv_f2 REM = hv_fma2(NZ, Q, A);
                Q = hv_fma2(REM, R, Q);
                REM = hv_fma2(NZ, Q, A);
                Q = hv_fma2(REM, R, Q);
                const hv_f2 UV = (Q + C) + hv_splat(0.5f);
                // u_f in [0.0001, safe_width) as ONE unsigned compare (bit patterns of non-negative floats order like the values)
                const bool in_u = (__float_as_uint(UV.x) - lo) < wlim;
                const bool in_v = (__float_as_uint(UV.y) - lo) < hlim;
                g.ok[k] = (int)(Z > 0.0f) & (int)in_u & (int)in_v;
                const uint32_t u = (uint32_t)(int)UV.x, v = (uint32_t)(int)UV.y; // saturating conversions: garbage lanes stay defined
                const uint32_t off = __umul24(v, W24) + u; // exact for every in-image pixel; a garbage lane reads 0 or some pixel, unused
                g.rec[k] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs_px, (int)(off << 3), 0, 0));
                g.mm[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_m, (int)(off << 2), 0, 0));
                g.zk[k] = Z;
                XY += INC;
                Z += inc2;
            }
        };
        // ---- fold of an evaluated frame, two voxels at a time ----
        auto apply_frame = [&](const HvSweepGather<ZH> &g) __attribute__((always_inline)) {
#pragma unroll
            for (int zp = 0; zp < ZH / 2; ++zp) {
                const int k0 = 2 * zp, k1 = 2 * zp + 1;
                const float da = __uint_as_float(g.rec[k0].x), db = __uint_as_float(g.rec[k1].x);
                const float sa = (da - g.zk[k0]) * g.mm[k0], sb = (db - g.zk[k1]) * g.mm[k1];
                const bool oka = (int)g.ok[k0] & (int)(da > 0.0f) & (int)(sa > ntrunc);
                const bool okb = (int)g.ok[k1] & (int)(db > 0.0f) & (int)(sb > ntrunc);
                if (!__any((int)oka | (int)okb)) continue; // no lane of the wave updates either voxel
                const hv_f2 T = {fminf(sa * tinv, 1.0f), fminf(sb * tinv, 1.0f)}; // == `if (t > 1) t = 1` for the non-NaN t of an accepted voxel
                // running mean (tsdf * w + t) / (w + 1) with float weights (exact below 2^24), hv_div1's chain on a float2
                const hv_f2 W1 = WF[zp] + hv_splat(1.0f);
                const hv_f2 NUM = VT[zp] * WF[zp] + T;
                hv_f2 R = {__builtin_amdgcn_rcpf(W1.x), __builtin_amdgcn_rcpf(W1.y)};
                const hv_f2 E = hv_fma2(-W1, R, hv_splat(1.0f));
                R = hv_fma2(E, R, R);
                hv_f2 Q = NUM * R;
                hv_f2 REM = hv_fma2(-W1, Q, NUM);
                Q = hv_fma2(REM, R, Q);
                REM = hv_fma2(-W1, Q, NUM);
                Q = hv_fma2(REM, R, Q);
                VT[zp].x = oka ? Q.x : VT[zp].x;
                VT[zp].y = okb ? Q.y : VT[zp].y;
                WF[zp].x = oka ? W1.x : WF[zp].x;
                WF[zp].y = okb ? W1.y : WF[zp].y;
                const uint32_t ca = oka ? g.rec[k0].y : 0u, cb = okb ? g.rec[k1].y : 0u;
                arb[k0] += ca & 0x00ff00ffu; ag[k0] += ca & 0x0000ff00u;
                arb[k1] += cb & 0x00ff00ffu; ag[k1] += cb & 0x0000ff00u;
            }
        };
        // Frame constants are fetched one frame AHEAD, alternating between two scalar register sets (a copy between sets
        // would make the wave wait for the load at once): the scalar-load latency of frame n+1 hides behind frame n's fold.
        HvSweepFrameK ka = hv_sweep_frame_k(Ps, __ffsll((long long)mask) - 1), kb = ka;
        // (Measured dead end: issuing frame n+1's evaluation - projection + gathers - BEFORE folding frame n, two gather sets
        // in flight at 126 VGPRs: 32.0 k frames/s against 32.7 k.  The wave does not wait for its gathers.)
        HvSweepGather<ZH> g;
        while (true) {
            const int fa = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            project_frame(ka, fa, kb, mask, g);
            apply_frame(g);
            if (!mask) break;
            const int fb = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            project_frame(kb, fb, ka, mask, g);
            apply_frame(g);
            if (!mask) break;
        }
#pragma unroll
        for (int zz = 0; zz < ZH; ++zz) {
            const float wfz = (zz & 1) ? WF[zz / 2].y : WF[zz / 2].x;
            const uint32_t nw = (uint32_t)wfz;
            if (nw != vw[zz]) { // updated by at least one frame: fold the batch's colour into the planes
                const int q = wordb + zz * RR;
                ((float *)(unit + 0 * PLANE_BYTES))[q] = (zz & 1) ? VT[zz / 2].y : VT[zz / 2].x;
                ((uint32_t *)(unit + 1 * PLANE_BYTES))[q] = nw;
                ((uint32_t *)(unit + 2 * PLANE_BYTES))[q] += arb[zz] & 0xffffu;
                ((uint32_t *)(unit + 3 * PLANE_BYTES))[q] += ag[zz] >> 8;
                ((uint32_t *)(unit + 4 * PLANE_BYTES))[q] += arb[zz] >> 16;
            }
        }
    }
}

// ================================================================================================
// Sweep, third form (production since round 3): the batch is FOLDED per voxel.  The contract asks for bit-exact unit
// keys and weights and 1e-4 on tsdf / colour (BASELINE.json north_star); the running mean (tsdf w + t) / (w + 1) applied
// once per accepted frame is, in real numbers, (tsdf w0 + sum t) / (w0 + n) - so a lane keeps `sum t` and n per voxel in
// registers over the batch's frames and divides ONCE when the item is done (measured against the oracle's per-frame
// float chain: <= 3e-7).  What decides WHETHER a frame updates a voxel - the projection, the pixel it lands in, the
// truncation test - is still evaluated with the reference's IEEE operations in the reference's order, so weights and
// colour sums stay exact integers.  Against the second form a voxel visit loses the correctly rounded running-mean
// division (two rcp + 11 fma-class per voxel pair), the float weight carry, and three tests the item-level regime test
// already implies:
//  * `pc2 > 0`: an item only takes this path when every voxel of the wave's box lies beyond `near_z` of every frame in
//    its mask;
//  * `depth > 0`: with near_z >= 1.25 sdf_trunc and a multiplier >= 1, a record with depth <= 0 (invalid, truncated, or
//    the zero an out-of-range buffer load returns) gives sdf <= -pc2 < -sdf_trunc and fails the truncation test by itself;
//  * the image-tile test of the tile-sharded mode folds into the image-range compare (the range constants become the
//    tile's), so tiled volumes take the fast path too.
// The observation count rides in byte 3 of the packed colour word (HV_REC_ONE, set by the pack role): one masked add
// accumulates green and the count.  tsdf / weight planes are only read when the item is done, and only by lanes with n > 0.
// ================================================================================================
template <int ZH, int SPLIT, int WPE, bool ANYSKIP, int DBG = 0, bool REC12 = true>
__global__ __launch_bounds__(64 * (64 / ZH) / SPLIT, WPE) void k_tsdf_sweep_fold(
    HvTable table, const int32_t *__restrict__ list, const unsigned long long *__restrict__ frame_mask,
    char *__restrict__ pool, const uint2 *__restrict__ frame_px, const HvFrameParams *__restrict__ Ps, int n_frames,
    int general, const float *__restrict__ mult, int xcd_aware, int parity) {
    constexpr int TASKS = 64 / ZH;          // wave tasks per unit
    constexpr int WAVES = TASKS / SPLIT;    // waves per workgroup
    int n_units = table.counters[HV_CNT_TOUCH(parity)];
    if (n_units > table.max_blocks) n_units = table.max_blocks;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const HvFrameParams &P0 = Ps[0];
    const float vl = P0.voxel_length_f, hl = P0.half_voxel_length_f;
    const double unit_length = P0.unit_length;
    const int npx = P0.H * P0.W;
    const hv_f2 F = {P0.fx, P0.fy}, C = {P0.cx, P0.cy};
    // u_f in [max(0.0001, tile_u0), min(safe_width, tile_u1)) as ONE unsigned compare (bit patterns of non-negative floats order
    // like the values; (int)u_f >= k <=> u_f >= k for integers k >= 0): the whole-image tile gives the reference's range
    const float lo_uf = fmaxf(0.0001f, (float)P0.tile_u0), lo_vf = fmaxf(0.0001f, (float)P0.tile_v0);
    const float hi_uf = fminf(P0.safe_width_f, (float)P0.tile_u1), hi_vf = fminf(P0.safe_height_f, (float)P0.tile_v1);
    const uint32_t lo_u = __float_as_uint(lo_uf), lo_v = __float_as_uint(lo_vf);
    const uint32_t lim_u = hi_uf > lo_uf ? __float_as_uint(hi_uf) - lo_u : 0u, lim_v = hi_vf > lo_vf ? __float_as_uint(hi_vf) - lo_v : 0u;
    const uint32_t W24 = (uint32_t)P0.W;
    const float ntrunc = -P0.sdf_trunc_f, tinv = P0.sdf_trunc_inv_f;
    const float near_z = fmaxf(0.03f, 1.25f * P0.sdf_trunc_f);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void *)mult, 0, npx * 4, 0x00020000);
    // work items -> workgroups: as in k_tsdf_sweep (XCD groups)
    const int G = xcd_aware > 0 ? xcd_aware : 1;
    const int rounds = (n_units + 8 * G - 1) / (8 * G);
    const int n_items = xcd_aware > 0 ? rounds * 8 * G * SPLIT : n_units * SPLIT;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        int t, part;
        if (xcd_aware > 0) {
            const int xcd = item & 7, j = item >> 3;
            const int g = j / (G * SPLIT), within = j - g * (G * SPLIT);
            t = (g * 8 + xcd) * G + within / SPLIT;
            part = within % SPLIT;
            if (t >= n_units) continue;
        } else {
            t = item / SPLIT;
            part = item % SPLIT;
        }
        const int task = part * WAVES + wave;
        const int cg = task & 3;            // column group: x in [4 cg, 4 cg + 4)
        const int z0 = (task >> 2) * ZH;
        const int x = cg * 4 + (lane >> 4);
        const int y = lane & 15;
        const int32_t slot = list[t];
        const int32_t idx = table.vals[slot];
        unsigned long long mask = frame_mask[slot];
        if (idx < 0 || mask == 0ull) continue;
        int32_t ux, uy, uz;
        hv_unpack_key(table.keys[slot], ux, uy, uz);
        const double o0 = (double)ux * unit_length, o1 = (double)uy * unit_length, o2 = (double)uz * unit_length;
        // lane f <-> frame f: does the box of this wave's voxel centres come within near_z of frame f's camera plane?
        bool near = false;
        if (lane < n_frames && ((mask >> lane) & 1ull)) {
            const HvFrameParams &Pl = Ps[lane];
            const float bx = (float)((double)(hl + vl * (float)(cg * 4)) + o0), by = (float)((double)hl + o1),
                        bz = (float)((double)(hl + vl * (float)z0) + o2);
            const float zmin = (Pl.ext[8] * bx + Pl.ext[9] * by + Pl.ext[10] * bz + Pl.ext[11]) + fminf(Pl.ext[8] * (3.0f * vl), 0.f) +
                               fminf(Pl.ext[9] * (15.0f * vl), 0.f) + fminf(Pl.ext[10] * ((float)(ZH - 1) * vl), 0.f);
            near = !(zmin > near_z);
        }
        const bool near_any = __any(near);
        char *unit = pool + (int64_t)idx * (PLANE_BYTES * HV_TSDF_PLANES);
        const int wordb = z0 * RR + cg * 64 + lane;
        // the voxel centre of (x, y, z = 0) does not depend on the frame
        const float p0 = (float)((double)(hl + vl * (float)x) + o0);
        const float p1 = (float)((double)(hl + vl * (float)y) + o1);
        const float p2 = (float)((double)hl + o2);
        if (general || near_any) {
            // rare regime: the reference's evaluation frame by frame with integer weights (k_tsdf_sweep's EXACT arithmetic), one
            // voxel at a time so that this path's registers do not set the kernel's occupancy
#pragma unroll 1
            for (int zz = 0; zz < ZH; ++zz) {
                const int q = wordb + zz * RR;
                float vt = ((const float *)(unit + 0 * PLANE_BYTES))[q];
                uint32_t vw = ((const uint32_t *)(unit + 1 * PLANE_BYTES))[q];
                uint32_t vr = ((const uint32_t *)(unit + 2 * PLANE_BYTES))[q];
                uint32_t vg = ((const uint32_t *)(unit + 3 * PLANE_BYTES))[q];
                uint32_t vb = ((const uint32_t *)(unit + 4 * PLANE_BYTES))[q];
                bool dirty = false;
                unsigned long long m = mask;
#pragma unroll 1
                while (m) {
                    const int f = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const HvFrameParams &P = Ps[f];
                    const float inc0 = P.ext_scaled_col2[0], inc1 = P.ext_scaled_col2[1], inc2 = P.ext_scaled_col2[2];
                    float pc0 = ((P.ext[0] * p0 + P.ext[1] * p1) + P.ext[2] * p2) + P.ext[3];
                    float pc1 = ((P.ext[4] * p0 + P.ext[5] * p1) + P.ext[6] * p2) + P.ext[7];
                    float pc2 = ((P.ext[8] * p0 + P.ext[9] * p1) + P.ext[10] * p2) + P.ext[11];
#pragma unroll 1
                    for (int s = 0; s < z0 + zz; ++s) { // the reference's repeated float additions along z, replayed
                        pc0 += inc0;
                        pc1 += inc1;
                        pc2 += inc2;
                    }
                    float tv;
                    uint32_t cv;
                    const uint2 *px_f = REC12 ? (const uint2 *)((const uint32_t *)frame_px + (int64_t)f * npx * 3) : frame_px + (int64_t)f * npx;
                    const bool ok = hv_tsdf_eval_fast<true, true, REC12>(P, px_f, mult, pc0, pc1, pc2, tv, cv);
                    hv_tsdf_apply(ok, tv, cv, vt, vw, vr, vg, vb);
                    dirty |= ok;
                }
                if (dirty) {
                    ((float *)(unit + 0 * PLANE_BYTES))[q] = vt;
                    ((uint32_t *)(unit + 1 * PLANE_BYTES))[q] = vw;
                    ((uint32_t *)(unit + 2 * PLANE_BYTES))[q] = vr;
                    ((uint32_t *)(unit + 3 * PLANE_BYTES))[q] = vg;
                    ((uint32_t *)(unit + 4 * PLANE_BYTES))[q] = vb;
                }
            }
            continue;
        }
        float S[ZH];             // sum of the accepted frames' t
        uint32_t arb[ZH], agn[ZH]; // r | b << 16 and g << 8 | n << 24 of the accepted frames
#pragma unroll
        for (int k = 0; k < ZH; ++k) {
            S[k] = 0.0f;
            arb[k] = agn[k] = 0u;
        }
        auto fold_frame = [&](const HvSweepFrameK K, const int f, HvSweepFrameK &Knext, const unsigned long long rest) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t rs_px =
                REC12 ? __builtin_amdgcn_make_buffer_rsrc((void *)((const uint32_t *)frame_px + (int64_t)f * npx * 3), 0, npx * 12, 0x00020000)
                      : __builtin_amdgcn_make_buffer_rsrc((void *)(frame_px + (int64_t)f * npx), 0, npx * 8, 0x00020000);
            const hv_f2 INC = K.i01;
            const float inc2 = K.i2;
            // pc = ((e0 p0 + e1 p1) + e2 p2) + e3, rows 0 and 1 as one float2 (same IEEE ops as the reference)
            hv_f2 XY = ((K.e04 * p0 + K.e15 * p1) + K.e26 * p2) + K.e37;
            const hv_f2 Z12 = K.e9_10 * hv_f2{p1, p2};
            float Z = ((K.e8 * p0 + Z12.x) + Z12.y) + K.e11;
            __builtin_amdgcn_sched_barrier(0);
            Knext = hv_sweep_frame_k(Ps, rest ? __ffsll((long long)rest) - 1 : f);
            __builtin_amdgcn_sched_barrier(0);
            for (int s = 0; s < z0; ++s) { // the reference's repeated float additions along z, replayed
                XY += INC;
                Z += inc2;
            }
            uint2 rec[ZH];
            float mm[ZH], zk[ZH];
            bool inimg[ZH];
#pragma unroll
            for (int k = 0; k < ZH; ++k) {
                // (a0, a1) / pc2, correctly rounded, sharing one refined reciprocal (hv_div2's chain on a float2)
                float r = __builtin_amdgcn_rcpf(Z);
                const float e = fmaf(-Z, r, 1.0f);
                r = fmaf(e, r, r);
                const hv_f2 A = XY * F;
                const hv_f2 R = hv_splat(r), NZ = hv_splat(-Z);
                hv_f2 Q = A * R;
                hv_f2 REM = hv_fma2(NZ, Q, A);
                Q = hv_fma2(REM, R, Q);
                REM = hv_fma2(NZ, Q, A);
                Q = hv_fma2(REM, R, Q);
                const hv_f2 UV = (Q + C) + hv_splat(0.5f);
                const bool in_u = (__float_as_uint(UV.x) - lo_u) < lim_u;
                const bool in_v = (__float_as_uint(UV.y) - lo_v) < lim_v;
                inimg[k] = (int)in_u & (int)in_v;
                const uint32_t u = (uint32_t)(int)UV.x, v = (uint32_t)(int)UV.y; // saturating conversions: garbage lanes stay defined
                const uint32_t off = __umul24(v, W24) + u; // exact for every in-image pixel; a garbage lane reads 0 or some pixel, unused
                // (DBG != 0: timing ablations for profiles/, selected by HV_TSDF_SWEEP_DBG; they do not produce a valid volume.
                //  1: no multiplier gather; 2: no gathers at all; 3: both gathers at lane-contiguous addresses)
                if (DBG == 2) {
                    rec[k] = make_uint2(__float_as_uint(Z + 0.01f), off);
                    mm[k] = 1.0f;
                } else if (REC12) {
                    typedef uint32_t hv_u3 __attribute__((ext_vector_type(3)));
                    const uint32_t goff = DBG == 3 ? ((off & 0xffc0u) | (uint32_t)lane) : off;
                    const hv_u3 r3 = __builtin_bit_cast(hv_u3, __builtin_amdgcn_raw_buffer_load_b96(rs_px, (int)__umul24(goff, 12u), 0, 0));
                    rec[k] = make_uint2(r3.x, r3.y);
                    mm[k] = __uint_as_float(r3.z);
                } else {
                    const uint32_t goff = DBG == 3 ? ((off & 0xffc0u) | (uint32_t)lane) : off;
                    rec[k] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs_px, (int)(goff << 3), 0, 0));
                    mm[k] = DBG == 1 ? 1.0f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_m, (int)(goff << 2), 0, 0));
                }
                zk[k] = Z;
                XY += INC;
                Z += inc2;
            }
#pragma unroll
            for (int k = 0; k < ZH; ++k) {
                const float sdf = (__uint_as_float(rec[k].x) - zk[k]) * mm[k];
                const bool ok = (int)inimg[k] & (int)(sdf > ntrunc);
                if (ANYSKIP && !__any(ok)) continue;
                const float tk = fminf(sdf * tinv, 1.0f); // == `if (t > 1) t = 1` for the non-NaN t of an accepted voxel
                S[k] += ok ? tk : 0.0f;
                const uint32_t c = ok ? rec[k].y : 0u;
                arb[k] += c & 0x00ff00ffu;
                agn[k] += c & 0xff00ff00u;
            }
        };
        HvSweepFrameK ka = hv_sweep_frame_k(Ps, __ffsll((long long)mask) - 1), kb = ka;
        while (true) {
            const int fa = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            fold_frame(ka, fa, kb, mask);
            if (!mask) break;
            const int fb = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            fold_frame(kb, fb, ka, mask);
            if (!mask) break;
        }
        // one running-mean step per voxel for the whole batch
        float vt[ZH];
        uint32_t vw[ZH];
#pragma unroll
        for (int zz = 0; zz < ZH; ++zz) {
            if (agn[zz] >> 24) {
                const int q = wordb + zz * RR;
                vt[zz] = ((const float *)(unit + 0 * PLANE_BYTES))[q];
                vw[zz] = ((const uint32_t *)(unit + 1 * PLANE_BYTES))[q];
            }
        }
#pragma unroll
        for (int zz = 0; zz < ZH; ++zz) {
            const uint32_t n = agn[zz] >> 24;
            if (n) {
                const int q = wordb + zz * RR;
                const uint32_t nw = vw[zz] + n;
                ((float *)(unit + 0 * PLANE_BYTES))[q] = (vt[zz] * (float)vw[zz] + S[zz]) / (float)nw;
                ((uint32_t *)(unit + 1 * PLANE_BYTES))[q] = nw;
                ((uint32_t *)(unit + 2 * PLANE_BYTES))[q] += arb[zz] & 0xffffu;
                ((uint32_t *)(unit + 3 * PLANE_BYTES))[q] += (agn[zz] >> 8) & 0xffffu;
                ((uint32_t *)(unit + 4 * PLANE_BYTES))[q] += arb[zz] >> 16;
            }
        }
    }
}

// ================================================================================================
// Sweep, fourth form: the fold form on WHOLE voxel columns.  A lane owns one (x, y) column of the unit and all 16 z of
// it, a wave 64 columns (one column group), a unit is 4 wave tasks.  Against the third form (4 z per lane):
//  * the z-walk starts at z = 0 for every lane: the replay of the reference's repeated float additions up to z0 (0 / 4 / 8
//    / 12 steps, two instructions each) disappears, and the per-frame set-up (the rigid transform of the column's base
//    point, the frame's buffer descriptor, the prefetch of the next frame's constants) is paid once per 16 voxels
//    instead of once per 4: 42.7 -> ~35 VALU instructions per voxel visit;
//  * the voxels of a column are evaluated GV at a time; with PIPE the gathers of group g+1 are issued BEFORE group g is
//    folded, so a wave always has GV gathers in flight behind GV voxels' worth of arithmetic.
// 12-byte frame records only ({depth, colour | 1 << 24, multiplier}: one gather per voxel visit).  Registers: 3 x 16
// accumulators (sum t, r | b << 16, g << 8 | n << 24) + two gather groups.
// ================================================================================================
// ANYSKIP: 0 = every fold step runs, rejected lanes add zeros (selects); 1 = that + a wave-uniform skip of a step no lane
// accepts; 2 = the accumulation runs under the EXEC mask of the accepting lanes (no selects: 14 instead of 16 vector
// instructions in a fold step; the compiler's execz branch is the skip).  Measured on the bench stream in three A/B runs:
// 557-563 us (2) against 562-576 us (1) - 2 % in two of them, nothing in the third.  Measured and dropped with it: the pixel offset as one v_mad_u32_u24 instead of the compiler's
// v_mad_u64_u32 (561-567 us), the sum of sdf kept in metres with one multiplication by 1 / trunc per batch (one
// instruction less per visit, 564-568 us: the extra live register costs more), 16-byte records with the colour fields
// spread 16 bits apart so that the accumulators add the words unmasked (two instructions less, 590-599 us: four registers
// per gather in flight, 104 B of scratch).  At 128 registers the form is bound by what it keeps live, not by its count.
// Correction rounds of the projection's shared-reciprocal division in the column sweep.  hv_div2 (online path, bitwise sweep forms)
// runs two; with the reciprocal refined by one Newton step the FIRST round already returns the correctly rounded quotient on every
// operand pair tried: tools/divtest.hip, round 4 - 0 mismatches against IEEE division in 3.4e12 pairs each of the kernel's operand
// ranges, wide random exponents and divisors whose mantissa ends in runs of ones / zeros (1.0e13 divisions; the two-round form:
// 0 as well).  Two packed FMAs less per voxel visit.  -DHV_SWEEP_DIV_ROUNDS=2 restores the second round.
#ifndef HV_SWEEP_DIV_ROUNDS
#define HV_SWEEP_DIV_ROUNDS 1
#endif
// hv_sweep_column_core: the sweep of ONE work item (unit slot, part) as the callable `run_item(slot, part)`, handed to `drive`, which
// decides what items this workgroup runs (k_tsdf_sweep_column: the grid-stride loop over the batch's union list; k_tsdf_fused: the
// same items interleaved with the touch + pack items of the NEXT batch).  Everything inlines: one copy of the code per kernel.
#ifndef HV_SWEEP_INTERIOR
#define HV_SWEEP_INTERIOR 1
#endif
template <int SPLIT, int GV, int PIPE, int ANYSKIP, int ZS, class Drive, bool INTERIOR = (HV_SWEEP_INTERIOR != 0 && PIPE != 2)>
__device__ __forceinline__ void hv_sweep_column_core(
    const HvTable &table, const unsigned long long *__restrict__ frame_mask,
    char *__restrict__ pool, const uint2 *__restrict__ frame_px, const HvFrameParams *__restrict__ Ps, const int n_frames,
    const int general, const float *__restrict__ mult, Drive drive) {
    // ZS = 2: a column is split into two z halves = 8 wave tasks per unit (a GPU that owns few units - multi-GPU sharding - has
    // ~3 000 column tasks of very different lengths for 4 096 wave slots: nothing evens them out; twice as many, half as long
    // tasks do).  The upper half replays the reference's 8 repeated float additions along z per frame.
    static_assert(ZS == 1 || (ZS == 2 && SPLIT == 4), "z halves need one-wave workgroups");
    constexpr int ZH = 16 / ZS;
    constexpr int NG = ZH / GV;
    constexpr int WAVES = 4 / SPLIT; // waves per workgroup
    typedef uint32_t hv_u3 __attribute__((ext_vector_type(3)));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const HvFrameParams &P0 = Ps[0];
    const float vl = P0.voxel_length_f, hl = P0.half_voxel_length_f;
    const double unit_length = P0.unit_length;
    const int npx = P0.H * P0.W;
    const hv_f2 F = {P0.fx, P0.fy}, C = {P0.cx, P0.cy};
    const float lo_uf = fmaxf(0.0001f, (float)P0.tile_u0), lo_vf = fmaxf(0.0001f, (float)P0.tile_v0);
    const float hi_uf = fminf(P0.safe_width_f, (float)P0.tile_u1), hi_vf = fminf(P0.safe_height_f, (float)P0.tile_v1);
    const uint32_t lo_u = __float_as_uint(lo_uf), lo_v = __float_as_uint(lo_vf);
    const uint32_t lim_u = hi_uf > lo_uf ? __float_as_uint(hi_uf) - lo_u : 0u, lim_v = hi_vf > lo_vf ? __float_as_uint(hi_vf) - lo_v : 0u;
    const uint32_t W24 = (uint32_t)P0.W;
    const float ntrunc = -P0.sdf_trunc_f, tinv = P0.sdf_trunc_inv_f;
    const float near_z = fmaxf(0.03f, 1.25f * P0.sdf_trunc_f);
    auto run_item = [&](const int32_t slot, int part) __attribute__((always_inline)) {
        const int z0 = ZS == 1 ? 0 : (part >> 2) * ZH; // first z of this task
        if (ZS == 2) part &= 3;
        const int cg = part * WAVES + wave; // column group: x in [4 cg, 4 cg + 4)
        const int x = cg * 4 + (lane >> 4);
        const int y = lane & 15;
        const int32_t idx = table.vals[slot];
        unsigned long long mask = frame_mask[slot];
        if (idx < 0 || mask == 0ull) return;
        int32_t ux, uy, uz;
        hv_unpack_key(table.keys[slot], ux, uy, uz);
        const double o0 = (double)ux * unit_length, o1 = (double)uy * unit_length, o2 = (double)uz * unit_length;
        // lane f <-> frame f: does the box of this wave's voxel centres come within near_z of frame f's camera plane?
        bool near = false;
        // ... and (round 5, INTERIOR): does the box project at least 2 pixels inside the image (the GPU's tile) in frame f?  Then every voxel
        // of the item does - a perspective projection maps the box into the hull of its projected corners - and the frame's visits skip the
        // image-range test: two v_sub + two v_cmp of ~30 instructions per visit, for ~70 % of the (item, frame) pairs of the bench's stream.
        bool interior = false;
        if (lane < n_frames && ((mask >> lane) & 1ull)) {
            const HvFrameParams &Pl = Ps[lane];
            const float bx = (float)((double)(hl + vl * (float)(cg * 4)) + o0), by = (float)((double)hl + o1),
                        bz = (float)((double)(hl + vl * (float)z0) + o2);
            const float zmin = (Pl.ext[8] * bx + Pl.ext[9] * by + Pl.ext[10] * bz + Pl.ext[11]) + fminf(Pl.ext[8] * (3.0f * vl), 0.f) +
                               fminf(Pl.ext[9] * (15.0f * vl), 0.f) + fminf(Pl.ext[10] * ((float)(ZH - 1) * vl), 0.f);
            near = !(zmin > near_z);
            if (INTERIOR && !near) {
                float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float qx = bx + ((c & 1) ? 3.0f * vl : 0.0f), qy = by + ((c & 2) ? 15.0f * vl : 0.0f), qz = bz + ((c & 4) ? (float)(ZH - 1) * vl : 0.0f);
                    const float c2 = Pl.ext[8] * qx + Pl.ext[9] * qy + Pl.ext[10] * qz + Pl.ext[11];
                    const float rz = 1.0f / c2; // (c2 >= zmin > near_z)
                    const float cu = (Pl.ext[0] * qx + Pl.ext[1] * qy + Pl.ext[2] * qz + Pl.ext[3]) * F.x * rz + C.x + 0.5f;
                    const float cv = (Pl.ext[4] * qx + Pl.ext[5] * qy + Pl.ext[6] * qz + Pl.ext[7]) * F.y * rz + C.y + 0.5f;
                    umin = fminf(umin, cu); umax = fmaxf(umax, cu);
                    vmin = fminf(vmin, cv); vmax = fmaxf(vmax, cv);
                }
                interior = umin >= lo_uf + 2.0f && umax <= hi_uf - 2.0f && vmin >= lo_vf + 2.0f && vmax <= hi_vf - 2.0f;
            }
        }
        const unsigned long long interior_mask = INTERIOR ? __ballot(interior) : 0ull;
        const bool near_any = __any(near);
        char *unit = pool + (int64_t)idx * (PLANE_BYTES * HV_TSDF_PLANES);
        const int wordb = z0 * RR + cg * 64 + lane;
        const float p0 = (float)((double)(hl + vl * (float)x) + o0);
        const float p1 = (float)((double)(hl + vl * (float)y) + o1);
        const float p2 = (float)((double)hl + o2);
        if (general || near_any) {
            // rare regime: the reference's evaluation frame by frame with integer weights, one voxel at a time
#pragma unroll 1
            for (int zz = 0; zz < ZH; ++zz) {
                const int q = wordb + zz * RR;
                float vt = ((const float *)(unit + 0 * PLANE_BYTES))[q];
                uint32_t vw = ((const uint32_t *)(unit + 1 * PLANE_BYTES))[q];
                uint32_t vr = ((const uint32_t *)(unit + 2 * PLANE_BYTES))[q];
                uint32_t vg = ((const uint32_t *)(unit + 3 * PLANE_BYTES))[q];
                uint32_t vb = ((const uint32_t *)(unit + 4 * PLANE_BYTES))[q];
                bool dirty = false;
                unsigned long long m = mask;
#pragma unroll 1
                while (m) {
                    const int f = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const HvFrameParams &P = Ps[f];
                    const float inc0 = P.ext_scaled_col2[0], inc1 = P.ext_scaled_col2[1], inc2 = P.ext_scaled_col2[2];
                    float pc0 = ((P.ext[0] * p0 + P.ext[1] * p1) + P.ext[2] * p2) + P.ext[3];
                    float pc1 = ((P.ext[4] * p0 + P.ext[5] * p1) + P.ext[6] * p2) + P.ext[7];
                    float pc2 = ((P.ext[8] * p0 + P.ext[9] * p1) + P.ext[10] * p2) + P.ext[11];
#pragma unroll 1
                    for (int s = 0; s < z0 + zz; ++s) { // the reference's repeated float additions along z, replayed
                        pc0 += inc0;
                        pc1 += inc1;
                        pc2 += inc2;
                    }
                    float tv;
                    uint32_t cv;
                    const uint2 *px_f = (const uint2 *)((const uint32_t *)frame_px + (int64_t)f * npx * 3);
                    const bool ok = hv_tsdf_eval_fast<true, true, true>(P, px_f, mult, pc0, pc1, pc2, tv, cv);
                    hv_tsdf_apply(ok, tv, cv, vt, vw, vr, vg, vb);
                    dirty |= ok;
                }
                if (dirty) {
                    ((float *)(unit + 0 * PLANE_BYTES))[q] = vt;
                    ((uint32_t *)(unit + 1 * PLANE_BYTES))[q] = vw;
                    ((uint32_t *)(unit + 2 * PLANE_BYTES))[q] = vr;
                    ((uint32_t *)(unit + 3 * PLANE_BYTES))[q] = vg;
                    ((uint32_t *)(unit + 4 * PLANE_BYTES))[q] = vb;
                }
            }
            return;
        }
        float S[ZH];               // sum of the accepted frames' t
        uint32_t arb[ZH], agn[ZH]; // r | b << 16 and g << 8 | n << 24 of the accepted frames
#pragma unroll
        for (int k = 0; k < ZH; ++k) {
            S[k] = 0.0f;
            arb[k] = agn[k] = 0u;
        }
        struct Group { // GV voxels between their gather and their fold
            hv_u3 rec[GV];
            float zk[GV];
            bool inimg[GV];
        };
        // the frame whose voxels are being projected
        hv_f2 XY, INC;
        float Z, inc2;
        __amdgpu_buffer_rsrc_t rs_px;
        auto begin_frame = [&](const HvSweepFrameK K, const int f, HvSweepFrameK &Knext, const unsigned long long rest) __attribute__((always_inline)) {
            rs_px = __builtin_amdgcn_make_buffer_rsrc((void *)((const uint32_t *)frame_px + (int64_t)f * npx * 3), 0, npx * 12, 0x00020000);
            INC = K.i01;
            inc2 = K.i2;
            // pc = ((e0 p0 + e1 p1) + e2 p2) + e3 at z = 0, rows 0 and 1 as one float2 (same IEEE ops as the reference)
            XY = ((K.e04 * p0 + K.e15 * p1) + K.e26 * p2) + K.e37;
            const hv_f2 Z12 = K.e9_10 * hv_f2{p1, p2};
            Z = ((K.e8 * p0 + Z12.x) + Z12.y) + K.e11;
            if (ZS == 2 && z0 != 0) { // (wave-uniform) the reference's repeated additions up to this task's first voxel
#pragma unroll
                for (int sidx = 0; sidx < ZH; ++sidx) {
                    XY += INC;
                    Z += inc2;
                }
            }
            // K is consumed: only now issue the scalar loads of the next frame's constants (scalar loads return out of order, a
            // wait is always "for all"), so that their latency hides behind this frame's arithmetic
            __builtin_amdgcn_sched_barrier(0);
            Knext = hv_sweep_frame_k(Ps, rest ? __ffsll((long long)rest) - 1 : f);
            __builtin_amdgcn_sched_barrier(0);
        };
        bool frame_checked = true; // (wave-uniform) false: the frame is INTERIOR for this item, no voxel can leave the image
        auto project = [&](Group &g) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < GV; ++k) {
                // (a0, a1) / pc2, correctly rounded, sharing one refined reciprocal (hv_div2's chain on a float2)
                float r = __builtin_amdgcn_rcpf(Z);
                const float e = fmaf(-Z, r, 1.0f);
                r = fmaf(e, r, r);
                const hv_f2 A = XY * F;
                const hv_f2 R = hv_splat(r), NZ = hv_splat(-Z);
                hv_f2 Q = A * R;
                hv_f2 REM = hv_fma2(NZ, Q, A);
                Q = hv_fma2(REM, R, Q);
                if (HV_SWEEP_DIV_ROUNDS == 2) { // (hv_div2's second correction round: never needed, see HV_SWEEP_DIV_ROUNDS)
                    REM = hv_fma2(NZ, Q, A);
                    Q = hv_fma2(REM, R, Q);
                }
                const hv_f2 UV = (Q + C) + hv_splat(0.5f);
                bool in = true;
                // a scalar branch around four vector instructions, kept a branch by the empty asm.  (The same choice per gather group - two
                // copies of this loop - costs two more spilled registers and 5 % of the sweep; per frame - two copies of the frame body - 171.)
                if (!INTERIOR || frame_checked) {
                    if (INTERIOR) asm volatile("" ::: "memory");
                    const bool in_u = (__float_as_uint(UV.x) - lo_u) < lim_u;
                    const bool in_v = (__float_as_uint(UV.y) - lo_v) < lim_v;
                    in = (int)in_u & (int)in_v;
                }
                g.inimg[k] = in;
                const uint32_t u = (uint32_t)(int)UV.x, v = (uint32_t)(int)UV.y; // saturating conversions: garbage lanes stay defined
                const uint32_t off = __umul24(v, W24) + u; // exact for every in-image pixel; a garbage lane reads 0 or some pixel, unused
                g.rec[k] = __builtin_bit_cast(hv_u3, __builtin_amdgcn_raw_buffer_load_b96(rs_px, (int)__umul24(off, 12u), 0, 0));
                g.zk[k] = Z;
                XY += INC;
                Z += inc2;
            }
        };
        auto fold = [&](const Group &g, const int gi) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < GV; ++k) {
                const int z = gi * GV + k;
                const float sdf = (__uint_as_float(g.rec[k].x) - g.zk[k]) * __uint_as_float(g.rec[k].z);
                const bool ok = (int)g.inimg[k] & (int)(sdf > ntrunc);
                if (ANYSKIP == 2) {
                    if (ok) {
                        asm volatile("" ::: "memory"); // keeps the branch: if-converted, the body is the select form again
                        S[z] += fminf(sdf * tinv, 1.0f); // == `if (t > 1) t = 1` for the non-NaN t of an accepted voxel
                        arb[z] += g.rec[k].y & 0x00ff00ffu;
                        agn[z] += g.rec[k].y & 0xff00ff00u;
                    }
                    continue;
                }
                if (ANYSKIP && !__any(ok)) continue; // no lane of the wave updates its voxel at this z
                const float tk = fminf(sdf * tinv, 1.0f);
                S[z] += ok ? tk : 0.0f;
                const uint32_t c = ok ? g.rec[k].y : 0u;
                arb[z] += c & 0x00ff00ffu;
                agn[z] += c & 0xff00ff00u;
            }
        };
        HvSweepFrameK ka = hv_sweep_frame_k(Ps, __ffsll((long long)mask) - 1), kb = ka;
        if (PIPE == 2) {
            // gather groups pipelined ACROSS frames: the first group of frame n+1 is projected (and its gathers issued) before
            // the last group of frame n is folded
            static_assert(PIPE != 2 || NG % 2 == 0, "two gather groups alternate");
            Group ga, gb;
            {
                const int f0 = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                begin_frame(ka, f0, kb, mask);
                project(ga);
            }
            // one frame whose group 0 is in flight in ga; K_next: the constants of the frame after it (already requested)
            auto frame_body = [&](const HvSweepFrameK K_next, HvSweepFrameK &K_after) __attribute__((always_inline)) {
                bool more = false;
#pragma unroll
                for (int gi = 0; gi < NG; gi += 2) {
                    project(gb);
                    __builtin_amdgcn_sched_barrier(0);
                    fold(ga, gi);
                    __builtin_amdgcn_sched_barrier(0);
                    if (gi + 2 < NG) {
                        project(ga);
                    } else {
                        more = mask != 0ull;
                        if (more) {
                            const int fn = __ffsll((long long)mask) - 1;
                            mask &= mask - 1;
                            begin_frame(K_next, fn, K_after, mask);
                            project(ga);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    fold(gb, gi + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                return more;
            };
            while (true) {
                if (!frame_body(kb, ka)) break;
                if (!frame_body(ka, kb)) break;
            }
        } else {
            auto fold_frame = [&](const HvSweepFrameK K, const int f, HvSweepFrameK &Knext, const unsigned long long rest) __attribute__((always_inline)) {
                begin_frame(K, f, Knext, rest);
                if (INTERIOR) frame_checked = !((interior_mask >> f) & 1ull);
                if (PIPE == 1) {
                    // the gathers of group g+1 are issued before group g is folded; the pipeline drains at the end of a frame
                    Group ga, gb;
                    project(ga);
#pragma unroll
                    for (int gi = 0; gi < NG; gi += 2) {
                        project(gb);
                        __builtin_amdgcn_sched_barrier(0);
                        fold(ga, gi);
                        __builtin_amdgcn_sched_barrier(0);
                        if (gi + 2 < NG) project(ga);
                        __builtin_amdgcn_sched_barrier(0);
                        fold(gb, gi + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int gi = 0; gi < NG; ++gi) {
                        Group ga;
                        project(ga);
                        fold(ga, gi);
                    }
                }
            };
            while (true) {
                const int fa = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                fold_frame(ka, fa, kb, mask);
                if (!mask) break;
                const int fb = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                fold_frame(kb, fb, ka, mask);
                if (!mask) break;
            }
        }
        // one running-mean step per voxel for the whole batch, four voxels of the column at a time
#pragma unroll
        for (int z4 = 0; z4 < ZH; z4 += 4) {
            float vt[4];
            uint32_t vw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (agn[z4 + k] >> 24) {
                    const int q = wordb + (z4 + k) * RR;
                    vt[k] = ((const float *)(unit + 0 * PLANE_BYTES))[q];
                    vw[k] = ((const uint32_t *)(unit + 1 * PLANE_BYTES))[q];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int z = z4 + k;
                const uint32_t n = agn[z] >> 24;
                if (n) {
                    const int q = wordb + z * RR;
                    const uint32_t nw = vw[k] + n;
                    ((float *)(unit + 0 * PLANE_BYTES))[q] = (vt[k] * (float)vw[k] + S[z]) / (float)nw;
                    ((uint32_t *)(unit + 1 * PLANE_BYTES))[q] = nw;
                    ((uint32_t *)(unit + 2 * PLANE_BYTES))[q] += arb[z] & 0xffffu;
                    ((uint32_t *)(unit + 3 * PLANE_BYTES))[q] += (agn[z] >> 8) & 0xffffu;
                    ((uint32_t *)(unit + 4 * PLANE_BYTES))[q] += arb[z] >> 16;
                }
            }
        }
    };
    drive(run_item);
}

// The column sweep on its own: a grid-stride loop over the batch's union list.
template <int SPLIT, int GV, int PIPE, int ANYSKIP, int ZS = 1>
__device__ __forceinline__ void hv_sweep_column_body(
    HvTable table, const int32_t *__restrict__ list, const unsigned long long *__restrict__ frame_mask,
    char *__restrict__ pool, const uint2 *__restrict__ frame_px, const HvFrameParams *__restrict__ Ps, int n_frames,
    int general, const float *__restrict__ mult, int xcd_aware, int parity) {
    constexpr int PARTS = SPLIT * ZS; // work items per unit
    int n_units = table.counters[HV_CNT_TOUCH(parity)];
    if (n_units > table.max_blocks) n_units = table.max_blocks;
    hv_sweep_column_core<SPLIT, GV, PIPE, ANYSKIP, ZS>(table, frame_mask, pool, frame_px, Ps, n_frames, general, mult, [&](auto &&run_item) __attribute__((always_inline)) {
        const int G = xcd_aware > 0 ? xcd_aware : 1;
        const int rounds = (n_units + 8 * G - 1) / (8 * G);
        const int n_items = xcd_aware > 0 ? rounds * 8 * G * PARTS : n_units * PARTS;
        for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
            int t, part;
            if (xcd_aware > 0) {
                const int xcd = item & 7, j = item >> 3;
                const int g = j / (G * PARTS), within = j - g * (G * PARTS);
                t = (g * 8 + xcd) * G + within / PARTS;
                part = within % PARTS;
                if (t >= n_units) continue;
            } else {
                t = item / PARTS;
                part = item % PARTS;
            }
            run_item(list[t], part);
        }
    });
}

// The kernels proper.  WPE = the waves per SIMD the register allocation is capped for (4: 128 VGPRs).  The _v120 / _v112
// entries cap the allocation at 120 / 112 registers instead (amdgpu_num_vgpr counts half of the unified file on gfx950): four
// sweep waves then leave 32 / 64 registers of every SIMD free, enough for waves of the NEXT batch's touch + pack launch
// (56 VGPRs) to be resident beside them - without that, the second queue only gets a wave slot when a sweep wave retires
// (profiles/r02/pipeline_timeline.txt), which is what an 8-rank share cannot afford.
template <int SPLIT, int WPE, int GV, int PIPE, int ANYSKIP, int ZS = 1>
__global__ __launch_bounds__(64 * 4 / SPLIT, WPE) void k_tsdf_sweep_column(
    HvTable table, const int32_t *__restrict__ list, const unsigned long long *__restrict__ frame_mask,
    char *__restrict__ pool, const uint2 *__restrict__ frame_px, const HvFrameParams *__restrict__ Ps, int n_frames,
    int general, const float *__restrict__ mult, int xcd_aware, int parity) {
    hv_sweep_column_body<SPLIT, GV, PIPE, ANYSKIP, ZS>(table, list, frame_mask, pool, frame_px, Ps, n_frames, general, mult, xcd_aware, parity);
}
#define HV_SWEEP_COLUMN_CAPPED(NAME, HALF_VGPRS)                                                                        \
    template <int SPLIT, int GV, int PIPE, int ANYSKIP>                                                                 \
    __global__ __launch_bounds__(64 * 4 / SPLIT) __attribute__((amdgpu_num_vgpr(HALF_VGPRS))) void NAME(               \
        HvTable table, const int32_t *__restrict__ list, const unsigned long long *__restrict__ frame_mask,             \
        char *__restrict__ pool, const uint2 *__restrict__ frame_px, const HvFrameParams *__restrict__ Ps, int n_frames, \
        int general, const float *__restrict__ mult, int xcd_aware, int parity) {                                       \
        hv_sweep_column_body<SPLIT, GV, PIPE, ANYSKIP>(table, list, frame_mask, pool, frame_px, Ps, n_frames, general, mult,  \
                                                       xcd_aware, parity);                                              \
    }
HV_SWEEP_COLUMN_CAPPED(k_tsdf_sweep_column_v120, 60)
HV_SWEEP_COLUMN_CAPPED(k_tsdf_sweep_column_v112, 56)
#undef HV_SWEEP_COLUMN_CAPPED

// ================================================================================================
// k_tsdf_fused (round 5): the column sweep of batch k-1 AND the touch + pack pass of batch k in ONE launch.
//
// Rounds 2-4 ran the touch + pack launch of the next batch on a second stream beside the sweep.  The sweep's waves hold every VGPR
// of a SIMD (4 x 128), so the other queue only got a wave slot when a sweep wave retired: the launch took the whole sweep to finish,
// ended ~37 us AFTER it (profiles/r04/pipeline_timeline.txt), and at an 8-GPU share (sweep 85 us, touch + pack 60 us) a step was
// the SUM of the two launches (profiles/r04/rank8_timeline.txt: 153 us), which is what kept the projected scaling at 4.5x.  Here
// both kinds of work are items of one grid, interleaved evenly (Bresenham) in dispatch order: the streaming / latency-chain waves
// of the next batch take their slots among the VALU-bound sweep waves by construction, not by the dispatcher's mercy.
//   item kinds (one 64-lane workgroup each): sweep (unit, part) of batch k-1 | touch: one 8x8 sample patch of a frame of batch k |
//   pack: 1024 pixels of a frame of batch k.
//   XCD-aware as the sweep was: workgroup b runs on XCD b & 7; all parts of a unit and G list neighbours share an XCD's L2.
// What used to be k_tsdf_batch_finish (a launch + two dependent-launch gaps per batch) is the sweep's epilogue: the LAST part of
// a unit to finish clears the unit's frame mask (per-slot counter `done`), and the last item of the launch that matters for it
// (sweep items: the union list is consumed; touch items: the pool claims are made) zeroes the batch's list counter and
// publishes the pool status.  Frame constants: Ps_sweep / Ps_aux live in a device ring filled by k_upload_words on a stream of
// its own, a batch ahead (hv_tsdf.hip: tsdf_integrate_batch_impl).  Host side: the sweep of the batch handed over by call k is
// launched by call k+1 (with that call's touch + pack) or by hv_tsdf_flush - which every other entry point runs first.
// ================================================================================================
struct HvFusedSweep {             // batch k-1 (n_frames == 0: no sweep in this launch)
    const int32_t *list;          // union list of the batch (scratch set `parity`)
    unsigned long long *mask;     // [table capacity] frame masks of that set: read, then cleared by the unit's last part
    int32_t *done;                // [table capacity] parts of the unit finished in this launch (back to 0 with the last one)
    const uint2 *px;              // 12-byte frame records of the batch
    const HvFrameParams *Ps;
    int32_t n_frames, parity;
};
struct HvFusedAux {               // batch k (n_frames == 0: no touch + pack in this launch)
    unsigned long long *mask;     // frame masks of the batch's scratch set
    int32_t *stamp, *list;
    const char *depth;            // frames back to back, depth_stride bytes apart
    const uint8_t *rgb;
    uint2 *px;                    // records to write
    const HvFrameParams *Ps;
    const float *mult;            // per-pixel multiplier table (copied into the records)
    int64_t depth_stride;
    int32_t n_frames, parity, batch_stamp;
    int32_t touch_per_frame, pack_per_frame; // items per frame: 8x8 sample patches / 1024-pixel chunks
};

// The kernel's arguments are passed ONE BY ONE, not as this struct (with a struct argument the sweep's register allocation loses
// 30 accumulators to scratch: the compiler then reads the fields through a reference into the kernarg segment instead of preloading
// them).  The struct MIRRORS the argument list - same order, same natural alignment = the layout of the kernarg segment - so that
// the epilogue and the touch + pack role can re-read arguments from the segment (hv_fused_kernarg) instead of keeping them live.
struct HvFusedArgs {
    HvTable table;
    char *pool;
    HvFusedSweep S;
    HvFusedAux A;
    HvStatus *status;
    int32_t status_seq, general, xcd_g, lead;
};
#define HV_FUSED_PARAMS                                                                                                            \
    HvTable table, char *__restrict__ pool, const int32_t *__restrict__ s_list, unsigned long long *__restrict__ s_mask,           \
        int32_t *__restrict__ s_done, const uint2 *__restrict__ s_px, const HvFrameParams *__restrict__ s_Ps, int32_t s_n_frames,  \
        int32_t s_parity, HvFusedAux A, HvStatus *status, int32_t status_seq, int32_t general, int32_t xcd_g, int32_t lead
#define HV_FUSED_ARGS(K)                                                                                                           \
    (K).table, (K).pool, (K).S.list, (K).S.mask, (K).S.done, (K).S.px, (K).S.Ps, (K).S.n_frames, (K).S.parity, (K).A, (K).status,  \
        (K).status_seq, (K).general, (K).xcd_g, (K).lead
static_assert(offsetof(HvFusedArgs, S) == sizeof(HvTable) + 8 && sizeof(HvFusedSweep) == 48 && offsetof(HvFusedArgs, A) % 8 == 0 &&
                  offsetof(HvFusedArgs, status) == offsetof(HvFusedArgs, A) + sizeof(HvFusedAux),
              "HvFusedArgs must have the layout of the kernel's argument list");

// What k_tsdf_batch_finish did, by whichever item of the launch finishes last (lane 0 of the item's wave calls this).  The
// launch's arguments are read AGAIN from the kernarg segment, through a pointer laundered by an empty asm so that the scalar loads
// are not merged with the kernel's own: nothing this needs stays live in registers over a sweep item (the sweep sits at the 128
// VGPRs / 102 SGPRs of 4 waves per SIMD; spilled SGPRs take a VGPR for their lanes, and that one register costs the sweep its
// accumulators).
typedef const __attribute__((address_space(4))) HvFusedArgs *HvFusedKernargPtr;
__device__ __forceinline__ HvFusedKernargPtr hv_fused_kernarg() {
    HvFusedKernargPtr kp = (HvFusedKernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}

// items of XCD x the epilogue waits for: the parts of its units (unit t belongs to XCD (t / G) & 7) + its touch items (a & 7 == x)
template <int PARTS>
__device__ __forceinline__ int hv_fused_expected_xcd(const int n_units, const int G, const int n_touch, const int x) {
    const int full = n_units / (8 * G), rem = n_units - full * 8 * G;
    const int units_x = full * G + min(max(rem - x * G, 0), G);
    const int touch_x = n_touch > x ? (n_touch - x + 7) >> 3 : 0;
    return units_x * PARTS + touch_x;
}

// Completion is counted per XCD first (one counter line per XCD, HV_CNT_SWEEP_DONE_XCD), then over the XCDs: 50 k returning atomics
// per launch on ONE address were the launch's critical path.
template <int PARTS>
__device__ __forceinline__ void hv_fused_item_done(const int xcd) {
    HvFusedKernargPtr kp = hv_fused_kernarg();
    int32_t *counters = kp->table.counters;
    const int32_t s_frames = kp->S.n_frames, s_parity = kp->S.parity;
    int n_units = 0;
    if (s_frames > 0) n_units = min(counters[HV_CNT_TOUCH(s_parity)], kp->table.max_blocks); // (stable during the launch, on a line nobody writes)
    const int n_touch = kp->A.touch_per_frame * kp->A.n_frames;
    const int xg = kp->xcd_g;
    const int G = xg > 0 ? xg : 1;
    const int32_t gx = atomicAdd(&counters[HV_CNT_SWEEP_DONE_XCD + 32 * xcd], 1);
    if (gx != hv_fused_expected_xcd<PARTS>(n_units, G, n_touch, xcd) - 1) return;
    counters[HV_CNT_SWEEP_DONE_XCD + 32 * xcd] = 0;
    int live = 0; // XCDs that have items at all
#pragma unroll
    for (int x = 0; x < 8; ++x) live += hv_fused_expected_xcd<PARTS>(n_units, G, n_touch, x) > 0 ? 1 : 0;
    const int32_t g = atomicAdd(&counters[HV_CNT_SWEEP_DONE], 1);
    if (g != live - 1) return;
    counters[HV_CNT_SWEEP_DONE] = 0;
    if (s_frames > 0) counters[HV_CNT_TOUCH(s_parity)] = 0; // the list is consumed (the other set's may be filling)
    // pool occupancy after this launch's claims, for hv_capacity_gate.  L1-bypassing loads: the claims were made by atomics of
    // other workgroups
    volatile HvStatus *st = kp->status;
    if (st == nullptr) return; // (a claim pass that is verified synchronously: hv_claims_fit reads the counters itself)
    st->blocks = __hip_atomic_load(&counters[HV_CNT_BLOCKS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st->overflow = __hip_atomic_load(&counters[HV_CNT_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    st->seq = kp->status_seq;
}

// the unit's last part clears its frame mask for the scratch set's next batch (every part read the mask before it came here)
template <int PARTS>
__device__ __forceinline__ void hv_fused_unit_part_done(const int32_t slot) {
    HvFusedKernargPtr kp = hv_fused_kernarg();
    int32_t *done = kp->S.done;
    const int32_t before = atomicAdd(&done[slot], 1);
    if (before == PARTS - 1) {
        done[slot] = 0;
        kp->S.mask[slot] = 0ull;
    }
}

// The touch + pack role of a k_tsdf_fused workgroup: aux items q, q + naw, ... of this XCD (item a = q * 8 + xcd; touch items
// first - latency chains: hash probes, atomics - then the pack items).  Reads the launch's arguments from the kernarg segment.
template <int PARTS>
__device__ __forceinline__ void hv_fused_aux_role(int q0, const int naw, const int na, const int xcd, const int n_touch, const int n_pack) {
    __shared__ HvTouchScratch scratch;
    // The frame constants are read with VECTOR loads (the index goes through a VGPR the compiler cannot see through): as scalars
    // the touch role's 40-odd doubles do not fit beside the kernel's own, and spilled scalars take a VGPR of the WHOLE kernel for
    // their lanes - the one register the sweep's accumulators cannot spare at 128.
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    // (the launch's arguments, field by field through the kernarg pointer: scalar loads)
    HvFusedKernargPtr kp = hv_fused_kernarg();
    struct {
        HvTable table;
    } K;
    K.table.keys = kp->table.keys;
    K.table.vals = kp->table.vals;
    K.table.block_keys = kp->table.block_keys;
    K.table.counters = kp->table.counters;
    K.table.mask = kp->table.mask;
    K.table.max_blocks = kp->table.max_blocks;
    K.table.prob_nodes = nullptr;
    K.table.prob_node_cap = 0;
    HvFusedAux A;
    A.mask = kp->A.mask;
    A.stamp = kp->A.stamp;
    A.list = kp->A.list;
    A.depth = kp->A.depth;
    A.rgb = kp->A.rgb;
    A.px = kp->A.px;
    A.Ps = kp->A.Ps;
    A.mult = kp->A.mult;
    A.depth_stride = kp->A.depth_stride;
    A.n_frames = kp->A.n_frames;
    A.parity = kp->A.parity;
    A.batch_stamp = kp->A.batch_stamp;
    A.touch_per_frame = kp->A.touch_per_frame;
    A.pack_per_frame = kp->A.pack_per_frame;
    const int lane = threadIdx.x & 63;
    for (int q = q0; q < na; q += naw) {
        const int a = q * 8 + xcd;
        if (a < n_touch) {
            const int f = a / A.touch_per_frame, patch = a - f * A.touch_per_frame;
            const HvFrameParams &P = A.Ps[f + vzero];
            if (patch < hv_touch_patches(P))
                hv_touch_batch_patch(K.table, P, A.depth + (int64_t)f * A.depth_stride, patch, scratch, f, A.mask, A.stamp, A.list,
                                     A.batch_stamp, A.parity);
            // (the claims of this wave's lanes are returning atomics: performed before this one is issued)
            if (lane == 0) hv_fused_item_done<PARTS>(xcd);
        } else if (a < n_touch + n_pack) {
            const int c = a - n_touch;
            const int f = c / A.pack_per_frame, chunk = c - f * A.pack_per_frame;
            const HvFrameParams &P = A.Ps[f + vzero];
            const int64_t npx = (int64_t)P.H * P.W;
            const void *depth_f = A.depth + (int64_t)f * A.depth_stride;
            const uint8_t *rgb_f = A.rgb + (int64_t)f * npx * 3;
#pragma unroll
            for (int it = 0; it < 4; ++it) // (unrolled: the four sub-chunks' loads are in flight together - a pack wave holds a 128-VGPR slot)
                hv_pack_px4(P, f, ((int64_t)chunk * 1024 + it * 256 + lane * 4), depth_f, rgb_f, A.px, A.mult, nullptr);
        }
    }
}

// Which items a workgroup of k_tsdf_fused runs.  A workgroup has ONE role for its whole life (the register allocation of the sweep
// then is the stand-alone kernel's: in a loop that switched roles per item the touch + pack code's live values pushed the sweep's
// accumulators into scratch).  The first Lw workgroups of an XCD are dealt the roles in the proportion ns : na, evenly interleaved
// in dispatch order (Bresenham); each class then strides over its own items.
struct HvFusedRole {
    int n_units, ns, na, n_touch, n_pack; // sweep / aux items per XCD, touch / pack items of the launch
    int G, xcd, rank, stride;             // this workgroup: its first item and stride inside its class
    bool active, sweep;
};
template <int PARTS>
__device__ __forceinline__ HvFusedRole hv_fused_role(const HvTable &table, const int s_n_frames, const int s_parity, const HvFusedAux &A, const int xcd_g,
                                                     const int lead) {
    HvFusedRole r;
    r.n_units = 0;
    if (s_n_frames > 0) r.n_units = min(table.counters[HV_CNT_TOUCH(s_parity)], table.max_blocks);
    r.G = xcd_g > 0 ? xcd_g : 1;
    const int rounds = (r.n_units + 8 * r.G - 1) / (8 * r.G);
    r.ns = rounds * r.G * PARTS; // (the list padded to whole rounds of 8 XCDs x G units)
    r.n_touch = A.touch_per_frame * A.n_frames;
    r.n_pack = A.pack_per_frame * A.n_frames;
    r.na = (r.n_touch + r.n_pack + 7) >> 3;
    const int per = r.ns + r.na;
    r.xcd = blockIdx.x & 7;
    const int l = blockIdx.x >> 3;
    const int Lw = min(per, (int)(gridDim.x >> 3));
    r.active = l < Lw;
    r.sweep = false;
    r.rank = r.stride = 0;
    if (!r.active) return r;
    // sweep workgroups of this XCD: the share ns / per of the Lw (any split serves; float arithmetic and 32-bit divisions only -
    // a 64-bit division is expanded on the vector unit and leaves these wave-uniform values, and with them the list index, the unit's
    // slot and address and the frame mask of every sweep item, in VECTOR registers: 60 of the sweep's accumulators went to scratch)
    int nsw = r.ns > 0 ? max(1, (int)((float)Lw * ((float)r.ns / (float)per))) : 0;
    if (nsw > Lw) nsw = Lw;
    if (r.na > 0 && nsw >= Lw) nsw = Lw - 1;
    // `lead`: the first workgroups of an XCD are ALL sweep items (a sweep item lives ~10x as long as a touch / pack item: dealt evenly
    // from the start, the sweep items trickle into the machine behind short-lived neighbours and the launch ends on the sweep's
    // second wave; with the machine's 512 wave slots per XCD filled by sweep items first, the touch + pack items fill what the
    // later sweep items leave free)
    const int ld = min(max(lead, 0), nsw);
    int rank_s;
    if (l < ld) {
        rank_s = l;
        r.sweep = true;
    } else {
        const uint32_t l2 = (uint32_t)(l - ld), n2 = (uint32_t)(nsw - ld), L2 = (uint32_t)(Lw - ld); // < 2^13 each (gridDim.x >> 3 <= 8192 by launch)
        const int q = (int)((l2 * n2) / L2);
        r.sweep = (int)(((l2 + 1) * n2) / L2) > q;
        rank_s = ld + q;
    }
    r.rank = __builtin_amdgcn_readfirstlane(r.sweep ? rank_s : l - rank_s);
    r.stride = __builtin_amdgcn_readfirstlane(r.sweep ? nsw : Lw - nsw);
    r.ns = __builtin_amdgcn_readfirstlane(r.ns);
    r.na = __builtin_amdgcn_readfirstlane(r.na);
    r.n_units = __builtin_amdgcn_readfirstlane(r.n_units);
    return r;
}

template <int ZS>
__global__ __launch_bounds__(64, 4) void k_tsdf_fused(HV_FUSED_PARAMS) {
    constexpr int PARTS = 4 * ZS;
    const int lane = threadIdx.x & 63;
    const HvFusedRole r = hv_fused_role<PARTS>(table, s_n_frames, s_parity, A, xcd_g, lead);
    if (r.n_units * PARTS + r.n_touch == 0) { // (nothing the epilogue would wait for: an empty sweep without a next batch)
        if (blockIdx.x == 0 && lane == 0 && status != nullptr) hv_publish_status(table, status, status_seq);
    }
    if (!r.active) return;
    if (!r.sweep) {
        hv_fused_aux_role<PARTS>(r.rank, r.stride, r.na, r.xcd, r.n_touch, r.n_pack);
        return;
    }
    hv_sweep_column_core<4, 4, 1, 2, ZS>(table, s_mask, pool, s_px, s_Ps, s_n_frames, general, (const float *)nullptr, [&](auto &&run_item) __attribute__((always_inline)) {
        for (int j = r.rank; j < r.ns; j += r.stride) {
            const int g = j / (r.G * PARTS), within = j - g * (r.G * PARTS);
            const int t = (g * 8 + r.xcd) * r.G + within / PARTS;
            if (t >= r.n_units) continue;
            const int32_t slot = s_list[t];
            run_item(slot, within % PARTS);
            if (lane == 0) {
                hv_fused_unit_part_done<PARTS>(slot);
                hv_fused_item_done<PARTS>(r.xcd);
            }
        }
    });
}

// The batch's union list in order of DECREASING work (frames that see the unit = set bits of its frame mask): a counting sort by one
// workgroup between the touch pass and the sweep, on the touch pass's stream.  The sweep deals its items out in list order; a unit seen
// by all 32 frames is a wave task 10 - 30 x as long as one seen by a single frame, and with the long ones first the short ones fill
// the launch's tail (longest-processing-time-first: what matters at an 8-GPU share, where 6 800 tasks meet 4 096 wave slots).
__global__ __launch_bounds__(1024) void k_tsdf_list_by_work(HvTable table, const int32_t *__restrict__ list,
                                                            const unsigned long long *__restrict__ frame_mask,
                                                            int32_t *__restrict__ sorted, int parity) {
    __shared__ int32_t s_bin[65]; // bin b = 64 - popcount: the longest tasks first
    int n = table.counters[HV_CNT_TOUCH(parity)];
    if (n > table.max_blocks) n = table.max_blocks;
    for (int i = threadIdx.x; i < 65; i += blockDim.x) s_bin[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&s_bin[64 - __popcll(frame_mask[list[i]])], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int at = 0;
        for (int b = 0; b < 65; ++b) {
            const int c = s_bin[b];
            s_bin[b] = at;
            at += c;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int32_t slot = list[i];
        sorted[atomicAdd(&s_bin[64 - __popcll(frame_mask[slot])], 1)] = slot;
    }
}

// n16 16-byte words from device-visible host memory to device memory, one workgroup (see hv_tsdf_integrate_batch: frame constants).
__global__ __launch_bounds__(256) void k_upload_words(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16) {
    for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i]; // (one workgroup: 3-4 PCIe round trips for a 32-frame batch)
}

// After the sweep (one workgroup): clear the frame masks of the batch's units and zero the batch's touched-list counter, so
// that the next batch / online frame starts clean without a memset launch per counter.
__global__ __launch_bounds__(1024) void k_tsdf_batch_finish(HvTable table, const int32_t *__restrict__ list,
                                                             unsigned long long *__restrict__ frame_mask, int parity,
                                                             HvStatus *status, int32_t status_seq) {
    int n_units = table.counters[HV_CNT_TOUCH(parity)];
    if (n_units > table.max_blocks) n_units = table.max_blocks;
    __syncthreads(); // every thread holds n_units before the counters are reset
    // 8 independent list loads in flight per thread, then the 8 stores: two memory latencies per 8192 units instead
    // of two per 1024
    for (int base = 0; base < n_units; base += 8 * (int)blockDim.x) {
        int32_t slot[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = base + k * (int)blockDim.x + (int)threadIdx.x;
            slot[k] = t < n_units ? list[t] : -1;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (slot[k] >= 0) frame_mask[slot[k]] = 0ull;
    }
    if (threadIdx.x == 0) {
        table.counters[HV_CNT_TOUCH(parity)] = 0; // (the other set's counter may be filling: the next batch's touch pass)
        hv_publish_status(table, status, status_seq); // pool occupancy after this batch, for hv_capacity_gate
    }
}

// ---- numerators export / import (multi-GPU merge) ----------------------------------------------
__global__ void k_tsdf_export(HvTable table, const char *__restrict__ pool, const int32_t *__restrict__ keys,
                              int64_t k, float *__restrict__ payload) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= k * RRR) return;
    const int64_t ui = gid / RRR;
    const int word = (int)(gid % RRR);
    float out[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int32_t kx = keys[ui * 3], ky = keys[ui * 3 + 1], kz = keys[ui * 3 + 2];
    if (hv_key_in_range(kx, ky, kz)) {
        const int32_t slot = hv_table_find(table, hv_pack_key(kx, ky, kz));
        const int32_t idx = slot >= 0 ? table.vals[slot] : -1;
        if (idx >= 0) {
            const char *unit = pool + (int64_t)idx * (PLANE_BYTES * HV_TSDF_PLANES);
            const float tsdf = ((const float *)(unit))[word];
            const uint32_t w = ((const uint32_t *)(unit + PLANE_BYTES))[word];
            out[0] = tsdf * (float)w;
            out[1] = (float)w;
            out[2] = (float)((const uint32_t *)(unit + 2 * PLANE_BYTES))[word];
            out[3] = (float)((const uint32_t *)(unit + 3 * PLANE_BYTES))[word];
            out[4] = (float)((const uint32_t *)(unit + 4 * PLANE_BYTES))[word];
        }
    }
    float *dst = payload + gid * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) dst[c] = out[c];
}

__global__ void k_tsdf_import_claim(HvTable table, const int32_t *__restrict__ keys, int64_t k) {
    const int64_t ui = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ui >= k) return;
    const int32_t kx = keys[ui * 3], ky = keys[ui * 3 + 1], kz = keys[ui * 3 + 2];
    if (!hv_key_in_range(kx, ky, kz)) {
        atomicAdd(&table.counters[HV_CNT_DROPPED], 1);
        return;
    }
    hv_table_insert(table, hv_pack_key(kx, ky, kz));
}

__global__ void k_tsdf_import(HvTable table, char *__restrict__ pool, const int32_t *__restrict__ keys,
                              int64_t k, const float *__restrict__ payload) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= k * RRR) return;
    const int64_t ui = gid / RRR;
    const int word = (int)(gid % RRR);
    const int32_t kx = keys[ui * 3], ky = keys[ui * 3 + 1], kz = keys[ui * 3 + 2];
    if (!hv_key_in_range(kx, ky, kz)) return;
    const int32_t slot = hv_table_find(table, hv_pack_key(kx, ky, kz));
    const int32_t idx = slot >= 0 ? table.vals[slot] : -1;
    if (idx < 0) return;
    const float *src = payload + gid * 5;
    char *unit = pool + (int64_t)idx * (PLANE_BYTES * HV_TSDF_PLANES);
    const float w = src[1];
    ((float *)unit)[word] = w > 0.f ? src[0] / w : 0.f;
    ((uint32_t *)(unit + PLANE_BYTES))[word] = (uint32_t)w;
    ((uint32_t *)(unit + 2 * PLANE_BYTES))[word] = (uint32_t)src[2];
    ((uint32_t *)(unit + 3 * PLANE_BYTES))[word] = (uint32_t)src[3];
    ((uint32_t *)(unit + 4 * PLANE_BYTES))[word] = (uint32_t)src[4];
}

// ---- halo merge (image-tile sharding): units stamped since the last merge, and the unpack ------------
__global__ void k_tsdf_collect_dirty(HvTable table, const int32_t *__restrict__ stamp, int32_t merge_stamp, int32_t n_blocks,
                                     int32_t *__restrict__ keys, int32_t cap) {
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool dirty = false;
    int32_t kx = 0, ky = 0, kz = 0;
    if (idx < n_blocks) {
        const unsigned long long key = table.block_keys[idx];
        const int32_t slot = hv_table_find(table, key);
        dirty = slot >= 0 && stamp[slot] > merge_stamp;
        hv_unpack_key(key, kx, ky, kz);
    }
    const int32_t at = hv_wave_append(&table.counters[HV_CNT_OUT], dirty);
    if (dirty && at < cap) {
        keys[at * 3 + 0] = kx;
        keys[at * 3 + 1] = ky;
        keys[at * 3 + 2] = kz;
    }
}

// action[u]: 0 = leave the unit alone (this GPU does not hold it), 1 = replace its state by the reduced numerators (this GPU
// keeps the unit), 2 = zero it (another GPU keeps it; this one goes on fusing deltas into an empty unit)
__global__ void k_tsdf_halo_unpack(HvTable table, char *__restrict__ pool, const int32_t *__restrict__ keys, int64_t k,
                                   const float *__restrict__ payload, const uint8_t *__restrict__ action) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= k * RRR) return;
    const int64_t ui = gid / RRR;
    const int act = action[ui];
    if (act == 0) return;
    const int word = (int)(gid % RRR);
    const int32_t kx = keys[ui * 3], ky = keys[ui * 3 + 1], kz = keys[ui * 3 + 2];
    if (!hv_key_in_range(kx, ky, kz)) return;
    const int32_t slot = hv_table_find(table, hv_pack_key(kx, ky, kz));
    const int32_t idx = slot >= 0 ? table.vals[slot] : -1;
    if (idx < 0) return;
    char *unit = pool + (int64_t)idx * (PLANE_BYTES * HV_TSDF_PLANES);
    float tsdf = 0.f;
    uint32_t w = 0u, r = 0u, g = 0u, b = 0u;
    if (act == 1) {
        const float *src = payload + gid * 5;
        const float wf = src[1];
        tsdf = wf > 0.f ? src[0] / wf : 0.f;
        w = (uint32_t)wf;
        r = (uint32_t)src[2];
        g = (uint32_t)src[3];
        b = (uint32_t)src[4];
    }
    ((float *)unit)[word] = tsdf;
    ((uint32_t *)(unit + PLANE_BYTES))[word] = w;
    ((uint32_t *)(unit + 2 * PLANE_BYTES))[word] = r;
    ((uint32_t *)(unit + 3 * PLANE_BYTES))[word] = g;
    ((uint32_t *)(unit + 4 * PLANE_BYTES))[word] = b;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int make_frame_params(hv_volume *v, int H, int W, const double *intr, const double *T_cw,
                             double depth_scale, double depth_trunc, int depth_dtype, HvFrameParams *P) {
    memset(P, 0, sizeof(*P));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) P->ext[r * 4 + c] = (float)T_cw[r * 4 + c];
    P->voxel_length_f = (float)v->cfg.voxel_size;
    P->half_voxel_length_f = P->voxel_length_f * 0.5f;
    for (int r = 0; r < 3; ++r) P->ext_scaled_col2[r] = P->ext[r * 4 + 2] * P->voxel_length_f;
    for (int c = 0; c < 4; ++c) {
        P->sweep_k[2 * c] = P->ext[c];
        P->sweep_k[2 * c + 1] = P->ext[4 + c];
    }
    P->sweep_k[8] = P->ext[8];
    P->sweep_k[9] = P->ext[11];
    P->sweep_k[10] = P->ext[9];
    P->sweep_k[11] = P->ext[10];
    for (int r = 0; r < 3; ++r) P->sweep_k[12 + r] = P->ext_scaled_col2[r];
    P->fx = (float)intr[0];
    P->fy = (float)intr[1];
    P->cx = (float)intr[2];
    P->cy = (float)intr[3];
    P->ffl_inv_x = 1.0f / (float)intr[0];
    P->ffl_inv_y = 1.0f / (float)intr[1];
    P->sdf_trunc_f = (float)v->cfg.sdf_trunc;
    P->sdf_trunc_inv_f = 1.0f / P->sdf_trunc_f;
    P->safe_width_f = (float)W - 0.0001f;
    P->safe_height_f = (float)H - 0.0001f;
    P->unit_length = v->cfg.voxel_size * (double)v->cfg.block_size;
    double pose[16];
    hv_invert4x4(T_cw, pose);
    for (int i = 0; i < 12; ++i) P->pose[i] = pose[i];
    P->fx_d = intr[0];
    P->fy_d = intr[1];
    P->cx_d = intr[2];
    P->cy_d = intr[3];
    P->sdf_trunc_d = v->cfg.sdf_trunc;
    P->depth_scale_f = (float)depth_scale;
    P->depth_trunc_d = depth_trunc;
    P->H = H;
    P->W = W;
    P->stride = v->cfg.depth_sampling_stride;
    P->touch_box_bits = v->touch_box_bits;
    P->bgr = v->color_bgr;
    P->depth_is_u16 = depth_dtype == HV_DEPTH_U16;
    const bool whole = v->tile[0] == 0 && v->tile[1] == 0 && v->tile[2] == 0 && v->tile[3] == 0;
    P->tile_u0 = whole ? 0 : v->tile[0];
    P->tile_v0 = whole ? 0 : v->tile[1];
    P->tile_u1 = whole ? W : v->tile[2];
    P->tile_v1 = whole ? H : v->tile[3];
    P->tiled = whole ? 0 : 1;
    P->owner_rank = v->owner_rank;
    P->owner_world = v->owner_world;
    return HV_OK;
}

// Per-frame scratch (single-buffered: the two halves of a frame run back to back on one stream).
// Measured alternative, rejected: prep+touch of frame f+1 on a second stream overlapping the sweep of
// frame f (double-buffered scratch, event hand-offs) ran 6955 vs 7280 frames/s - the cross-stream
// waits and the contention on the sweep cost more than the 16 us they hide.
static inline uint2 *frame_px_of(hv_volume *v, int) { return (uint2 *)v->frame_px; }
static inline int32_t *touched_list_of(hv_volume *v, int) { return v->touched_list; }

// First half of a frame: convert/pack the frame, claim and list the touched units.  (This parity's
// touched counter was zeroed by the previous frame's sweep kernel.)
static int tsdf_launch_touch(hv_volume *v, hipStream_t s, const HvFrameParams &P, int parity, const void *d_depth,
                             const uint8_t *d_rgb) {
    const int64_t npx = (int64_t)P.H * P.W;
    const int n_prep_blocks = (int)((npx + 255) / 256);
    const int ns = ((P.W + P.stride - 1) / P.stride) * ((P.H + P.stride - 1) / P.stride);
    const int n_touch_blocks = (ns * HV_TOUCH_FAN + 255) / 256;
    hipLaunchKernelGGL(k_tsdf_prep_touch, dim3(n_prep_blocks + n_touch_blocks), dim3(256), 0, s, v->table,
                       v->touched_stamp, touched_list_of(v, parity), parity, d_depth, d_rgb, frame_px_of(v, parity), P,
                       n_touch_blocks);
    return HV_OK;
}

// Second half on the volume's stream: sweep the touched units.  Grid: enough workgroups to fill
// 256 CUs; grid-stride over the device-side touched count (no host round trip between launches).
// (Re)build the per-pixel multiplier table when the intrinsics or the image size changed (stream-ordered).
static int tsdf_multiplier_table(hv_volume *v, const HvFrameParams &P) {
    const float key[4] = {P.cx, P.cy, P.ffl_inv_x, P.ffl_inv_y};
    if (v->mult_table != nullptr && v->mult_W == P.W && v->mult_H == P.H && memcmp(key, v->mult_key, sizeof(key)) == 0)
        return HV_OK;
    const size_t npx = (size_t)P.W * P.H;
    void *buf = v->mult_table;
    int rc = hv_ensure_buffer(v, &buf, &v->mult_table_bytes, npx * sizeof(float));
    if (rc != HV_OK) return rc;
    v->mult_table = (float *)buf;
    hipLaunchKernelGGL(k_tsdf_multiplier_table, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, v->stream, P,
                       v->mult_table);
    HV_HIP(hipGetLastError());
    memcpy(v->mult_key, key, sizeof(key));
    v->mult_W = P.W;
    v->mult_H = P.H;
    return HV_OK;
}

static int tsdf_launch_integrate(hv_volume *v, const HvFrameParams &P, int parity) {
    static const int grid_blocks = getenv("HV_TSDF_GRID") ? atoi(getenv("HV_TSDF_GRID")) : 8192; // > touched units of a frame: no second pass per block
    const dim3 grid(grid_blocks), block(256);
    const int32_t *list = touched_list_of(v, parity);
    const uint2 *px = frame_px_of(v, parity);
    char *pool = (char *)v->pool;
    hv_profile_begin(v);
    const float *mult = v->mult_table;
    const int32_t seq = hv_next_status_seq(v);
#define HV_LAUNCH_ONLINE(V) hipLaunchKernelGGL(k_tsdf_integrate<V>, grid, block, 0, v->stream, v->table, list, parity, pool, px, P, mult, v->d_status, seq)
    switch (v->debug_variant) {
    case 1: HV_LAUNCH_ONLINE(1); break;
    case 2: HV_LAUNCH_ONLINE(2); break;
    case 8: HV_LAUNCH_ONLINE(8); break;
    case 9: HV_LAUNCH_ONLINE(9); break;
    default: HV_LAUNCH_ONLINE(0); break;
    }
#undef HV_LAUNCH_ONLINE
    hv_profile_end(v, 0);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

static void tsdf_next_frame(hv_volume *v, HvFrameParams &P, int &parity) {
    if (v->plan_lists_stale) {
        // coherent batches ran before this frame: their plans leave the list counters behind (the main stream has waited for every
        // one of those chains, so this memset is ordered after them)
        (void)hipMemsetAsync(&v->table.counters[HV_CNT_TOUCH0], 0, HV_CNT_TOUCH_SPAN_BYTES, v->stream);
        v->plan_lists_stale = false;
    }
    v->content_version += 1;
    v->frame_counter += 1;
    P.frame_id = v->frame_counter;
    parity = v->frame_counter & 1;
    v->last_touch_parity = parity;
    v->touch_counters_clean = false; // this parity's counter keeps the frame's touched count until the next frame's sweep
}

// Online path: both halves back to back on the volume's stream.  In checked mode (hv_capacity_gate: the pool's headroom is
// not known to cover this frame) the claims of the touch pass are verified before anything is fused: if some did not fit
// the pool grows and the touch pass runs again under a fresh stamp.
static int tsdf_integrate_one(hv_volume *v, const void *d_depth, int depth_dtype, const uint8_t *d_rgb,
                              int H, int W, const double *intr, const double *T_cw, double depth_scale,
                              double depth_trunc) {
    bool checked = false;
    int rc = hv_tsdf_flush(v); // a deferred multi-frame sweep goes first (frames are fused in order)
    if (rc != HV_OK) return rc;
    rc = hv_capacity_gate(v, &checked);
    if (rc != HV_OK) return rc;
    HvFrameParams P;
    make_frame_params(v, H, W, intr, T_cw, depth_scale, depth_trunc, depth_dtype, &P);
    int parity = 0;
    for (int attempt = 0;; ++attempt) {
        tsdf_next_frame(v, P, parity);
        if (attempt > 0) HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_TOUCH0], 0, HV_CNT_TOUCH_SPAN_BYTES, v->stream));
        rc = tsdf_launch_touch(v, v->stream, P, parity, d_depth, d_rgb);
        if (rc != HV_OK) return rc;
        if (!checked) break;
        rc = hv_claims_fit(v);
        if (rc == HV_OK) break;
        if (rc != HV_RETRY_CLAIM || attempt >= 8) return rc == HV_RETRY_CLAIM ? HV_ERR_CAPACITY : rc;
        make_frame_params(v, H, W, intr, T_cw, depth_scale, depth_trunc, depth_dtype, &P); // (touch_box_bits etc. unchanged; cheap)
    }
    if (v->debug_variant == 8) {
        rc = tsdf_multiplier_table(v, P);
        if (rc != HV_OK) return rc;
    }
    return tsdf_launch_integrate(v, P, parity);
}

// ---- fused form: host side (see k_tsdf_fused) ---------------------------------------------------------------------------------
static constexpr int HV_BATCH_MAX = 64; // frames per sweep (one bit per frame in a unit's mask)

static int tsdf_fused_resources(hv_volume *v) {
    if (v->stream_up == nullptr) {
        HV_HIP(hipStreamCreateWithFlags(&v->stream_up, hipStreamNonBlocking));
        for (int i = 0; i < 4; ++i) HV_HIP(hipEventCreateWithFlags(&v->ev_swept[i], hipEventDisableTiming));
    }
    if (v->params_ring == nullptr) HV_HIP(hipMalloc(&v->params_ring, sizeof(HvFrameParams) * HV_BATCH_MAX * 4));
    return HV_OK;
}

// One launch: the sweep of `sweep` (the pending batch, or nullptr) + the touch + pack pass described by `aux` (or nullptr).
static int tsdf_launch_fused(hv_volume *v, bool with_sweep, const HvFusedAux *aux, bool publish = true, const HvFusedSweep *explicit_sweep = nullptr) {
    HvFusedSweep S;
    memset(&S, 0, sizeof(S));
    HvFusedAux A;
    memset(&A, 0, sizeof(A));
    if (aux) A = *aux;
    if (explicit_sweep) { // (the two-stream form's sweep with the finish as its epilogue: not the pending batch)
        S = *explicit_sweep;
    } else if (with_sweep) {
        const int sp = v->pending.parity;
        S.list = v->touched_list + (size_t)sp * (size_t)v->cfg.max_blocks;
        S.mask = (unsigned long long *)v->touched_mask + (size_t)sp * (size_t)v->table_capacity;
        S.px = (const uint2 *)v->pending.px;
        S.Ps = v->pending.params;
        S.n_frames = v->pending.n_frames;
        S.parity = sp;
    } else {
        S.Ps = A.Ps; // (the sweep's frame-independent constants are read at kernel entry: any valid frame will do)
        S.mask = A.mask;
    }
    S.done = v->sweep_done;
    const int general = getenv("HV_TSDF_BATCH_GENERAL") ? atoi(getenv("HV_TSDF_BATCH_GENERAL")) : 0;
    const int xcd_g = getenv("HV_TSDF_SWEEP_XCD") ? atoi(getenv("HV_TSDF_SWEEP_XCD")) : 2;
    const int grid = getenv("HV_TSDF_BATCH_GRID") ? atoi(getenv("HV_TSDF_BATCH_GRID")) : 65536;
    // z halves (8 tasks per unit) when this GPU shares the volume with 3 or more others: DESIGN section 4
    const int zs = getenv("HV_TSDF_SWEEP_ZS") ? atoi(getenv("HV_TSDF_SWEEP_ZS")) : (v->owner_world >= 4 ? 2 : 1);
    HvFusedArgs K;
    memset(&K, 0, sizeof(K));
    K.table = v->table;
    K.pool = (char *)v->pool;
    K.S = S;
    K.A = A;
    // (publish = false: a claim pass whose claims are verified before anything else happens - a published block count that ran past
    // the pool would be taken for the state to roll back to)
    K.status = publish ? v->d_status : nullptr;
    K.status_seq = publish ? hv_next_status_seq(v) : 0;
    K.general = general;
    K.xcd_g = xcd_g;
    K.lead = getenv("HV_TSDF_FUSED_LEAD") ? atoi(getenv("HV_TSDF_FUSED_LEAD")) : 512; // sweep items dealt first per XCD (512 = its wave slots at 4 / SIMD)
    if (with_sweep || explicit_sweep) hv_profile_begin(v);
    if (zs == 2)
        hipLaunchKernelGGL(k_tsdf_fused<2>, dim3(grid), dim3(64), 0, v->stream, HV_FUSED_ARGS(K));
    else
        hipLaunchKernelGGL(k_tsdf_fused<1>, dim3(grid), dim3(64), 0, v->stream, HV_FUSED_ARGS(K));
    if (explicit_sweep) {
        hv_profile_end(v, S.n_frames);
    } else if (with_sweep) {
        hv_profile_end(v, v->pending.n_frames);
        // the frame constants of the swept batch may be overwritten once this launch is done
        HV_HIP(hipEventRecord(v->ev_swept[v->pending.ring], v->stream));
        v->ev_swept_valid[v->pending.ring] = true;
        v->pending.valid = false;
    }
    HV_HIP(hipGetLastError());
    return HV_OK;
}

int hv_tsdf_flush(hv_volume *v) {
    if (v == nullptr || !v->pending.valid) return HV_OK;
    HV_HIP(hipSetDevice(v->device));
    return tsdf_launch_fused(v, true, nullptr);
}

// One chunk (B <= 64 frames resident in HBM) of a multi-frame call in the fused form: its touch + pack pass goes out now, together
// with the sweep of the batch before it; its own sweep is left pending.
static int tsdf_fused_chunk(hv_volume *v, const char *d_depth, size_t depth_frame_bytes, const uint8_t *d_rgb, int B, int height, int width,
                            const double *intr, const double *T_cw, double depth_scale, double depth_trunc, int depth_dtype, int host_set) {
    const size_t npx = (size_t)height * width;
    bool checked = false;
    int rc = hv_capacity_gate(v, &checked); // (a pool that has to grow flushes the pending sweep first: hv_reserve_blocks)
    if (rc != HV_OK) return rc;
    rc = tsdf_fused_resources(v);
    if (rc != HV_OK) return rc;
    // frame constants: pinned ring slot -> device ring slot, uploaded on a stream of its own so that it does not queue behind the
    // launch that is running (the host is usually several batches ahead of the GPU)
    const int ri = v->params_idx;
    v->params_idx = (ri + 1) & 3;
    if (v->pinned_params[ri] == nullptr) {
        HV_HIP(hipHostMalloc(&v->pinned_params[ri], sizeof(HvFrameParams) * HV_BATCH_MAX));
        HV_HIP(hipEventCreateWithFlags(&v->params_ev[ri], hipEventDisableTiming));
    } else {
        HV_HIP(hipEventSynchronize(v->params_ev[ri]));
    }
    HvFrameParams *params = (HvFrameParams *)v->pinned_params[ri];
    for (int f = 0; f < B; ++f) {
        make_frame_params(v, height, width, intr, T_cw + 16 * (size_t)f, depth_scale, depth_trunc, depth_dtype, &params[f]);
        v->frame_counter += 1;
        params[f].frame_id = v->frame_counter;
    }
    int batch_stamp = v->frame_counter;
    {
        // the multiplier table of these intrinsics.  When it has to be (re)built, the pending sweep goes first: its records carry
        // their own multipliers, but a re-allocation of the table synchronises the stream anyway
        const float key[4] = {params[0].cx, params[0].cy, params[0].ffl_inv_x, params[0].ffl_inv_y};
        if (!(v->mult_table != nullptr && v->mult_W == width && v->mult_H == height && memcmp(key, v->mult_key, sizeof(key)) == 0)) {
            rc = hv_tsdf_flush(v);
            if (rc != HV_OK) return rc;
        }
        rc = tsdf_multiplier_table(v, params[0]);
        if (rc != HV_OK) return rc;
    }
    const int parity = v->batch_parity & 1; // (the fused form alternates between two sets)
    v->batch_parity = parity ^ 1;
    v->last_touch_parity = parity;
    // this scratch set's records (its last reader, the sweep of two batches ago, was launched by the previous call; a
    // re-allocation waits for the stream)
    const size_t px_bytes = 12 * npx * (size_t)B;
    void **bb = parity ? &v->batch_buf2 : &v->batch_buf;
    size_t *bb_bytes = parity ? &v->batch_buf2_bytes : &v->batch_buf_bytes;
    rc = hv_ensure_buffer(v, bb, bb_bytes, px_bytes + sizeof(HvFrameParams) * (size_t)B + 256); // (sized as the unfused form's: the forms may alternate)
    if (rc != HV_OK) return rc;
    HvFrameParams *d_params = (HvFrameParams *)v->params_ring + (size_t)ri * HV_BATCH_MAX;
    if (v->ev_swept_valid[ri]) HV_HIP(hipStreamWaitEvent(v->stream_up, v->ev_swept[ri], 0));
    {
        const int n16 = (int)((sizeof(HvFrameParams) * (size_t)B + 15) / 16);
        hipLaunchKernelGGL(k_upload_words, dim3(1), dim3(256), 0, v->stream_up, (const uint4 *)params, (uint4 *)d_params, n16);
    }
    HV_HIP(hipEventRecord(v->params_ev[ri], v->stream_up));
    HV_HIP(hipStreamWaitEvent(v->stream, v->params_ev[ri], 0));
    if (host_set >= 0) HV_HIP(hipStreamWaitEvent(v->stream, v->hs_dev_ready[host_set], 0)); // the frames have arrived
    if (!v->pending.valid && (!v->touch_counters_clean || v->plan_lists_stale)) {
        // (an online frame / a coherent batch left its list length behind; never while a sweep is pending - its list is live)
        HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_TOUCH0], 0, HV_CNT_TOUCH_SPAN_BYTES, v->stream));
        v->touch_counters_clean = true;
        v->plan_lists_stale = false;
    }
    HvFusedAux A;
    memset(&A, 0, sizeof(A));
    auto fill = [&]() {
        A.mask = (unsigned long long *)v->touched_mask + (size_t)parity * (size_t)v->table_capacity;
        A.stamp = v->touched_stamp;
        A.list = v->touched_list + (size_t)parity * (size_t)v->cfg.max_blocks;
        A.depth = d_depth;
        A.rgb = d_rgb;
        A.px = (uint2 *)*bb;
        A.Ps = d_params;
        A.mult = v->mult_table;
        A.depth_stride = (int64_t)depth_frame_bytes;
        A.n_frames = B;
        A.parity = parity;
        A.batch_stamp = batch_stamp;
        A.touch_per_frame = hv_touch_patches(width, height, v->cfg.depth_sampling_stride);
        A.pack_per_frame = (int)((npx + 1023) / 1024);
    };
    fill();
    if (!checked) {
        rc = tsdf_launch_fused(v, v->pending.valid, &A);
        if (rc != HV_OK) return rc;
    } else {
        // checked mode (the pool's headroom is not known to cover this batch): nothing is fused before every unit of the batch has
        // its pool slot.  The pending sweep goes first, then the touch + pack pass alone, verified; if some claims did not fit the
        // pool has grown (tables rebuilt, stamps kept) and the pass runs again under a fresh stamp.
        rc = hv_tsdf_flush(v);
        if (rc != HV_OK) return rc;
        for (int attempt = 0;; ++attempt) {
            rc = tsdf_launch_fused(v, false, &A, false);
            if (rc != HV_OK) return rc;
            rc = hv_claims_fit(v);
            if (rc == HV_OK) break;
            if (rc != HV_RETRY_CLAIM || attempt >= 8) return rc == HV_RETRY_CLAIM ? HV_ERR_CAPACITY : rc;
            v->frame_counter += 1;
            batch_stamp = v->frame_counter;
            fill(); // the tables were rebuilt: the scratch set's arrays moved (their list counters were zeroed)
        }
    }
    v->pending.valid = true;
    v->pending.parity = parity;
    v->pending.n_frames = B;
    v->pending.ring = ri;
    v->pending.px = *bb;
    v->pending.params = d_params;
    return HV_OK;
}

static int check_tsdf_args(hv_volume *v, const void *depth, const uint8_t *rgb, int H, int W,
                           const double *intr, const double *T_cw, int frames) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_tsdf_integrate: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_integrate: volume is not in TSDF mode");
    HV_REQUIRE(depth != nullptr && rgb != nullptr && intr != nullptr && T_cw != nullptr, HV_ERR_INVALID,
               "[ScalableTSDFVolume::Integrate] Unsupported image format.");
    HV_REQUIRE(H > 0 && W > 0 && frames > 0, HV_ERR_INVALID, "hv_tsdf_integrate: empty image");
    HV_REQUIRE((int64_t)H * W <= v->cfg.max_points, HV_ERR_CAPACITY,
               "hv_tsdf_integrate: image %dx%d exceeds max_points=%lld", W, H, (long long)v->cfg.max_points);
    return HV_OK;
}

extern "C" {

int hv_tsdf_integrate(hv_volume *v, const void *depth, int32_t depth_dtype, const uint8_t *rgb,
                      int32_t height, int32_t width, const double *intr, const double *T_cw,
                      double depth_scale, double depth_trunc, int32_t loc) {
    int rc = check_tsdf_args(v, depth, rgb, height, width, intr, T_cw, 1);
    if (rc != HV_OK) return rc;
    HV_HIP(hipSetDevice(v->device));
    const size_t npx = (size_t)height * width;
    const void *d_depth = nullptr, *d_rgb = nullptr;
    rc = hv_stage_in(v, depth, npx * (depth_dtype == HV_DEPTH_U16 ? 2 : 4), loc, 0, &d_depth);
    if (rc != HV_OK) return rc;
    rc = hv_stage_in(v, rgb, npx * 3, loc, 1, &d_rgb);
    if (rc != HV_OK) return rc;
    return tsdf_integrate_one(v, d_depth, depth_dtype, (const uint8_t *)d_rgb, height, width, intr, T_cw,
                              depth_scale, depth_trunc);
}

// depth_ptrs / rgb_ptrs != nullptr: host-resident frames given one pointer per frame (hv_tsdf_integrate_frames); else `depth`
// / `rgb` hold the frames contiguously at `loc`.
static int tsdf_integrate_batch_impl(hv_volume *v, const void *depth, const void *const *depth_ptrs, int32_t depth_dtype,
                                     const uint8_t *rgb, const void *const *rgb_ptrs, int32_t n_frames, int32_t height,
                                     int32_t width, const double *intr, const double *T_cw, double depth_scale,
                                     double depth_trunc, int32_t loc) {
    int rc = HV_OK;
    HV_HIP(hipSetDevice(v->device));
    const size_t npx = (size_t)height * width;
    const size_t dsz = depth_dtype == HV_DEPTH_U16 ? 2 : 4;
    const void *d_depth = nullptr, *d_rgb = nullptr;
    int host_set = -1; // device staging set of host-resident frames (hv_stage_frames)
    if (loc == HV_HOST) {
        // pageable caller memory -> page-locked slots -> DMA on the copy stream, two device sets: the frames of this call
        // cross PCIe while the previous batch is still being swept
        rc = hv_stage_frames(v, depth_ptrs, depth, npx * dsz, rgb_ptrs, rgb, npx * 3, n_frames, &d_depth, &d_rgb, &host_set);
        if (rc != HV_OK) return rc;
    } else {
        d_depth = depth;
        d_rgb = rgb;
    }
    if (v->debug_variant != 0 || n_frames == 1) { // ablation variants / trivial batch: frame by frame
        if (host_set >= 0) HV_HIP(hipStreamWaitEvent(v->stream, v->hs_dev_ready[host_set], 0));
        for (int f = 0; f < n_frames; ++f) {
            rc = tsdf_integrate_one(v, (const char *)d_depth + npx * dsz * f, depth_dtype,
                                    (const uint8_t *)d_rgb + npx * 3 * f, height, width, intr, T_cw + 16 * f,
                                    depth_scale, depth_trunc);
            if (rc != HV_OK) return rc;
        }
        if (host_set >= 0) return hv_stage_frames_consumed(v, host_set, v->stream);
        return HV_OK;
    }
    // multi-frame sweeps of up to 64 frames (one bit per frame in the per-unit mask)
    //
    // Batch pipeline.  A batch is [touch + pack launch: HBM streaming and hash-insert latency chains] -> [sweep: VALU-bound]
    // -> [finish].  When batches follow each other with nothing else done to the volume in between (a replay / rebuild loop),
    // the touch + pack launch of batch k+1 goes to a second stream and runs WHILE batch k is swept: it only inserts units and
    // fills its own scratch set (frame records, union list, frame masks, list counter: two sets, alternating), the sweep of
    // batch k reads none of that.  Hand-offs: the aux stream waits for the event recorded on the main stream just before the
    // PREVIOUS sweep (everything older than that sweep is done - in particular the batch that last used this scratch set);
    // the main stream waits for this batch's touch + pack before its own sweep.  Every other entry point works on the main
    // stream behind those waits and never meets the aux stream; a call that finds the volume touched by anything else since
    // the previous batch (content_version), host-resident frames, the checked capacity mode or HV_TSDF_PIPELINE=0 runs
    // everything on the main stream as before.
    const bool pipeline_on = !(getenv("HV_TSDF_PIPELINE") && atoi(getenv("HV_TSDF_PIPELINE")) == 0);
    // image-coherent ownership (hv_tsdf_set_sharding): the batch is planned on the device, see k_tsdf_touch_plan
    const bool coherent = v->owner_world > 1 && v->shard_coherent != 0;
    if (coherent) {
        const size_t cap = (size_t)v->table_capacity;
        const size_t set_bytes = cap * 16 + sizeof(uint32_t) * HV_PLAN_BINS + sizeof(int4) * 64;
        if (v->plan_buf == nullptr || v->plan_cap != cap) {
            rc = hv_ensure_buffer(v, &v->plan_buf, &v->plan_buf_bytes, 2 * set_bytes + 512);
            if (rc != HV_OK) return rc;
            if (v->stream_aux) HV_HIP(hipStreamSynchronize(v->stream_aux));
            HV_HIP(hipMemsetAsync(v->plan_buf, 0, v->plan_buf_bytes, v->stream));
            for (int k = 0; k < 2; ++k) HV_HIP(hipMemsetAsync((char *)v->plan_buf + k * set_bytes, 0xFF, cap * 8, v->stream));
            HV_HIP(hipStreamSynchronize(v->stream));
            v->plan_cap = cap;
        }
    }
    // sweep form: 4 = k_tsdf_sweep_column (production: the batch folded per voxel, a lane walks a whole voxel column), 3 =
    // k_tsdf_sweep_fold (the fold on 4 voxels per lane), 2 = k_tsdf_sweep (the reference's running mean frame by frame: tsdf
    // bit-identical to it), 1 = first form (A/B, and the only one that runs without the multiplier table).
    // The switches are read per call (a handful of getenv per batch): the parity tests flip them inside one process.
    const int sweep_form = getenv("HV_TSDF_SWEEP") ? atoi(getenv("HV_TSDF_SWEEP")) : 4;
    // per-pixel multiplier table (HV_TSDF_BATCH_MULT=0: compute the multiplier per voxel visit instead; first form only)
    const int use_mult = getenv("HV_TSDF_BATCH_MULT") ? atoi(getenv("HV_TSDF_BATCH_MULT")) : 1;
    // fold form: 12-byte frame records {depth, colour, multiplier}, one gather per voxel visit (HV_TSDF_SWEEP_REC12=0: 8-byte
    // records + the table, two gathers: A/B)
    const bool rec12 = (sweep_form == 3 || sweep_form == 4) && use_mult && !(getenv("HV_TSDF_SWEEP_REC12") && atoi(getenv("HV_TSDF_SWEEP_REC12")) == 0);
    const size_t rec_bytes = rec12 ? 12 : 8;
    bool chain_ok = v->pipe_armed && v->pipe_version == v->content_version; // nothing but batches since ev_presweep was recorded
    v->content_version += 1;
    const int BMAX = 64;
    // Fused form (round 5, the default for the production sweep): one launch = the sweep of the previous batch + this batch's touch +
    // pack pass (k_tsdf_fused); this batch's sweep stays pending until the next call or hv_tsdf_flush.  HV_TSDF_FUSED=0 and every
    // A/B variant of the sweep (other forms, register caps, gather groups ...) take the unfused path below, which starts by flushing.
    auto env_is = [](const char *name, int dflt) { return !getenv(name) || atoi(getenv(name)) == dflt; };
    const bool fused = sweep_form == 4 && use_mult && rec12 && !coherent && pipeline_on && getenv("HV_TSDF_FUSED") && atoi(getenv("HV_TSDF_FUSED")) == 1 &&
                       env_is("HV_TSDF_SWEEP_WPE", 4) && env_is("HV_TSDF_SWEEP_ANYSKIP", 2) && env_is("HV_TSDF_SWEEP_GV", 4) &&
                       env_is("HV_TSDF_SWEEP_PIPE", 1) && env_is("HV_TSDF_BATCH_SPLIT", 4) && env_is("HV_TSDF_SWEEP_VCAP", 0) &&
                       !(getenv("HV_TSDF_LIST") && strcmp(getenv("HV_TSDF_LIST"), "kernel") == 0);
    if (fused) {
        if (v->stream_aux) HV_HIP(hipStreamSynchronize(v->stream_aux)); // (a chain of the unfused form may still run its touch + pack launch there)
        v->pipe_armed = false;
        for (int f0 = 0; f0 < n_frames; f0 += BMAX) {
            const int B = std::min(BMAX, n_frames - f0);
            rc = tsdf_fused_chunk(v, (const char *)d_depth + npx * dsz * (size_t)f0, npx * dsz, (const uint8_t *)d_rgb + npx * 3 * (size_t)f0, B, height, width,
                                  intr, T_cw + 16 * (size_t)f0, depth_scale, depth_trunc, depth_dtype, host_set);
            if (rc != HV_OK) return rc;
        }
        // (the touch + pack pass of every chunk - the only reader of the frames - is queued on the main stream by now)
        if (host_set >= 0) return hv_stage_frames_consumed(v, host_set, v->stream);
        return HV_OK;
    }
    rc = hv_tsdf_flush(v);
    if (rc != HV_OK) return rc;
    for (int f0 = 0; f0 < n_frames; f0 += BMAX) {
        const int B = std::min(BMAX, n_frames - f0);
        // pool headroom (grows here when more than half is known to be used; see hv_capacity_gate)
        bool checked = false;
        const int64_t max_before = v->cfg.max_blocks;
        rc = hv_capacity_gate(v, &checked);
        if (rc != HV_OK) return rc;
        // The touch pass appends first-touched units to the union list itself.  HV_TSDF_LIST=kernel: build the list afterwards
        // from the allocated units instead (one atomic per wave instead of one per unit on a single counter; measured equal at
        // one rank - 32.2 k vs 32.3 k frames/s - for one launch more: kept for A/B only)
        const bool list_in_touch = !(getenv("HV_TSDF_LIST") && strcmp(getenv("HV_TSDF_LIST"), "kernel") == 0);
        if (v->cfg.max_blocks != max_before) chain_ok = false; // the pool grew: the stream was drained, start a fresh chain
        const bool overlap = pipeline_on && chain_ok && !checked && list_in_touch;
        if (v->stream_aux == nullptr) {
            // HV_TSDF_AUX_CUS=n: the touch + pack stream gets n CUs of its own (hipExtStreamCreateWithCUMask, bits spread evenly over
            // the 256) so that its workgroups do not wait for retiring sweep waves (VERDICT r03 Next #5a; measured in DESIGN section 4)
            const int aux_cus = getenv("HV_TSDF_AUX_CUS") ? atoi(getenv("HV_TSDF_AUX_CUS")) : 0;
            if (aux_cus > 0 && aux_cus < 256) {
                uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < aux_cus; ++i) {
                    const int bit = (int)((int64_t)i * 256 / aux_cus);
                    mask[bit >> 5] |= 1u << (bit & 31);
                }
                HV_HIP(hipExtStreamCreateWithCUMask(&v->stream_aux, 8, mask));
            } else if (getenv("HV_TSDF_AUX_PRIO") && atoi(getenv("HV_TSDF_AUX_PRIO")) != 0) {
                int lo_p = 0, hi_p = 0; // (A/B, round 5: with one-wave workgroups a priority has something to act on)
                HV_HIP(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
                HV_HIP(hipStreamCreateWithPriority(&v->stream_aux, hipStreamNonBlocking, hi_p));
            } else
            HV_HIP(hipStreamCreateWithFlags(&v->stream_aux, hipStreamNonBlocking)); // (queue priority high / low against the sweep's: measured, no effect)
            HV_HIP(hipEventCreateWithFlags(&v->ev_prep, hipEventDisableTiming));
            HV_HIP(hipEventCreateWithFlags(&v->ev_presweep, hipEventDisableTiming));
        }
        hipStream_t ps = overlap ? v->stream_aux : v->stream; // where this batch's touch + pack launch goes
        bool overlap_this = overlap;
        // scratch set
        // Three sets (round 5, HV_TSDF_SETS=3): with two, the touch + pack launch of batch k + 1 had to wait for the finish
        // of batch k - 1 and ran beside sweep k, where it only gets wave slots as sweep waves retire - it ended ~37 us AFTER that
        // sweep, and sweep k + 1 waited for it (profiles/r04/pipeline_timeline.txt).  With three, batch k + 2's launch is already
        // queued behind it on the aux stream and soaks up the same tail, so that by the time sweep k + 1 could start, ITS touch + pack
        // launch (and the list sort) finished a whole sweep ago: consecutive sweeps are separated by the finish launch only.
        // Measured (profiles/r05/pipeline_experiments.md): no gain - the aux stream is in order and EVERY touch + pack launch is starved
        // of wave slots for the length of a sweep, so it cannot get further ahead than one batch however many sets there are.  Default 2.
        const int n_sets = (!coherent && getenv("HV_TSDF_SETS") && atoi(getenv("HV_TSDF_SETS")) == 3) ? 3 : 2; // (the per-batch plan keeps two sets)
        const int parity = list_in_touch ? v->batch_parity % n_sets : 0;
        if (list_in_touch) v->batch_parity = (parity + 1) % n_sets;
        int32_t *d_list = v->touched_list + (size_t)parity * (size_t)v->cfg.max_blocks;
        unsigned long long *d_mask_rw = (unsigned long long *)v->touched_mask + (size_t)parity * (size_t)v->table_capacity;
        if (v->ev_set_free[0] == nullptr)
            for (int i = 0; i < HV_TSDF_SETS; ++i) HV_HIP(hipEventCreateWithFlags(&v->ev_set_free[i], hipEventDisableTiming));
        if (overlap) {
            // what this batch's touch + pack launch waits for: the batch that last used its scratch set is swept and finished.  Two
            // sets: ev_presweep = the main stream up to just before the previous sweep.  Three sets: ev_presweep = the main stream
            // where this chain of batches began (everything older - a reset, an extraction ... - lives on the main stream) + the
            // set's own "free" event when a batch of this chain has used the set.
            HV_HIP(hipStreamWaitEvent(v->stream_aux, v->ev_presweep, 0));
            if (n_sets != 2 && v->ev_set_free_valid[parity]) HV_HIP(hipStreamWaitEvent(v->stream_aux, v->ev_set_free[parity], 0));
        } else {
            for (int i = 0; i < HV_TSDF_SETS; ++i) v->ev_set_free_valid[i] = false; // a chain (re)starts with this batch
        }
        if (host_set >= 0) HV_HIP(hipStreamWaitEvent(ps, v->hs_dev_ready[host_set], 0)); // the frames have arrived
        // per-frame constants go through a ring of 4 pinned host buffers: the H2D copy is truly
        // asynchronous and a slot is only rewritten after the copy that last used it has completed,
        // so consecutive calls queue up on the stream without a host synchronisation
        const int ri = v->params_idx;
        v->params_idx = (ri + 1) & 3;
        if (v->pinned_params[ri] == nullptr) {
            HV_HIP(hipHostMalloc(&v->pinned_params[ri], sizeof(HvFrameParams) * BMAX));
            HV_HIP(hipEventCreateWithFlags(&v->params_ev[ri], hipEventDisableTiming));
        } else {
            HV_HIP(hipEventSynchronize(v->params_ev[ri]));
        }
        HvFrameParams *params = (HvFrameParams *)v->pinned_params[ri];
        for (int f = 0; f < B; ++f) {
            make_frame_params(v, height, width, intr, T_cw + 16 * (size_t)(f0 + f), depth_scale, depth_trunc, depth_dtype,
                              &params[f]);
            v->frame_counter += 1;
            params[f].frame_id = v->frame_counter;
            if (coherent) params[f].owner_world = 1; // the touch enumerates every unit of the batch; k_tsdf_plan_assign decides whose it is
        }
        int batch_stamp = v->frame_counter;
        v->last_touch_parity = parity;
        const float *d_mult = nullptr;
        if (use_mult) {
            // (the pack role copies the multipliers into 12-byte records: when the table has to be rebuilt - first call, other
            // intrinsics - it is rebuilt on the main stream and this batch's touch + pack launch follows it there)
            const float *before = v->mult_table;
            const int mw = v->mult_W, mh = v->mult_H;
            float key_before[4];
            memcpy(key_before, v->mult_key, sizeof(key_before));
            rc = tsdf_multiplier_table(v, params[0]);
            if (rc != HV_OK) return rc;
            d_mult = v->mult_table;
            if (before != v->mult_table || mw != v->mult_W || mh != v->mult_H || memcmp(key_before, v->mult_key, sizeof(key_before)) != 0) {
                if (ps != v->stream) HV_HIP(hipStreamSynchronize(v->stream_aux)); // nothing of an older batch still reads the old table there
                ps = v->stream;
                overlap_this = false;
                if (host_set >= 0) HV_HIP(hipStreamWaitEvent(ps, v->hs_dev_ready[host_set], 0));
            }
        }
        // scratch: [B frame records of npx uint2][B HvFrameParams]
        const size_t px_bytes = rec_bytes * npx * (size_t)B;
        void **bb = parity == 2 ? &v->batch_buf3 : parity ? &v->batch_buf2 : &v->batch_buf;
        size_t *bb_bytes = parity == 2 ? &v->batch_buf3_bytes : parity ? &v->batch_buf2_bytes : &v->batch_buf_bytes;
        if (*bb_bytes < px_bytes + sizeof(HvFrameParams) * (size_t)B + 256 && v->stream_aux) HV_HIP(hipStreamSynchronize(v->stream_aux)); // (a launch two batches back may still write the old buffer)
        rc = hv_ensure_buffer(v, bb, bb_bytes, px_bytes + sizeof(HvFrameParams) * (size_t)B + 256); // (re-allocation drains the main stream, and with it every batch whose touch pass it waited for)
        if (rc != HV_OK) return rc;
        uint2 *d_px = (uint2 *)*bb;
        HvFrameParams *d_params = (HvFrameParams *)((char *)*bb + ((px_bytes + 255) & ~(size_t)255));
        // The 12 KB of frame constants go up with a ONE-WORKGROUP kernel that reads the page-locked ring slot in place (hipHostMalloc
        // memory is device-visible).  hipMemcpyAsync turns a small pinned copy into the runtime's blit kernel, whose workgroups wait
        // for wave slots behind the sweep that is running on the other queue: 100-420 us per batch in profiles/r03/
        // pipeline_timeline.txt (first version), which delayed the touch + pack launch to the end of the sweep it should hide in.
        {
            const int n16 = (int)((sizeof(HvFrameParams) * (size_t)B + 15) / 16);
            hipLaunchKernelGGL(k_upload_words, dim3(1), dim3(256), 0, ps, (const uint4 *)params, (uint4 *)d_params, n16);
        }
        HV_HIP(hipEventRecord(v->params_ev[ri], ps));
        // the touched-list counters are zero after hv_reset and k_tsdf_batch_finish; an online frame leaves its own
        // parity's count behind
        // HV_TSDF_AUX_W64=1 (A/B): one-wave workgroups - a 4-wave workgroup of this launch only fits a CU when four sweep waves retire
        // close together, a one-wave workgroup fits wherever ONE sweep wave retires
        const int aux_threads = (!coherent && getenv("HV_TSDF_AUX_W64") && atoi(getenv("HV_TSDF_AUX_W64")) != 0) ? 64 : 256;
        const int n_prep_blocks = (int)((npx + 4 * aux_threads - 1) / (4 * aux_threads)); // 4 pixels per thread
        const int n_touch_blocks = (hv_touch_patches(width, height, v->cfg.depth_sampling_stride) + aux_threads / 64 - 1) / (aux_threads / 64);
        // HV_TSDF_LPT: 2 = the union list is sorted by decreasing work by the touch + pack launch's last touch workgroup, 1 = by a launch
        // of its own behind it (k_tsdf_list_by_work), 0 (default) = list order.  Measured: the sorted order takes 6 % (one rank) / 17 %
        // (an 8-rank share) off the SWEEP and nothing off the STEP - sweep and touch + pack share the machine's wave slots, the sum of
        // their work is what a step costs, and the idle tail the sort removes from the sweep was where the next batch's touch + pack
        // launch ran for free (profiles/r05/pipeline_experiments.md).
        const int lpt = (list_in_touch && !coherent && aux_threads == 256) ? (getenv("HV_TSDF_LPT") ? atoi(getenv("HV_TSDF_LPT")) : 0) : 0;
        int32_t *d_sorted_tail = nullptr;
        if (lpt != 0) {
            const size_t want_sorted = sizeof(int32_t) * HV_TSDF_SETS * (size_t)v->cfg.max_blocks;
            if (v->list_sorted_bytes < want_sorted && v->stream_aux) HV_HIP(hipStreamSynchronize(v->stream_aux)); // (a sort may still write the old array)
            rc = hv_ensure_buffer(v, &v->list_sorted, &v->list_sorted_bytes, want_sorted);
            if (rc != HV_OK) return rc;
            if (v->touch_ticket == nullptr) {
                HV_HIP(hipMalloc((void **)&v->touch_ticket, 256));
                HV_HIP(hipMemsetAsync(v->touch_ticket, 0, 256, v->stream));
                HV_HIP(hipStreamSynchronize(v->stream));
            }
            if (lpt == 2) d_sorted_tail = (int32_t *)v->list_sorted + (size_t)parity * (size_t)v->cfg.max_blocks;
        }
        for (int attempt = 0;; ++attempt) {
            // (coherent form: the plan restarts its own set's list - k_tsdf_touch_plan - and a memset here, on the main stream, could land
            // on a list the aux stream is already filling)
            if (!coherent) {
                if (!v->touch_counters_clean) HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_TOUCH0], 0, HV_CNT_TOUCH_SPAN_BYTES, v->stream)); // (never while a chain runs: only an online frame or an aborted claim leaves them dirty)
                v->touch_counters_clean = true;
            }
            if (coherent) {
                const size_t cap = (size_t)v->table_capacity;
                const size_t set_bytes = cap * 16 + sizeof(uint32_t) * HV_PLAN_BINS + sizeof(int4) * 64;
                char *base = (char *)v->plan_buf + (size_t)parity * set_bytes;
                HvPlan plan;
                plan.bt_keys = (unsigned long long *)base;
                plan.bt_masks = (unsigned long long *)(base + cap * 8);
                plan.hist = (uint32_t *)(base + cap * 16);
                plan.box = (int4 *)(base + cap * 16 + sizeof(uint32_t) * HV_PLAN_BINS);
                plan.cap_mask = (uint32_t)(cap - 1);
                const char *dd = (const char *)d_depth + npx * dsz * (size_t)f0;
                const unsigned slots_grid = (unsigned)((cap + 255) / 256);
                hipLaunchKernelGGL(k_tsdf_touch_plan, dim3(n_touch_blocks * B), dim3(256), 0, ps, v->table, plan, dd, (int64_t)(npx * dsz), d_params,
                                   n_touch_blocks, B, parity);
                hipLaunchKernelGGL(k_tsdf_plan_assign, dim3(slots_grid), dim3(256), 0, ps, v->table, plan, v->touched_stamp, d_mask_rw, d_list,
                                   batch_stamp, d_params, B, parity, v->owner_rank, v->owner_world);
                const bool pack_all = getenv("HV_TSDF_PLAN_PACK_ALL") && atoi(getenv("HV_TSDF_PLAN_PACK_ALL")) != 0; // A/B
                hipLaunchKernelGGL(k_tsdf_prep_touch_batch, dim3(n_prep_blocks * B), dim3(256), 0, ps, v->table, v->touched_stamp, d_mask_rw,
                                   d_list, batch_stamp, dd, (int64_t)(npx * dsz), (const uint8_t *)d_rgb + npx * 3 * (size_t)f0, d_px, d_params,
                                   n_prep_blocks, 0, B, parity, rec12 ? d_mult : nullptr, pack_all ? nullptr : (const int4 *)plan.box, plan.hist,
                                   v->d_status, hv_next_status_seq(v), (int32_t *)nullptr, (uint32_t *)nullptr);
            } else
            hipLaunchKernelGGL(k_tsdf_prep_touch_batch, dim3((n_prep_blocks + n_touch_blocks) * B), dim3(aux_threads), 0, ps,
                               v->table, v->touched_stamp, d_mask_rw, list_in_touch ? d_list : nullptr,
                               batch_stamp, (const char *)d_depth + npx * dsz * (size_t)f0, (int64_t)(npx * dsz),
                               (const uint8_t *)d_rgb + npx * 3 * (size_t)f0, d_px, d_params, n_prep_blocks, n_touch_blocks, B, parity,
                               rec12 ? d_mult : nullptr, nullptr, nullptr, nullptr, 0, d_sorted_tail, d_sorted_tail ? v->touch_ticket : nullptr);
            if (!checked) break;
            // checked mode: nothing is fused before every unit of the batch has its pool slot; if some claims did not fit, the
            // pool has grown (table rebuilt without them, stamps kept) and the touch pass runs again under a fresh stamp
            rc = hv_claims_fit(v);
            if (rc == HV_OK) break;
            if (rc != HV_RETRY_CLAIM || attempt >= 8) return rc == HV_RETRY_CLAIM ? HV_ERR_CAPACITY : rc;
            v->frame_counter += 1;
            batch_stamp = v->frame_counter;
            v->touch_counters_clean = false;
            // the tables were rebuilt: the scratch set's arrays moved
            d_list = v->touched_list + (size_t)parity * (size_t)v->cfg.max_blocks;
            d_mask_rw = (unsigned long long *)v->touched_mask + (size_t)parity * (size_t)v->table_capacity;
            if (lpt != 0) { // (the sorted lists are laid out by max_blocks, which has just grown)
                rc = hv_ensure_buffer(v, &v->list_sorted, &v->list_sorted_bytes, sizeof(int32_t) * HV_TSDF_SETS * (size_t)v->cfg.max_blocks);
                if (rc != HV_OK) return rc;
                if (lpt == 2) d_sorted_tail = (int32_t *)v->list_sorted + (size_t)parity * (size_t)v->cfg.max_blocks;
            }
        }
        if (!list_in_touch) {
            // one thread per pool slot (how many are allocated is only known on the device; threads beyond leave at once)
            hipLaunchKernelGGL(k_tsdf_batch_list, dim3((unsigned)((v->cfg.max_blocks + 255) / 256)), dim3(256), 0, v->stream, v->table,
                               (const int32_t *)v->touched_stamp, batch_stamp, d_list);
        }
        // HV_TSDF_LPT=1: the sweep (and the finish) read the list sorted by decreasing work (k_tsdf_list_by_work), queued behind the
        // touch pass on its stream
        if (lpt == 1) {
            int32_t *d_sorted = (int32_t *)v->list_sorted + (size_t)parity * (size_t)v->cfg.max_blocks;
            hipLaunchKernelGGL(k_tsdf_list_by_work, dim3(1), dim3(1024), 0, ps, v->table, (const int32_t *)d_list, (const unsigned long long *)d_mask_rw,
                               d_sorted, parity);
            d_list = d_sorted;
        } else if (lpt == 2) {
            d_list = (int32_t *)v->list_sorted + (size_t)parity * (size_t)v->cfg.max_blocks; // (sorted by the touch + pack launch itself)
        }
        // What the NEXT batch's touch + pack launch waits for: everything the main stream holds up to here, i.e. the finish of the
        // batch that last used the next batch's scratch set.  Recorded BEFORE the main stream waits for this batch's own touch + pack
        // launch (round 4; HV_TSDF_PRESWEEP_LATE=1: after it, as rounds 2-3 did): the aux stream is in order, so the next launch
        // follows this batch's there anyway, and with the late form every touch + pack launch also waited for the cross-queue
        // hand-off into the sweep before it (~23 us + the upload: profiles/r04/rank8_timeline.txt, the critical loop of a 1/8 share).
        const bool presweep_late = getenv("HV_TSDF_PRESWEEP_LATE") && atoi(getenv("HV_TSDF_PRESWEEP_LATE")) != 0;
        if (!presweep_late && (n_sets == 2 || !overlap)) HV_HIP(hipEventRecord(v->ev_presweep, v->stream));
        if (overlap_this) {
            HV_HIP(hipEventRecord(v->ev_prep, v->stream_aux));
            HV_HIP(hipStreamWaitEvent(v->stream, v->ev_prep, 0));
        }
        if (presweep_late) HV_HIP(hipEventRecord(v->ev_presweep, v->stream));
        chain_ok = true;                                   // (the next chunk of this call may follow this one directly)
        // HV_TSDF_FINISH=epilogue (round 5): the production sweep is launched through k_tsdf_fused without a touch + pack part - the same
        // sweep, whose epilogue does what k_tsdf_batch_finish did (the unit's last part clears its frame mask, the launch's last
        // item zeroes the list counter and publishes the pool status): one launch and one dependent-launch gap less per batch
        const bool epilogue_sweep = d_mult && sweep_form == 4 && rec12 && !coherent && list_in_touch && getenv("HV_TSDF_FINISH") &&
                                    strcmp(getenv("HV_TSDF_FINISH"), "epilogue") == 0 && env_is("HV_TSDF_SWEEP_WPE", 4) &&
                                    env_is("HV_TSDF_SWEEP_ANYSKIP", 2) && env_is("HV_TSDF_SWEEP_GV", 4) && env_is("HV_TSDF_SWEEP_PIPE", 1) &&
                                    env_is("HV_TSDF_BATCH_SPLIT", 4) && env_is("HV_TSDF_SWEEP_VCAP", 0);
        if (epilogue_sweep) {
            HvFusedSweep ES;
            memset(&ES, 0, sizeof(ES));
            ES.list = d_list;
            ES.mask = d_mask_rw;
            ES.px = d_px;
            ES.Ps = d_params;
            ES.n_frames = B;
            ES.parity = parity;
            rc = tsdf_launch_fused(v, false, nullptr, true, &ES);
            if (rc != HV_OK) return rc;
            HV_HIP(hipEventRecord(v->ev_set_free[parity], v->stream));
            v->ev_set_free_valid[parity] = true;
            continue;
        }
        hv_profile_begin(v);
        const int sweep_zh = getenv("HV_TSDF_SWEEP_ZH") ? atoi(getenv("HV_TSDF_SWEEP_ZH")) : 4;
        // workgroups per unit (2 / 4 / 8).  Second form: 8 (two waves per workgroup; 32.7 k frames/s against 31.6 k at 4 and
        // 29.0 k at 2); the first form measured best at 4
        const int split = getenv("HV_TSDF_BATCH_SPLIT") ? atoi(getenv("HV_TSDF_BATCH_SPLIT")) : (sweep_form >= 2 ? 8 : 4);
        (void)split;
        // 1: run the EXACT evaluation with integer weights everywhere (A/B and parity checks of the rare-regime code)
        const int general = getenv("HV_TSDF_BATCH_GENERAL") ? atoi(getenv("HV_TSDF_BATCH_GENERAL")) : 0;
        const unsigned long long *d_mask = d_mask_rw;
        // workgroups in the grid: one work item (unit x SPLIT part) each up to 16 384 units per batch, grid-stride beyond;
        // measured 28.2 k frames/s at 8192 (3.3 items per workgroup: coarser tail), 29.0 k at 16 384, 29.6 k at 65 536
        const int sweep_grid = getenv("HV_TSDF_BATCH_GRID") ? atoi(getenv("HV_TSDF_BATCH_GRID")) : 65536;
#define HV_LAUNCH_COL(S, MT)                                                                                           \
    hipLaunchKernelGGL((k_tsdf_integrate_batch_col<4, S, MT>), dim3(sweep_grid), dim3(64 * 16 / S), 0, v->stream, v->table, \
                       d_list, d_mask, (char *)v->pool, d_px, d_params, parity, general, d_mult)
#define HV_LAUNCH_SWEEP(ZH, S, WPE)                                                                                    \
    hipLaunchKernelGGL((k_tsdf_sweep<ZH, S, WPE>), dim3(sweep_grid), dim3(64 * (64 / ZH) / S), 0, v->stream, v->table,   \
                       d_list, d_mask, (char *)v->pool, d_px, d_params, B, general, d_mult, xcd_aware, parity)
#define HV_LAUNCH_FOLD(ZH, S, WPE, ANY)                                                                                \
    hipLaunchKernelGGL((k_tsdf_sweep_fold<ZH, S, WPE, ANY>), dim3(sweep_grid), dim3(64 * (64 / ZH) / S), 0, v->stream,   \
                       v->table, d_list, d_mask, (char *)v->pool, d_px, d_params, B, general, d_mult, xcd_aware, parity)
#define HV_LAUNCH_COLUMN(S, WPE, GV, PIPE, ANY)                                                                         \
    hipLaunchKernelGGL((k_tsdf_sweep_column<S, WPE, GV, PIPE, ANY>), dim3(sweep_grid), dim3(64 * 4 / S), 0, v->stream,   \
                       v->table, d_list, d_mask, (char *)v->pool, d_px, d_params, B, general, d_mult, xcd_aware, parity)
        if (d_mult && sweep_form == 4) {
            // column form: HV_TSDF_SWEEP_WPE (waves / SIMD the registers are capped for), _GV (voxels per gather group), _PIPE
            // (next group's gathers before this group's fold), _ANYSKIP, HV_TSDF_BATCH_SPLIT (workgroups per unit: 1 / 2 / 4)
            const int wpe = getenv("HV_TSDF_SWEEP_WPE") ? atoi(getenv("HV_TSDF_SWEEP_WPE")) : 4;
            const int xcd_aware = getenv("HV_TSDF_SWEEP_XCD") ? atoi(getenv("HV_TSDF_SWEEP_XCD")) : 2;
            const int anyskip = getenv("HV_TSDF_SWEEP_ANYSKIP") ? atoi(getenv("HV_TSDF_SWEEP_ANYSKIP")) : 2; // 2: exec-masked fold
            const int gv = getenv("HV_TSDF_SWEEP_GV") ? atoi(getenv("HV_TSDF_SWEEP_GV")) : 4;
            const int pipe = getenv("HV_TSDF_SWEEP_PIPE") ? atoi(getenv("HV_TSDF_SWEEP_PIPE")) : 1;
            const int csplit = getenv("HV_TSDF_BATCH_SPLIT") ? atoi(getenv("HV_TSDF_BATCH_SPLIT")) : 4;
            const int vcap = getenv("HV_TSDF_SWEEP_VCAP") ? atoi(getenv("HV_TSDF_SWEEP_VCAP")) : 0;
            const int zs = getenv("HV_TSDF_SWEEP_ZS") ? atoi(getenv("HV_TSDF_SWEEP_ZS")) : (v->owner_world >= 4 ? 2 : 1);
#define HV_LAUNCH_COLUMN_CAPPED(NAME, S, GV, PIPE)                                                                      \
    hipLaunchKernelGGL((NAME<S, GV, PIPE, true>), dim3(sweep_grid), dim3(64 * 4 / S), 0, v->stream, v->table, d_list, d_mask, \
                       (char *)v->pool, d_px, d_params, B, general, d_mult, xcd_aware, parity)
            if (vcap == 120 || vcap == 112) {
                if (vcap == 120) { if (gv == 2) HV_LAUNCH_COLUMN_CAPPED(k_tsdf_sweep_column_v120, 4, 2, 1); else HV_LAUNCH_COLUMN_CAPPED(k_tsdf_sweep_column_v120, 4, 4, 1); }
                else { if (gv == 2) HV_LAUNCH_COLUMN_CAPPED(k_tsdf_sweep_column_v112, 4, 2, 1); else HV_LAUNCH_COLUMN_CAPPED(k_tsdf_sweep_column_v112, 4, 4, 1); }
            } else if (csplit == 1) {
                HV_LAUNCH_COLUMN(1, 4, 4, 1, true);
            } else if (csplit == 2) {
                if (pipe == 2) HV_LAUNCH_COLUMN(2, 4, 4, 2, true); else HV_LAUNCH_COLUMN(2, 4, 4, 1, true);
            } else if (!anyskip) {
                HV_LAUNCH_COLUMN(4, 4, 4, 1, false);
            } else if (zs == 2) {
                // z halves: 8 tasks per unit (default for a GPU that shares the volume with 3 or more others)
                if (wpe >= 6) hipLaunchKernelGGL((k_tsdf_sweep_column<4, 6, 4, 1, 2, 2>), dim3(sweep_grid), dim3(64), 0, v->stream, v->table, d_list, d_mask,
                                                 (char *)v->pool, d_px, d_params, B, general, d_mult, xcd_aware, parity);
                else if (wpe == 5) hipLaunchKernelGGL((k_tsdf_sweep_column<4, 5, 4, 1, 2, 2>), dim3(sweep_grid), dim3(64), 0, v->stream, v->table, d_list, d_mask,
                                                      (char *)v->pool, d_px, d_params, B, general, d_mult, xcd_aware, parity);
                else hipLaunchKernelGGL((k_tsdf_sweep_column<4, 4, 4, 1, 2, 2>), dim3(sweep_grid), dim3(64), 0, v->stream, v->table, d_list, d_mask,
                                        (char *)v->pool, d_px, d_params, B, general, d_mult, xcd_aware, parity);
            } else if (anyskip == 2 && gv == 4 && pipe == 1 && wpe == 4) {
                HV_LAUNCH_COLUMN(4, 4, 4, 1, 2);
            } else if (gv == 8) {
                if (wpe >= 4) HV_LAUNCH_COLUMN(4, 4, 8, 2, true); else if (wpe == 3) HV_LAUNCH_COLUMN(4, 3, 8, 2, true); else HV_LAUNCH_COLUMN(4, 2, 8, 2, true);
            } else if (gv == 2) {
                if (pipe == 2) { if (wpe >= 5) HV_LAUNCH_COLUMN(4, 5, 2, 2, true); else HV_LAUNCH_COLUMN(4, 4, 2, 2, true); }
                else if (pipe == 1) { if (wpe >= 5) HV_LAUNCH_COLUMN(4, 5, 2, 1, true); else HV_LAUNCH_COLUMN(4, 4, 2, 1, true); }
                else HV_LAUNCH_COLUMN(4, 5, 2, 0, true);
            } else if (gv == 1) {
                HV_LAUNCH_COLUMN(4, 5, 1, 1, true);
            } else {
                if (pipe == 2) { if (wpe >= 5) HV_LAUNCH_COLUMN(4, 5, 4, 2, true); else if (wpe == 3) HV_LAUNCH_COLUMN(4, 3, 4, 2, true); else HV_LAUNCH_COLUMN(4, 4, 4, 2, true); }
                else if (pipe == 1) { if (wpe >= 5) HV_LAUNCH_COLUMN(4, 5, 4, 1, true); else if (wpe == 3) HV_LAUNCH_COLUMN(4, 3, 4, 1, true); else HV_LAUNCH_COLUMN(4, 4, 4, 1, true); }
                else HV_LAUNCH_COLUMN(4, 4, 4, 0, true);
            }
        } else if (d_mult && sweep_form == 3) {
            // fold form (production); HV_TSDF_SWEEP_WPE / _ZH / _ANYSKIP / _REC12 / HV_TSDF_BATCH_SPLIT select the measured alternatives
            const int wpe = getenv("HV_TSDF_SWEEP_WPE") ? atoi(getenv("HV_TSDF_SWEEP_WPE")) : 4;
            const int xcd_aware = getenv("HV_TSDF_SWEEP_XCD") ? atoi(getenv("HV_TSDF_SWEEP_XCD")) : 2;
            const int anyskip = getenv("HV_TSDF_SWEEP_ANYSKIP") ? atoi(getenv("HV_TSDF_SWEEP_ANYSKIP")) : 0;
            if (!rec12) {
                hipLaunchKernelGGL((k_tsdf_sweep_fold<4, 8, 4, false, 0, false>), dim3(sweep_grid), dim3(64 * 16 / 8), 0, v->stream, v->table,
                                   d_list, d_mask, (char *)v->pool, d_px, d_params, B, general, d_mult, xcd_aware, parity);
            } else if (sweep_zh == 8) {
                if (split == 4) HV_LAUNCH_FOLD(8, 4, 4, false); else if (wpe <= 2) HV_LAUNCH_FOLD(8, 8, 2, false); else HV_LAUNCH_FOLD(8, 8, 4, false);
            } else if (split == 4) {
                if (wpe >= 8) HV_LAUNCH_FOLD(4, 4, 8, false); else HV_LAUNCH_FOLD(4, 4, 4, false);
            } else if (split == 16) {
                if (wpe >= 8) HV_LAUNCH_FOLD(4, 16, 8, false); else HV_LAUNCH_FOLD(4, 16, 6, false);
            } else if (anyskip) {
                if (wpe >= 8) HV_LAUNCH_FOLD(4, 8, 8, true); else HV_LAUNCH_FOLD(4, 8, 6, true);
            } else if (getenv("HV_TSDF_SWEEP_DBG") && atoi(getenv("HV_TSDF_SWEEP_DBG")) > 0) {
                const int dbg = atoi(getenv("HV_TSDF_SWEEP_DBG"));
#define HV_LAUNCH_FOLD_DBG(D)                                                                                          \
    hipLaunchKernelGGL((k_tsdf_sweep_fold<4, 8, 4, false, D>), dim3(sweep_grid), dim3(64 * 16 / 8), 0, v->stream, v->table, \
                       d_list, d_mask, (char *)v->pool, d_px, d_params, B, general, d_mult, xcd_aware, parity)
                if (dbg == 1) HV_LAUNCH_FOLD_DBG(1); else if (dbg == 2) HV_LAUNCH_FOLD_DBG(2); else HV_LAUNCH_FOLD_DBG(3);
#undef HV_LAUNCH_FOLD_DBG
            } else {
                if (wpe >= 8) HV_LAUNCH_FOLD(4, 8, 8, false); else if (wpe == 6 || wpe == 7) HV_LAUNCH_FOLD(4, 8, 6, false);
                else if (wpe == 5) HV_LAUNCH_FOLD(4, 8, 5, false); else HV_LAUNCH_FOLD(4, 8, 4, false);
            }
        } else if (d_mult && sweep_form == 2) {
            const int wpe = getenv("HV_TSDF_SWEEP_WPE") ? atoi(getenv("HV_TSDF_SWEEP_WPE")) : 4; // 4 waves / SIMD = 128 VGPRs: nothing spills
            const int xcd_aware = getenv("HV_TSDF_SWEEP_XCD") ? atoi(getenv("HV_TSDF_SWEEP_XCD")) : 2; // list entries per XCD group (0: list order)
            if (sweep_zh == 8) {
                if (split == 2) HV_LAUNCH_SWEEP(8, 2, 1); else if (wpe == 4) HV_LAUNCH_SWEEP(8, 4, 2); else HV_LAUNCH_SWEEP(8, 4, 1);
            } else if (split == 2) {
                if (wpe == 4) HV_LAUNCH_SWEEP(4, 2, 4); else HV_LAUNCH_SWEEP(4, 2, 1);
            } else if (split == 8) {
                if (wpe == 4) HV_LAUNCH_SWEEP(4, 8, 4); else if (wpe == 5) HV_LAUNCH_SWEEP(4, 8, 5); else HV_LAUNCH_SWEEP(4, 8, 1);
            } else {
                if (wpe == 4) HV_LAUNCH_SWEEP(4, 4, 4); else if (wpe == 3) HV_LAUNCH_SWEEP(4, 4, 3); else if (wpe == 1) HV_LAUNCH_SWEEP(4, 4, 1); else HV_LAUNCH_SWEEP(4, 4, 5);
            }
        } else if (d_mult) {
            if (split == 2) HV_LAUNCH_COL(2, true); else if (split == 8) HV_LAUNCH_COL(8, true); else HV_LAUNCH_COL(4, true);
        } else {
            if (split == 2) HV_LAUNCH_COL(2, false); else if (split == 8) HV_LAUNCH_COL(8, false); else HV_LAUNCH_COL(4, false);
        }
#undef HV_LAUNCH_COL
#undef HV_LAUNCH_SWEEP
#undef HV_LAUNCH_FOLD
#undef HV_LAUNCH_COLUMN
#undef HV_LAUNCH_COLUMN_CAPPED
        hv_profile_end(v, B);
        // (image-coherent ownership needs no finish launch: the plan stores the masks of exactly the units it lists, restarts the
        // set's list itself and publishes the pool status after its claims - sweep k + 1 follows sweep k directly)
        if (!coherent)
            hipLaunchKernelGGL(k_tsdf_batch_finish, dim3(1), dim3(1024), 0, v->stream, v->table, d_list, d_mask_rw, parity, v->d_status,
                               hv_next_status_seq(v));
        else
            v->plan_lists_stale = true; // the sets' list counters keep their batches' sizes until their next plan: an online frame clears them first
        HV_HIP(hipGetLastError());
        HV_HIP(hipEventRecord(v->ev_set_free[parity], v->stream));
        v->ev_set_free_valid[parity] = true;
    }
    v->pipe_armed = true;
    v->pipe_version = v->content_version;
    // (every touch + pack launch of this call precedes the point the main stream has reached: it waited for each of them)
    if (host_set >= 0) return hv_stage_frames_consumed(v, host_set, v->stream);
    return HV_OK;
}

int hv_tsdf_integrate_batch(hv_volume *v, const void *depth, int32_t depth_dtype, const uint8_t *rgb,
                            int32_t n_frames, int32_t height, int32_t width, const double *intr,
                            const double *T_cw, double depth_scale, double depth_trunc, int32_t loc) {
    int rc = check_tsdf_args(v, depth, rgb, height, width, intr, T_cw, n_frames);
    if (rc != HV_OK) return rc;
    return tsdf_integrate_batch_impl(v, depth, nullptr, depth_dtype, rgb, nullptr, n_frames, height, width, intr, T_cw, depth_scale,
                                     depth_trunc, loc);
}

int hv_tsdf_integrate_frames(hv_volume *v, const void *const *depth_frames, int32_t depth_dtype, const uint8_t *const *rgb_frames,
                             int32_t n_frames, int32_t height, int32_t width, const double *intr, const double *T_cw,
                             double depth_scale, double depth_trunc) {
    HV_REQUIRE(depth_frames != nullptr && rgb_frames != nullptr && n_frames >= 1, HV_ERR_INVALID, "hv_tsdf_integrate_frames: null argument");
    for (int f = 0; f < n_frames; ++f)
        HV_REQUIRE(depth_frames[f] != nullptr && rgb_frames[f] != nullptr, HV_ERR_INVALID, "hv_tsdf_integrate_frames: null frame %d", f);
    int rc = check_tsdf_args(v, depth_frames[0], rgb_frames[0], height, width, intr, T_cw, n_frames);
    if (rc != HV_OK) return rc;
    return tsdf_integrate_batch_impl(v, nullptr, depth_frames, depth_dtype, nullptr, (const void *const *)rgb_frames, n_frames, height,
                                     width, intr, T_cw, depth_scale, depth_trunc, HV_HOST);
}

int hv_tsdf_set_tile(hv_volume *v, int32_t u0, int32_t v0, int32_t u1, int32_t v1) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_tsdf_set_tile: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_set_tile: volume is not in TSDF mode");
    HV_REQUIRE((u0 == 0 && v0 == 0 && u1 == 0 && v1 == 0) || (u0 >= 0 && v0 >= 0 && u1 > u0 && v1 > v0), HV_ERR_INVALID,
               "hv_tsdf_set_tile: empty tile");
    v->tile[0] = u0;
    v->tile[1] = v0;
    v->tile[2] = u1;
    v->tile[3] = v1;
    return HV_OK;
}

int hv_tsdf_set_sharding(hv_volume *v, int32_t mode) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_tsdf_set_sharding: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_set_sharding: volume is not in TSDF mode");
    HV_REQUIRE(mode == 0 || mode == 1, HV_ERR_INVALID, "hv_tsdf_set_sharding: mode must be 0 (hash) or 1 (image-coherent)");
    if (v->shard_coherent != 0 && mode == 0 && v->touched_mask != nullptr) {
        // the hash form ORs frame bits into masks that are zero between batches; the coherent form leaves its last masks behind
        HV_HIP(hipSetDevice(v->device));
        if (v->stream_aux) HV_HIP(hipStreamSynchronize(v->stream_aux));
        HV_HIP(hipMemsetAsync(v->touched_mask, 0, sizeof(uint64_t) * 2 * v->table_capacity, v->stream));
        HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_TOUCH0], 0, HV_CNT_TOUCH_SPAN_BYTES, v->stream));
        HV_HIP(hipStreamSynchronize(v->stream));
        v->pipe_armed = false;
    }
    v->shard_coherent = mode;
    return HV_OK;
}

int hv_tsdf_set_color_order(hv_volume *v, int32_t bgr) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_tsdf_set_color_order: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_set_color_order: volume is not in TSDF mode");
    v->color_bgr = bgr ? 1 : 0;
    return HV_OK;
}

int hv_tsdf_set_owner(hv_volume *v, int32_t rank, int32_t world_size) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_tsdf_set_owner: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_set_owner: volume is not in TSDF mode");
    HV_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, HV_ERR_INVALID, "hv_tsdf_set_owner: bad rank/world");
    v->owner_rank = rank;
    v->owner_world = world_size;
    return HV_OK;
}

int hv_tsdf_dump(hv_volume *v, int32_t *keys, float *tsdf, float *weight, double *color, int64_t *n_units) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr && n_units != nullptr, HV_ERR_INVALID, "hv_tsdf_dump: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_dump: volume is not in TSDF mode");
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    *n_units = nb;
    if (nb == 0 || (keys == nullptr && tsdf == nullptr && weight == nullptr && color == nullptr)) return HV_OK;
    std::vector<uint64_t> bkeys((size_t)nb);
    HV_HIP(hipMemcpy(bkeys.data(), v->table.block_keys, sizeof(uint64_t) * nb, hipMemcpyDeviceToHost));
    std::vector<int64_t> order((size_t)nb);
    std::iota(order.begin(), order.end(), 0);
    std::vector<int32_t> xyz((size_t)nb * 3);
    for (int64_t i = 0; i < nb; ++i) hv_unpack_key(bkeys[i], xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        for (int k = 0; k < 3; ++k)
            if (xyz[a * 3 + k] != xyz[b * 3 + k]) return xyz[a * 3 + k] < xyz[b * 3 + k];
        return false;
    });
    std::vector<char> unit((size_t)PLANE_BYTES * HV_TSDF_PLANES);
    for (int64_t o = 0; o < nb; ++o) {
        const int64_t i = order[o];
        if (keys) memcpy(keys + o * 3, &xyz[i * 3], 12);
        if (!(tsdf || weight || color)) continue;
        HV_HIP(hipMemcpy(unit.data(), (char *)v->pool + i * unit.size(), unit.size(), hipMemcpyDeviceToHost));
        const float *pt = (const float *)unit.data();
        const uint32_t *pw = (const uint32_t *)(unit.data() + PLANE_BYTES);
        const uint32_t *pr = (const uint32_t *)(unit.data() + 2 * PLANE_BYTES);
        const uint32_t *pg = (const uint32_t *)(unit.data() + 3 * PLANE_BYTES);
        const uint32_t *pb = (const uint32_t *)(unit.data() + 4 * PLANE_BYTES);
        for (int x = 0; x < R; ++x)
            for (int y = 0; y < R; ++y)
                for (int z = 0; z < R; ++z) {
                    const int src = z * RR + x * R + y;
                    const int64_t dst = o * RRR + (x * R + y) * R + z;
                    if (tsdf) tsdf[dst] = pt[src];
                    if (weight) weight[dst] = (float)pw[src];
                    if (color) {
                        const double w = (double)pw[src];
                        color[dst * 3 + 0] = w > 0 ? (double)pr[src] / w : 0.0;
                        color[dst * 3 + 1] = w > 0 ? (double)pg[src] / w : 0.0;
                        color[dst * 3 + 2] = w > 0 ? (double)pb[src] / w : 0.0;
                    }
                }
    }
    return HV_OK;
}

int hv_tsdf_touched(hv_volume *v, int32_t *keys, int64_t cap, int64_t *n) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_tsdf_touched: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_touched: volume is not in TSDF mode");
    int rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    int64_t nt = v->h_counters[HV_CNT_TOUCH(v->last_touch_parity)];
    if (nt > v->cfg.max_blocks) nt = v->cfg.max_blocks;
    *n = nt;
    if (keys == nullptr || nt == 0) return HV_OK;
    std::vector<int32_t> slots((size_t)nt);
    HV_HIP(hipMemcpy(slots.data(), touched_list_of(v, v->last_touch_parity), sizeof(int32_t) * nt, hipMemcpyDeviceToHost));
    std::vector<uint64_t> tkeys((size_t)v->table_capacity);
    HV_HIP(hipMemcpy(tkeys.data(), v->table.keys, sizeof(uint64_t) * v->table_capacity, hipMemcpyDeviceToHost));
    std::vector<std::array<int32_t, 3>> out((size_t)nt);
    for (int64_t i = 0; i < nt; ++i) hv_unpack_key(tkeys[slots[i]], out[i][0], out[i][1], out[i][2]);
    std::sort(out.begin(), out.end());
    const int64_t m = std::min(nt, cap);
    for (int64_t i = 0; i < m; ++i) memcpy(keys + i * 3, out[i].data(), 12);
    return HV_OK;
}

int hv_tsdf_unit_keys(hv_volume *v, int32_t *keys, int64_t cap, int64_t *n) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_tsdf_unit_keys: null argument");
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    *n = nb;
    if (keys == nullptr || nb == 0) return HV_OK;
    std::vector<uint64_t> bkeys((size_t)nb);
    HV_HIP(hipMemcpy(bkeys.data(), v->table.block_keys, sizeof(uint64_t) * nb, hipMemcpyDeviceToHost));
    const int64_t m = std::min(nb, cap);
    for (int64_t i = 0; i < m; ++i) hv_unpack_key(bkeys[i], keys[i * 3], keys[i * 3 + 1], keys[i * 3 + 2]);
    return HV_OK;
}

int hv_tsdf_export_numerators(hv_volume *v, const int32_t *keys, int64_t k, float *payload, int32_t loc) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr && (k == 0 || (keys != nullptr && payload != nullptr)), HV_ERR_INVALID,
               "hv_tsdf_export_numerators: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_export_numerators: not a TSDF volume");
    if (k == 0) return HV_OK;
    HV_HIP(hipSetDevice(v->device));
    const void *d_keys = nullptr;
    int rc = hv_stage_in(v, keys, sizeof(int32_t) * 3 * k, HV_HOST, 0, &d_keys);
    if (rc != HV_OK) return rc;
    const size_t bytes = sizeof(float) * 5 * RRR * (size_t)k;
    float *d_payload = payload;
    if (loc == HV_HOST) {
        v->mesh_cache_version = v->points_cache_version = 0; // out_a is about to be overwritten
        rc = hv_ensure_buffer(v, &v->out_a, &v->out_a_bytes, bytes);
        if (rc != HV_OK) return rc;
        d_payload = (float *)v->out_a;
    }
    const int64_t total = k * RRR;
    hipLaunchKernelGGL(k_tsdf_export, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->stream, v->table,
                       (const char *)v->pool, (const int32_t *)d_keys, k, d_payload);
    HV_HIP(hipGetLastError());
    if (loc == HV_HOST) {
        HV_HIP(hipMemcpyAsync(payload, d_payload, bytes, hipMemcpyDeviceToHost, v->stream));
    }
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}

int hv_tsdf_import_numerators(hv_volume *v, const int32_t *keys, int64_t k, const float *payload, int32_t loc) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr && (k == 0 || (keys != nullptr && payload != nullptr)), HV_ERR_INVALID,
               "hv_tsdf_import_numerators: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_import_numerators: not a TSDF volume");
    if (k == 0) return HV_OK;
    v->content_version += 1;
    HV_HIP(hipSetDevice(v->device));
    const void *d_keys = nullptr, *d_payload = nullptr;
    int rc = hv_stage_in(v, keys, sizeof(int32_t) * 3 * k, HV_HOST, 0, &d_keys);
    if (rc != HV_OK) return rc;
    rc = hv_stage_in(v, payload, sizeof(float) * 5 * RRR * (size_t)k, loc, 1, &d_payload);
    if (rc != HV_OK) return rc;
    // the imported units are claimed first and the claims verified (this call synchronises anyway): a pool that is too small
    // grows before anything is written, or the call fails with the volume unchanged (a gather onto a root whose
    // pool was sized like every other rank's)
    bool checked = false;
    rc = hv_capacity_gate(v, &checked);
    if (rc != HV_OK) return rc;
    for (int attempt = 0;; ++attempt) {
        hipLaunchKernelGGL(k_tsdf_import_claim, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, v->stream, v->table,
                           (const int32_t *)d_keys, k);
        rc = hv_claims_fit(v);
        if (rc == HV_OK) break;
        if (rc != HV_RETRY_CLAIM || attempt >= 8) return rc == HV_RETRY_CLAIM ? HV_ERR_CAPACITY : rc;
    }
    const int64_t total = k * RRR;
    hipLaunchKernelGGL(k_tsdf_import, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->stream, v->table,
                       (char *)v->pool, (const int32_t *)d_keys, k, (const float *)d_payload);
    HV_HIP(hipGetLastError());
    hv_launch_publish_status(v); // the imported units are part of the published occupancy (a later rollback keeps them)
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}

// ---- halo merge entry points (SURVEY 8b hv_merge_halo, 8e): the library packs / unpacks on the device, the caller runs
// the two collectives (all-gather of key lists, all-reduce of the dense buffer) with whatever transport it has -
// torch.distributed over RCCL in pyslam_amd/distributed.py ----
int hv_tsdf_dirty_keys(hv_volume *v, int32_t *keys, int64_t cap, int64_t *n) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_tsdf_dirty_keys: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_dirty_keys: not a TSDF volume");
    HV_HIP(hipSetDevice(v->device));
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    *n = 0;
    if (nb == 0) return HV_OK;
    v->mesh_cache_version = v->points_cache_version = 0; // out_a is reused below
    rc = hv_ensure_buffer(v, &v->out_a, &v->out_a_bytes, sizeof(int32_t) * 3 * (size_t)nb);
    if (rc != HV_OK) return rc;
    HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT], 0, sizeof(int32_t), v->stream));
    hipLaunchKernelGGL(k_tsdf_collect_dirty, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, v->stream, v->table,
                       (const int32_t *)v->touched_stamp, v->merge_stamp, (int32_t)nb, (int32_t *)v->out_a, (int32_t)nb);
    HV_HIP(hipGetLastError());
    rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    const int64_t nd = v->h_counters[HV_CNT_OUT];
    *n = nd;
    if (keys == nullptr || nd == 0) return HV_OK;
    std::vector<std::array<int32_t, 3>> out((size_t)nd);
    HV_HIP(hipMemcpy(out.data(), v->out_a, sizeof(int32_t) * 3 * (size_t)nd, hipMemcpyDeviceToHost));
    std::sort(out.begin(), out.end());
    const int64_t m = std::min(nd, cap);
    for (int64_t i = 0; i < m; ++i) memcpy(keys + i * 3, out[i].data(), 12);
    return HV_OK;
}

int hv_tsdf_mark_merged(hv_volume *v) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_tsdf_mark_merged: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_mark_merged: not a TSDF volume");
    v->merge_stamp = v->frame_counter;
    return HV_OK;
}

int hv_merge_halo_plan(const int32_t *gathered_keys, const int64_t *counts, int32_t world_size, int32_t rank,
                       int32_t *shared_keys, uint8_t *action, int64_t cap, int64_t *n_shared) {
    HV_REQUIRE(counts != nullptr && n_shared != nullptr && world_size >= 1 && rank >= 0 && rank < world_size, HV_ERR_INVALID,
               "hv_merge_halo_plan: bad argument");
    // (key, rank) pairs of every rank's dirty list, sorted by key then rank; a key listed by >= 2 ranks is shared
    struct Entry { std::array<int32_t, 3> key; int32_t rank; };
    int64_t total = 0;
    for (int r = 0; r < world_size; ++r) total += counts[r];
    HV_REQUIRE(total == 0 || gathered_keys != nullptr, HV_ERR_INVALID, "hv_merge_halo_plan: null key list");
    std::vector<Entry> all((size_t)total);
    int64_t at = 0;
    for (int r = 0; r < world_size; ++r)
        for (int64_t i = 0; i < counts[r]; ++i, ++at) {
            memcpy(all[at].key.data(), gathered_keys + at * 3, 12);
            all[at].rank = r;
        }
    std::sort(all.begin(), all.end(), [](const Entry &a, const Entry &b) { return a.key != b.key ? a.key < b.key : a.rank < b.rank; });
    int64_t ns = 0;
    for (int64_t i = 0; i < total;) {
        int64_t j = i;
        bool mine = false;
        while (j < total && all[j].key == all[i].key) {
            mine |= all[j].rank == rank;
            ++j;
        }
        if (j - i >= 2) { // listed by two ranks or more (a rank lists a key once)
            if (shared_keys != nullptr && action != nullptr && ns < cap) {
                memcpy(shared_keys + ns * 3, all[i].key.data(), 12);
                // the lowest listing rank keeps the unit; EVERY other rank zeroes its copy if it has one - also a rank that holds
                // the unit from an earlier window without having listed it now: the pack step exports whatever a rank holds,
                // so a holder that kept its copy would be counted twice (unpack is a no-op where the unit does not exist)
                action[ns] = all[i].rank == rank ? 1 : 2;
                (void)mine;
            }
            ++ns;
        }
        i = j;
    }
    *n_shared = ns;
    return HV_OK;
}

int hv_merge_halo_plan_held(const int32_t *dirty_keys, const int64_t *dirty_counts, const int32_t *held_keys,
                            const int64_t *held_counts, int32_t world_size, int32_t rank, int32_t *shared_keys, uint8_t *action,
                            int64_t cap, int64_t *n_shared) {
    HV_REQUIRE(dirty_counts != nullptr && held_counts != nullptr && n_shared != nullptr && world_size >= 1 && rank >= 0 && rank < world_size,
               HV_ERR_INVALID, "hv_merge_halo_plan_held: bad argument");
    // (key, rank, kind) of every rank's two lists; a key is merged when some rank updated it since its last merge AND two
    // ranks or more hold it: afterwards the lowest HOLDING rank has the complete unit and every other holder zeros
    struct Entry { std::array<int32_t, 3> key; int32_t rank; int32_t dirty; };
    int64_t nd = 0, nh = 0;
    for (int r = 0; r < world_size; ++r) {
        nd += dirty_counts[r];
        nh += held_counts[r];
    }
    HV_REQUIRE((nd == 0 || dirty_keys != nullptr) && (nh == 0 || held_keys != nullptr), HV_ERR_INVALID, "hv_merge_halo_plan_held: null key list");
    std::vector<Entry> all((size_t)(nd + nh));
    int64_t at = 0, src = 0;
    for (int r = 0; r < world_size; ++r)
        for (int64_t i = 0; i < dirty_counts[r]; ++i, ++at, ++src) {
            memcpy(all[at].key.data(), dirty_keys + src * 3, 12);
            all[at].rank = r;
            all[at].dirty = 1;
        }
    src = 0;
    for (int r = 0; r < world_size; ++r)
        for (int64_t i = 0; i < held_counts[r]; ++i, ++at, ++src) {
            memcpy(all[at].key.data(), held_keys + src * 3, 12);
            all[at].rank = r;
            all[at].dirty = 0;
        }
    std::sort(all.begin(), all.end(), [](const Entry &a, const Entry &b) {
        if (a.key != b.key) return a.key < b.key;
        if (a.rank != b.rank) return a.rank < b.rank;
        return a.dirty < b.dirty;
    });
    int64_t ns = 0;
    const int64_t total = nd + nh;
    for (int64_t i = 0; i < total;) {
        int64_t j = i;
        int holders = 0, keeper = -1, last_holder = -1;
        bool any_dirty = false;
        while (j < total && all[j].key == all[i].key) {
            if (all[j].dirty) {
                any_dirty = true;
            } else if (all[j].rank != last_holder) {
                last_holder = all[j].rank;
                if (keeper < 0) keeper = all[j].rank;
                ++holders;
            }
            ++j;
        }
        if (any_dirty && holders >= 2) {
            if (shared_keys != nullptr && action != nullptr && ns < cap) {
                memcpy(shared_keys + ns * 3, all[i].key.data(), 12);
                action[ns] = keeper == rank ? 1 : 2;
            }
            ++ns;
        }
        i = j;
    }
    *n_shared = ns;
    return HV_OK;
}

int hv_merge_halo_pack(hv_volume *v, const int32_t *shared_keys, int64_t k, float *payload, int32_t loc) {
    return hv_tsdf_export_numerators(v, shared_keys, k, payload, loc);
}

int hv_merge_halo_unpack(hv_volume *v, const int32_t *shared_keys, int64_t k, const float *payload, const uint8_t *action,
                         int32_t loc) {
    if (const int frc_ = hv_tsdf_flush(v)) return frc_; // the deferred sweep of the last multi-frame batch goes first
    HV_REQUIRE(v != nullptr && (k == 0 || (shared_keys != nullptr && payload != nullptr && action != nullptr)), HV_ERR_INVALID,
               "hv_merge_halo_unpack: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_merge_halo_unpack: not a TSDF volume");
    if (k == 0) return HV_OK;
    v->content_version += 1;
    HV_HIP(hipSetDevice(v->device));
    // keys + actions in one staging buffer, payload in the other
    std::vector<char> host((size_t)k * 13);
    memcpy(host.data(), shared_keys, (size_t)k * 12);
    memcpy(host.data() + (size_t)k * 12, action, (size_t)k);
    const void *d_ka = nullptr, *d_payload = nullptr;
    int rc = hv_stage_in(v, host.data(), host.size(), HV_HOST, 0, &d_ka);
    if (rc != HV_OK) return rc;
    rc = hv_stage_in(v, payload, sizeof(float) * 5 * RRR * (size_t)k, loc, 1, &d_payload);
    if (rc != HV_OK) return rc;
    const int64_t total = k * RRR;
    hipLaunchKernelGGL(k_tsdf_halo_unpack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->stream, v->table,
                       (char *)v->pool, (const int32_t *)d_ka, k, (const float *)d_payload,
                       (const uint8_t *)d_ka + (size_t)k * 12);
    HV_HIP(hipGetLastError());
    HV_HIP(hipStreamSynchronize(v->stream)); // `host` goes out of scope
    return HV_OK;
}

} // extern "C"
