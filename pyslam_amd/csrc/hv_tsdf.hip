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

// One voxel update (UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier's inner body) in two phases: the evaluation
// (hv_tsdf_eval_fast) decides whether the voxel is updated and with what - it needs only the frame -, hv_tsdf_apply folds it into the
// voxel state.  Splitting them lets the kernels fetch voxel planes only for lanes that really update something.
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
// gathers in flight, no voxel-plane traffic; predicated, the multiplier computed per visit: this kernel waits on memory, the table's
// extra gather costs more than the VALU work it saves - 8.5 k frames/s against 7.9 k, 6.2-7.2 k with a branching evaluation); phase 2
// read-modify-writes only the 16-byte pieces that hold an updated voxel.  pc[][] is advanced by ZB z-steps.
template <int ZB>
__device__ __forceinline__ void hv_tsdf_slabs(const HvFrameParams &P, const uint2 *__restrict__ frame_px, char *__restrict__ unit, int wordb,
                                              float (&pc)[4][3], float inc0, float inc1, float inc2) {
    float tv[ZB][4];
    uint32_t cv[ZB][4];
    unsigned mask = 0;
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
                if (hv_tsdf_eval_fast<true, false>(P, frame_px, nullptr, pc[c][0], pc[c][1], pc[c][2], tv[zz][c], cv[zz][c]))
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
                if (hv_tsdf_eval_fast<false, false>(P, frame_px, nullptr, pc[c][0], pc[c][1], pc[c][2], tv[zz][c], cv[zz][c]))
                    mask |= 1u << (zz * 4 + c);
                pc[c][0] += inc0;
                pc[c][1] += inc1;
                pc[c][2] += inc2;
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

// The online sweep: evaluate first - predicated, all gathers of a lane in flight -, then fetch only the pieces that are updated.
// (The roofline ablations of rounds 1-2 - no plane traffic / plane traffic only - are recorded in profiles/r01, r02.)
__global__ __launch_bounds__(256) void k_tsdf_integrate(HvTable table, const int32_t *__restrict__ list,
                                                         int parity, char *__restrict__ pool,
                                                         const uint2 *__restrict__ frame_px, HvFrameParams P,
                                                         HvStatus *status, int32_t status_seq) {
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
        // 2 z-slabs evaluated per batch: (107 VGPRs, 4 waves/SIMD) measured 7269 frames/s vs 6342 for 4
        // (180 VGPRs, 2 waves/SIMD) on the headline config
        constexpr int ZB = 2;
#pragma unroll
        for (int zb = 0; zb < 4; zb += ZB) hv_tsdf_slabs<ZB>(P, frame_px, unit, (z0 + zb) * RR + x * R + y0, pc, inc0, inc1, inc2);
    }
}

// Pack role of the multi-frame path: pixels [i0, i0 + 4) of frame f -> frame records (4 pixels per thread: one 16-byte depth load -
// 8 for uint16 -, three dwords of RGB, two / three 16-byte record stores).
__device__ __forceinline__ void hv_pack_px4(const HvFrameParams &P, const int f, const int64_t i0, const void *depth_f,
                                            const uint8_t *rgb_f, uint2 *__restrict__ frame_px, const float *__restrict__ mult12) {
    const int64_t npx = (int64_t)P.H * P.W;
    if (i0 >= npx) return;
    if (P.tiled && (P.W & 3) == 0) {
        // tile-sharded volume: only voxels that project into this GPU's tile gather a record (the sweep's image-range test
        // uses the tile's bounds), so only the tile's columns and rows are packed (+ 4 pixels: a garbage lane may read
        // outside, its value is never used)
        const int u = (int)(i0 % P.W), v = (int)(i0 / P.W);
        if (u + 3 < P.tile_u0 - 4 || u >= P.tile_u1 + 4 || v < P.tile_v0 - 4 || v >= P.tile_v1 + 4) return;
    }
    uint2 *dst = frame_px + (int64_t)f * npx + i0;
    // mult12 != nullptr: 12-byte records {depth, colour, multiplier} (the column sweep gathers a voxel's pixel with ONE load; the
    // multiplier comes from the per-pixel table, which is built before this launch); else 8-byte {depth, colour} records beside the
    // table (the bitwise sweep form)
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

// Touch role of the multi-frame path: one wave = one 8x8 patch of frame f's depth samples; every distinct unit the patch opens gets
// bit f of its frame mask and - at its first touch in the batch - its place in the batch's union list.
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
                           const int32_t old = atomicExch(&stamp[slot], batch_stamp);
                           if (old != batch_stamp) {
                               const int32_t at = atomicAdd(&table.counters[HV_CNT_TOUCH(parity)], 1);
                               if (at < table.max_blocks) list[at] = slot;
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
//   sweep                    per union unit: a lane's voxels are folded over every frame bit in ascending (= chronological) order
//                            in registers and stored once - k_tsdf_sweep_column (production) or k_tsdf_sweep (bitwise)
//   k_tsdf_batch_finish      clears the batch's frame masks and list counter
// Identical results to B successive hv_tsdf_integrate calls: a unit is updated by frame f iff frame
// f touched it, and its frames are applied in order.  Plane traffic per frame drops by ~B x (the
// union of 32 consecutive frames' units is ~1.6x one frame's); what remains is the per-voxel math
// and the frame gathers.
// (Rounds 2-5 also carried a first column form, a 4-voxels-per-lane fold form, a launch fused with the next batch's touch + pack pass,
// a longest-task-first list order, a third scratch set and an image-coherent ownership plan; each measured slower than or equal to
// what is kept - profiles/r02 .. r05, sweep_forms.jsonl, pipeline_experiments.md, simulate_ranks_coherent.jsonl - and taken out in
// round 6.)
// ================================================================================================
__global__ __launch_bounds__(256) void k_tsdf_prep_touch_batch(HvTable table, int32_t *__restrict__ stamp,
                                                                unsigned long long *__restrict__ frame_mask,
                                                                int32_t *__restrict__ list, int batch_stamp,
                                                                const char *__restrict__ depth_raw, int64_t depth_stride,
                                                                const uint8_t *__restrict__ rgb,
                                                                uint2 *__restrict__ frame_px,
                                                                const HvFrameParams *__restrict__ Ps, int n_prep_blocks,
                                                                int n_touch_blocks, int n_frames, int parity,
                                                                const float *__restrict__ mult12) {
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
        hv_pack_px4(P, f, ((int64_t)bx * blockDim.x + threadIdx.x) * 4, depth_f, rgb_f, frame_px, mult12);
        return;
    }
    // ---- touch role: one wave per 8x8 sample patch (hv_touch_patch) ----
    __shared__ HvTouchScratch scratch[4];
    const int patch = bx * 4 + (int)(threadIdx.x / HV_WAVE);
    if (patch < hv_touch_patches(P))
        hv_touch_batch_patch(table, P, depth_f, patch, scratch[threadIdx.x / HV_WAVE], f, frame_mask, stamp, list, batch_stamp, parity);
}

// ================================================================================================
// Sweep, BITWISE form (HV_TSDF_SWEEP=2; the production form further down folds a batch per voxel and is not bitwise).  Lane = one
// voxel column, wave = 64 columns x 4 z, frames applied in order in registers with the reference's running mean after every frame:
// tsdf, weight and colour bit-identical to B successive online frames and to the oracle.  What the form does about instructions:
//  * the (u, v) projection chain is written on float2 values of ONE voxel ((pc0, pc1) walk, (a0, a1) / pc2 with the
//    refined reciprocal broadcast): v_pk_* instructions;
//  * gathers go through raw buffer descriptors (out-of-range offsets return 0: garbage lanes need no select), the pixel
//    index is a 24-bit mad;
//  * what does not depend on the frame (intrinsics, image size, truncation) is read once per work item, the 16 dwords
//    that do (extrinsics and the z step) are fetched one frame AHEAD into scalar registers;
//  * the near-camera-plane regime is detected once per work item (lane f tests frame f) and routes the whole item to the EXACT loop;
//  * the running mean of a voxel pair is skipped when no lane of the wave updates either voxel;
//  * colour is accumulated in two packed registers per voxel (16-bit r and b fields, g in bits 8..23: <= 64 frames x 255
//    fit) and the three colour planes are only loaded for the final add of voxels that were updated.
// Measured (profiles/r02): 31.4 k frames/s at 128 VGPRs / 4 waves per SIMD.
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
// Sweep, production form: the batch is FOLDED per voxel, on WHOLE voxel columns.  The contract asks for bit-exact unit keys and
// weights and 1e-4 on tsdf / colour (BASELINE.json north_star); the running mean (tsdf w + t) / (w + 1) applied once per accepted
// frame is, in real numbers, (tsdf w0 + sum t) / (w0 + n) - so a lane keeps `sum t` and n per voxel in registers over the batch's
// frames and divides ONCE when the item is done (measured against the oracle's per-frame float chain: <= 3e-7).  What decides
// WHETHER a frame updates a voxel - the projection, the pixel it lands in, the truncation test - is still evaluated with the
// reference's IEEE operations in the reference's order, so weights and colour sums stay exact integers.  Against the bitwise form a
// voxel visit loses the correctly rounded running-mean division, the float weight carry, and three tests the item-level regime test
// already implies:
//  * `pc2 > 0`: an item only takes this path when every voxel of the wave's box lies beyond `near_z` of every frame in its mask;
//  * `depth > 0`: with near_z >= 1.25 sdf_trunc and a multiplier >= 1, a record with depth <= 0 (invalid, truncated, or
//    the zero an out-of-range buffer load returns) gives sdf <= -pc2 < -sdf_trunc and fails the truncation test by itself;
//  * the image-tile test of the tile-sharded mode folds into the image-range compare (the range constants become the tile's).
// The observation count rides in byte 3 of the packed colour word (HV_REC_ONE, set by the pack role): one masked add accumulates
// green and the count.  tsdf / weight planes are only read when the item is done, and only by lanes with n > 0.
// A lane owns one (x, y) column of the unit and all 16 z of it, a wave 64 columns (one column group), a unit is 4 wave tasks:
//  * the z-walk starts at z = 0 for every lane: no replay of the reference's repeated float additions, and the per-frame set-up (the
//    rigid transform of the column's base point, the frame's buffer descriptor, the prefetch of the next frame's constants) is paid
//    once per 16 voxels: ~27 VALU instructions per voxel visit;
//  * the voxels of a column are evaluated 4 at a time; the gathers of group g+1 are issued BEFORE group g is folded, so a wave always
//    has 4 gathers in flight behind 4 voxels' worth of arithmetic;
//  * the accumulation runs under the EXEC mask of the accepting lanes (no selects; the compiler's execz branch skips a step no lane
//    accepts).
// 12-byte frame records ({depth, colour | 1 << 24, multiplier}: one gather per voxel visit).  Registers: 3 x 16 accumulators (sum t,
// r | b << 16, g << 8 | n << 24) + two gather groups: AT the 128 registers of 4 waves per SIMD.
// (Measured and dropped, rounds 3-5 - profiles/r03 .. r05: gather groups of 1 / 2 / 8 voxels, gathers pipelined across frames, the
// select form of the accumulation, 1 / 2 workgroups per unit, register caps of 120 / 112 / 5-6 waves per SIMD, 16-byte records.)
// Correction rounds of the projection's shared-reciprocal division.  hv_div2 (online path, bitwise form) runs two; with the reciprocal
// refined by one Newton step the FIRST round already returns the correctly rounded quotient on every operand pair tried:
// tools/divtest.hip, round 4 - 0 mismatches against IEEE division in 3.4e12 pairs each of the kernel's operand ranges, wide random
// exponents and divisors whose mantissa ends in runs of ones / zeros.  Two packed FMAs less per voxel visit.
// -DHV_SWEEP_DIV_ROUNDS=2 restores the second round.
#ifndef HV_SWEEP_DIV_ROUNDS
#define HV_SWEEP_DIV_ROUNDS 1
#endif
// hv_sweep_column_core: the sweep of ONE work item (unit slot, part) as the callable `run_item(slot, part)`, handed to `drive`, which
// decides what items this workgroup runs (k_tsdf_sweep_column: the grid-stride loop over the batch's union list).  Everything inlines.
template <int ZS, class Drive>
__device__ __forceinline__ void hv_sweep_column_core(
    const HvTable &table, const unsigned long long *__restrict__ frame_mask,
    char *__restrict__ pool, const uint2 *__restrict__ frame_px, const HvFrameParams *__restrict__ Ps, const int n_frames,
    const int general, const float *__restrict__ mult, Drive drive) {
    constexpr int SPLIT = 4;       // workgroups per unit: one wave each
    constexpr int GV = 4;          // voxels per gather group
    constexpr bool INTERIOR = true;
    // ZS = 2: a column is split into two z halves = 8 wave tasks per unit (a GPU that owns few units - multi-GPU sharding - has
    // ~3 000 column tasks of very different lengths for 4 096 wave slots: nothing evens them out; twice as many, half as long
    // tasks do).  The upper half replays the reference's 8 repeated float additions along z per frame.
    static_assert(ZS == 1 || ZS == 2, "whole columns or z halves");
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
                if (ok) { // the accumulation runs under the EXEC mask of the accepting lanes
                    asm volatile("" ::: "memory"); // keeps the branch: if-converted, the body is the select form again
                    S[z] += fminf(sdf * tinv, 1.0f); // == `if (t > 1) t = 1` for the non-NaN t of an accepted voxel
                    arb[z] += g.rec[k].y & 0x00ff00ffu;
                    agn[z] += g.rec[k].y & 0xff00ff00u;
                }
            }
        };
        HvSweepFrameK ka = hv_sweep_frame_k(Ps, __ffsll((long long)mask) - 1), kb = ka;
        auto fold_frame = [&](const HvSweepFrameK K, const int f, HvSweepFrameK &Knext, const unsigned long long rest) __attribute__((always_inline)) {
            begin_frame(K, f, Knext, rest);
            if (INTERIOR) frame_checked = !((interior_mask >> f) & 1ull);
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

// The column sweep's kernel: a grid-stride loop over the batch's union list, XCD-aware (see k_tsdf_sweep).  4 waves per SIMD = 128
// registers.  ZS = 2 (z halves, 8 one-wave tasks per unit) is the default for a GPU that shares the volume with 3 or more others.
template <int ZS>
__global__ __launch_bounds__(64, 4) void k_tsdf_sweep_column(
    HvTable table, const int32_t *__restrict__ list, const unsigned long long *__restrict__ frame_mask,
    char *__restrict__ pool, const uint2 *__restrict__ frame_px, const HvFrameParams *__restrict__ Ps, int n_frames,
    int general, const float *__restrict__ mult, int xcd_aware, int parity) {
    constexpr int PARTS = 4 * ZS; // work items per unit
    int n_units = table.counters[HV_CNT_TOUCH(parity)];
    if (n_units > table.max_blocks) n_units = table.max_blocks;
    hv_sweep_column_core<ZS>(table, frame_mask, pool, frame_px, Ps, n_frames, general, mult, [&](auto &&run_item) __attribute__((always_inline)) {
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
    const dim3 grid(8192), block(256); // > touched units of a frame: no second pass per block
    hv_profile_begin(v);
    hipLaunchKernelGGL(k_tsdf_integrate, grid, block, 0, v->stream, v->table, (const int32_t *)touched_list_of(v, parity), parity, (char *)v->pool,
                       (const uint2 *)frame_px_of(v, parity), P, v->d_status, hv_next_status_seq(v));
    hv_profile_end(v, 0);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

static void tsdf_next_frame(hv_volume *v, HvFrameParams &P, int &parity) {
    v->content_version += 1;
    v->frame_counter += 1;
    P.frame_id = v->frame_counter;
    parity = v->frame_counter & 1;
    v->last_touch_parity = parity;
    v->touch_counters_clean = false; // this parity's counter keeps the frame's touched count until the next frame's sweep
}

// hv_tsdf_set_rectify_maps: B device-resident frames through the camera's undistort / rectify maps on stream `s` (one launch), into
// the volume's rect_buf; depth / rgb then point at the rectified copies.  The buffer is single: the launch that reads it (touch + pack
// of the same batch) follows on the same stream, and the next batch's rectify launch follows that one (same stream) or the event the
// aux stream waits for (ev_presweep, recorded on the main stream behind it).
static int tsdf_rectify(hv_volume *v, hipStream_t s, const void **depth, int depth_dtype, const uint8_t **rgb, int B, int H, int W) {
    if (v->rect_W == 0) return HV_OK;
    HV_REQUIRE(v->rect_W == W && v->rect_H == H, HV_ERR_INVALID, "hv_tsdf_integrate: frames are %dx%d, the rectify maps %dx%d", W, H, v->rect_W, v->rect_H);
    const size_t npx = (size_t)H * W, dsz = depth_dtype == HV_DEPTH_U16 ? 2 : 4;
    const size_t depth_bytes = (npx * dsz * (size_t)B + 255) & ~(size_t)255;
    const size_t want = depth_bytes + npx * 3 * (size_t)B;
    if (v->rect_buf_bytes < want && v->stream_aux) HV_HIP(hipStreamSynchronize(v->stream_aux)); // (a launch on the aux stream may still read the old buffer)
    int rc = hv_ensure_buffer(v, &v->rect_buf, &v->rect_buf_bytes, want);
    if (rc != HV_OK) return rc;
    rc = hv_rectify_frames_device(v, s, *depth, depth_dtype, *rgb, B, H, W, v->rect_buf, (uint8_t *)v->rect_buf + depth_bytes);
    if (rc != HV_OK) return rc;
    *depth = v->rect_buf;
    *rgb = (const uint8_t *)v->rect_buf + depth_bytes;
    return HV_OK;
}

// Online path: both halves back to back on the volume's stream.  In checked mode (hv_capacity_gate: the pool's headroom is
// not known to cover this frame) the claims of the touch pass are verified before anything is fused: if some did not fit
// the pool grows and the touch pass runs again under a fresh stamp.
static int tsdf_integrate_one(hv_volume *v, const void *d_depth, int depth_dtype, const uint8_t *d_rgb,
                              int H, int W, const double *intr, const double *T_cw, double depth_scale,
                              double depth_trunc) {
    bool checked = false;
    int rc = hv_capacity_gate(v, &checked);
    if (rc != HV_OK) return rc;
    rc = tsdf_rectify(v, v->stream, &d_depth, depth_dtype, &d_rgb, 1, H, W);
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
    return tsdf_launch_integrate(v, P, parity);
}

static constexpr int HV_BATCH_MAX = 64; // frames per sweep (one bit per frame in a unit's mask)

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
    if (n_frames == 1) { // trivial batch: the online path
        if (host_set >= 0) HV_HIP(hipStreamWaitEvent(v->stream, v->hs_dev_ready[host_set], 0));
        rc = tsdf_integrate_one(v, d_depth, depth_dtype, (const uint8_t *)d_rgb, height, width, intr, T_cw, depth_scale, depth_trunc);
        if (rc != HV_OK) return rc;
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
    // the previous batch (content_version), the checked capacity mode or HV_TSDF_PIPELINE=0 runs everything on the main stream.
    const bool pipeline_on = !(getenv("HV_TSDF_PIPELINE") && atoi(getenv("HV_TSDF_PIPELINE")) == 0);
    // sweep form: 4 = k_tsdf_sweep_column (production: the batch folded per voxel, a lane walks a whole voxel column, tsdf within
    // 5e-6 of the per-frame chain), 2 = k_tsdf_sweep (the reference's running mean frame by frame: tsdf bit-identical to it).
    // The switches are read per call: the parity tests flip them inside one process.
    const int sweep_form = getenv("HV_TSDF_SWEEP") && atoi(getenv("HV_TSDF_SWEEP")) == 2 ? 2 : 4;
    const bool rec12 = sweep_form == 4; // 12-byte frame records {depth, colour, multiplier}; the bitwise form: 8-byte records + the table
    const size_t rec_bytes = rec12 ? 12 : 8;
    // 1: run the EXACT evaluation with integer weights everywhere (parity checks of the rare-regime code)
    const int general = getenv("HV_TSDF_BATCH_GENERAL") ? atoi(getenv("HV_TSDF_BATCH_GENERAL")) : 0;
    // z halves (8 tasks per unit) when this GPU shares the volume with 3 or more others: DESIGN section 4
    const int zs = getenv("HV_TSDF_SWEEP_ZS") ? atoi(getenv("HV_TSDF_SWEEP_ZS")) : (v->owner_world >= 4 ? 2 : 1);
    const int xcd_aware = 2;      // list entries per XCD group (k_tsdf_sweep: measured best, FETCH_SIZE halves)
    const int sweep_grid = 65536; // one work item each up to 16 384 units per batch, grid-stride beyond (measured: 29.6 k frames/s vs 28.2 k at 8192)
    bool chain_ok = v->pipe_armed && v->pipe_version == v->content_version; // nothing but batches since ev_presweep was recorded
    v->content_version += 1;
    const int BMAX = HV_BATCH_MAX;
    for (int f0 = 0; f0 < n_frames; f0 += BMAX) {
        const int B = std::min(BMAX, n_frames - f0);
        // pool headroom (grows here when more than half is known to be used; see hv_capacity_gate)
        bool checked = false;
        const int64_t max_before = v->cfg.max_blocks;
        rc = hv_capacity_gate(v, &checked);
        if (rc != HV_OK) return rc;
        if (v->cfg.max_blocks != max_before) chain_ok = false; // the pool grew: the stream was drained, start a fresh chain
        const bool overlap = pipeline_on && chain_ok && !checked;
        if (v->stream_aux == nullptr) {
            HV_HIP(hipStreamCreateWithFlags(&v->stream_aux, hipStreamNonBlocking)); // (queue priorities, CU masks: measured, no gain - profiles/r04, r05)
            HV_HIP(hipEventCreateWithFlags(&v->ev_prep, hipEventDisableTiming));
            HV_HIP(hipEventCreateWithFlags(&v->ev_presweep, hipEventDisableTiming));
        }
        hipStream_t ps = overlap ? v->stream_aux : v->stream; // where this batch's touch + pack launch goes
        bool overlap_this = overlap;
        // scratch set (two, alternating)
        const int parity = v->batch_parity & 1;
        v->batch_parity = parity ^ 1;
        int32_t *d_list = v->touched_list + (size_t)parity * (size_t)v->cfg.max_blocks;
        unsigned long long *d_mask_rw = (unsigned long long *)v->touched_mask + (size_t)parity * (size_t)v->table_capacity;
        // what this batch's touch + pack launch waits for: the batch that last used its scratch set is swept and finished =
        // the main stream up to just before the previous sweep
        if (overlap) HV_HIP(hipStreamWaitEvent(v->stream_aux, v->ev_presweep, 0));
        if (host_set >= 0) HV_HIP(hipStreamWaitEvent(ps, v->hs_dev_ready[host_set], 0)); // the frames have arrived
        // per-frame constants go through a ring of 4 pinned host buffers: a slot is only rewritten after the upload that last
        // used it has completed, so consecutive calls queue up on the stream without a host synchronisation
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
        }
        int batch_stamp = v->frame_counter;
        v->last_touch_parity = parity;
        {
            // (the pack role copies the multipliers into the 12-byte records: when the table has to be rebuilt - first call, other
            // intrinsics - it is rebuilt on the main stream and this batch's touch + pack launch follows it there)
            const float *before = v->mult_table;
            const int mw = v->mult_W, mh = v->mult_H;
            float key_before[4];
            memcpy(key_before, v->mult_key, sizeof(key_before));
            rc = tsdf_multiplier_table(v, params[0]);
            if (rc != HV_OK) return rc;
            if (before != v->mult_table || mw != v->mult_W || mh != v->mult_H || memcmp(key_before, v->mult_key, sizeof(key_before)) != 0) {
                if (ps != v->stream) HV_HIP(hipStreamSynchronize(v->stream_aux)); // nothing of an older batch still reads the old table there
                ps = v->stream;
                overlap_this = false;
                if (host_set >= 0) HV_HIP(hipStreamWaitEvent(ps, v->hs_dev_ready[host_set], 0));
            }
        }
        const float *d_mult = v->mult_table;
        // scratch: [B frame records of npx][B HvFrameParams]
        const size_t px_bytes = rec_bytes * npx * (size_t)B;
        void **bb = parity ? &v->batch_buf2 : &v->batch_buf;
        size_t *bb_bytes = parity ? &v->batch_buf2_bytes : &v->batch_buf_bytes;
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
        const int n_prep_blocks = (int)((npx + 1023) / 1024); // 4 pixels per thread
        const int n_touch_blocks = (hv_touch_patches(width, height, v->cfg.depth_sampling_stride) + 3) / 4;
        // this chunk's frames (through the camera's rectify maps first, when the volume has them: hv_tsdf_set_rectify_maps)
        const void *c_depth = (const char *)d_depth + npx * dsz * (size_t)f0;
        const uint8_t *c_rgb = (const uint8_t *)d_rgb + npx * 3 * (size_t)f0;
        rc = tsdf_rectify(v, ps, &c_depth, depth_dtype, &c_rgb, B, height, width);
        if (rc != HV_OK) return rc;
        for (int attempt = 0;; ++attempt) {
            // the touched-list counters are zero after hv_reset and k_tsdf_batch_finish; an online frame or an aborted claim pass
            // leaves its own parity's count behind (never while a chain runs)
            if (!v->touch_counters_clean) HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_TOUCH0], 0, HV_CNT_TOUCH_SPAN_BYTES, v->stream));
            v->touch_counters_clean = true;
            hipLaunchKernelGGL(k_tsdf_prep_touch_batch, dim3((n_prep_blocks + n_touch_blocks) * B), dim3(256), 0, ps,
                               v->table, v->touched_stamp, d_mask_rw, d_list, batch_stamp,
                               (const char *)c_depth, (int64_t)(npx * dsz), c_rgb, d_px, (const HvFrameParams *)d_params, n_prep_blocks, n_touch_blocks, B, parity,
                               rec12 ? d_mult : (const float *)nullptr);
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
        }
        // What the NEXT batch's touch + pack launch waits for: everything the main stream holds up to here, i.e. the finish of the
        // batch that last used the next batch's scratch set.  Recorded BEFORE the main stream waits for this batch's own touch + pack
        // launch: the aux stream is in order, so the next launch follows this batch's there anyway, and recorded after it every touch +
        // pack launch also waited for the cross-queue hand-off into the sweep before it (profiles/r04/rank8_timeline.txt).
        HV_HIP(hipEventRecord(v->ev_presweep, v->stream));
        if (overlap_this) {
            HV_HIP(hipEventRecord(v->ev_prep, v->stream_aux));
            HV_HIP(hipStreamWaitEvent(v->stream, v->ev_prep, 0));
        }
        chain_ok = true; // (the next chunk of this call may follow this one directly)
        hv_profile_begin(v);
        const unsigned long long *d_mask = d_mask_rw;
        if (sweep_form == 2)
            hipLaunchKernelGGL((k_tsdf_sweep<4, 8, 4>), dim3(sweep_grid), dim3(64 * 16 / 8), 0, v->stream, v->table, (const int32_t *)d_list, d_mask,
                               (char *)v->pool, (const uint2 *)d_px, (const HvFrameParams *)d_params, B, general, d_mult, xcd_aware, parity);
        else if (zs == 2)
            hipLaunchKernelGGL(k_tsdf_sweep_column<2>, dim3(sweep_grid), dim3(64), 0, v->stream, v->table, (const int32_t *)d_list, d_mask,
                               (char *)v->pool, (const uint2 *)d_px, (const HvFrameParams *)d_params, B, general, d_mult, xcd_aware, parity);
        else
            hipLaunchKernelGGL(k_tsdf_sweep_column<1>, dim3(sweep_grid), dim3(64), 0, v->stream, v->table, (const int32_t *)d_list, d_mask,
                               (char *)v->pool, (const uint2 *)d_px, (const HvFrameParams *)d_params, B, general, d_mult, xcd_aware, parity);
        hv_profile_end(v, B);
        hipLaunchKernelGGL(k_tsdf_batch_finish, dim3(1), dim3(1024), 0, v->stream, v->table, (const int32_t *)d_list, d_mask_rw, parity, v->d_status,
                           hv_next_status_seq(v));
        HV_HIP(hipGetLastError());
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

int hv_tsdf_set_color_order(hv_volume *v, int32_t bgr) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_tsdf_set_color_order: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_set_color_order: volume is not in TSDF mode");
    v->color_bgr = bgr ? 1 : 0;
    return HV_OK;
}

int hv_tsdf_set_owner(hv_volume *v, int32_t rank, int32_t world_size) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_tsdf_set_owner: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_set_owner: volume is not in TSDF mode");
    HV_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, HV_ERR_INVALID, "hv_tsdf_set_owner: bad rank/world");
    v->owner_rank = rank;
    v->owner_world = world_size;
    return HV_OK;
}

int hv_tsdf_dump(hv_volume *v, int32_t *keys, float *tsdf, float *weight, double *color, int64_t *n_units) {
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
    HV_REQUIRE(v != nullptr && (k == 0 || (keys != nullptr && payload != nullptr)), HV_ERR_INVALID,
               "hv_tsdf_import_numerators: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_import_numerators: not a TSDF volume");
    if (k == 0) return HV_OK;
    v->content_version += 1;
    v->extract_epoch += 1; // (writes voxels without stamping their units)
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
    HV_REQUIRE(v != nullptr && (k == 0 || (shared_keys != nullptr && payload != nullptr && action != nullptr)), HV_ERR_INVALID,
               "hv_merge_halo_unpack: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_merge_halo_unpack: not a TSDF volume");
    if (k == 0) return HV_OK;
    v->content_version += 1;
    v->extract_epoch += 1; // (writes voxels without stamping their units)
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

// ... with the plan hv_merge_halo_plan_device left in the volume (hv_halo.hip): units [first, first + count) of it, device payload,
// nothing staged, nothing waited for - the caller orders its collective against the volume's stream
int hv_merge_halo_pack_planned(hv_volume *v, int64_t first, int64_t count, float *d_payload) {
    HV_REQUIRE(v != nullptr && (count == 0 || d_payload != nullptr), HV_ERR_INVALID, "hv_merge_halo_pack_planned: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_merge_halo_pack_planned: not a TSDF volume");
    HV_REQUIRE(first >= 0 && count >= 0 && first + count <= v->halo_plan_n, HV_ERR_INVALID, "hv_merge_halo_pack_planned: range outside the plan");
    if (count == 0) return HV_OK;
    HV_HIP(hipSetDevice(v->device));
    const int64_t total = count * RRR;
    hipLaunchKernelGGL(k_tsdf_export, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->stream, v->table, (const char *)v->pool,
                       (const int32_t *)v->halo_plan + 3 * first, count, d_payload);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

int hv_merge_halo_unpack_planned(hv_volume *v, int64_t first, int64_t count, const float *d_payload) {
    HV_REQUIRE(v != nullptr && (count == 0 || d_payload != nullptr), HV_ERR_INVALID, "hv_merge_halo_unpack_planned: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_merge_halo_unpack_planned: not a TSDF volume");
    HV_REQUIRE(first >= 0 && count >= 0 && first + count <= v->halo_plan_n, HV_ERR_INVALID, "hv_merge_halo_unpack_planned: range outside the plan");
    if (count == 0) return HV_OK;
    v->content_version += 1;
    v->extract_epoch += 1; // (writes voxels without stamping their units)
    HV_HIP(hipSetDevice(v->device));
    const int64_t total = count * RRR;
    hipLaunchKernelGGL(k_tsdf_halo_unpack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->stream, v->table, (char *)v->pool,
                       (const int32_t *)v->halo_plan + 3 * first, count, d_payload,
                       (const uint8_t *)v->halo_plan + (size_t)v->halo_plan_n * 12 + first);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

} // extern "C"
