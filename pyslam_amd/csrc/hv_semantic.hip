// libpyslam_hipvol.so — semantic block grids of pySLAM's cpp/volumetric on the GPU block hash (gfx950):
//   VOXEL_SEMANTIC_GRID                = VoxelBlockSemanticGrid              (voting payload,
//                                        VoxelSemanticData, voxel_data_semantic.h:106-202)
//   VOXEL_SEMANTIC_PROBABILISTIC_GRID  = VoxelBlockSemanticProbabilisticGrid (log-probability payload,
//                                        VoxelSemanticDataProbabilistic, voxel_data_semantic.h:249-672)
//
// Label fusion is order dependent in both payloads (voting: a conflicting observation decrements the
// confidence counter and may switch the label, :175-191; probabilistic: float log-evidence sums and
// the incremental arg-max with its tie rules, :358-417), so — exactly as for the plain grid — points
// are grouped per voxel with a *stable* device radix sort and the head thread of every voxel run
// folds its points in point-index order: labels, counters / log-probabilities, counts and the float64
// position sums come out bit-identical to the reference's sequential branch.
//
// Records: voting 64 B {count, object_id+1, class_id+1, confidence_counter, position_sum f64[3],
// color_sum f32[3], pad}; probabilistic 128 B with 6 inline label slots + a chain of overflow nodes (hv_semantic.h).
#include <algorithm>
#include <array>
#include <cmath>
#include <numeric>
#include <type_traits>

#include "hv_common.h"
#include "hv_query.h"
#include "hv_semantic.h"
#include "hv_bins.h"
#include "hv_unproject.h"
#include <rocprim/device/device_radix_sort.hpp>

static constexpr uint32_t HV_SORT_SENTINEL = 0xFFFFFFFFu;

__host__ __device__ static inline int32_t sem_floor_div(int32_t a, int32_t b) {
    const int64_t aa = a, bb = b;
    return (int32_t)((aa >= 0) ? (aa / bb) : ((aa - bb + 1) / bb));
}

// get_voxel_key_inv<Tpos,Tpos>(x, inv_voxel_size_) as update_voxel calls it (voxel_block_grid.hpp:473):
// float points -> float product; double points -> double product with the float member promoted.
__device__ __forceinline__ int32_t sem_voxel_coord(float x, float inv) { return (int32_t)floorf(x * inv); }
__device__ __forceinline__ int32_t sem_voxel_coord(double x, float inv) { return (int32_t)floor(x * (double)inv); }

template <typename PT>
__global__ __launch_bounds__(256) void k_sem_keys(HvTable table, const PT *__restrict__ pts, int64_t n, HvSemParams G,
                                                   uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                   const uint32_t *__restrict__ valid_mask_keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vals_out[i] = (uint32_t)i;
    if (valid_mask_keys != nullptr && valid_mask_keys[i] == HV_SORT_SENTINEL) { // pixel rejected by the unprojection
        keys_out[i] = HV_SORT_SENTINEL;
        return;
    }
    const PT p[3] = {pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2]};
    uint32_t key = HV_SORT_SENTINEL;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) ok = ok && isfinite((double)p[a]) && fabs((double)p[a] * (double)G.inv_voxel_size) < 1.0e9;
    if (ok) {
        int32_t b[3], l[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int32_t v = sem_voxel_coord(p[a], G.inv_voxel_size);
            b[a] = sem_floor_div(v, G.bs);
            l[a] = (int32_t)((int64_t)v - (int64_t)b[a] * G.bs);
        }
        if (hv_key_in_range(b[0], b[1], b[2])) {
            const unsigned long long bkey = hv_pack_key(b[0], b[1], b[2]);
            if (G.owner_world > 1 && hv_owner_of(bkey, G.owner_world) != G.owner_rank) { // another GPU's block: not this GPU's point
                keys_out[i] = HV_SORT_SENTINEL;
                return;
            }
            const int32_t slot = hv_table_insert(table, bkey);
            if (slot >= 0) key = ((uint32_t)slot << G.local_bits) | (uint32_t)(l[0] + l[1] * G.bs + l[2] * G.bs * G.bs);
        }
    }
    if (key == HV_SORT_SENTINEL) atomicAdd(&table.counters[HV_CNT_DROPPED], 1);
    keys_out[i] = key;
}

// What the fold learns about one point.  Two sources: the caller's arrays (points float32 | float64, colours none | uint8 | float32,
// class / instance ids, depths - the binding's integrate overloads, volumetric_grid_module.h:131-467), or the packed 32-byte record
// the frame entry point's bin pass leaves per pixel (round 6: two 16-byte loads per point where the arrays take seven scalar ones).
struct HvSemPoint {
    double x, y, z;
    float c0, c1, c2;
    int32_t cls, obj;
    float depth;
};
template <typename PT, int COLOR_KIND>
struct HvSemArrays {
    const PT *__restrict__ pts;
    const void *__restrict__ cols;
    const int32_t *__restrict__ class_ids;
    const int32_t *__restrict__ instance_ids;
    const float *__restrict__ depths;
    static constexpr bool kColors = COLOR_KIND != HV_COLOR_NONE;
    __device__ __forceinline__ bool has_labels() const { return class_ids != nullptr; }
    __device__ __forceinline__ bool has_depth() const { return depths != nullptr; }
    __device__ __forceinline__ HvSemPoint get(int64_t p) const {
        HvSemPoint q;
        q.x = (double)pts[p * 3 + 0];
        q.y = (double)pts[p * 3 + 1];
        q.z = (double)pts[p * 3 + 2];
        q.c0 = q.c1 = q.c2 = 0.f;
        const float inv_255 = 1.0f / 255.0f; // voxel_data.h:82
        if (COLOR_KIND == HV_COLOR_U8) {
            const uint8_t *c = (const uint8_t *)cols + p * 3;
            q.c0 = (float)c[0] * inv_255;
            q.c1 = (float)c[1] * inv_255;
            q.c2 = (float)c[2] * inv_255;
        } else if (COLOR_KIND == HV_COLOR_F32) {
            const float *c = (const float *)cols + p * 3;
            q.c0 = c[0];
            q.c1 = c[1];
            q.c2 = c[2];
        }
        q.cls = class_ids ? class_ids[p] : 0;
        q.obj = instance_ids ? instance_ids[p] : 0;
        q.depth = depths ? depths[p] : 0.0f;
        return q;
    }
};
// record: {x, y, z, rgb8 | class, instance, depth, -}; colours through the fold's 256-entry table of (float)(c / 255.0) (depth.py:76)
struct HvSemRecs {
    const float4 *__restrict__ rec;
    const float *lut; // LDS
    bool labels, depth;
    static constexpr bool kColors = true;
    __device__ __forceinline__ bool has_labels() const { return labels; }
    __device__ __forceinline__ bool has_depth() const { return depth; }
    __device__ __forceinline__ HvSemPoint get(int64_t p) const {
        const float4 a = rec[2 * p], b = rec[2 * p + 1];
        HvSemPoint q;
        q.x = (double)a.x;
        q.y = (double)a.y;
        q.z = (double)a.z;
        const uint32_t c = __float_as_uint(a.w);
        q.c0 = lut[c & 255u];
        q.c1 = lut[(c >> 8) & 255u];
        q.c2 = lut[(c >> 16) & 255u];
        q.cls = (int32_t)__float_as_uint(b.x);
        q.obj = (int32_t)__float_as_uint(b.y);
        q.depth = b.z;
        return q;
    }
};

// ... and the same records after a wave has fetched its bin's into LDS, one lane per point, all loads in flight together (a run's head
// lane then adds from LDS instead of chasing two global loads per point, one point after the other); indexed by sorted position
struct HvSemStaged {
    const float4 *stage; // LDS
    const float *lut;    // LDS
    bool labels, depth;
    static constexpr bool kColors = true;
    __device__ __forceinline__ bool has_labels() const { return labels; }
    __device__ __forceinline__ bool has_depth() const { return depth; }
    __device__ __forceinline__ HvSemPoint get(int64_t e) const {
        const float4 a = stage[2 * e], b = stage[2 * e + 1];
        HvSemPoint q;
        q.x = (double)a.x;
        q.y = (double)a.y;
        q.z = (double)a.z;
        const uint32_t c = __float_as_uint(a.w);
        q.c0 = lut[c & 255u];
        q.c1 = lut[(c >> 8) & 255u];
        q.c2 = lut[(c >> 16) & 255u];
        q.cls = (int32_t)__float_as_uint(b.x);
        q.obj = (int32_t)__float_as_uint(b.y);
        q.depth = b.z;
        return q;
    }
};
static constexpr int HV_SEMB_STAGE = 128; // bins up to this size fold from staged records (32 bytes each in the sort's spare windows)

// update_voxel_direct (voxel_block_grid.hpp:524-614) for a SemanticVoxelWithDepth payload: one voxel's points folded in
// point-index order.  `next(j)` yields the point index of the run's j-th entry, or -1 at its end.  VOX = HvSemVoxel (voting),
// HvProbVoxel (probabilistic) or one of the two "*2" payloads (hv_semantic.h).
template <typename VOX, typename SRC, typename Next>
__device__ __forceinline__ void sem_fold_run(const HvTable &table, VOX *__restrict__ pool, int64_t vid, const HvSemParams &G,
                                             const SRC &src, unsigned long long *__restrict__ occ, Next next) {
    // the voxel is read once, folded in registers and written once: with the label state updated in memory point by point the
    // loads of the next point could not be issued before the stores of this one (they may alias): one memory round trip per point
    VOX acc = pool[vid];
    int32_t count = acc.count;
    if (count == 0) atomicOr(&occ[vid >> 6], 1ull << (vid & 63)); // first point of this voxel (or the first after a reset)
    int overflowed = 0;
    const bool labels = src.has_labels(), with_depth = src.has_depth();
    for (int j = 0;; ++j) {
        const int64_t p = next(j);
        if (p < 0) break;
        const HvSemPoint q = src.get(p);
        acc.pos[0] += q.x;
        acc.pos[1] += q.y;
        acc.pos[2] += q.z;
        if (SRC::kColors) {
            acc.col[0] += q.c0;
            acc.col[1] += q.c1;
            acc.col[2] += q.c2;
        }
        if (labels) {
            const int32_t obj = q.obj, cls = q.cls;
            if constexpr (HvPay<VOX>::kind == 0) {
                HvSemVoxel *sv = (HvSemVoxel *)&acc;
                const bool gate = with_depth ? (q.depth < G.depth_threshold) : true; // *_with_depth, voxel_data_semantic.h:168-198
                if (count == 0) {
                    if (gate) { // initialize_semantics
                        sv->obj1 = obj + 1;
                        sv->cls1 = cls + 1;
                        sv->counter = 1;
                    }
                } else if (gate) { // update_semantics
                    if (sv->obj1 == obj + 1 && sv->cls1 == cls + 1) {
                        sv->counter++;
                    } else {
                        sv->counter--;
                        if (sv->counter <= 0) {
                            sv->obj1 = obj + 1;
                            sv->cls1 = cls + 1;
                            sv->counter = 1;
                        }
                    }
                }
            } else if constexpr (HvPay<VOX>::kind == 2) {
                sem2_fold(&acc, count == 0, with_depth ? (q.depth < G.depth_threshold) : true, obj, cls);
            } else if constexpr (HvPay<VOX>::kind == 3) {
                overflowed += prob2_fold(&acc, table, count == 0, obj, cls, prob2_observation_log_prob(with_depth, q.depth, G));
            } else {
                const float lp = prob_observation_log_prob(with_depth, q.depth, G);
                if (!prob_fold(&acc, table, count == 0, obj, cls, lp)) ++overflowed;
            }
        }
        count = count == 0 ? 1 : count + 1;
    }
    acc.count = count;
    pool[vid] = acc;
    if (overflowed) atomicAdd(&table.counters[HV_CNT_LABEL_OVERFLOW], overflowed);
}

// radix path: the run of sorted key i (head thread only)
template <typename VOX, typename PT, int COLOR_KIND>
__global__ __launch_bounds__(256) void k_sem_reduce(HvTable table, VOX *__restrict__ pool,
                                                     const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                     int64_t n, HvSemParams G, const PT *__restrict__ pts,
                                                     const void *__restrict__ cols, const int32_t *__restrict__ class_ids,
                                                     const int32_t *__restrict__ instance_ids,
                                                     const float *__restrict__ depths, unsigned long long *__restrict__ occ) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = keys[i];
    if (key == HV_SORT_SENTINEL) return;
    if (i > 0 && keys[i - 1] == key) return;
    const int32_t idx = table.vals[(int32_t)(key >> G.local_bits)];
    if (idx < 0) return;
    const int64_t vid = (int64_t)idx * G.nvox + (key & ((1u << G.local_bits) - 1u));
    const HvSemArrays<PT, COLOR_KIND> src{pts, cols, class_ids, instance_ids, depths};
    sem_fold_run<VOX>(table, pool, vid, G, src, occ,
                      [&](int j) -> int64_t { return (i + j < n && keys[i + j] == key) ? (int64_t)vals[i + j] : -1; });
}

// ---- per-keyframe bin path (hv_bins.h): bin -> fold (+ the big bins' ranges), 3 launches instead of the radix sort's ~20 (rounds 4-5:
// count -> offsets -> scatter -> fold + tasks, and a separate unprojection launch in front of them) ----
// entries are (local voxel index << IB | point index) with IB = 32 - local_bits (23 for 8^3 blocks: 8 M points per call)
__device__ __forceinline__ bool sem_point_block(const HvSemParams &G, double px, double py, double pz, bool is_f64, unsigned long long &bkey,
                                                uint32_t &lidx, bool &foreign) {
    const double p[3] = {px, py, pz};
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) ok = ok && isfinite(p[a]) && fabs(p[a] * (double)G.inv_voxel_size) < 1.0e9;
    if (!ok) return false;
    int32_t b[3], l[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int32_t v = is_f64 ? sem_voxel_coord(p[a], G.inv_voxel_size) : sem_voxel_coord((float)p[a], G.inv_voxel_size);
        b[a] = sem_floor_div(v, G.bs);
        l[a] = (int32_t)((int64_t)v - (int64_t)b[a] * G.bs);
    }
    if (!hv_key_in_range(b[0], b[1], b[2])) return false;
    bkey = hv_pack_key(b[0], b[1], b[2]);
    foreign = G.owner_world > 1 && hv_owner_of(bkey, G.owner_world) != G.owner_rank;
    lidx = (uint32_t)(l[0] + l[1] * G.bs + l[2] * G.bs * G.bs);
    return !foreign;
}

template <typename PT>
__global__ __launch_bounds__(HV_BIN_THREADS) void k_semb_bin(HvTable table, HvBins B, const PT *__restrict__ pts, int64_t n, HvSemParams G,
                                                   const uint32_t *__restrict__ valid_mask_keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool has = false;
    unsigned long long bkey = 0ull;
    uint32_t lidx = 0;
    if (i < n && !(valid_mask_keys != nullptr && valid_mask_keys[i] == HV_SORT_SENTINEL)) {
        bool foreign = false;
        has = sem_point_block(G, (double)pts[i * 3 + 0], (double)pts[i * 3 + 1], (double)pts[i * 3 + 2], sizeof(PT) == 8, bkey, lidx, foreign);
        if (!has && !foreign) atomicAdd(&table.counters[HV_CNT_DROPPED], 1);
    }
    hv_bins_push(table, B, has, bkey, lidx, (uint32_t)i);
}

// The frame entry point's bin pass: the thread unprojects its pixel (depth2pointcloud + world transform, hv_unproject.h), packs
// position, colour, labels and depth into the pixel's 32-byte record and goes on with the key - the unprojection launch, its 24-byte
// point / colour rows and the fold's seven scalar gathers per point are gone.
__global__ __launch_bounds__(HV_BIN_THREADS) void k_semb_bin_frame(HvTable table, HvBins B, HvSemParams G, HvUnprojectParams U,
                                                        const void *__restrict__ depth_raw, const uint8_t *__restrict__ rgb,
                                                        const int32_t *__restrict__ cls_img, const int32_t *__restrict__ obj_img,
                                                        float4 *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool has = false;
    unsigned long long bkey = 0ull;
    uint32_t lidx = 0;
    float pt[3];
    if (i < (int64_t)U.H * U.W && hv_unproject_point(U, depth_raw, i, pt)) {
        bool foreign = false;
        has = sem_point_block(G, (double)pt[0], (double)pt[1], (double)pt[2], false, bkey, lidx, foreign);
        if (!has && !foreign) atomicAdd(&table.counters[HV_CNT_DROPPED], 1);
        if (has) {
            const uint8_t *c = rgb + i * 3;
            const uint32_t packed = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
            const int32_t cls = cls_img ? cls_img[i] : 0, obj = obj_img ? obj_img[i] : 0;
            // depths = camera z of the point = the pixel's depth (…voxel_semantic_grid.py:418-424)
            const float d = U.depth_is_u16 ? (float)((const uint16_t *)depth_raw)[i] : ((const float *)depth_raw)[i];
            rec[2 * i] = make_float4(pt[0], pt[1], pt[2], __uint_as_float(packed));
            rec[2 * i + 1] = make_float4(__uint_as_float((uint32_t)cls), __uint_as_float((uint32_t)obj), d, 0.f);
        }
    }
    hv_bins_push(table, B, has, bkey, lidx, (uint32_t)i);
}

// Entries s_src[0 .. m) of ONE block into (voxel, point index) order, by a wave, in its LDS window: counting sort by voxel
// (histogram, prefix, scatter) and a rank inside each voxel's short run - a handful of passes over m entries instead of the
// O(log^2) passes of a bitonic sort with a wave barrier each (which cost 9 us for a 256-entry bucket).  Runs longer than 48 (coarse
// voxels) take the bitonic sort.  off[] needs nvox + 1 words; the sorted entries end up in s_src.
__device__ __forceinline__ void semb_sort_window(uint32_t *s_src, uint32_t *s_dst, uint32_t *off, int m, int idx_bits, int nvox) {
    const int lane = hv_lane_id();
    for (int v = lane; v <= nvox; v += HV_WAVE) off[v] = 0u;
    hv_wave_lds_sync();
    for (int e = lane; e < m; e += HV_WAVE) atomicAdd(&off[s_src[e] >> idx_bits], 1u);
    hv_wave_lds_sync();
    // exclusive prefix over the voxels (lane l: voxels [l * per, (l + 1) * per)), the longest run on the way
    const int per = (nvox + HV_WAVE - 1) / HV_WAVE;
    uint32_t sum = 0, longest = 0;
    for (int k = 0; k < per; ++k) {
        const int v = lane * per + k;
        const uint32_t c = v < nvox ? off[v] : 0u;
        sum += c;
        longest = max(longest, c);
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < HV_WAVE; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, o));
    if (longest > 48u) { // long runs: the rank pass below is quadratic in the run length
        int m2 = HV_WAVE;
        while (m2 < m) m2 <<= 1;
        for (int e = m + lane; e < m2; e += HV_WAVE) s_src[e] = 0xFFFFFFFFu;
        hv_wave_lds_sync();
        hv_vgb_bitonic_wave(s_src, m2);
        return;
    }
    uint32_t run = incl - sum;
    hv_wave_lds_sync();
    for (int k = 0; k < per; ++k) {
        const int v = lane * per + k;
        if (v < nvox) {
            const uint32_t c = off[v];
            off[v] = run; // start of voxel v's run (the scatter advances it to the run's end = the next run's start)
            run += c;
        }
    }
    hv_wave_lds_sync();
    for (int e = lane; e < m; e += HV_WAVE) {
        const uint32_t ent = s_src[e];
        s_dst[atomicAdd(&off[ent >> idx_bits], 1u)] = ent;
    }
    hv_wave_lds_sync();
    for (int q = lane; q < m; q += HV_WAVE) {
        const uint32_t ent = s_dst[q];
        const uint32_t v = ent >> idx_bits;
        const int a0 = v == 0u ? 0 : (int)off[v - 1], b0 = (int)off[v];
        int rank = 0;
        for (int r = a0; r < b0; ++r) rank += s_dst[r] < ent;
        s_src[a0 + rank] = ent;
    }
    hv_wave_lds_sync();
}

// (register budget of the two fold kernels: 4 waves per SIMD = 128 registers.  With the probabilistic voxel held in registers - round
// 5, hv_semantic.h - the allocator would take 130-134 and drop to 3 waves)
#ifndef HV_SEMB_FOLD_MIN_WAVES
#define HV_SEMB_FOLD_MIN_WAVES 4
#endif
// The fold's point source on the device: the arrays as they are, or the records with the workgroup's colour table (by all 256 threads).
template <typename PT, int COLOR_KIND>
__device__ __forceinline__ HvSemArrays<PT, COLOR_KIND> sem_src_prepare(const HvSemArrays<PT, COLOR_KIND> &src, float *) { return src; }
__device__ __forceinline__ HvSemRecs sem_src_prepare(const HvSemRecs &src, float *lut) {
    lut[threadIdx.x] = (float)((double)threadIdx.x / 255.0); // depth.py:76 (uint8 image / 255.0), cast to float32 by voxel_semantic_grid.py
    __syncthreads();
    HvSemRecs r = src;
    r.lut = lut;
    return r;
}

// One wave per touched block: its bin is brought into (voxel, point index) order in the wave's LDS window (semb_sort_window)
// and the head lane of every voxel run folds the run in point order - the reference's sequential order, bit-identical to the radix
// path.  Bins beyond the window go to k_semb_fold_tasks (or, without a task list, through voxel ranges / point windows here).
template <typename VOX, typename SRC>
__global__ __launch_bounds__(256, HV_SEMB_FOLD_MIN_WAVES) void k_semb_fold_wave(HvTable table, VOX *__restrict__ pool, HvBins B, HvSemParams G, SRC src_in,
                                                         unsigned long long *__restrict__ occ, int64_t n_points, HvStatus *status, int32_t status_seq,
                                                         int32_t *__restrict__ task_count, int4 *__restrict__ tasks, int task_cap, int wcap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[]; // per wave: [wcap src][wcap dst][nvox + 64 offsets]; wcap = the window, a power of two (HV_SEM_WCAP)
    __shared__ float s_lut[256];
    const SRC src = sem_src_prepare(src_in, s_lut);
    const int idx_bits = B.idx_bits, parity = B.parity;
    const HvBinLists L = hv_bins_lists(B);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        task_count[parity ^ 1] = 0;        // the next call's task list
        hv_bins_clear_next(B);             // the next call's list lengths
        table.counters[HV_CNT_OUT2] = 0;   // (hv_bins_push's largest-bin mark: the association uses this counter too)
        hv_publish_status(table, status, status_seq);
    }
    const int wave = threadIdx.x >> 6, lane = hv_lane_id();
    const int per_wave = 2 * wcap + G.nvox + 64;
    uint32_t *s = s_dyn + wave * per_wave, *s_dst = s + wcap, *off = s + 2 * wcap;
    const uint32_t idx_mask = (1u << idx_bits) - 1u;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    auto fold_sorted = [&](int m, int64_t block_base) {
        for (int e = lane; e < m; e += HV_WAVE) {
            const uint32_t lidx = s[e] >> idx_bits;
            if (e > 0 && (s[e - 1] >> idx_bits) == lidx) continue; // not the head of its voxel's run
            sem_fold_run<VOX>(table, pool, block_base + lidx, G, src, occ,
                              [&](int j) -> int64_t { return (e + j < m && (s[e + j] >> idx_bits) == lidx) ? (int64_t)(s[e + j] & idx_mask) : -1; });
        }
    };
    const bool can_stage = (size_t)(wcap + G.nvox + 64) * sizeof(uint32_t) >= (size_t)HV_SEMB_STAGE * 32;
    hv_bins_for_each(table, B, L, blockIdx.x * 4 + wave, gridDim.x * 4, [&](const HvBinHeader &h) {
        const int32_t slot = h.slot, nb = h.nb, idx = h.idx;
        hv_wave_lds_sync(); // the window of the previous bin is no longer read
        if (lane == 0) B.cnt[slot] = 0; // clean for the next call
        if (idx < 0) return;            // (the block did not get a pool slot: overflow, reported by the caller)
        const int64_t block_base = (int64_t)idx * G.nvox;
        if (nb <= wcap) {
            if (nb <= HV_WAVE) {
                // a bin of one entry per lane (a 2 mm keyframe: 12 points per block on average) is ranked in registers - an entry's
                // place is the number of smaller entries, one readlane + compare per entry of the bin - instead of the counting sort's
                // passes over the block's 512 voxel offsets (clear, histogram, prefix, scatter, rank: six wave barriers) (round 6)
                const uint32_t mine = lane < nb ? h.head : 0xFFFFFFFFu; // (entries are distinct: voxel << idx_bits | point index)
                int rank = 0;
                for (int j = 0; j < nb; ++j) rank += ((uint32_t)__builtin_amdgcn_readlane((int)mine, j) < mine) ? 1 : 0;
                if (lane < nb) s[rank] = mine;
                hv_wave_lds_sync();
            } else {
                if (lane < nb) s[lane] = h.head; // (the header brought the bin's first 64 entries)
                for (int e = HV_WAVE + lane; e < nb; e += HV_WAVE) s[e] = hv_bins_entry(B, slot, e);
                hv_wave_lds_sync();
                semb_sort_window(s, s_dst, off, nb, idx_bits, G.nvox);
            }
            if constexpr (std::is_same<SRC, HvSemRecs>::value) {
                if (nb <= HV_SEMB_STAGE && can_stage) {
                    // every lane fetches the records of ITS sorted entries (all of the bin's in flight at once) into the windows the sort
                    // no longer needs, and - as the head of a voxel run - asks for the voxel's line, which then arrives with the records
                    float4 *stage = (float4 *)s_dst;
                    int32_t warm = 0;
                    static_assert(HV_SEMB_STAGE == 2 * HV_WAVE, "two sorted entries per lane");
                    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 ra0 = zero4, rb0 = zero4, ra1 = zero4, rb1 = zero4; // (named registers: an array indexed under a condition went to scratch)
                    const int e0 = lane, e1 = lane + HV_WAVE;
                    if (e0 < nb) {
                        const uint32_t ent = s[e0];
                        const int64_t p = ent & idx_mask;
                        ra0 = src.rec[2 * p];
                        rb0 = src.rec[2 * p + 1];
                        if (e0 == 0 || (s[e0 - 1] >> idx_bits) != (ent >> idx_bits)) warm += pool[block_base + (ent >> idx_bits)].count;
                    }
                    if (e1 < nb) {
                        const uint32_t ent = s[e1];
                        const int64_t p = ent & idx_mask;
                        ra1 = src.rec[2 * p];
                        rb1 = src.rec[2 * p + 1];
                        if ((s[e1 - 1] >> idx_bits) != (ent >> idx_bits)) warm += pool[block_base + (ent >> idx_bits)].count;
                    }
                    if (e0 < nb) {
                        stage[2 * e0] = ra0;
                        stage[2 * e0 + 1] = rb0;
                    }
                    if (e1 < nb) {
                        stage[2 * e1] = ra1;
                        stage[2 * e1 + 1] = rb1;
                    }
                    asm volatile("" ::"v"(warm)); // (the voxel lines have arrived)
                    hv_wave_lds_sync();
                    const HvSemStaged staged{stage, src.lut, src.labels, src.depth};
                    for (int e = lane; e < nb; e += HV_WAVE) {
                        const uint32_t lidx = s[e] >> idx_bits;
                        if (e > 0 && (s[e - 1] >> idx_bits) == lidx) continue; // not the head of its voxel's run
                        sem_fold_run<VOX>(table, pool, block_base + lidx, G, staged, occ,
                                          [&](int j) -> int64_t { return (e + j < nb && (s[e + j] >> idx_bits) == lidx) ? (int64_t)(e + j) : -1; });
                    }
                    return;
                }
            }
            fold_sorted(nb, block_base);
            return;
        }
        if (tasks != nullptr) {
            // A bin beyond the window (a wall at 0.7 m puts ~3 000 points of a 640x480 keyframe into one 8 cm block) would be a
            // long serial job for this wave while the other waves of the launch have long left: it is cut into its voxel-index
            // ranges of 64 and handed to k_semb_fold_tasks, one wave per range.
            const int n_ranges = (G.nvox + 63) / 64;
            int32_t at = 0;
            if (lane == 0) at = atomicAdd(&task_count[parity], n_ranges);
            at = __shfl(at, 0);
            if (at + n_ranges <= task_cap) {
                if (lane < n_ranges) tasks[at + lane] = make_int4(idx, lane, nb, slot);
                return;
            }
            if (lane == 0) atomicSub(&task_count[parity], n_ranges); // (no room: this wave does it itself, below)
        }
        // voxel-index ranges narrow enough for a range's entries to fit the window; a range that still overflows (very many points
        // in few voxels) is folded in point-index windows (a voxel's points still arrive in order)
        int parts = 2;
        while (nb / parts > wcap / 2 && parts < G.nvox) parts <<= 1;
        const int width = (G.nvox + parts - 1) / parts;
        for (int lo = 0; lo < G.nvox; lo += width) {
            const uint32_t hi = (uint32_t)(lo + width);
            for (int64_t w = -1; w < n_points; w += wcap) { // w = -1: the whole range at once
                hv_wave_lds_sync();
                int m = 0;
                bool overflow = false;
                for (int e0 = 0; e0 < nb; e0 += HV_WAVE) {
                    const int e = e0 + lane;
                    const uint32_t ent = e < nb ? hv_bins_entry(B, slot, e) : 0u;
                    const uint32_t li = ent >> idx_bits;
                    const int64_t p = ent & idx_mask;
                    const bool in = e < nb && li >= (uint32_t)lo && li < hi && (w < 0 || (p >= w && p < w + wcap));
                    const unsigned long long bm = __ballot(in);
                    if (m + __popcll(bm) > wcap) {
                        overflow = true;
                        break;
                    }
                    if (in) s[m + __popcll(bm & lt)] = ent;
                    m += __popcll(bm);
                }
                if (overflow) {
                    w = -(int64_t)wcap; // next: w = 0
                    continue;
                }
                if (m > 0) {
                    hv_wave_lds_sync();
                    semb_sort_window(s, s_dst, off, m, idx_bits, G.nvox);
                    fold_sorted(m, block_base);
                }
                if (w < 0) break;
            }
        }
    });
}

// The deferred big bins: one wave per (block, range of 64 voxel indices).  The range's entries are picked out of the bin
// (ballot compaction, four loads in flight), sorted and folded like a small bin; a range that overflows the window goes through
// point-index windows.
template <typename VOX, typename SRC>
__global__ __launch_bounds__(256, HV_SEMB_FOLD_MIN_WAVES) void k_semb_fold_tasks(HvTable table, VOX *__restrict__ pool, HvBins B, const int32_t *__restrict__ task_count,
                                                          const int4 *__restrict__ tasks, int task_cap, HvSemParams G, SRC src_in,
                                                          unsigned long long *__restrict__ occ, int64_t n_points, int wcap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    __shared__ float s_lut[256];
    const SRC src = sem_src_prepare(src_in, s_lut);
    const int idx_bits = B.idx_bits;
    const int n_tasks = min(task_count[B.parity], task_cap);
    const int wave = threadIdx.x >> 6, lane = hv_lane_id();
    const int per_wave = 2 * wcap + G.nvox + 64;
    uint32_t *s = s_dyn + wave * per_wave, *s_dst = s + wcap, *off = s + 2 * wcap;
    const uint32_t idx_mask = (1u << idx_bits) - 1u;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int t = blockIdx.x * 4 + wave; t < n_tasks; t += gridDim.x * 4) {
        const int4 task = tasks[t];
        const int64_t block_base = (int64_t)task.x * G.nvox;
        const uint32_t lo = (uint32_t)task.y * 64u, hi = lo + 64u;
        const int nb = task.z, slot = task.w;
        auto fold_sorted = [&](int m) {
            for (int e = lane; e < m; e += HV_WAVE) {
                const uint32_t lidx = s[e] >> idx_bits;
                if (e > 0 && (s[e - 1] >> idx_bits) == lidx) continue;
                sem_fold_run<VOX>(table, pool, block_base + lidx, G, src, occ,
                                  [&](int j) -> int64_t { return (e + j < m && (s[e + j] >> idx_bits) == lidx) ? (int64_t)(s[e + j] & idx_mask) : -1; });
            }
        };
        for (int64_t w = -1; w < n_points; w += wcap) { // w = -1: the whole range at once; on overflow: point-index windows
            hv_wave_lds_sync();
            int m = 0;
            bool overflow = false;
            for (int e0 = 0; e0 < nb && !overflow; e0 += 4 * HV_WAVE) {
                uint32_t ent[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { // four independent loads in flight
                    const int e = e0 + k * HV_WAVE + lane;
                    ent[k] = e < nb ? hv_bins_entry(B, slot, e) : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = e0 + k * HV_WAVE + lane;
                    const uint32_t li = ent[k] >> idx_bits;
                    const int64_t p = ent[k] & idx_mask;
                    const bool in = e < nb && li >= lo && li < hi && (w < 0 || (p >= w && p < w + wcap));
                    const unsigned long long bm = __ballot(in);
                    if (m + __popcll(bm) > wcap) {
                        overflow = true;
                        break;
                    }
                    if (in) s[m + __popcll(bm & lt)] = ent[k];
                    m += __popcll(bm);
                }
            }
            if (overflow) { // (only possible for w = -1: a window holds at most WCAP distinct point indices)
                w = -(int64_t)wcap; // next iteration: w = 0
                continue;
            }
            if (m > 0) {
                hv_wave_lds_sync();
                semb_sort_window(s, s_dst, off, m, idx_bits, G.nvox);
                fold_sorted(m);
            }
            if (w < 0) break; // the range fitted: done
        }
    }
}

// get_voxels(min_count, min_confidence) (voxel_block_grid.hpp:785-817, semantic branch) and, with Q.kind 1 / 2,
// get_voxels_in_bb (:944-1013) / get_voxels_in_camera_frustrum (:1019-1195) for semantic voxels.
template <typename VOX>
__global__ __launch_bounds__(256) void k_sem_collect(HvTable table, const VOX *__restrict__ pool, int64_t n_blocks,
                                                      HvSemParams G, HvQuery Q, int min_count, float min_confidence,
                                                      double *__restrict__ out_pts, float *__restrict__ out_cols,
                                                      int32_t *__restrict__ out_cls, int32_t *__restrict__ out_obj,
                                                      float *__restrict__ out_conf, int64_t cap,
                                                      const unsigned long long *__restrict__ occ) {
    // one wave per block of the pool, its occupied voxels only (sem_for_occupied; occ == nullptr: every voxel).  The rows a wave
    // finds are collected in its LDS window first (voxel indices) and get their place in the output with ONE returning atomic per
    // ~450 rows: one per block visited was 120 k atomics on a single counter per call at 2 mm - 0.35 of the kernel's 0.48 ms.
    constexpr int BUF = 512;
    __shared__ uint32_t s_buf[4][BUF];
    const int wave = threadIdx.x / HV_WAVE, lane = hv_lane_id();
    uint32_t *buf = s_buf[wave];
    int n_buf = 0;       // rows waiting in the window (wave-uniform)
    int64_t n_found = 0; // size query (out_pts == nullptr): rows of this wave, added to the counter once
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    auto flush = [&]() {
        if (n_buf == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int32_t base = 0;
        if (lane == 0) base = atomicAdd(&table.counters[HV_CNT_OUT], n_buf);
        base = __shfl(base, 0);
        for (int i = lane; i < n_buf; i += HV_WAVE) {
            const int64_t at = (int64_t)base + i;
            if (at >= cap) continue;
            const VOX *v = pool + buf[i];
            const int32_t count = v->count;
            const double c = (double)count;
            const float cf = (float)count;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                out_pts[at * 3 + k] = v->pos[k] / c;
                out_cols[at * 3 + k] = v->col[k] / cf;
            }
            out_cls[at] = sem_class_id(v, table.prob_nodes);
            out_obj[at] = sem_object_id(v, table.prob_nodes);
            out_conf[at] = sem_confidence(v, table.prob_nodes);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        n_buf = 0;
    };
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / HV_WAVE);
    for (int64_t b = (int64_t)blockIdx.x * (blockDim.x / HV_WAVE) + wave; b < n_blocks; b += n_waves) {
        int32_t bk[3] = {0, 0, 0};
        if (Q.kind != 0) {
            hv_unpack_key(table.block_keys[b], bk[0], bk[1], bk[2]);
            bool out = false; // the block's key range against the query's: wave-uniform
#pragma unroll
            for (int a = 0; a < 3; ++a) out = out || bk[a] < Q.bmin[a] || bk[a] > Q.bmax[a];
            if (out) continue;
        }
        sem_for_occupied(occ, b, G.nvox, false, [&](int64_t gid, bool active) {
            bool pred = false;
            if (active) {
                const VOX *v = pool + gid;
                const int32_t count = v->count;
                pred = count >= min_count && sem_confidence(v, table.prob_nodes) >= min_confidence;
                if (pred && Q.kind != 0) {
                    const int l = (int)(gid - b * G.nvox);
                    const int32_t lc[3] = {l % G.bs, (l / G.bs) % G.bs, l / (G.bs * G.bs)};
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const int32_t vk = bk[a] * G.bs + lc[a];
                        if (vk < Q.vmin[a] || vk > Q.vmax[a]) pred = false;
                    }
                    if (pred) {
                        const double c = (double)count;
                        const double p0 = v->pos[0] / c, p1 = v->pos[1] / c, p2 = v->pos[2] / c;
                        if (Q.kind == 1) { // BoundingBox3D::contains, bounding_boxes_3d.cpp:207-210
                            pred = p0 >= Q.bb[0] && p0 <= Q.bb[3] && p1 >= Q.bb[1] && p1 <= Q.bb[4] && p2 >= Q.bb[2] && p2 <= Q.bb[5];
                        } else {
                            float uvd[3];
                            pred = hv_frustum_contains_d(Q, p0, p1, p2, uvd);
                        }
                    }
                }
            }
            const unsigned long long m = __ballot(pred);
            if (m == 0ull) return;
            if (out_pts == nullptr) {
                n_found += __popcll(m);
                return;
            }
            if (pred) buf[n_buf + __popcll(m & lt)] = (uint32_t)gid;
            n_buf += __popcll(m);
            if (n_buf > BUF - HV_WAVE) flush();
        });
    }
    if (out_pts != nullptr) flush();
    else if (lane == 0 && n_found) atomicAdd(&table.counters[HV_CNT_OUT], (int32_t)n_found);
}

// ------------------------------------------------------------------------------------------------
static int sem_sort_bits(const hv_volume *v) {
    int slot_bits = 0;
    while ((1ull << slot_bits) < v->table_capacity) slot_bits++;
    return std::min(32, slot_bits + v->local_bits + 1);
}

// The fold's LDS window per wave (entries): bins beyond it go to the task kernel as voxel ranges.  512 leaves room for the 16 waves
// per CU the kernel's registers allow (1024: 12 waves; measured round 4)
static int sem_wcap() { return 512; }
static size_t sem_fold_lds(int wcap, int nvox) { return 4 * sizeof(uint32_t) * (size_t)(2 * wcap + nvox + 64); } // two windows + per-voxel offsets per wave
static bool sem_bins_usable(const hv_volume *v, int64_t n) {
    const char *force = getenv("HV_SEM_PATH"); // HV_SEM_PATH=sort keeps the device-wide radix sort: A/B, tests
    const int nvox = v->cfg.block_size * v->cfg.block_size * v->cfg.block_size;
    return hv_bins_usable(v, n, 32 - v->local_bits) && sem_fold_lds(sem_wcap(), nvox) <= 64 * 1024 && !(force && strcmp(force, "sort") == 0);
}

// The bin path's launches after the capacity gate: `bin(B)` launches the bin pass (arrays or frame), then the fold over `src`.
template <typename VOX, typename SRC, typename BinLaunch>
static int sem_bins_run(hv_volume *v, int64_t n, bool checked, const SRC &src, BinLaunch bin) {
    const HvSemParams G = sem_params(v);
    const int idx_bits = 32 - v->local_bits;
    const int wcap = sem_wcap();
    const size_t lds_bytes = sem_fold_lds(wcap, G.nvox);
    int rc = HV_OK;
    HvBins B{};
    for (int attempt = 0;; ++attempt) {
        rc = hv_bins_ensure(v);
        if (rc != HV_OK) return rc;
        B = hv_bins_begin(v, idx_bits);
        bin(B);
        if (!checked) break;
        rc = hv_claims_fit(v); // blocks that did not fit: grow and claim again (the bins restart from clean arrays)
        if (rc == HV_OK) break;
        v->bins_clean = false;
        if (rc != HV_RETRY_CLAIM || attempt >= 8) return rc == HV_RETRY_CLAIM ? HV_ERR_CAPACITY : rc;
    }
    v->bins.parity ^= 1;
    const int32_t seq = hv_next_status_seq(v);
    const unsigned fold_grid = hv_bins_fold_grid(v, k_semb_fold_wave<VOX, SRC>, lds_bytes); // persistent waves (hv_bins_for_each)
    VOX *bpool = (VOX *)v->pool;
    // task list of the big bins: [2 counters (one per parity)][tasks]; at most n / wcap bins are big
    const int task_cap = (int)std::min<int64_t>((n / wcap + 1) * ((G.nvox + 63) / 64), 1 << 22);
    const size_t task_bytes = 256 + sizeof(int4) * (size_t)task_cap;
    if (v->semb_tasks == nullptr || v->semb_tasks_bytes < task_bytes) {
        rc = hv_ensure_buffer(v, &v->semb_tasks, &v->semb_tasks_bytes, task_bytes);
        if (rc != HV_OK) return rc;
        HV_HIP(hipMemsetAsync(v->semb_tasks, 0, 256, v->stream));
    }
    int32_t *task_count = (int32_t *)v->semb_tasks;
    int4 *tasks = (int4 *)((char *)v->semb_tasks + 256);
    const bool use_tasks = true;
    hipLaunchKernelGGL((k_semb_fold_wave<VOX, SRC>), dim3(fold_grid), dim3(256), lds_bytes, v->stream, v->table, bpool, B, G, src, v->occ, n,
                       v->d_status, seq, task_count, use_tasks ? tasks : (int4 *)nullptr, task_cap, wcap);
    if (use_tasks)
        hipLaunchKernelGGL((k_semb_fold_tasks<VOX, SRC>), dim3(1024), dim3(256), lds_bytes, v->stream, v->table, bpool, B,
                           (const int32_t *)task_count, (const int4 *)tasks, task_cap, G, src, v->occ, n, wcap);
    HV_HIP(hipGetLastError());
    v->frame_counter += 1;
    return HV_OK;
}

template <typename VOX, typename PT>
static int sem_integrate(hv_volume *v, const PT *d_pts, int64_t n, const void *d_cols, int color_kind,
                         const int32_t *d_cls, const int32_t *d_inst, const float *d_depths,
                         const uint32_t *valid_mask_keys = nullptr) {
    const HvSemParams G = sem_params(v);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    bool checked = false;
    int rc = hv_capacity_gate(v, &checked); // may grow the pool: the sort-key width follows the table
    if (rc != HV_OK) return rc;
    // Per-keyframe bin path (rounds 4-6): needs the point index and the local voxel index in one 32-bit entry.
    if (sem_bins_usable(v, n)) {
        auto bin = [&](const HvBins &B) {
            hipLaunchKernelGGL(k_semb_bin<PT>, dim3((unsigned)((n + HV_BIN_THREADS - 1) / HV_BIN_THREADS)), dim3(HV_BIN_THREADS), 0, v->stream, v->table, B, d_pts, n, G, valid_mask_keys);
        };
        if (color_kind == HV_COLOR_U8)
            return sem_bins_run<VOX>(v, n, checked, HvSemArrays<PT, HV_COLOR_U8>{d_pts, d_cols, d_cls, d_inst, d_depths}, bin);
        if (color_kind == HV_COLOR_F32)
            return sem_bins_run<VOX>(v, n, checked, HvSemArrays<PT, HV_COLOR_F32>{d_pts, d_cols, d_cls, d_inst, d_depths}, bin);
        return sem_bins_run<VOX>(v, n, checked, HvSemArrays<PT, HV_COLOR_NONE>{d_pts, d_cols, d_cls, d_inst, d_depths}, bin);
    }
    size_t bytes = 0;
    HV_HIP(rocprim::radix_sort_pairs(nullptr, bytes, v->sort_keys_in, v->sort_keys_out, v->sort_vals_in, v->sort_vals_out,
                                     (size_t)n, 0, sem_sort_bits(v), v->stream));
    rc = hv_ensure_buffer(v, &v->sort_tmp, &v->sort_tmp_bytes, bytes);
    if (rc != HV_OK) return rc;
    for (int attempt = 0;; ++attempt) {
        hipLaunchKernelGGL(k_sem_keys<PT>, dim3(blocks), dim3(256), 0, v->stream, v->table, d_pts, n, G, v->sort_keys_in,
                           v->sort_vals_in, valid_mask_keys);
        if (!checked) break;
        rc = hv_claims_fit(v); // blocks that did not fit: grow, claim again (sort keys embed table slots)
        if (rc == HV_OK) break;
        if (rc != HV_RETRY_CLAIM || attempt >= 8) return rc == HV_RETRY_CLAIM ? HV_ERR_CAPACITY : rc;
        HV_HIP(rocprim::radix_sort_pairs(nullptr, bytes, v->sort_keys_in, v->sort_keys_out, v->sort_vals_in, v->sort_vals_out,
                                         (size_t)n, 0, sem_sort_bits(v), v->stream));
        rc = hv_ensure_buffer(v, &v->sort_tmp, &v->sort_tmp_bytes, bytes);
        if (rc != HV_OK) return rc;
    }
    bytes = v->sort_tmp_bytes;
    HV_HIP(rocprim::radix_sort_pairs(v->sort_tmp, bytes, v->sort_keys_in, v->sort_keys_out, v->sort_vals_in,
                                     v->sort_vals_out, (size_t)n, 0, sem_sort_bits(v), v->stream));
    VOX *pool = (VOX *)v->pool;
    if (color_kind == HV_COLOR_U8) {
        hipLaunchKernelGGL((k_sem_reduce<VOX, PT, HV_COLOR_U8>), dim3(blocks), dim3(256), 0, v->stream, v->table, pool,
                           v->sort_keys_out, v->sort_vals_out, n, G, d_pts, d_cols, d_cls, d_inst, d_depths, v->occ);
    } else if (color_kind == HV_COLOR_F32) {
        hipLaunchKernelGGL((k_sem_reduce<VOX, PT, HV_COLOR_F32>), dim3(blocks), dim3(256), 0, v->stream, v->table, pool,
                           v->sort_keys_out, v->sort_vals_out, n, G, d_pts, d_cols, d_cls, d_inst, d_depths, v->occ);
    } else {
        hipLaunchKernelGGL((k_sem_reduce<VOX, PT, HV_COLOR_NONE>), dim3(blocks), dim3(256), 0, v->stream, v->table, pool,
                           v->sort_keys_out, v->sort_vals_out, n, G, d_pts, d_cols, d_cls, d_inst, d_depths, v->occ);
    }
    hv_launch_publish_status(v); // pool occupancy for the next call's hv_capacity_gate
    HV_HIP(hipGetLastError());
    v->frame_counter += 1;
    return HV_OK;
}

// stage a host array into a dedicated slice of the volume's scratch (5 inputs -> one growing buffer)
static int sem_stage(hv_volume *v, const void *src, size_t bytes, int32_t loc, char *&cursor, const void **dev) {
    if (src == nullptr) {
        *dev = nullptr;
        return HV_OK;
    }
    if (loc == HV_DEVICE) {
        *dev = src;
        return HV_OK;
    }
    const int rc = hv_h2d(v, cursor, src, bytes); // (waits for the copy when the source is page-locked)
    if (rc != HV_OK) return rc;
    *dev = cursor;
    cursor += (bytes + 255) & ~(size_t)255;
    return HV_OK;
}

template <typename VOX>
static int sem_dump(hv_volume *v, int64_t nb, const std::vector<int64_t> &order, const std::vector<std::array<int32_t, 3>> &xyz,
                    int32_t *keys, int32_t *ints, float *conf, double *pos_sums, float *col_sums, int32_t *label_counts,
                    int32_t *labels, float *log_probs, int32_t max_labels, float *obj_conf, float *cls_conf) {
    const int nvox = sem_params(v).nvox;
    std::vector<VOX> host((size_t)nb * nvox);
    HV_HIP(hipMemcpy(host.data(), v->pool, sizeof(VOX) * host.size(), hipMemcpyDeviceToHost));
    std::vector<HvProbNode> host_nodes; // the label maps' overflow nodes handed out so far
    if (v->table.prob_nodes != nullptr) {
        int32_t used = 0;
        HV_HIP(hipMemcpy(&used, v->table.counters + HV_CNT_PROB_NODES, sizeof(int32_t), hipMemcpyDeviceToHost));
        host_nodes.resize((size_t)std::min(std::max(used, 0), v->table.prob_node_cap) + 1);
        HV_HIP(hipMemcpy(host_nodes.data(), v->table.prob_nodes, sizeof(HvProbNode) * (host_nodes.size() - 1), hipMemcpyDeviceToHost));
    }
    const HvProbNode *nodes = host_nodes.empty() ? nullptr : host_nodes.data();
    for (int64_t o = 0; o < nb; ++o) {
        const int64_t i = order[o];
        if (keys) memcpy(keys + o * 3, xyz[i].data(), 12);
        for (int l = 0; l < nvox; ++l) {
            const VOX *x = &host[(size_t)i * nvox + l];
            const size_t at = (size_t)o * nvox + l;
            if (ints) {
                int32_t *d = ints + at * 4;
                d[0] = x->count; d[1] = sem_object_id(x, nodes); d[2] = sem_class_id(x, nodes); d[3] = sem_confidence_counter(x, nodes);
            }
            if (conf) conf[at] = sem_confidence(x, nodes);
            if (obj_conf) obj_conf[at] = sem_object_confidence(x, nodes);
            if (cls_conf) cls_conf[at] = sem_class_confidence(x, nodes);
            if (pos_sums) memcpy(pos_sums + at * 3, x->pos, 24);
            if (col_sums) memcpy(col_sums + at * 3, x->col, 12);
            if constexpr (HvPay<VOX>::maps) { // (the "*2" payload: entries (id, which map) of its two marginal maps)
                const HvProbVoxel *p = (const HvProbVoxel *)x;
                const int nlab = prob_nlab(p->meta);
                if (label_counts) label_counts[at] = nlab;
                for (int k = 0; k < max_labels; ++k) { // (insertion order; the first max_labels pairs of longer maps)
                    const HvProbPair q = k < nlab ? prob_get(p, nodes, k) : HvProbPair{-1, -1, 0.0f};
                    if (labels) {
                        labels[(at * max_labels + k) * 2 + 0] = q.obj;
                        labels[(at * max_labels + k) * 2 + 1] = q.cls;
                    }
                    if (log_probs) log_probs[at * max_labels + k] = q.logp;
                }
            } else if (label_counts) {
                label_counts[at] = 0;
            }
        }
    }
    return HV_OK;
}

static int sem_run_collect(hv_volume *v, const HvQuery &Q, int32_t min_count, float min_confidence, double *points, float *colors,
                           int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap, int64_t *n) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_get_voxels_semantic: null argument");
    HV_REQUIRE(hv_is_semantic(v), HV_ERR_MODE, "hv_get_voxels_semantic: volume is not a semantic grid");
    HV_HIP(hipSetDevice(v->device));
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    *n = 0;
    if (nb == 0) return HV_OK;
    const bool want = points && colors && class_ids && object_ids && confidences && cap > 0;
    double *d_pts = nullptr;
    float *d_cols = nullptr, *d_conf = nullptr;
    int32_t *d_cls = nullptr, *d_obj = nullptr;
    if (want) {
        rc = hv_ensure_buffer(v, &v->out_a, &v->out_a_bytes, (size_t)cap * (24 + 12 + 4 + 4 + 4) + 1024);
        if (rc != HV_OK) return rc;
        d_pts = (double *)v->out_a;
        d_cols = (float *)(d_pts + 3 * cap);
        d_cls = (int32_t *)(d_cols + 3 * cap);
        d_obj = d_cls + cap;
        d_conf = (float *)(d_obj + cap);
    }
    HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT], 0, sizeof(int32_t), v->stream));
    const dim3 grid((unsigned)std::min<int64_t>((nb + 3) / 4, 8192)); // a wave per block, grid-stride
    // (min_count <= 0 asks for voxels that never took a point as well: the occupancy bits cannot be used then)
    const unsigned long long *occ = min_count >= 1 ? v->occ : nullptr;
    HV_SEM_DISPATCH(v, hipLaunchKernelGGL(k_sem_collect<VOX>, grid, dim3(256), 0, v->stream, v->table, (const VOX *)v->pool, nb, sem_params(v), Q, min_count,
                                          min_confidence, d_pts, d_cols, d_cls, d_obj, d_conf, want ? cap : 0, occ));
    HV_HIP(hipGetLastError());
    rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    *n = v->h_counters[HV_CNT_OUT];
    if (want) {
        const int64_t m = std::min(*n, cap);
        if (m > 0) {
            HV_HIP(hipMemcpyAsync(points, d_pts, 24 * m, hipMemcpyDeviceToHost, v->stream));
            HV_HIP(hipMemcpyAsync(colors, d_cols, 12 * m, hipMemcpyDeviceToHost, v->stream));
            HV_HIP(hipMemcpyAsync(class_ids, d_cls, 4 * m, hipMemcpyDeviceToHost, v->stream));
            HV_HIP(hipMemcpyAsync(object_ids, d_obj, 4 * m, hipMemcpyDeviceToHost, v->stream));
            HV_HIP(hipMemcpyAsync(confidences, d_conf, 4 * m, hipMemcpyDeviceToHost, v->stream));
            HV_HIP(hipStreamSynchronize(v->stream));
        }
    }
    return HV_OK;
}

extern "C" {

int hv_set_depth_threshold(hv_volume *v, float depth_threshold) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_set_depth_threshold: null volume");
    v->sem_depth_threshold = depth_threshold;
    return HV_OK;
}

int hv_set_depth_decay_rate(hv_volume *v, float depth_decay_rate) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_set_depth_decay_rate: null volume");
    v->sem_depth_decay_rate = depth_decay_rate;
    return HV_OK;
}

int hv_label_overflows(hv_volume *v, int64_t *n) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_label_overflows: null argument");
    int rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    *n = v->h_counters[HV_CNT_LABEL_OVERFLOW];
    return HV_OK;
}

int hv_prob_nodes_used(hv_volume *v, int64_t *n) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_prob_nodes_used: null argument");
    int rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    *n = v->h_counters[HV_CNT_PROB_NODES];
    return HV_OK;
}

int hv_integrate_points_semantic(hv_volume *v, const void *points, int32_t point_dtype, int64_t n, const void *colors,
                                 int32_t color_dtype, const int32_t *class_ids, const int32_t *instance_ids,
                                 const float *depths, int32_t loc) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_integrate_points_semantic: null volume");
    HV_REQUIRE(hv_is_semantic(v), HV_ERR_MODE, "hv_integrate_points_semantic: volume is not a semantic grid");
    if (n == 0) return HV_OK;
    HV_REQUIRE(points != nullptr && n > 0, HV_ERR_INVALID, "points must be a contiguous Nx3 array");
    HV_REQUIRE(point_dtype == 0 || point_dtype == 1, HV_ERR_INVALID, "points must be float32 or float64");
    HV_REQUIRE(color_dtype == HV_COLOR_NONE || color_dtype == HV_COLOR_U8 || color_dtype == HV_COLOR_F32, HV_ERR_INVALID,
               "Colors must be uint8 or float32");
    HV_REQUIRE(color_dtype == HV_COLOR_NONE || colors != nullptr, HV_ERR_INVALID, "points and colors must have the same size");
    HV_REQUIRE(class_ids != nullptr || instance_ids == nullptr, HV_ERR_INVALID,
               "instance_ids but no class_ids is not supported"); // voxel_block_grid.hpp:57-59
    HV_REQUIRE(n <= v->cfg.max_points, HV_ERR_CAPACITY, "hv_integrate_points_semantic: %lld points exceed max_points=%lld",
               (long long)n, (long long)v->cfg.max_points);
    HV_HIP(hipSetDevice(v->device));
    const size_t psz = point_dtype == 1 ? 8 : 4;
    const size_t need = 3 * psz * n + 12 * (size_t)n + 3 * 4 * (size_t)n + 5 * 256;
    if (loc == HV_HOST) {
        int rc = hv_ensure_buffer(v, &v->stage_a, &v->stage_a_bytes, need);
        if (rc != HV_OK) return rc;
    }
    char *cursor = (char *)v->stage_a;
    const void *d_pts, *d_cols, *d_cls, *d_inst, *d_dep;
    int rc = sem_stage(v, points, 3 * psz * n, loc, cursor, &d_pts);
    if (rc == HV_OK) rc = sem_stage(v, color_dtype == HV_COLOR_NONE ? nullptr : colors, (color_dtype == HV_COLOR_U8 ? 3 : 12) * (size_t)n, loc, cursor, &d_cols);
    if (rc == HV_OK) rc = sem_stage(v, class_ids, 4 * (size_t)n, loc, cursor, &d_cls);
    if (rc == HV_OK) rc = sem_stage(v, instance_ids, 4 * (size_t)n, loc, cursor, &d_inst);
    if (rc == HV_OK) rc = sem_stage(v, depths, 4 * (size_t)n, loc, cursor, &d_dep);
    if (rc != HV_OK) return rc;
    if (point_dtype == 1) {
        HV_SEM_DISPATCH(v, return sem_integrate<VOX, double>(v, (const double *)d_pts, n, d_cols, color_dtype, (const int32_t *)d_cls, (const int32_t *)d_inst,
                                                             (const float *)d_dep));
    }
    HV_SEM_DISPATCH(v, return sem_integrate<VOX, float>(v, (const float *)d_pts, n, d_cols, color_dtype, (const int32_t *)d_cls, (const int32_t *)d_inst,
                                                        (const float *)d_dep));
    return HV_OK; // (not reached)
}

int hv_integrate_rgbd_semantic(hv_volume *v, const float *depth, const uint8_t *rgb, const int32_t *class_ids_image,
                               const int32_t *object_ids_image, int32_t height, int32_t width, const double *intr,
                               const double *T_cw, double min_depth, double max_depth, int32_t use_depths, int32_t loc) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_integrate_rgbd_semantic: null volume");
    HV_REQUIRE(hv_is_semantic(v), HV_ERR_MODE, "hv_integrate_rgbd_semantic: volume is not a semantic grid");
    HV_REQUIRE(depth != nullptr && rgb != nullptr && intr != nullptr && T_cw != nullptr && height > 0 && width > 0,
               HV_ERR_INVALID, "hv_integrate_rgbd_semantic: null or empty input");
    HV_REQUIRE(class_ids_image != nullptr || object_ids_image == nullptr, HV_ERR_INVALID,
               "instance_ids but no class_ids is not supported");
    const int64_t npx = (int64_t)height * width;
    HV_REQUIRE(npx <= v->cfg.max_points, HV_ERR_CAPACITY, "hv_integrate_rgbd_semantic: image exceeds max_points");
    HV_HIP(hipSetDevice(v->device));
    // the frame's depth and colour planes on the device (staged on the volume's stream when they come from the host)
    const void *d_depth = nullptr, *d_rgb = nullptr;
    int rc = hv_stage_in(v, depth, (size_t)npx * 4, loc, 0, &d_depth);
    if (rc != HV_OK) return rc;
    rc = hv_stage_in(v, rgb, (size_t)npx * 3, loc, 1, &d_rgb);
    if (rc != HV_OK) return rc;
    // label planes: staged behind each other in the output scratch when they come from the host
    const int32_t *d_cls = class_ids_image, *d_obj = object_ids_image;
    if (loc == HV_HOST && class_ids_image != nullptr) {
        const size_t plane = (sizeof(int32_t) * (size_t)npx + 255) & ~(size_t)255;
        rc = hv_ensure_buffer(v, &v->out_c, &v->out_c_bytes, 2 * plane);
        if (rc != HV_OK) return rc;
        char *st = (char *)v->out_c;
        bool pinned_src = false;
        if ((rc = hv_h2d_lazy(v, st, class_ids_image, sizeof(int32_t) * npx, &pinned_src)) != HV_OK) return rc;
        d_cls = (const int32_t *)st;
        if (object_ids_image != nullptr) {
            if ((rc = hv_h2d_lazy(v, st + plane, object_ids_image, sizeof(int32_t) * npx, &pinned_src)) != HV_OK) return rc;
            d_obj = (const int32_t *)(st + plane);
        }
        if ((rc = hv_h2d_fence(v, pinned_src)) != HV_OK) return rc;
    }
    if (sem_bins_usable(v, npx)) {
        // production: the bin pass unprojects the pixel itself and leaves a 32-byte record per point for the fold
        bool checked = false;
        rc = hv_capacity_gate(v, &checked);
        if (rc != HV_OK) return rc;
        rc = hv_ensure_buffer(v, &v->bin_rec, &v->bin_rec_bytes, (size_t)32 * (size_t)v->cfg.max_points);
        if (rc != HV_OK) return rc;
        const HvSemParams G = sem_params(v);
        const HvUnprojectParams U = unproject_params(HV_DEPTH_F32, 1.0, height, width, intr, T_cw, min_depth, max_depth);
        auto bin = [&](const HvBins &B) {
            hipLaunchKernelGGL(k_semb_bin_frame, dim3((unsigned)((npx + HV_BIN_THREADS - 1) / HV_BIN_THREADS)), dim3(HV_BIN_THREADS), 0, v->stream, v->table, B, G, U, d_depth,
                               (const uint8_t *)d_rgb, d_cls, d_obj, (float4 *)v->bin_rec);
        };
        const HvSemRecs src{(const float4 *)v->bin_rec, nullptr, d_cls != nullptr, use_depths != 0};
        HV_SEM_DISPATCH(v, return sem_bins_run<VOX>(v, npx, checked, src, bin));
    }
    // (inputs beyond the bin path's limits: unprojected rows, then the radix path)
    rc = hv_unproject_frame(v, d_depth, HV_DEPTH_F32, 1.0, (const uint8_t *)d_rgb, height, width, intr, T_cw, min_depth, max_depth, HV_DEVICE, nullptr);
    if (rc != HV_OK) return rc;
    // depths = camera z of the point = the pixel's depth (…voxel_semantic_grid.py:418-424)
    const float *d_depths = use_depths ? (const float *)d_depth : nullptr;
    HV_SEM_DISPATCH(v, return sem_integrate<VOX, float>(v, v->scratch_points, npx, v->scratch_colors, HV_COLOR_F32, d_cls, d_obj, d_depths, v->sort_keys_out));
    return HV_OK; // (not reached)
}

int hv_get_voxels_semantic(hv_volume *v, int32_t min_count, float min_confidence, double *points, float *colors,
                           int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap, int64_t *n) {
    HvQuery Q;
    memset(&Q, 0, sizeof(Q));
    return sem_run_collect(v, Q, min_count, min_confidence, points, colors, class_ids, object_ids, confidences, cap, n);
}

int hv_get_voxels_semantic_in_bb(hv_volume *v, const double *bbox, int32_t min_count, float min_confidence, double *points,
                                 float *colors, int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap,
                                 int64_t *n) {
    HV_REQUIRE(v != nullptr && bbox != nullptr, HV_ERR_INVALID, "hv_get_voxels_semantic_in_bb: null argument");
    HvQuery Q;
    memset(&Q, 0, sizeof(Q));
    Q.kind = 1;
    for (int k = 0; k < 6; ++k) Q.bb[k] = bbox[k];
    const HvSemParams G = sem_params(v);
    HvGridParams GP{G.inv_voxel_size, G.bs, G.nvox, G.local_bits};
    fill_key_range(Q, GP);
    return sem_run_collect(v, Q, min_count, min_confidence, points, colors, class_ids, object_ids, confidences, cap, n);
}

int hv_get_voxels_semantic_in_frustum(hv_volume *v, const float *intr_f32, int32_t width, int32_t height, const double *T_cw,
                                      float depth_max, float depth_min, int32_t min_count, float min_confidence, double *points,
                                      float *colors, int32_t *class_ids, int32_t *object_ids, float *confidences, int64_t cap,
                                      int64_t *n) {
    HV_REQUIRE(v != nullptr && intr_f32 != nullptr && T_cw != nullptr, HV_ERR_INVALID, "hv_get_voxels_semantic_in_frustum: null argument");
    HvQuery Q;
    memset(&Q, 0, sizeof(Q));
    Q.kind = 2;
    fill_frustum_query(Q, v, intr_f32, width, height, T_cw, depth_max, depth_min);
    const HvSemParams G = sem_params(v);
    HvGridParams GP{G.inv_voxel_size, G.bs, G.nvox, G.local_bits};
    fill_key_range(Q, GP);
    return sem_run_collect(v, Q, min_count, min_confidence, points, colors, class_ids, object_ids, confidences, cap, n);
}

static int sem_dump_blocks(hv_volume *v, int32_t *keys, int32_t *ints, float *conf, double *pos_sums, float *col_sums, int32_t *label_counts,
                           int32_t *labels, float *log_probs, int32_t max_labels, float *obj_conf, float *cls_conf, int64_t *n_blocks) {
    HV_REQUIRE(v != nullptr && n_blocks != nullptr && max_labels >= 0, HV_ERR_INVALID, "hv_dump_blocks_semantic: null argument");
    HV_REQUIRE(hv_is_semantic(v), HV_ERR_MODE, "hv_dump_blocks_semantic: wrong mode");
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    *n_blocks = nb;
    if (nb == 0 || (!keys && !ints && !conf && !pos_sums && !col_sums && !label_counts && !labels && !log_probs && !obj_conf && !cls_conf)) return HV_OK;
    std::vector<uint64_t> bkeys((size_t)nb);
    HV_HIP(hipMemcpy(bkeys.data(), v->table.block_keys, sizeof(uint64_t) * nb, hipMemcpyDeviceToHost));
    std::vector<std::array<int32_t, 3>> xyz((size_t)nb);
    for (int64_t i = 0; i < nb; ++i) hv_unpack_key(bkeys[i], xyz[i][0], xyz[i][1], xyz[i][2]);
    std::vector<int64_t> order((size_t)nb);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return xyz[a] < xyz[b]; });
    HV_SEM_DISPATCH(v, return sem_dump<VOX>(v, nb, order, xyz, keys, ints, conf, pos_sums, col_sums, label_counts, labels, log_probs, max_labels, obj_conf, cls_conf));
    return HV_OK; // (not reached)
}

int hv_dump_blocks_semantic2(hv_volume *v, int32_t *keys, int32_t *ints, float *conf, double *pos_sums, float *col_sums,
                             int32_t *label_counts, int32_t *labels, float *log_probs, int32_t max_labels, int64_t *n_blocks) {
    return sem_dump_blocks(v, keys, ints, conf, pos_sums, col_sums, label_counts, labels, log_probs, max_labels, nullptr, nullptr, n_blocks);
}

int hv_dump_marginals_semantic(hv_volume *v, float *object_confidences, float *class_confidences, int64_t *n_blocks) {
    return sem_dump_blocks(v, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, object_confidences, class_confidences, n_blocks);
}

int hv_dump_blocks_semantic(hv_volume *v, int32_t *keys, int32_t *ints, double *pos_sums, float *col_sums,
                            int64_t *n_blocks) {
    return hv_dump_blocks_semantic2(v, keys, ints, nullptr, pos_sums, col_sums, nullptr, nullptr, nullptr, 0, n_blocks);
}

} // extern "C"
