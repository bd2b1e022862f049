// libpyslam_hipvol.so — operations of pySLAM's semantic block grids beyond integrate / get_voxels
// (both payloads, gfx950):
//   carve                                voxel_grid_carving.h:47-79 via iterate_voxels_in_camera_frustrum
//                                        (voxel_block_grid.hpp:1335-1540)
//   assign_object_ids_to_instance_ids    voxel_semantic_data_association.h:70-373
//   remap_instance_ids                   image_utils.h:69-163
//   get_object_segments (+ PCA OBB)      voxel_block_semantic_grid.hpp:217-267, bounding_boxes_3d.cpp:373-553
//   merge_segments / remove_segment / remove_low_confidence_segments   voxel_block_semantic_grid.hpp:119-196
//   remove_low_count_voxels / remove_low_confidence_voxels / size      voxel_block_grid.hpp:625-676, 1557-1572
//
// Association on the GPU: one thread per allocated voxel tests the frustum and the class / depth
// gates, then votes (instance_id, object_id) into a small device hash with wave-aggregated atomics
// (ballot + shuffle: one atomic per distinct pair per wave); voxels without an object id are appended
// to a pending list (wave-ballot compaction).  The few distinct pairs go to the host, which applies
// the reference's winner / min_votes / min_vote_ratio rules verbatim and hands the final
// instance -> object table back for the pending voxels.  New object ids come from a process-wide
// counter like the reference's (voxel_semantic_shared_data.h:26-34); they are assigned in ascending
// instance-id order (the reference assigns them in its hash-map iteration order: same set of new
// ids, possibly permuted between the instances that need one in the same call).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "hv_common.h"
#include "hv_query.h"
#include "hv_semantic.h"
#include <rocprim/device/device_radix_sort.hpp>

static constexpr int32_t HV_OBJ_PENDING = INT32_MIN;  // vote cast by a voxel that waits for a new object id
static constexpr int32_t HV_OBJ_SEEN = INT32_MIN + 1; // "this instance id occurs in the image" marker
static constexpr uint32_t HV_VOTE_CAP = 1u << 16;     // vote-table slots (distinct (instance, object) pairs per frame)
static constexpr uint64_t HV_VOTE_EMPTY = ~0ull;

static std::atomic<int32_t> g_next_object_id{1}; // VoxelSemanticSharedData::next_object_id: the value the device counters start from
// The counter itself lives in device memory (one per GPU of the process): k_sem_assoc_rules hands the ids out without a host round
// trip.  hv_peek / hv_set_next_object_id read / write it (synchronising the device).
static int32_t *g_dev_next_id[64] = {};
static int g_dev_last = -1; // GPU whose counter was used last (hv_peek_next_object_id reads that one)

static int hv_dev_next_id(int device, int32_t **out) {
    HV_REQUIRE(device >= 0 && device < 64, HV_ERR_INVALID, "device ordinal out of range");
    if (g_dev_next_id[device] == nullptr) {
        HV_HIP(hipSetDevice(device));
        HV_HIP(hipMalloc((void **)&g_dev_next_id[device], sizeof(int32_t)));
        const int32_t start = g_next_object_id.load();
        HV_HIP(hipMemcpy(g_dev_next_id[device], &start, sizeof(int32_t), hipMemcpyHostToDevice));
    }
    g_dev_last = device;
    *out = g_dev_next_id[device];
    return HV_OK;
}

template <typename VOX> __device__ __forceinline__ void sem_reset(VOX *v) {
    uint4 *q = (uint4 *)v;
    uint32_t chain = 0u; // a probabilistic voxel keeps its overflow nodes for the label map it grows next
    if constexpr (HvPay<VOX>::maps) chain = ((const HvProbVoxel *)v)->next;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(VOX) / 16); ++i) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (HvPay<VOX>::maps) {
        if (chain != 0u) ((HvProbVoxel *)v)->next = chain;
    }
}

// The visit predicate of iterate_voxels_in_camera_frustrum (min_count = 1, min_confidence = 0).
template <typename VOX>
__device__ __forceinline__ bool sem_visit(const HvQuery &Q, const HvTable &table, const VOX *v, int64_t b, int l,
                                          const HvSemParams &G, float *uvd) {
    const int32_t count = v->count;
    if (count < 1) return false;
    if (!sem_confidence_not_negative(v, table.prob_nodes)) return false; // (confidence >= 0)
    int32_t bk[3];
    hv_unpack_key(table.block_keys[b], bk[0], bk[1], bk[2]);
    const int32_t lc[3] = {l % G.bs, (l / G.bs) % G.bs, l / (G.bs * G.bs)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (bk[a] < Q.bmin[a] || bk[a] > Q.bmax[a]) return false;
        const int32_t vk = bk[a] * G.bs + lc[a];
        if (vk < Q.vmin[a] || vk > Q.vmax[a]) return false;
    }
    const double c = (double)count;
    return hv_frustum_contains_d(Q, v->pos[0] / c, v->pos[1] / c, v->pos[2] / c, uvd);
}

// What the association vote asks of a voxel, read in ONE batch of loads (round 6).  The pointer-based accessors above look at a voxel
// in global memory field by field behind branches - count, then the label map, then the block key, then the position sums - and the
// compiler keeps that order: four to five dependent trips to the cache per visit before the image is even looked at.  Here the record
// (64 / 128 bytes) is loaded whole into registers beside the block key, everything is decided from the copy (the inline label slots
// through constant indices: hv_semantic.h), and only a map with overflow nodes, an invalid cache or a non-finite log-probability
// goes back to the pointer-based functions (same answers by construction: they are the fallback).
struct HvVoteHot {
    int32_t count, cls, obj;
    bool conf_ok;
    double pos[3];
};
__device__ __forceinline__ HvVoteHot vote_hot(const HvSemVoxel *v, const void *) {
    const uint4 a = *(const uint4 *)v; // count, obj1, cls1, counter
    HvVoteHot h;
    h.count = (int32_t)a.x;
    h.obj = (int32_t)a.y - 1;
    h.cls = (int32_t)a.z - 1;
    h.pos[0] = v->pos[0];
    h.pos[1] = v->pos[1];
    h.pos[2] = v->pos[2];
    // get_confidence() >= 0 (sem_confidence above): counter / count clamped to 1, 0 for an empty voxel
    const float r = (float)(int32_t)a.w / (float)h.count;
    h.conf_ok = h.count == 0 || (r < 1.0f ? r : 1.0f) >= 0.0f;
    return h;
}
__device__ __forceinline__ HvVoteHot vote_hot(const HvProbVoxel *v, const void *nodes_) {
    const HvProbNode *nodes = (const HvProbNode *)nodes_;
    const HvProbVoxel r = *v; // eight 16-byte loads in flight together; only constant indices below: the copy stays in registers
    HvVoteHot h;
    h.count = r.count;
    h.pos[0] = r.pos[0];
    h.pos[1] = r.pos[1];
    h.pos[2] = r.pos[2];
    const int nlab = prob_nlab(r.meta), best = prob_best(r.meta);
    bool finite = true;
#pragma unroll
    for (int k = 0; k < HV_PROB_K; ++k) {
        const float lp = r.logp[k];
        finite = finite && (k >= nlab || lp - lp == 0.0f);
    }
    if (nlab == 0) {
        h.cls = h.obj = -1;
        h.conf_ok = true;
    } else if (nlab <= HV_PROB_K && best >= 0 && finite) { // the common case: decided from the copy
        const HvProbPair p = prob_slot_get(&r, best);
        h.cls = p.cls;
        h.obj = p.obj;
        h.conf_ok = true;
    } else { // overflow nodes / no cached best pair / a NaN or an infinity in the map
        h.cls = sem_class_id(v, nodes);
        h.obj = sem_object_id(v, nodes);
        h.conf_ok = sem_confidence_not_negative(v, nodes);
    }
    return h;
}
__device__ __forceinline__ HvVoteHot vote_hot(const HvSem2Voxel *v, const void *) {
    const uint4 a = *(const uint4 *)v; // count, obj1, cls1, obj_counter
    HvVoteHot h;
    h.count = (int32_t)a.x;
    h.obj = (int32_t)a.y - 1;
    h.cls = (int32_t)a.z - 1;
    h.pos[0] = v->pos[0];
    h.pos[1] = v->pos[1];
    h.pos[2] = v->pos[2];
    // get_confidence() >= 0: the smaller of the two counter / count ratios, each clamped to 1 (0 for an empty voxel)
    const float o = sem2_ratio((int32_t)a.w, h.count), c = sem2_ratio(v->cls_counter, h.count);
    h.conf_ok = h.count == 0 || (c < o ? c : o) >= 0.0f;
    return h;
}
__device__ __forceinline__ HvVoteHot vote_hot(const HvProb2Voxel *v, const void *nodes_) {
    const HvProbNode *nodes = (const HvProbNode *)nodes_;
    const HvProbVoxel r = *v;
    HvVoteHot h;
    h.count = r.count;
    h.pos[0] = r.pos[0];
    h.pos[1] = r.pos[1];
    h.pos[2] = r.pos[2];
    const int n = prob_nlab(r.meta);
    bool finite = true;
#pragma unroll
    for (int k = 0; k < HV_PROB_K; ++k) {
        const float lp = r.logp[k];
        finite = finite && (k >= n || lp - lp == 0.0f);
    }
    if (n == 0) {
        h.cls = h.obj = -1;
        h.conf_ok = true;
    } else if (n <= HV_PROB_K && finite) { // the common case: both most likely ids from the copy (cached entry, else the argmax of its map)
        auto most_likely = [&](int which, int c) { // (scalars only: nothing of the copy may be reached through a run-time index)
            int32_t id = -1;
            float lp = -INFINITY;
            bool won = false;
#pragma unroll
            for (int k = 0; k < HV_PROB_K; ++k) {
                const bool mine = k < n && r.cls[k] == which;
                const bool take = mine && (c >= 0 ? k == c : (r.logp[k] > lp || (won && r.logp[k] == lp && r.obj[k] < id)));
                id = take ? r.obj[k] : id;
                lp = take ? r.logp[k] : lp;
                won = won || take;
            }
            return id;
        };
        h.obj = most_likely(0, prob2_best_obj(r.meta));
        h.cls = most_likely(1, prob2_best_cls(r.meta));
        h.conf_ok = true;
    } else { // overflow nodes / a NaN or an infinity in the maps
        h.cls = sem_class_id(v, nodes);
        h.obj = sem_object_id(v, nodes);
        h.conf_ok = sem_confidence_not_negative(v, nodes);
    }
    return h;
}
// sem_visit on the copy
__device__ __forceinline__ bool sem_visit_hot(const HvQuery &Q, unsigned long long block_key, int l, const HvSemParams &G, const HvVoteHot &h, float *uvd) {
    if (h.count < 1 || !h.conf_ok) return false;
    int32_t bk[3];
    hv_unpack_key(block_key, bk[0], bk[1], bk[2]);
    const int32_t lc[3] = {l % G.bs, (l / G.bs) % G.bs, l / (G.bs * G.bs)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (bk[a] < Q.bmin[a] || bk[a] > Q.bmax[a]) return false;
        const int32_t vk = bk[a] * G.bs + lc[a];
        if (vk < Q.vmin[a] || vk > Q.vmax[a]) return false;
    }
    const double c = (double)h.count;
    return hv_frustum_contains_d(Q, h.pos[0] / c, h.pos[1] / c, h.pos[2] / c, uvd);
}

// Block-level frustum cull for the per-voxel scans below (a wave = 64 consecutive voxels of ONE block: 64 divides bs^3 for the
// supported block sizes).  A voxel's averaged position lies in its cell, hence in the block's box; if all eight corners of the
// box (grown by half a voxel against rounding) are on the outer side of one of the frustum's six planes - depth_min, depth_max,
// u >= 0, u < W, v >= 0, v < H, each linear in camera coordinates - no voxel of the block can pass CameraFrustrum::contains and
// the wave leaves without reading a single voxel record (a keyframe sees a fraction of the map; the scan is otherwise
// bs^3 x 64..128 bytes per allocated block whether it is in view or not).  Lanes 0-7 test one corner each.
__device__ __forceinline__ bool sem_block_outside_frustum_key(const HvQuery &Q, unsigned long long block_key, const HvSemParams &G) {
    int32_t bk[3];
    hv_unpack_key(block_key, bk[0], bk[1], bk[2]);
    const int lane = hv_lane_id();
    const double vs = 1.0 / (double)G.inv_voxel_size, ext = (double)G.bs * vs;
    double p[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = (double)bk[a] * ext + (((lane >> a) & 1) ? ext + 0.5 * vs : -0.5 * vs);
    double pc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) pc[r] = (Q.R[r * 3 + 0] * p[0] + Q.R[r * 3 + 1] * p[1] + Q.R[r * 3 + 2] * p[2]) + Q.t[r];
    const double fu = (double)Q.fx * pc[0] + (double)Q.cx * pc[2], fv = (double)Q.fy * pc[1] + (double)Q.cy * pc[2];
    const unsigned long long corners = 0xffull;
    const bool out0 = pc[2] < (double)Q.depth_min, out1 = pc[2] > (double)Q.depth_max;
    const bool out2 = fu < 0.0, out3 = fu - (double)Q.width * pc[2] >= 0.0;
    const bool out4 = fv < 0.0, out5 = fv - (double)Q.height * pc[2] >= 0.0;
    return (__ballot(out0) & corners) == corners || (__ballot(out1) & corners) == corners || (__ballot(out2) & corners) == corners ||
           (__ballot(out3) & corners) == corners || (__ballot(out4) & corners) == corners || (__ballot(out5) & corners) == corners;
}

__device__ __forceinline__ bool sem_block_outside_frustum(const HvQuery &Q, const HvTable &table, int64_t b, const HvSemParams &G) {
    return sem_block_outside_frustum_key(Q, table.block_keys[b], G);
}

template <typename VOX>
__global__ __launch_bounds__(256) void k_sem_carve(HvTable table, VOX *__restrict__ pool, int64_t n_blocks, HvSemParams G,
                                                    HvQuery Q, const float *__restrict__ depth, const unsigned long long *__restrict__ occ) {
    // one wave per block, its occupied voxels only (sem_for_occupied)
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / HV_WAVE);
    for (int64_t b = (int64_t)blockIdx.x * (blockDim.x / HV_WAVE) + threadIdx.x / HV_WAVE; b < n_blocks; b += n_waves) {
        if ((G.nvox & 63) == 0 && sem_block_outside_frustum(Q, table, b, G)) continue; // wave-uniform
        sem_for_occupied(occ, b, G.nvox, false, [&](int64_t gid, bool active) {
            if (!active) return;
            VOX *v = pool + gid;
            float uvd[3];
            if (!sem_visit(Q, table, v, b, (int)(gid - b * G.nvox), G, uvd)) return;
            const float image_depth = depth[(int64_t)(int)uvd[1] * Q.width + (int)uvd[0]];
            if (image_depth <= 0.0f || !isfinite(image_depth)) return;
            if (uvd[2] < image_depth - Q.carve_threshold) sem_reset(v);
        });
    }
}

// ---- association -------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t vote_key(int32_t inst, int32_t obj) {
    return ((uint64_t)(uint32_t)inst << 32) | (uint64_t)(uint32_t)obj;
}

__device__ inline void vote_add(unsigned long long *__restrict__ vkeys, int32_t *__restrict__ vcounts, int32_t *overflow,
                                uint64_t key, int32_t n) {
    uint32_t s = hv_slot_hash(key) & (HV_VOTE_CAP - 1);
    for (uint32_t probe = 0; probe < HV_VOTE_CAP; ++probe) {
        unsigned long long k = vkeys[s];
        if (k == HV_VOTE_EMPTY) {
            k = atomicCAS(&vkeys[s], HV_VOTE_EMPTY, (unsigned long long)key);
            if (k == HV_VOTE_EMPTY) k = key;
        }
        if (k == key) {
            atomicAdd(&vcounts[s], n);
            return;
        }
        s = (s + 1) & (HV_VOTE_CAP - 1);
    }
    atomicAdd(overflow, 1);
}

// Workgroup-level vote table in LDS: a keyframe's votes fall on a handful of (instance, object) pairs, so with one global
// atomic per distinct key per WAVE every wave of the scan queued behind the same few addresses.  The waves of a
// workgroup add into HV_VOTE_LOCAL LDS slots instead (linear probing; a key that finds no slot goes to the global table
// directly), and the persistent workgroup flushes its slots once at the end of its share of the scan.
static constexpr int HV_VOTE_LOCAL = 64;
struct HvVoteLocal {
    unsigned long long keys[HV_VOTE_LOCAL];
    int32_t counts[HV_VOTE_LOCAL];
};

__device__ __forceinline__ void vote_local_init(HvVoteLocal &L) {
    for (int i = threadIdx.x; i < HV_VOTE_LOCAL; i += blockDim.x) {
        L.keys[i] = HV_VOTE_EMPTY;
        L.counts[i] = 0;
    }
    __syncthreads();
}

__device__ inline void vote_local_add(HvVoteLocal &L, unsigned long long *__restrict__ vkeys, int32_t *__restrict__ vcounts,
                                      int32_t *overflow, uint64_t key, int32_t n) {
    uint32_t s = hv_slot_hash(key) & (HV_VOTE_LOCAL - 1);
    for (int probe = 0; probe < HV_VOTE_LOCAL; ++probe) {
        unsigned long long k = L.keys[s];
        if (k == HV_VOTE_EMPTY) {
            k = atomicCAS(&L.keys[s], HV_VOTE_EMPTY, (unsigned long long)key);
            if (k == HV_VOTE_EMPTY) k = key;
        }
        if (k == key) {
            atomicAdd(&L.counts[s], n);
            return;
        }
        s = (s + 1) & (HV_VOTE_LOCAL - 1);
    }
    vote_add(vkeys, vcounts, overflow, key, n);
}

// every lane with key != EMPTY votes once; one LDS atomic per distinct key per wave
__device__ __forceinline__ void vote_wave_local(HvVoteLocal &L, unsigned long long *__restrict__ vkeys, int32_t *__restrict__ vcounts,
                                                int32_t *overflow, uint64_t key) {
    const int lane = hv_lane_id();
    unsigned long long remaining = __ballot(key != HV_VOTE_EMPTY);
    while (remaining) {
        const int first = __ffsll((long long)remaining) - 1;
        const unsigned long long fkey = __shfl((unsigned long long)key, first);
        const unsigned long long same = __ballot(key == fkey);
        if (lane == first) vote_local_add(L, vkeys, vcounts, overflow, fkey, (int32_t)__popcll(same));
        remaining &= ~same;
    }
}

__device__ __forceinline__ void vote_local_flush(HvVoteLocal &L, unsigned long long *__restrict__ vkeys, int32_t *__restrict__ vcounts,
                                                 int32_t *overflow) {
    __syncthreads();
    for (int i = threadIdx.x; i < HV_VOTE_LOCAL; i += blockDim.x)
        if (L.keys[i] != HV_VOTE_EMPTY) vote_add(vkeys, vcounts, overflow, L.keys[i], L.counts[i]);
}

struct HvAssocParams {
    int32_t use_depth, do_carving;
    float depth_threshold;
    int32_t pending_cap;
};

// process_point, voxel_semantic_data_association.h:190-246.  One wave per block of the pool: the block's box is tested against the
// frustum once, then only the voxels whose occupancy bit is set are visited (a 2 mm ScanNet keyframe faces 120 k blocks = 61 M
// voxel slots of which 7 % hold a voxel: the thread-per-slot form spent 0.7 - 1.0 ms per keyframe on per-wave overhead - cull,
// ballots, appends - for 950 k waves of mostly empty slots).
template <typename VOX, int MINWG>
__global__ __launch_bounds__(256, MINWG) void k_sem_assoc_vote(HvTable table, VOX *__restrict__ pool, int64_t n_blocks,
                                                         HvSemParams G, HvQuery Q, const int32_t *__restrict__ cls_img,
                                                         const int32_t *__restrict__ inst_img,
                                                         const float *__restrict__ depth, HvAssocParams A,
                                                         unsigned long long *__restrict__ vkeys,
                                                         int32_t *__restrict__ vcounts, int2 *__restrict__ pending,
                                                         const unsigned long long *__restrict__ occ, int64_t n_px) {
    __shared__ HvVoteLocal s_votes;
    // Voxels waiting for their object id are collected per wave in LDS and get their places in the pending list with ONE returning
    // atomic per ~200 of them (round 6): a 2 mm keyframe leaves a few hundred thousand such voxels behind, and one atomic per visit
    // with a pending lane - tens of thousands on a single word, ~10 ns each, serialised - was most of this kernel.
    constexpr int PEND_BUF = 256;
    __shared__ int2 s_pend[4][PEND_BUF];
    int2 *pend = s_pend[threadIdx.x / HV_WAVE];
    int n_pend = 0; // (wave-uniform)
    const unsigned long long lane_lt = hv_lane_id() == 0 ? 0ull : (~0ull >> (64 - hv_lane_id()));
    auto flush_pending = [&]() {
        if (n_pend == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int32_t base = 0;
        if (hv_lane_id() == 0) base = atomicAdd(&table.counters[HV_CNT_AUX], n_pend);
        base = __shfl(base, 0);
        for (int k = hv_lane_id(); k < n_pend; k += HV_WAVE)
            if (base + k < A.pending_cap) pending[base + k] = pend[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        n_pend = 0;
    };
    // The occupied voxels of the blocks in view are QUEUED per wave in LDS and visited 64 at a time, every lane busy (round 6).  A
    // 2 mm keyframe faces 160 k blocks holding 15 occupied voxels each on average: visited block by block, a wave went through its
    // two dependent round trips (record -> image pixels) once per block with a quarter of its lanes, 160 k times per keyframe; from
    // the queue it does so once per 64 voxels.  Votes and pending voxels are order-free (counts; a list the apply pass scatters).
    constexpr int Q_CAP = 256;
    __shared__ int32_t s_queue[4][Q_CAP];
    int32_t *queue = s_queue[threadIdx.x / HV_WAVE];
    int n_q = 0; // (wave-uniform)
    vote_local_init(s_votes);
    if (n_blocks < 0) n_blocks = min(table.counters[HV_CNT_BLOCKS], table.max_blocks); // (the host does not wait to learn it)
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / HV_WAVE);
    // one voxel of block b (pool slot gid): process_point, voxel_semantic_data_association.h:190-246
    auto visit_b = [&](int64_t gid, bool active, int64_t b) {
        uint64_t key = HV_VOTE_EMPTY;
        bool is_pending = false;
        int32_t inst = -1;
        if (active) {
            VOX *v = pool + gid;
            const unsigned long long bkey = table.block_keys[b]; // (requested with the record)
            const HvVoteHot h = vote_hot(v, table.prob_nodes);
            float uvd[3];
            if (sem_visit_hot(Q, bkey, (int)(gid - b * G.nvox), G, h, uvd)) {
                // the three pixels in one trip as well
                const int64_t px = (int64_t)(int)uvd[1] * Q.width + (int)uvd[0];
                const int32_t image_class = cls_img[px];
                const float image_depth = A.use_depth ? depth[px] : 1.0f;
                inst = inst_img[px];
                const int32_t point_class = h.cls;
                bool go = image_class >= 0 && point_class >= 0 && point_class == image_class && inst >= 0;
                if (go && A.use_depth) {
                    if (image_depth <= 0.0f || !isfinite(image_depth)) {
                        go = false;
                    } else if (A.do_carving && uvd[2] < image_depth - A.depth_threshold) {
                        sem_reset(v);
                        go = false;
                    } else if (uvd[2] > image_depth + A.depth_threshold) {
                        go = false;
                    }
                }
                if (go) {
                    int32_t obj = h.obj;
                    if (obj < 0) {
                        if (inst == 0) {
                            obj = 0;
                            sem_set_object_id(v, table, 0);
                        } else {
                            obj = HV_OBJ_PENDING;
                            is_pending = true;
                        }
                    }
                    key = vote_key(inst, obj);
                }
            }
        }
        vote_wave_local(s_votes, vkeys, vcounts, &table.counters[HV_CNT_OUT2], key);
        const unsigned long long pm = __ballot(is_pending);
        if (pm) {
            if (is_pending) pend[n_pend + __popcll(pm & lane_lt)] = make_int2((int32_t)gid, inst);
            n_pend += __popcll(pm);
            if (n_pend > PEND_BUF - HV_WAVE) flush_pending();
        }
    };
    auto drain_queue = [&](bool all) { // ONE instance of the visit in the kernel
        hv_wave_lds_sync();
        const int upto = all ? n_q : (n_q & ~(HV_WAVE - 1));
        for (int i = 0; i < upto; i += HV_WAVE) {
            const bool active = i + hv_lane_id() < n_q;
            const int32_t gid = active ? queue[i + hv_lane_id()] : 0;
            visit_b((int64_t)gid, active, (int64_t)(gid / G.nvox));
        }
        const int rem = n_q - min(upto, n_q); // (< 64) stays queued, moved to the front
        const int32_t keep = hv_lane_id() < rem ? queue[upto + hv_lane_id()] : 0;
        hv_wave_lds_sync();
        if (hv_lane_id() < rem) queue[hv_lane_id()] = keep;
        n_q = rem;
        hv_wave_lds_sync();
    };
    auto enqueue = [&](int64_t gid, bool active) {
        const unsigned long long m = __ballot(active);
        if (active) queue[n_q + (int)__popcll(m & lane_lt)] = (int32_t)gid;
        n_q += (int)__popcll(m);
        if (n_q > Q_CAP - HV_WAVE) drain_queue(false);
    };
    // the key and the occupancy words of the NEXT block are requested before this block is worked on: a block is a chain of
    // dependent round trips (key -> cull, words -> records -> image pixels) and a wave holds too many registers for the SIMD to
    // hide them with other waves
    const bool words = sem_occ_words_usable(occ, G.nvox);
    const int W = G.nvox >> 6, lane = hv_lane_id();
    int64_t b = (int64_t)blockIdx.x * (blockDim.x / HV_WAVE) + threadIdx.x / HV_WAVE;
    unsigned long long next_key = b < n_blocks ? table.block_keys[b] : 0ull;
    unsigned long long next_word = (words && b < n_blocks && lane < W) ? occ[b * W + lane] : 0ull;
    for (; b < n_blocks; b += n_waves) {
        const unsigned long long bkey = next_key, word = next_word;
        if (b + n_waves < n_blocks) {
            next_key = table.block_keys[b + n_waves];
            next_word = (words && lane < W) ? occ[(b + n_waves) * W + lane] : 0ull;
        }
        if (words && !__any(word != 0ull)) continue;                                      // nothing ever landed in this block
        if ((G.nvox & 63) == 0 && sem_block_outside_frustum_key(Q, bkey, G)) continue; // wave-uniform
        if (words) sem_for_occupied_word(word, b, G.nvox, enqueue);
        else sem_for_occupied(occ, b, G.nvox, false, enqueue);
    }
    drain_queue(true);
    flush_pending();
    // the image scan of voxel_semantic_data_association.h:316-331 - every instance id with a valid class - into the same workgroup table:
    // a launch of its own until round 6 (18 us at 1296x968, 10 us at 640x480 for a pass over two images); here it fills the tail of this
    // kernel, where most waves have run out of blocks (the table takes the markers in any order)
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n_px; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + threadIdx.x;
        uint64_t key = HV_VOTE_EMPTY;
        if (i < n_px) {
            const int32_t inst_px = inst_img[i];
            if (inst_px >= 0 && cls_img[i] >= 0) key = vote_key(inst_px, HV_OBJ_SEEN);
        }
        vote_wave_local(s_votes, vkeys, vcounts, &table.counters[HV_CNT_OUT2], key);
    }
    vote_local_flush(s_votes, vkeys, vcounts, &table.counters[HV_CNT_OUT2]);
}

__global__ __launch_bounds__(256) void k_sem_assoc_compact(HvTable table, unsigned long long *__restrict__ vkeys,
                                                            int32_t *__restrict__ vcounts,
                                                            unsigned long long *__restrict__ out_keys,
                                                            int32_t *__restrict__ out_counts) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool pred = s < HV_VOTE_CAP && vkeys[s] != HV_VOTE_EMPTY;
    const int32_t at = hv_wave_append(&table.counters[HV_CNT_OUT], pred);
    if (pred) {
        out_keys[at] = vkeys[s];
        out_counts[at] = vcounts[s];
        vkeys[s] = HV_VOTE_EMPTY; // the table is empty again for the next keyframe (no 768 KB of memsets per call)
        vcounts[s] = 0;
    }
}

// Multi-GPU, device-resident exchange of the pair lists (hv_assoc_pairs_export / _import).  Message of one GPU, int64 words:
// [0] = number of pairs n (<= cap), [1 .. cap] = keys, [1 + cap .. 2 cap] = votes.
__global__ __launch_bounds__(256) void k_assoc_pairs_export(HvTable table, const unsigned long long *__restrict__ ckeys,
                                                             const int32_t *__restrict__ ccounts, long long *__restrict__ msg, int cap,
                                                             int32_t *__restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_all = table.counters[HV_CNT_OUT];
    const int n = min(n_all, cap);
    if (i == 0) {
        msg[0] = n;
        if (n_all > cap) atomicOr(flags, 1); // HV_ASSOC_TOO_MANY_PAIRS: latched by the decide stage's last kernel
    }
    if (i < n) {
        msg[1 + i] = (long long)ckeys[i];
        msg[1 + cap + i] = ccounts[i];
    }
}

// every GPU's pairs into the (empty) vote table: equal (instance, object) pairs of different GPUs - and the image markers every GPU
// contributes - add up to ONE pair, so that the merged list is as long as a single GPU's would be
__global__ __launch_bounds__(256) void k_assoc_pairs_import(HvTable table, const long long *__restrict__ msgs, int world, int cap,
                                                             unsigned long long *__restrict__ vkeys, int32_t *__restrict__ vcounts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (r >= world) return;
    const long long *msg = msgs + (size_t)r * (1 + 2 * (size_t)cap);
    const int n = (int)min((long long)cap, max(0ll, msg[0]));
    if (i < n) vote_add(vkeys, vcounts, &table.counters[HV_CNT_OUT2], (uint64_t)msg[1 + i], (int32_t)msg[1 + cap + i]);
}

__device__ __forceinline__ int32_t map_lookup(const int32_t *__restrict__ map_inst, const int32_t *__restrict__ map_obj,
                                              int32_t n_map, int32_t inst, int32_t missing) {
    int lo = 0, hi = n_map - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int32_t k = map_inst[mid];
        if (k == inst) return map_obj[mid];
        if (k < inst) lo = mid + 1;
        else hi = mid - 1;
    }
    return missing;
}

// The reference's voting rules (voxel_semantic_data_association.h:268-361) on the compacted (instance, object, votes) pairs of one
// keyframe, on the device: one workgroup sorts the pairs by (instance, object) in LDS, one thread walks them - a keyframe has a few
// dozen pairs.  New object ids come from the process-wide counter in device memory, handed out in ascending instance order.
// map = {inst[HV_VOTE_CAP], obj[HV_VOTE_CAP]} sorted by instance id, *n_map its size; *flags |= 1 when the pairs did not fit the
// workgroup (the host reports it when the map is fetched).  No host round trip between the vote and the remap / integrate.
static constexpr int HV_RULES_MAX = 4096;
// capacity flags of one association (S.flags; latched into HvStatus::assoc_flags by k_sem_assoc_apply, reported by the next call)
static constexpr int32_t HV_ASSOC_TOO_MANY_PAIRS = 1, HV_ASSOC_VOTE_TABLE_FULL = 2, HV_ASSOC_PENDING_FULL = 4;
__global__ __launch_bounds__(1024) void k_sem_assoc_rules(HvTable table, const unsigned long long *__restrict__ pkeys,
                                                           const int32_t *__restrict__ pcounts, float min_vote_ratio, int32_t min_votes,
                                                           int32_t *__restrict__ next_object_id, int32_t *__restrict__ map_inst,
                                                           int32_t *__restrict__ map_obj, int32_t *__restrict__ n_map_out,
                                                           int32_t *__restrict__ flags) {
    __shared__ unsigned long long s_key[HV_RULES_MAX];
    __shared__ int32_t s_cnt[HV_RULES_MAX];
    int n = table.counters[HV_CNT_OUT];
    if (n > HV_RULES_MAX) {
        if (threadIdx.x == 0) {
            *flags = HV_ASSOC_TOO_MANY_PAIRS; // (this call's flags: k_sem_assoc_apply adds the vote's and latches them into the status word)
            *n_map_out = 0;
        }
        return;
    }
    if (threadIdx.x == 0) *flags = 0;
    int m2 = 1;
    while (m2 < n) m2 <<= 1;
    for (int i = threadIdx.x; i < m2; i += blockDim.x) {
        if (i < n) {
            const unsigned long long k = pkeys[i]; // instance << 32 | object (two's complement)
            s_key[i] = (k & 0xffffffff00000000ull) | (unsigned long long)((uint32_t)k ^ 0x80000000u); // unsigned order == (inst, signed obj)
            s_cnt[i] = pcounts[i];
        } else {
            s_key[i] = ~0ull;
            s_cnt[i] = 0;
        }
    }
    __syncthreads();
    for (int k = 2; k <= m2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < m2; t += blockDim.x) {
                const int x = t ^ j;
                if (x > t) {
                    const unsigned long long a = s_key[t], b = s_key[x];
                    if ((a > b) == ((t & k) == 0)) {
                        s_key[t] = b;
                        s_key[x] = a;
                        const int32_t c = s_cnt[t];
                        s_cnt[t] = s_cnt[x];
                        s_cnt[x] = c;
                    }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x != 0) return;
    int32_t next_id = *next_object_id;
    int n_map = 0;
    for (int i = 0; i < n;) {
        const int32_t inst = (int32_t)(uint32_t)(s_key[i] >> 32);
        int j = i;
        int32_t pending = 0;
        bool seen = false;
        while (j < n && (int32_t)(uint32_t)(s_key[j] >> 32) == inst) { // PENDING and SEEN sort first (the two smallest object values)
            const int32_t obj = (int32_t)((uint32_t)s_key[j] ^ 0x80000000u);
            if (obj == HV_OBJ_PENDING) pending += s_cnt[j];
            else if (obj == HV_OBJ_SEEN) seen = true;
            else break;
            ++j;
        }
        // the instance's votes in ascending object-id order (std::map), the new id - if voxels wait for one - at its place in that order
        int32_t new_id = -1;
        if (pending > 0) new_id = next_id++;
        int max_votes = 0, winning = -1, total_votes = 0;
        bool have_votes = pending > 0, new_done = pending <= 0;
        auto take = [&](int32_t obj, int32_t cnt) {
            total_votes += cnt;
            if (cnt > max_votes) {
                max_votes = cnt;
                winning = obj;
            }
        };
        while (j < n && (int32_t)(uint32_t)(s_key[j] >> 32) == inst) {
            const int32_t obj = (int32_t)((uint32_t)s_key[j] ^ 0x80000000u);
            int32_t cnt = s_cnt[j];
            ++j;
            while (j < n && s_key[j] == s_key[j - 1]) cnt += s_cnt[j++]; // (the same pair from several lists: multi-GPU merge)
            if (!new_done && new_id < obj) {
                take(new_id, pending);
                new_done = true;
            }
            if (!new_done && new_id == obj) { // (cannot happen with a monotone counter; the reference would add the counts)
                cnt += pending;
                new_done = true;
            }
            have_votes = true;
            take(obj, cnt);
        }
        if (!new_done) take(new_id, pending);
        bool present = false;
        int32_t result = -1;
        if (have_votes) {
            present = true;
            if (total_votes >= min_votes) {
                const float ratio = (float)max_votes / (float)total_votes;
                result = ratio < min_vote_ratio ? -1 : winning;
            }
        }
        if (seen) {
            if (inst == 0) {
                result = 0;
                present = true;
            } else if (!present) {
                result = -1;
                present = true;
            }
        }
        if (present) {
            map_inst[n_map] = inst;
            map_obj[n_map] = result;
            ++n_map;
        }
        i = j;
    }
    *next_object_id = next_id;
    *n_map_out = n_map;
}

// deferred set_object_id of the voxels that waited for the vote, voxel_semantic_data_association.h:344-361
// (the pending count and the map's size are read on the device: nothing of the association goes through the host)
template <typename VOX>
__global__ __launch_bounds__(256) void k_sem_assoc_apply(HvTable table, VOX *__restrict__ pool, const int2 *__restrict__ pending,
                                                          int32_t pending_cap, const int32_t *__restrict__ map_inst,
                                                          const int32_t *__restrict__ map_obj, const int32_t *__restrict__ n_map_p,
                                                          int32_t *__restrict__ flags, HvStatus *status) {
    const int32_t n_pending = min(table.counters[HV_CNT_AUX], pending_cap);
    const int32_t n_map = *n_map_p;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the device flow (assign -> remap_instance_ids -> integrate) never fetches the map, and the next vote / fold clear
        // the shared counters - so what overflowed in THIS association is folded into its flags word here and latched into the pinned
        // status word, which the next call's gate reads without a synchronisation
        int32_t f = *flags | flags[1]; // flags[1]: what the stages before the rules kernel found (hv_assoc_pairs_export)
        flags[1] = 0;
        if (table.counters[HV_CNT_OUT2] != 0) f |= HV_ASSOC_VOTE_TABLE_FULL;
        if (table.counters[HV_CNT_AUX] > pending_cap) f |= HV_ASSOC_PENDING_FULL;
        *flags = f;
        if (f != 0) {
            volatile HvStatus *s = status;
            s->assoc_flags = s->assoc_flags | f;
            __threadfence_system();
        }
    }
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pending; i += gridDim.x * blockDim.x) {
        const int2 p = pending[i];
        const int32_t final_id = map_lookup(map_inst, map_obj, n_map, p.y, -1);
        if (final_id >= 0) sem_set_object_id(pool + p.x, table, final_id);
    }
}

// n_map_p != nullptr: the map's size lives on the device (the map of the volume's last association)
__global__ __launch_bounds__(256) void k_remap_instance_ids(const int32_t *__restrict__ in, int64_t n,
                                                             const int32_t *__restrict__ map_inst,
                                                             const int32_t *__restrict__ map_obj, int32_t n_map,
                                                             const int32_t *__restrict__ n_map_p, int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (n_map_p != nullptr) n_map = *n_map_p;
    // an EMPTY map hands the image back as it is - the early return of the module's binding, image_utils_module.h:52-58, which is what
    // pySLAM calls (the C++ template behind it would set every id to -1, image_utils.h:108-141)
    out[i] = n_map > 0 ? map_lookup(map_inst, map_obj, n_map, in[i], -1) : in[i];
}

// ---- segments ------------------------------------------------------------------------------------
template <typename VOX>
__global__ __launch_bounds__(256) void k_seg_collect(HvTable table, const VOX *__restrict__ pool, int64_t n_blocks, int nvox,
                                                      int min_count, float min_confidence,
                                                      unsigned long long *__restrict__ out_keys, int64_t cap,
                                                      const unsigned long long *__restrict__ occ) {
    // a wave per block, occupied voxels only; the keys a wave finds wait in its LDS window and take their place in the output with
    // one returning atomic per ~450 (k_sem_collect has the measurement)
    constexpr int BUF = 512;
    __shared__ unsigned long long s_buf[4][BUF];
    const int wave = threadIdx.x / HV_WAVE, lane = hv_lane_id();
    unsigned long long *buf = s_buf[wave];
    int n_buf = 0;
    int64_t n_found = 0;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    auto flush = [&]() {
        if (n_buf == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int32_t base = 0;
        if (lane == 0) base = atomicAdd(&table.counters[HV_CNT_OUT], n_buf);
        base = __shfl(base, 0);
        for (int i = lane; i < n_buf; i += HV_WAVE)
            if ((int64_t)base + i < cap) out_keys[(int64_t)base + i] = buf[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        n_buf = 0;
    };
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / HV_WAVE);
    for (int64_t b = (int64_t)blockIdx.x * (blockDim.x / HV_WAVE) + wave; b < n_blocks; b += n_waves) {
        sem_for_occupied(occ, b, nvox, false, [&](int64_t gid, bool active) {
            bool pred = false;
            int32_t obj = -1;
            if (active) {
                const VOX *v = pool + gid;
                // NB: strict '>' on the count, voxel_block_semantic_grid.hpp:224
                if (v->count > min_count && sem_confidence(v, table.prob_nodes) >= min_confidence) {
                    obj = sem_object_id(v, table.prob_nodes);
                    pred = obj >= 0;
                }
            }
            const unsigned long long m = __ballot(pred);
            if (m == 0ull) return;
            if (out_keys == nullptr) {
                n_found += __popcll(m);
                return;
            }
            if (pred) buf[n_buf + __popcll(m & lt)] = ((unsigned long long)(uint32_t)obj << 32) | (unsigned long long)(uint32_t)gid;
            n_buf += __popcll(m);
            if (n_buf > BUF - HV_WAVE) flush();
        });
    }
    if (out_keys != nullptr) flush();
    else if (lane == 0 && n_found) atomicAdd(&table.counters[HV_CNT_OUT], (int32_t)n_found);
}

template <typename VOX>
__global__ __launch_bounds__(256) void k_seg_rows(const VOX *__restrict__ pool, const unsigned long long *__restrict__ keys,
                                                   int64_t n, double *__restrict__ out_pts, float *__restrict__ out_cols,
                                                   int32_t *__restrict__ out_obj, int32_t *__restrict__ out_cls,
                                                   float *__restrict__ out_conf, const void *__restrict__ nodes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const VOX *v = pool + (uint32_t)(k & 0xffffffffull);
    const double c = (double)v->count;
    const float cf = (float)v->count;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        out_pts[i * 3 + a] = v->pos[a] / c;
        out_cols[i * 3 + a] = v->col[a] / cf;
    }
    out_obj[i] = (int32_t)(k >> 32);
    out_cls[i] = sem_class_id(v, nodes);
    out_conf[i] = sem_confidence(v, nodes);
}

// op 0 merge_segments(a <- b), 1 remove_segment(a), 2 remove_low_confidence_segments(int a),
// 3 remove_low_count_voxels(a), 4 remove_low_confidence_voxels(fa)
template <typename VOX>
__global__ __launch_bounds__(256) void k_sem_segment_op(HvTable table, VOX *__restrict__ pool, int64_t n_voxels, int op, int32_t a, int32_t b,
                                                         float fa, const unsigned long long *__restrict__ occ) {
    const void *nodes = table.prob_nodes;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_voxels) return;
    if (!sem_maybe_occupied(occ, gid)) return; // a voxel that never took a point: every op leaves its zero record as it is
    VOX *v = pool + gid;
    if (op == 0) {
        if (sem_object_id(v, nodes) == b) sem_set_object_id(v, table, a);
    } else if (op == 1) {
        if (sem_object_id(v, nodes) == a) sem_reset(v);
    } else if (op == 2) {
        if (sem_confidence(v, nodes) < (float)a) sem_reset(v);
    } else if (op == 3) {
        if (v->count < a) sem_reset(v);
    } else if (op == 4) {
        if (sem_confidence(v, nodes) < fa) sem_reset(v);
    }
}

template <typename VOX>
__global__ __launch_bounds__(256) void k_sem_count_nonempty(HvTable table, const VOX *__restrict__ pool, int64_t n_blocks, int nvox,
                                                             const unsigned long long *__restrict__ occ) {
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / HV_WAVE);
    for (int64_t b = (int64_t)blockIdx.x * (blockDim.x / HV_WAVE) + threadIdx.x / HV_WAVE; b < n_blocks; b += n_waves)
        sem_for_occupied(occ, b, nvox, false, [&](int64_t gid, bool active) {
            (void)hv_wave_append(&table.counters[HV_CNT_OUT], active && pool[gid].count > 0);
        });
}

// ---- host: PCA oriented bounding box, bounding_boxes_3d.cpp:373-553 ------------------------------
namespace {

struct Vec3 {
    double v[3];
};

void jacobi_eigen3(const double A[9], double evals[3], double evecs[9]) {
    double a[3][3], q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) a[r][c] = A[r * 3 + c];
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int r = p + 1; r < 3; ++r) {
                if (a[p][r] == 0.0) continue;
                const double theta = (a[r][r] - a[p][p]) / (2.0 * a[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[k][p], akr = a[k][r];
                    a[k][p] = c * akp - s * akr;
                    a[k][r] = s * akp + c * akr;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[p][k], ark = a[r][k];
                    a[p][k] = c * apk - s * ark;
                    a[r][k] = s * apk + c * ark;
                }
                for (int k = 0; k < 3; ++k) {
                    const double qkp = q[k][p], qkr = q[k][r];
                    q[k][p] = c * qkp - s * qkr;
                    q[k][r] = s * qkp + c * qkr;
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        evals[i] = a[i][i];
        for (int k = 0; k < 3; ++k) evecs[k * 3 + i] = q[k][i]; // column i = eigenvector i
    }
}

// Eigen::Quaterniond(R) (rotation matrix -> quaternion), output {w, x, y, z}
void quat_from_matrix(const double R[9], double q[4]) {
    auto m = [&](int r, int c) { return R[r * 3 + c]; };
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m(2, 1) - m(1, 2)) * t;
        q[2] = (m(0, 2) - m(2, 0)) * t;
        q[3] = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m(k, j) - m(j, k)) * t;
        q[1 + j] = (m(j, i) + m(i, j)) * t;
        q[1 + k] = (m(k, i) + m(i, k)) * t;
    }
}

void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
double norm3(const double a[3]) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// extents of the points in frame (c, R): fills center / size / quaternion of obb
void obb_from_frame(const double *pts, int64_t n, const double c[3], double R[9], double *obb) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i) {
        const double d[3] = {pts[i * 3] - c[0], pts[i * 3 + 1] - c[1], pts[i * 3 + 2] - c[2]};
        for (int a = 0; a < 3; ++a) {
            const double l = R[0 * 3 + a] * d[0] + R[1 * 3 + a] * d[1] + R[2 * 3 + a] * d[2]; // R^T d
            mn[a] = std::min(mn[a], l);
            mx[a] = std::max(mx[a], l);
        }
    }
    double cl[3];
    for (int a = 0; a < 3; ++a) {
        obb[7 + a] = 2.0 * (0.5 * (mx[a] - mn[a]));
        cl[a] = 0.5 * (mx[a] + mn[a]);
    }
    for (int r = 0; r < 3; ++r) obb[r] = c[r] + (R[r * 3 + 0] * cl[0] + R[r * 3 + 1] * cl[1] + R[r * 3 + 2] * cl[2]);
    quat_from_matrix(R, obb + 3);
}

// compute_obb_pca_3d; obb = {center xyz, quaternion wxyz, size xyz}
void compute_obb_pca(const double *pts, int64_t n, double *obb) {
    for (int i = 0; i < 10; ++i) obb[i] = 0.0;
    obb[3] = 1.0;
    if (n == 0) return;
    if (n == 1) {
        obb[0] = pts[0]; obb[1] = pts[1]; obb[2] = pts[2];
        return;
    }
    if (n == 2) {
        double c[3], diff[3];
        for (int a = 0; a < 3; ++a) {
            c[a] = 0.5 * (pts[a] + pts[3 + a]);
            diff[a] = pts[3 + a] - pts[a];
        }
        const double dn = norm3(diff);
        if (dn < 1e-10) {
            obb[0] = c[0]; obb[1] = c[1]; obb[2] = c[2];
            return;
        }
        double a1[3] = {diff[0] / dn, diff[1] / dn, diff[2] / dn}, a2[3], a3[3];
        const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0};
        cross3(std::fabs(a1[0]) < 0.9 ? ex : ey, a1, a2);
        double nn = norm3(a2);
        for (int a = 0; a < 3; ++a) a2[a] /= nn;
        cross3(a1, a2, a3);
        nn = norm3(a3);
        for (int a = 0; a < 3; ++a) a3[a] /= nn;
        double R[9];
        for (int r = 0; r < 3; ++r) { R[r * 3 + 0] = a1[r]; R[r * 3 + 1] = a2[r]; R[r * 3 + 2] = a3[r]; }
        double c01[3];
        cross3(a1, a2, c01);
        if (c01[0] * a3[0] + c01[1] * a3[1] + c01[2] * a3[2] < 0.0)
            for (int r = 0; r < 3; ++r) R[r * 3 + 2] = -R[r * 3 + 2];
        obb_from_frame(pts, n, c, R, obb);
        return;
    }
    // centroid and covariance in one pass (Welford)
    double centroid[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t k = 0; k < n; ++k) {
        double delta[3], delta2[3];
        for (int a = 0; a < 3; ++a) delta[a] = pts[k * 3 + a] - centroid[a];
        for (int a = 0; a < 3; ++a) centroid[a] += delta[a] / (double)(k + 1);
        for (int a = 0; a < 3; ++a) delta2[a] = pts[k * 3 + a] - centroid[a];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) cov[r * 3 + c] += delta[r] * delta2[c];
    }
    for (int i = 0; i < 9; ++i) cov[i] /= (double)n;
    // SelfAdjointEigenSolver reads the lower triangle: symmetrise from it
    double sym[9] = {cov[0], cov[3], cov[6], cov[3], cov[4], cov[7], cov[6], cov[7], cov[8]};
    double evals[3], evecs[9];
    jacobi_eigen3(sym, evals, evecs);
    // SelfAdjointEigenSolver returns ascending eigenvalues; the reference then sorts descending with a
    // swap sort over {0,1,2}
    int asc[3] = {0, 1, 2};
    std::sort(asc, asc + 3, [&](int x, int y) { return evals[x] < evals[y]; });
    double ev[3] = {evals[asc[0]], evals[asc[1]], evals[asc[2]]};
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (ev[order[j]] > ev[order[i]]) std::swap(order[i], order[j]);
    double R[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) R[r * 3 + c] = evecs[r * 3 + asc[order[c]]];
    const double c0[3] = {R[0], R[3], R[6]}, c1[3] = {R[1], R[4], R[7]}, c2[3] = {R[2], R[5], R[8]};
    double cr[3];
    cross3(c0, c1, cr);
    if (cr[0] * c2[0] + cr[1] * c2[1] + cr[2] * c2[2] < 0.0)
        for (int r = 0; r < 3; ++r) R[r * 3 + 2] = -R[r * 3 + 2];
    obb_from_frame(pts, n, centroid, R, obb);
}

struct HvSegmentsCache {
    std::vector<double> pts;
    std::vector<float> cols;
    std::vector<int32_t> row_obj;
    std::vector<int32_t> ids;  // per object {object_id, class_id, n_points}
    std::vector<float> conf;   // per object {min, max}
    std::vector<double> obb;   // per object {center xyz, quaternion wxyz, size xyz}
};

template <typename VOX> int sem_launch_segment_op(hv_volume *v, int op, int32_t a, int32_t b, float fa) {
    HV_HIP(hipSetDevice(v->device));
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    if (nb == 0) return HV_OK;
    const int64_t total = nb * sem_params(v).nvox;
    hipLaunchKernelGGL(k_sem_segment_op<VOX>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->stream, v->table, (VOX *)v->pool,
                       total, op, a, b, fa, (op == 0 && b < 0) ? nullptr : v->occ); // (merging INTO the voxels without an object id reaches empty ones too)
    HV_HIP(hipGetLastError());
    return HV_OK;
}

} // namespace

void hv_segments_cache_free(void *cache) { delete static_cast<HvSegmentsCache *>(cache); }

int hv_sem_carve(hv_volume *v, const HvQuery &Q, const float *d_depth, int64_t nb) {
    const HvSemParams G = sem_params(v);
    const dim3 grid((unsigned)std::min<int64_t>((nb + 3) / 4, 8192)); // a wave per block, grid-stride
    HV_SEM_DISPATCH(v, hipLaunchKernelGGL(k_sem_carve<VOX>, grid, dim3(256), 0, v->stream, v->table, (VOX *)v->pool, nb, G, Q, d_depth, v->occ));
    HV_HIP(hipGetLastError());
    return HV_OK;
}

int hv_sem_segment_op(hv_volume *v, int op, int32_t a, int32_t b, float fa) {
    HV_SEM_DISPATCH(v, return sem_launch_segment_op<VOX>(v, op, a, b, fa));
    return HV_OK; // (not reached)
}

int hv_sem_size(hv_volume *v, int64_t *n) {
    HV_HIP(hipSetDevice(v->device));
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    *n = 0;
    if (nb == 0) return HV_OK;
    const int nvox = sem_params(v).nvox;
    HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT], 0, sizeof(int32_t), v->stream));
    const dim3 grid((unsigned)std::min<int64_t>((nb + 3) / 4, 8192));
    HV_SEM_DISPATCH(v, hipLaunchKernelGGL(k_sem_count_nonempty<VOX>, grid, dim3(256), 0, v->stream, v->table, (const VOX *)v->pool, nb, nvox, v->occ));
    HV_HIP(hipGetLastError());
    rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    *n = v->h_counters[HV_CNT_OUT];
    return HV_OK;
}

extern "C" {

int32_t hv_peek_next_object_id(void) {
    if (g_dev_last >= 0 && g_dev_next_id[g_dev_last] != nullptr) {
        int32_t value = 0;
        if (hipSetDevice(g_dev_last) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
            hipMemcpy(&value, g_dev_next_id[g_dev_last], sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess)
            g_next_object_id.store(value);
    }
    return g_next_object_id.load();
}
void hv_set_next_object_id(int32_t id) {
    g_next_object_id.store(id);
    for (int d = 0; d < 64; ++d)
        if (g_dev_next_id[d] != nullptr && hipSetDevice(d) == hipSuccess && hipDeviceSynchronize() == hipSuccess)
            (void)hipMemcpy(g_dev_next_id[d], &id, sizeof(int32_t), hipMemcpyHostToDevice);
}

int hv_merge_segments(hv_volume *v, int32_t instance_id1, int32_t instance_id2) {
    HV_REQUIRE(v != nullptr && hv_is_semantic(v), HV_ERR_MODE, "hv_merge_segments: not a semantic grid");
    return hv_sem_segment_op(v, 0, instance_id1, instance_id2, 0.f);
}
int hv_remove_segment(hv_volume *v, int32_t object_id) {
    HV_REQUIRE(v != nullptr && hv_is_semantic(v), HV_ERR_MODE, "hv_remove_segment: not a semantic grid");
    return hv_sem_segment_op(v, 1, object_id, 0, 0.f);
}
int hv_remove_low_confidence_segments(hv_volume *v, int32_t min_confidence) {
    HV_REQUIRE(v != nullptr && hv_is_semantic(v), HV_ERR_MODE, "hv_remove_low_confidence_segments: not a semantic grid");
    return hv_sem_segment_op(v, 2, min_confidence, 0, 0.f);
}
int hv_remove_low_confidence_voxels(hv_volume *v, float min_confidence) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_remove_low_confidence_voxels: null volume");
    if (!hv_is_semantic(v)) return HV_OK; // no-op for non-semantic payloads, voxel_block_grid.hpp:650-676
    return hv_sem_segment_op(v, 4, 0, 0, min_confidence);
}

int hv_remap_instance_ids(hv_volume *v, const int32_t *instance_ids, int32_t height, int32_t width, const int32_t *map_inst,
                          const int32_t *map_obj, int64_t n_map, int32_t *out, int32_t loc) {
    HV_REQUIRE(v != nullptr && instance_ids != nullptr && out != nullptr, HV_ERR_INVALID, "hv_remap_instance_ids: null argument");
    HV_REQUIRE(n_map == 0 || (map_inst != nullptr && map_obj != nullptr), HV_ERR_INVALID, "hv_remap_instance_ids: null map");
    HV_HIP(hipSetDevice(v->device));
    const int64_t n = (int64_t)height * width;
    if (n == 0) return HV_OK;
    // sort the map by instance id on the host (tiny), upload next to the image
    std::vector<std::pair<int32_t, int32_t>> m((size_t)n_map);
    for (int64_t i = 0; i < n_map; ++i) m[i] = {map_inst[i], map_obj[i]};
    std::sort(m.begin(), m.end());
    std::vector<int32_t> flat((size_t)n_map * 2);
    for (int64_t i = 0; i < n_map; ++i) { flat[i] = m[i].first; flat[n_map + i] = m[i].second; }
    const size_t img_bytes = sizeof(int32_t) * (size_t)n;
    const size_t map_bytes = sizeof(int32_t) * 2 * (size_t)n_map;
    int rc = hv_ensure_buffer(v, &v->stage_b, &v->stage_b_bytes, 2 * img_bytes + map_bytes + 512);
    if (rc != HV_OK) return rc;
    char *base = (char *)v->stage_b;
    int32_t *d_map = (int32_t *)base;
    const int32_t *d_in = instance_ids;
    int32_t *d_out = out;
    char *cursor = base + ((map_bytes + 255) & ~(size_t)255);
    if (n_map > 0) HV_HIP(hipMemcpyAsync(d_map, flat.data(), map_bytes, hipMemcpyHostToDevice, v->stream));
    if (loc == HV_HOST) {
        HV_HIP(hipMemcpyAsync(cursor, instance_ids, img_bytes, hipMemcpyHostToDevice, v->stream));
        d_in = (const int32_t *)cursor;
        d_out = (int32_t *)(cursor + img_bytes);
    }
    hipLaunchKernelGGL(k_remap_instance_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, v->stream, d_in, n, d_map,
                       d_map + n_map, (int32_t)n_map, (const int32_t *)nullptr, d_out);
    HV_HIP(hipGetLastError());
    if (loc == HV_HOST) HV_HIP(hipMemcpyAsync(out, d_out, img_bytes, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream)); // `flat` is host memory
    return HV_OK;
}

// ---- association, in stages (all asynchronous on the volume's stream unless they hand data to the host) ------------------------
namespace {
struct AssocScratch {
    unsigned long long *vkeys, *ckeys;
    int32_t *vcounts, *ccounts, *map_inst, *map_obj, *n_map, *flags, *flags_vote;
    int2 *pending;
    int64_t pending_cap;
};

int assoc_scratch(hv_volume *v, AssocScratch *S) {
    const HvSemParams G = sem_params(v);
    const int64_t total = (int64_t)v->cfg.max_blocks * G.nvox; // (the pool's size, not its fill: no stage waits for the block count)
    S->pending_cap = std::max<int64_t>(1, std::min<int64_t>(total, 8 * (int64_t)v->cfg.max_points));
    // device scratch: [vote keys][vote counts][compact keys][compact counts][final map inst | obj][n_map, flags][pending]
    const size_t off_counts = sizeof(uint64_t) * HV_VOTE_CAP;
    const size_t off_ckeys = off_counts + sizeof(int32_t) * HV_VOTE_CAP;
    const size_t off_ccounts = off_ckeys + sizeof(uint64_t) * HV_VOTE_CAP;
    const size_t off_map = off_ccounts + sizeof(int32_t) * HV_VOTE_CAP;
    const size_t off_misc = off_map + sizeof(int32_t) * 2 * HV_VOTE_CAP;
    const size_t off_pending = off_misc + 256;
    const size_t scratch_bytes = off_pending + sizeof(int2) * (size_t)S->pending_cap;
    int rc = hv_ensure_buffer(v, &v->assoc_buf, &v->assoc_buf_bytes, scratch_bytes);
    if (rc != HV_OK) return rc;
    char *sb = (char *)v->assoc_buf;
    S->vkeys = (unsigned long long *)sb;
    S->vcounts = (int32_t *)(sb + off_counts);
    S->ckeys = (unsigned long long *)(sb + off_ckeys);
    S->ccounts = (int32_t *)(sb + off_ccounts);
    S->map_inst = (int32_t *)(sb + off_map);
    S->map_obj = S->map_inst + HV_VOTE_CAP;
    S->n_map = (int32_t *)(sb + off_misc);
    S->flags = S->n_map + 1;
    S->flags_vote = S->n_map + 2; // set by stages before the rules kernel (which writes `flags` afresh), folded in by k_sem_assoc_apply
    S->pending = (int2 *)(sb + off_pending);
    return HV_OK;
}
} // namespace

// Stage 1: the per-voxel votes and the image's instance ids -> compacted (instance << 32 | object, votes) pairs in device memory
// (how many: counter HV_CNT_OUT; voxels waiting for an object id: the pending list, counter HV_CNT_AUX).
int hv_assoc_vote(hv_volume *v, const float *intr_f32, int32_t width, int32_t height, const double *T_cw, float depth_max,
                  float depth_min, const int32_t *class_ids_image, const int32_t *instance_ids_image, const float *depth_image,
                  float depth_threshold, int32_t do_carving, int32_t loc) {
    HV_REQUIRE(v != nullptr && intr_f32 != nullptr && T_cw != nullptr, HV_ERR_INVALID, "hv_assoc_vote: null argument");
    HV_REQUIRE(hv_is_semantic(v), HV_ERR_MODE, "hv_assoc_vote: not a semantic grid");
    HV_REQUIRE(class_ids_image != nullptr && instance_ids_image != nullptr && width > 0 && height > 0, HV_ERR_INVALID,
               "hv_assoc_vote: empty label images");
    HV_HIP(hipSetDevice(v->device));
    bool checked_unused = false;
    int rc = hv_capacity_gate(v, &checked_unused); // consumes the published pool state (no synchronisation); refuses after an overflow
    if (rc != HV_OK) return rc;
    // how many blocks the pool holds is read by the kernel itself; the grid is sized by what the host knows without waiting
    const int64_t nb = std::min<int64_t>(v->cfg.max_blocks, std::max<int64_t>(v->known_blocks, 1024));
    const HvSemParams G = sem_params(v);
    const int64_t n_px = (int64_t)width * height;
    AssocScratch S;
    rc = assoc_scratch(v, &S);
    if (rc != HV_OK) return rc;
    // images: three planes staged back to back when they come from the host
    const int32_t *d_cls = class_ids_image, *d_inst = instance_ids_image;
    const float *d_depth = depth_image;
    if (loc == HV_HOST) {
        const size_t plane = (sizeof(int32_t) * (size_t)n_px + 255) & ~(size_t)255;
        rc = hv_ensure_buffer(v, &v->stage_b, &v->stage_b_bytes, 3 * plane);
        if (rc != HV_OK) return rc;
        char *st = (char *)v->stage_b;
        bool pinned_src = false; // (page-locked sources - the front's registered ring - are read by the DMA engine later: hv_h2d)
        if ((rc = hv_h2d_lazy(v, st, class_ids_image, sizeof(int32_t) * n_px, &pinned_src)) != HV_OK) return rc;
        if ((rc = hv_h2d_lazy(v, st + plane, instance_ids_image, sizeof(int32_t) * n_px, &pinned_src)) != HV_OK) return rc;
        d_cls = (const int32_t *)st;
        d_inst = (const int32_t *)(st + plane);
        if (depth_image != nullptr) {
            if ((rc = hv_h2d_lazy(v, st + 2 * plane, depth_image, sizeof(float) * n_px, &pinned_src)) != HV_OK) return rc;
            d_depth = (const float *)(st + 2 * plane);
        }
        if ((rc = hv_h2d_fence(v, pinned_src)) != HV_OK) return rc;
    }
    if (v->assoc_clean != v->assoc_buf || v->assoc_clean_bytes != v->assoc_buf_bytes) { // a new allocation, or a call that failed before its compaction
        HV_HIP(hipMemsetAsync(S.vkeys, 0xFF, sizeof(uint64_t) * HV_VOTE_CAP, v->stream));
        HV_HIP(hipMemsetAsync(S.vcounts, 0, sizeof(int32_t) * HV_VOTE_CAP, v->stream));
        HV_HIP(hipMemsetAsync(S.n_map, 0, 256, v->stream));
    }
    v->assoc_clean = nullptr;
    static_assert(HV_CNT_OUT2 == HV_CNT_OUT + 1 && HV_CNT_AUX == HV_CNT_OUT + 2, "one memset clears OUT, OUT2, AUX");
    HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT], 0, sizeof(int32_t) * 3, v->stream));
    HvAssocParams A;
    A.use_depth = depth_image != nullptr ? 1 : 0;
    A.do_carving = (do_carving && A.use_depth) ? 1 : 0;
    A.depth_threshold = depth_threshold;
    A.pending_cap = (int32_t)std::min<int64_t>(S.pending_cap, INT32_MAX);
    v->assoc_pending_cap = A.pending_cap;
    { // (always: nb >= 1, and the kernel scans the image as well)
        HvQuery Q;
        memset(&Q, 0, sizeof(Q));
        Q.kind = 2;
        Q.min_count = 1;
        fill_frustum_query(Q, v, intr_f32, width, height, T_cw, depth_max, depth_min);
        HvGridParams GP{};
        GP.inv_voxel_size = G.inv_voxel_size;
        GP.bs = G.bs;
        GP.nvox = G.nvox;
        GP.local_bits = G.local_bits;
        fill_key_range(Q, GP);
        // (four blocks per wave, one per 16-lane group, was measured in round 5 - profiles/r05/README.md: +2-3 % at 2 mm, -11 % at 1 cm - and dropped)
        const dim3 grid((unsigned)std::min<int64_t>((nb + 3) / 4, 4096)); // persistent: 16 workgroups per CU
        // the probabilistic payload's copy of the record takes 149 registers: three workgroups per CU, nothing spilled (measured
        // against four with 13 spilled registers: the vote 13 % slower, round 6); the voting payload fits four
        // (the 128-byte records - HvPay<VOX>::maps - take three, the 64-byte ones four)
        HV_SEM_DISPATCH(v, hipLaunchKernelGGL((k_sem_assoc_vote<VOX, HvPay<VOX>::maps ? 3 : 4>), grid, dim3(256), 0, v->stream, v->table, (VOX *)v->pool, (int64_t)-1, G, Q,
                                              d_cls, d_inst, d_depth, A, S.vkeys, S.vcounts, S.pending, v->occ, n_px));
    }
    hipLaunchKernelGGL(k_sem_assoc_compact, dim3(HV_VOTE_CAP / 256), dim3(256), 0, v->stream, v->table, S.vkeys, S.vcounts, S.ckeys,
                       S.ccounts);
    HV_HIP(hipGetLastError());
    v->assoc_clean = v->assoc_buf; // the compaction is queued: every slot it finds is empty again
    v->assoc_clean_bytes = v->assoc_buf_bytes;
    v->assoc_state = 1;
    return HV_OK;
}

// Multi-GPU: every GPU votes with the voxels it owns; the pair lists are exchanged (all-gather, a few hundred bytes), concatenated
// and set on every GPU, which then decides identically (the rules kernel adds the counts of equal pairs).
int hv_assoc_pairs_fetch(hv_volume *v, uint64_t *pair_keys, int32_t *pair_counts, int64_t cap, int64_t *n_pairs) {
    HV_REQUIRE(v != nullptr && n_pairs != nullptr, HV_ERR_INVALID, "hv_assoc_pairs_fetch: null argument");
    HV_REQUIRE(v->assoc_state >= 1, HV_ERR_INVALID, "hv_assoc_pairs_fetch: call hv_assoc_vote first");
    HV_HIP(hipSetDevice(v->device));
    int rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    HV_REQUIRE(v->h_counters[HV_CNT_OUT2] == 0, HV_ERR_CAPACITY, "hv_assoc_vote: more than %u distinct (instance, object) pairs", HV_VOTE_CAP);
    const int64_t n = v->h_counters[HV_CNT_OUT];
    *n_pairs = n;
    if (pair_keys == nullptr || pair_counts == nullptr || n == 0) return HV_OK;
    AssocScratch S;
    rc = assoc_scratch(v, &S);
    if (rc != HV_OK) return rc;
    const int64_t m = std::min(n, cap);
    HV_HIP(hipMemcpyAsync(pair_keys, S.ckeys, sizeof(uint64_t) * m, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipMemcpyAsync(pair_counts, S.ccounts, sizeof(int32_t) * m, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}

int hv_assoc_pairs_set(hv_volume *v, const uint64_t *pair_keys, const int32_t *pair_counts, int64_t n_pairs) {
    HV_REQUIRE(v != nullptr && (n_pairs == 0 || (pair_keys != nullptr && pair_counts != nullptr)), HV_ERR_INVALID, "hv_assoc_pairs_set: null argument");
    HV_REQUIRE(v->assoc_state >= 1, HV_ERR_INVALID, "hv_assoc_pairs_set: call hv_assoc_vote first");
    HV_REQUIRE(n_pairs >= 0 && n_pairs <= HV_RULES_MAX, HV_ERR_CAPACITY, "hv_assoc_pairs_set: more than %d pairs", HV_RULES_MAX);
    HV_HIP(hipSetDevice(v->device));
    AssocScratch S;
    int rc = assoc_scratch(v, &S);
    if (rc != HV_OK) return rc;
    const int32_t n32 = (int32_t)n_pairs;
    if (n_pairs > 0) {
        HV_HIP(hipMemcpyAsync(S.ckeys, pair_keys, sizeof(uint64_t) * n_pairs, hipMemcpyHostToDevice, v->stream));
        HV_HIP(hipMemcpyAsync(S.ccounts, pair_counts, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice, v->stream));
    }
    HV_HIP(hipMemcpyAsync(&v->table.counters[HV_CNT_OUT], &n32, sizeof(int32_t), hipMemcpyHostToDevice, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream)); // the arguments are host memory
    return HV_OK;
}

// The same exchange with the lists staying in device memory (round 5; the host forms above cost two synchronisations and four
// copies per keyframe, which the single-GPU flow no longer has): export this GPU's pairs as a fixed-size message into a device
// buffer of the caller (int64 [1 + 2 cap]), all-gather the messages (RCCL, on the device), import all of them.  Asynchronous on the
// volume's stream.  Equal pairs of different GPUs are added up on the way in (the vote table de-duplicates), so the merged list is
// no longer than a single GPU's - an 8-GPU keyframe cannot overflow the decide stage by repetition.
int hv_assoc_pairs_export(hv_volume *v, int64_t *d_msg, int64_t cap) {
    HV_REQUIRE(v != nullptr && d_msg != nullptr && cap > 0 && cap <= HV_RULES_MAX, HV_ERR_INVALID, "hv_assoc_pairs_export: bad argument (cap <= %d)", HV_RULES_MAX);
    HV_REQUIRE(v->assoc_state >= 1, HV_ERR_INVALID, "hv_assoc_pairs_export: call hv_assoc_vote first");
    HV_HIP(hipSetDevice(v->device));
    AssocScratch S;
    int rc = assoc_scratch(v, &S);
    if (rc != HV_OK) return rc;
    hipLaunchKernelGGL(k_assoc_pairs_export, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, v->stream, v->table, S.ckeys, S.ccounts,
                       (long long *)d_msg, (int)cap, S.flags_vote);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

int hv_assoc_pairs_import(hv_volume *v, const int64_t *d_msgs, int32_t world, int64_t cap) {
    HV_REQUIRE(v != nullptr && d_msgs != nullptr && world > 0 && cap > 0 && cap <= HV_RULES_MAX, HV_ERR_INVALID, "hv_assoc_pairs_import: bad argument");
    HV_REQUIRE(v->assoc_state >= 1, HV_ERR_INVALID, "hv_assoc_pairs_import: call hv_assoc_vote first");
    HV_HIP(hipSetDevice(v->device));
    AssocScratch S;
    int rc = assoc_scratch(v, &S);
    if (rc != HV_OK) return rc;
    // (the vote table is empty: the vote stage's compaction clears what it reads)
    HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT], 0, sizeof(int32_t), v->stream));
    hipLaunchKernelGGL(k_assoc_pairs_import, dim3((unsigned)((cap + 255) / 256), (unsigned)world), dim3(256), 0, v->stream, v->table,
                       (const long long *)d_msgs, (int)world, (int)cap, S.vkeys, S.vcounts);
    hipLaunchKernelGGL(k_sem_assoc_compact, dim3(HV_VOTE_CAP / 256), dim3(256), 0, v->stream, v->table, S.vkeys, S.vcounts, S.ckeys, S.ccounts);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

// Stage 2: the reference's winner / min_votes / min_vote_ratio rules on the pairs, new object ids, the deferred set_object_id of the
// voxels that waited - all on the device.  The map stays in device memory (hv_remap_instance_ids_last reads it there).
int hv_assoc_decide(hv_volume *v, float min_vote_ratio, int32_t min_votes) {
    HV_REQUIRE(v != nullptr && hv_is_semantic(v), HV_ERR_MODE, "hv_assoc_decide: not a semantic grid");
    HV_REQUIRE(v->assoc_state >= 1, HV_ERR_INVALID, "hv_assoc_decide: call hv_assoc_vote first");
    HV_HIP(hipSetDevice(v->device));
    AssocScratch S;
    int rc = assoc_scratch(v, &S);
    if (rc != HV_OK) return rc;
    int32_t *d_next = nullptr;
    rc = hv_dev_next_id(v->device, &d_next);
    if (rc != HV_OK) return rc;
    hipLaunchKernelGGL(k_sem_assoc_rules, dim3(1), dim3(1024), 0, v->stream, v->table, S.ckeys, S.ccounts, min_vote_ratio, min_votes, d_next,
                       S.map_inst, S.map_obj, S.n_map, S.flags);
    const dim3 grid(256);
    HV_SEM_DISPATCH(v, hipLaunchKernelGGL(k_sem_assoc_apply<VOX>, grid, dim3(256), 0, v->stream, v->table, (VOX *)v->pool, S.pending, v->assoc_pending_cap, S.map_inst,
                                          S.map_obj, S.n_map, S.flags, v->d_status));
    HV_HIP(hipGetLastError());
    v->assoc_state = 2;
    return HV_OK;
}

// The map of the last association, sorted by instance id (this is where an overflow of the vote table / pending list / rules
// workgroup is reported).
int hv_assoc_map_fetch(hv_volume *v, int32_t *map_inst, int32_t *map_obj, int64_t cap, int64_t *n_map) {
    HV_REQUIRE(v != nullptr && n_map != nullptr, HV_ERR_INVALID, "hv_assoc_map_fetch: null argument");
    HV_REQUIRE(v->assoc_state == 2, HV_ERR_INVALID, "hv_assoc_map_fetch: call hv_assoc_decide first");
    HV_HIP(hipSetDevice(v->device));
    AssocScratch S;
    int rc = assoc_scratch(v, &S);
    if (rc != HV_OK) return rc;
    int32_t misc[2] = {0, 0};
    HV_HIP(hipMemcpyAsync(misc, S.n_map, sizeof(misc), hipMemcpyDeviceToHost, v->stream));
    rc = hv_read_counters(v); // synchronises the stream
    if (rc != HV_OK) return rc;
    // misc[1] = the flags of the association whose map this is (k_sem_assoc_rules + k_sem_assoc_apply; the shared counters may belong
    // to a later fold by now).  Reported here, so the latch of the status word is cleared: the caller has been told.
    if (misc[1] != 0) v->h_status->assoc_flags = 0;
    HV_REQUIRE((misc[1] & HV_ASSOC_VOTE_TABLE_FULL) == 0, HV_ERR_CAPACITY,
               "hv_assign_object_ids_to_instance_ids: more than %u distinct (instance, object) pairs", HV_VOTE_CAP);
    HV_REQUIRE((misc[1] & HV_ASSOC_PENDING_FULL) == 0, HV_ERR_CAPACITY,
               "hv_assign_object_ids_to_instance_ids: pending list overflow (capacity %d)", v->assoc_pending_cap);
    HV_REQUIRE((misc[1] & HV_ASSOC_TOO_MANY_PAIRS) == 0, HV_ERR_CAPACITY, "hv_assign_object_ids_to_instance_ids: more than %d (instance, object) pairs in one keyframe",
               HV_RULES_MAX);
    *n_map = misc[0];
    const int64_t m = std::min<int64_t>(misc[0], cap);
    if (m > 0 && map_inst != nullptr && map_obj != nullptr) {
        HV_HIP(hipMemcpyAsync(map_inst, S.map_inst, sizeof(int32_t) * m, hipMemcpyDeviceToHost, v->stream));
        HV_HIP(hipMemcpyAsync(map_obj, S.map_obj, sizeof(int32_t) * m, hipMemcpyDeviceToHost, v->stream));
        HV_HIP(hipStreamSynchronize(v->stream));
    }
    return HV_OK;
}

// remap_instance_ids with the map of the volume's last association, straight from device memory (no dict, no upload)
int hv_remap_instance_ids_last(hv_volume *v, const int32_t *instance_ids, int32_t height, int32_t width, int32_t *out, int32_t loc) {
    HV_REQUIRE(v != nullptr && instance_ids != nullptr && out != nullptr, HV_ERR_INVALID, "hv_remap_instance_ids_last: null argument");
    HV_REQUIRE(v->assoc_state == 2, HV_ERR_INVALID, "hv_remap_instance_ids_last: call hv_assoc_decide first");
    HV_HIP(hipSetDevice(v->device));
    const int64_t n = (int64_t)height * width;
    if (n == 0) return HV_OK;
    AssocScratch S;
    int rc = assoc_scratch(v, &S);
    if (rc != HV_OK) return rc;
    const int32_t *d_in = instance_ids;
    int32_t *d_out = out;
    const size_t img_bytes = sizeof(int32_t) * (size_t)n;
    if (loc == HV_HOST) {
        rc = hv_ensure_buffer(v, &v->stage_b, &v->stage_b_bytes, 2 * img_bytes + 512);
        if (rc != HV_OK) return rc;
        HV_HIP(hipMemcpyAsync(v->stage_b, instance_ids, img_bytes, hipMemcpyHostToDevice, v->stream));
        d_in = (const int32_t *)v->stage_b;
        d_out = (int32_t *)((char *)v->stage_b + img_bytes);
    }
    hipLaunchKernelGGL(k_remap_instance_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, v->stream, d_in, n, S.map_inst, S.map_obj, 0,
                       (const int32_t *)S.n_map, d_out);
    HV_HIP(hipGetLastError());
    if (loc == HV_HOST) {
        HV_HIP(hipMemcpyAsync(out, d_out, img_bytes, hipMemcpyDeviceToHost, v->stream));
        HV_HIP(hipStreamSynchronize(v->stream));
    }
    return HV_OK;
}

// One semantic keyframe in ONE host call (round 6): the stages above in the integrator's order, on device-resident images.  At
// 640x480 / 1 cm the device needs ~150 us for a keyframe and the Python front spent ~200 us issuing its five calls (tensor wrappers,
// stream bookkeeping, argument marshalling) - the flow was bound by the host.
int hv_semantic_fuse_keyframe(hv_volume *v, const float *depth, const uint8_t *rgb, const int32_t *class_ids_image,
                              const int32_t *instance_ids_image, int32_t height, int32_t width, const float *frustum_intr_f32,
                              float frustum_depth_max, float frustum_depth_min, const double *intr, const double *T_cw,
                              int32_t filter_shadow_points, int32_t use_instance_ids, float assoc_depth_threshold, int32_t do_carving,
                              float min_vote_ratio, int32_t min_votes, double min_depth, double max_depth, int32_t use_depths) {
    HV_REQUIRE(v != nullptr && depth != nullptr && rgb != nullptr && intr != nullptr && T_cw != nullptr && frustum_intr_f32 != nullptr,
               HV_ERR_INVALID, "hv_semantic_fuse_keyframe: null argument");
    HV_REQUIRE(hv_is_semantic(v), HV_ERR_MODE, "hv_semantic_fuse_keyframe: not a semantic volume");
    HV_REQUIRE(height > 0 && width > 0, HV_ERR_INVALID, "hv_semantic_fuse_keyframe: bad image size");
    HV_HIP(hipSetDevice(v->device));
    const size_t npx = (size_t)height * width;
    int rc = hv_ensure_buffer(v, &v->kf_buf, &v->kf_buf_bytes, npx * (sizeof(float) + sizeof(int32_t)));
    if (rc != HV_OK) return rc;
    float *d_filtered = (float *)v->kf_buf;
    int32_t *d_obj = (int32_t *)(d_filtered + npx);
    const float *d_depth = depth;
    if (filter_shadow_points) { // kVolumetricIntegrationVoxelGridShadowPointsFilter, volumetric_integrator_voxel_semantic_grid.py:334
        rc = hv_filter_shadow_points(v, depth, height, width, 2, 2, -1.0f, d_filtered, HV_DEVICE);
        if (rc != HV_OK) return rc;
        d_depth = d_filtered;
    }
    const int32_t *d_obj_in = nullptr;
    if (use_instance_ids && instance_ids_image != nullptr) { // :340-372 (no class image: empty map, every id -> -1)
        if (class_ids_image != nullptr) {
            rc = hv_assoc_vote(v, frustum_intr_f32, width, height, T_cw, frustum_depth_max, frustum_depth_min, class_ids_image, instance_ids_image,
                               d_depth, assoc_depth_threshold, do_carving, HV_DEVICE);
            if (rc != HV_OK) return rc;
            rc = hv_assoc_decide(v, min_vote_ratio, min_votes);
            if (rc != HV_OK) return rc;
            rc = hv_remap_instance_ids_last(v, instance_ids_image, height, width, d_obj, HV_DEVICE);
            if (rc != HV_OK) return rc;
        } else {
            HV_HIP(hipMemsetAsync(d_obj, 0xFF, sizeof(int32_t) * npx, v->stream));
        }
        d_obj_in = d_obj;
    } else if (do_carving) { // :373-380
        rc = hv_carve(v, frustum_intr_f32, width, height, T_cw, frustum_depth_max, frustum_depth_min, d_depth, assoc_depth_threshold, HV_DEVICE);
        if (rc != HV_OK) return rc;
    }
    return hv_integrate_rgbd_semantic(v, d_depth, rgb, class_ids_image, d_obj_in, height, width, intr, T_cw, min_depth, max_depth, use_depths, HV_DEVICE);
}

// The reference's one call = vote + decide + fetch.
int hv_assign_object_ids_to_instance_ids(hv_volume *v, const float *intr_f32, int32_t width, int32_t height, const double *T_cw,
                                         float depth_max, float depth_min, const int32_t *class_ids_image,
                                         const int32_t *instance_ids_image, const float *depth_image, float depth_threshold,
                                         int32_t do_carving, float min_vote_ratio, int32_t min_votes, int32_t *map_inst,
                                         int32_t *map_obj, int64_t cap, int64_t *n_map, int32_t loc) {
    HV_REQUIRE(v != nullptr && intr_f32 != nullptr && T_cw != nullptr && n_map != nullptr, HV_ERR_INVALID,
               "hv_assign_object_ids_to_instance_ids: null argument");
    HV_REQUIRE(hv_is_semantic(v), HV_ERR_MODE, "hv_assign_object_ids_to_instance_ids: not a semantic grid");
    *n_map = 0;
    // "class IDs or semantic instances image is empty": the reference returns an empty map
    if (class_ids_image == nullptr || instance_ids_image == nullptr || width <= 0 || height <= 0) return HV_OK;
    int rc = hv_assoc_vote(v, intr_f32, width, height, T_cw, depth_max, depth_min, class_ids_image, instance_ids_image, depth_image,
                           depth_threshold, do_carving, loc);
    if (rc != HV_OK) return rc;
    rc = hv_assoc_decide(v, min_vote_ratio, min_votes);
    if (rc != HV_OK) return rc;
    return hv_assoc_map_fetch(v, map_inst, map_obj, cap, n_map);
}

int hv_object_segments_compute(hv_volume *v, int32_t min_count, float min_confidence, int64_t *n_rows, int64_t *n_objects) {
    HV_REQUIRE(v != nullptr && n_rows != nullptr && n_objects != nullptr, HV_ERR_INVALID, "hv_object_segments_compute: null argument");
    HV_REQUIRE(hv_is_semantic(v), HV_ERR_MODE, "hv_object_segments_compute: not a semantic grid");
    HV_HIP(hipSetDevice(v->device));
    if (v->segments_cache == nullptr) v->segments_cache = new HvSegmentsCache();
    HvSegmentsCache &C = *static_cast<HvSegmentsCache *>(v->segments_cache);
    C = HvSegmentsCache();
    *n_rows = 0;
    *n_objects = 0;
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    if (nb == 0) return HV_OK;
    const int nvox = sem_params(v).nvox;
    const dim3 grid((unsigned)std::min<int64_t>((nb + 3) / 4, 8192));
    // pass 1: count; pass 2: emit (object id, voxel index) keys
    int64_t m = 0;
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long *d_keys = nullptr;
        if (pass == 1) {
            rc = hv_ensure_buffer(v, &v->out_b, &v->out_b_bytes, sizeof(uint64_t) * 2 * (size_t)m + 256);
            if (rc != HV_OK) return rc;
            d_keys = (unsigned long long *)v->out_b;
        }
        HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT], 0, sizeof(int32_t), v->stream));
        // (min_count < 0 would admit voxels that never took a point: no occupancy bits then)
        HV_SEM_DISPATCH(v, hipLaunchKernelGGL(k_seg_collect<VOX>, grid, dim3(256), 0, v->stream, v->table, (const VOX *)v->pool, nb, nvox, min_count, min_confidence, d_keys, m,
                                              min_count >= 0 ? v->occ : nullptr));
        HV_HIP(hipGetLastError());
        rc = hv_read_counters(v);
        if (rc != HV_OK) return rc;
        m = v->h_counters[HV_CNT_OUT];
        if (m == 0) return HV_OK;
    }
    unsigned long long *d_keys = (unsigned long long *)v->out_b, *d_sorted = d_keys + m;
    size_t tmp_bytes = 0;
    HV_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys, d_sorted, (size_t)m, 0, 64, v->stream));
    rc = hv_ensure_buffer(v, &v->sort_tmp, &v->sort_tmp_bytes, tmp_bytes);
    if (rc != HV_OK) return rc;
    tmp_bytes = v->sort_tmp_bytes;
    HV_HIP(rocprim::radix_sort_keys(v->sort_tmp, tmp_bytes, d_keys, d_sorted, (size_t)m, 0, 64, v->stream));
    rc = hv_ensure_buffer(v, &v->out_a, &v->out_a_bytes, (size_t)m * (24 + 12 + 4 + 4 + 4) + 1024);
    if (rc != HV_OK) return rc;
    double *d_pts = (double *)v->out_a;
    float *d_cols = (float *)(d_pts + 3 * m);
    int32_t *d_obj = (int32_t *)(d_cols + 3 * m);
    int32_t *d_cls = d_obj + m;
    float *d_conf = (float *)(d_cls + m);
    const dim3 rgrid((unsigned)((m + 255) / 256));
    HV_SEM_DISPATCH(v, hipLaunchKernelGGL(k_seg_rows<VOX>, rgrid, dim3(256), 0, v->stream, (const VOX *)v->pool, d_sorted, m, d_pts, d_cols, d_obj, d_cls, d_conf,
                                          (const void *)v->table.prob_nodes));
    HV_HIP(hipGetLastError());
    C.pts.resize((size_t)m * 3);
    C.cols.resize((size_t)m * 3);
    C.row_obj.resize((size_t)m);
    std::vector<int32_t> cls((size_t)m);
    std::vector<float> conf((size_t)m);
    HV_HIP(hipMemcpyAsync(C.pts.data(), d_pts, 24 * m, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipMemcpyAsync(C.cols.data(), d_cols, 12 * m, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipMemcpyAsync(C.row_obj.data(), d_obj, 4 * m, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipMemcpyAsync(cls.data(), d_cls, 4 * m, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipMemcpyAsync(conf.data(), d_conf, 4 * m, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    // rows are grouped by object id (ascending): per-object summary + PCA box on the host
    for (int64_t i = 0; i < m;) {
        int64_t j = i;
        float cmin = conf[i], cmax = conf[i];
        while (j < m && C.row_obj[j] == C.row_obj[i]) {
            cmin = std::min(cmin, conf[j]);
            cmax = std::max(cmax, conf[j]);
            ++j;
        }
        C.ids.push_back(C.row_obj[i]);
        C.ids.push_back(cls[i]); // class of the object's first voxel (the reference: first in *its* iteration order)
        C.ids.push_back((int32_t)(j - i));
        C.conf.push_back(cmin);
        C.conf.push_back(cmax);
        C.obb.resize(C.obb.size() + 10);
        compute_obb_pca(C.pts.data() + i * 3, j - i, C.obb.data() + C.obb.size() - 10);
        i = j;
    }
    *n_rows = m;
    *n_objects = (int64_t)C.ids.size() / 3;
    return HV_OK;
}

int hv_object_segments_fetch(hv_volume *v, double *points, float *colors, int32_t *row_object_ids, int32_t *object_ids,
                             float *confidences, double *obbs) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_object_segments_fetch: null volume");
    HV_REQUIRE(v->segments_cache != nullptr, HV_ERR_INVALID, "hv_object_segments_fetch: call hv_object_segments_compute first");
    const HvSegmentsCache &C = *static_cast<HvSegmentsCache *>(v->segments_cache);
    if (points) memcpy(points, C.pts.data(), sizeof(double) * C.pts.size());
    if (colors) memcpy(colors, C.cols.data(), sizeof(float) * C.cols.size());
    if (row_object_ids) memcpy(row_object_ids, C.row_obj.data(), sizeof(int32_t) * C.row_obj.size());
    if (object_ids) memcpy(object_ids, C.ids.data(), sizeof(int32_t) * C.ids.size());
    if (confidences) memcpy(confidences, C.conf.data(), sizeof(float) * C.conf.size());
    if (obbs) memcpy(obbs, C.obb.data(), sizeof(double) * C.obb.size());
    return HV_OK;
}

// OrientedBoundingBox3D::compute_from_points(points, PCA) for callers that hold their own point sets
int hv_compute_obb_pca(const double *points, int64_t n, double *obb) {
    HV_REQUIRE(obb != nullptr && (n == 0 || points != nullptr), HV_ERR_INVALID, "hv_compute_obb_pca: null argument");
    compute_obb_pca(points, n, obb);
    return HV_OK;
}

} // extern "C"
