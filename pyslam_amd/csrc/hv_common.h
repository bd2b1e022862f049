// Shared host/device definitions of libpyslam_hipvol.so (gfx950 / CDNA4 only).
//
// One sparse block hash serves both fusion modes:
//   VOXEL_GRID  block = 8^3 voxels  of {count, position_sum[3], color_sum[3]}  (cpp/volumetric semantics)
//   TSDF        unit  = 16^3 voxels of {tsdf, weight, rgb sums}                (Open3D semantics)
// Keys are the reference's BlockKey / Open3D's unit index: three int32 packed 21 bits per axis
// into one 64-bit word so that a slot can be claimed with a single 64-bit CAS.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "hipvol.h"

#define HV_WAVE 64

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
void hv_set_error(const char *fmt, ...);

#define HV_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            hv_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,          \
                         __LINE__);                                                                \
            return HV_ERR_DEVICE;                                                                  \
        }                                                                                          \
    } while (0)

#define HV_REQUIRE(cond, code, ...)                                                                \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            hv_set_error(__VA_ARGS__);                                                             \
            return (code);                                                                         \
        }                                                                                          \
    } while (0)

// ------------------------------------------------------------------------------------------------
// packed block keys
// ------------------------------------------------------------------------------------------------
static constexpr int HV_KEY_BITS = 21;
static constexpr int32_t HV_KEY_BIAS = 1 << (HV_KEY_BITS - 1);          // 2^20
static constexpr uint64_t HV_KEY_MASK = (1ull << HV_KEY_BITS) - 1;
static constexpr uint64_t HV_EMPTY_KEY = ~0ull;

__host__ __device__ inline bool hv_key_in_range(int32_t x, int32_t y, int32_t z) {
    return x >= -HV_KEY_BIAS && x < HV_KEY_BIAS && y >= -HV_KEY_BIAS && y < HV_KEY_BIAS &&
           z >= -HV_KEY_BIAS && z < HV_KEY_BIAS;
}
__host__ __device__ inline uint64_t hv_pack_key(int32_t x, int32_t y, int32_t z) {
    return (uint64_t)(uint32_t)(x + HV_KEY_BIAS) | ((uint64_t)(uint32_t)(y + HV_KEY_BIAS) << HV_KEY_BITS) |
           ((uint64_t)(uint32_t)(z + HV_KEY_BIAS) << (2 * HV_KEY_BITS));
}
__host__ __device__ inline void hv_unpack_key(uint64_t k, int32_t &x, int32_t &y, int32_t &z) {
    x = (int32_t)(k & HV_KEY_MASK) - HV_KEY_BIAS;
    y = (int32_t)((k >> HV_KEY_BITS) & HV_KEY_MASK) - HV_KEY_BIAS;
    z = (int32_t)((k >> (2 * HV_KEY_BITS)) & HV_KEY_MASK) - HV_KEY_BIAS;
}
// slot hash (internal; the reference's BlockKeyHash is only reproduced for export)
__host__ __device__ inline uint32_t hv_slot_hash(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k;
}
// Owner rank of a unit for multi-GPU unit-ownership sharding (decorrelated from the slot hash).
__host__ __device__ inline int32_t hv_owner_of(uint64_t k, int32_t world) {
    k ^= k >> 29;
    k *= 0x9E3779B97F4A7C15ull;
    k ^= k >> 32;
    return (int32_t)((k >> 7) % (uint64_t)world);
}
// BlockKeyHash / VoxelKeyHash of the reference, voxel_hashing.h:51-58,106-113 (libstdc++ identity
// std::hash<int32_t>, sign-extended to size_t).
__host__ __device__ inline uint64_t hv_reference_hash(int32_t x, int32_t y, int32_t z) {
    return (uint64_t)(int64_t)x ^ ((uint64_t)(int64_t)y << 1) ^ ((uint64_t)(int64_t)z << 2);
}

// Device view of the block hash.  keys[slot] = packed key or HV_EMPTY_KEY; vals[slot] = pool index
// (written by the claiming thread; readers in *later* kernels only).  block_keys[pool index] = key.
struct HvTable {
    unsigned long long *keys;
    int32_t *vals;
    unsigned long long *block_keys;
    int32_t *counters; // [HV_CNT_*]
    uint32_t mask;     // capacity - 1
    int32_t max_blocks;
    void *prob_nodes;       // probabilistic semantic payload: overflow nodes of the per-voxel label maps (HvProbNode[prob_node_cap])
    int32_t prob_node_cap;  // (0 / nullptr in every other mode)
};

// Device view of a call's point bins (grid modes; hv_bins.h).
struct HvBins {
    int32_t *cnt;                // [table_capacity] points of this call per slot; zero between calls (the fold clears what it reads)
    uint32_t *inl;               // [table_capacity][HV_BIN_K0]
    int32_t *touched;            // [HV_BIN_LISTS][touched_cap] slots that took points in this call
    int32_t *len;                // [2 parities][HV_BIN_LISTS * HV_BIN_LEN_STRIDE] list lengths
    unsigned long long *pg_keys; // [pg_mask + 1]  epoch << 40 | slot << 16 | page number
    uint32_t *pg_data;           // [pg_mask + 1][HV_BIN_PG]
    uint32_t pg_mask;
    uint32_t epoch;              // this call's (1 .. HV_BIN_EPOCHS)
    int32_t touched_cap;
    int32_t idx_bits;            // entry = local voxel index << idx_bits | point index
    int32_t parity;              // which set of list lengths this call appends to (the fold zeroes the other one)
};

enum {
    HV_CNT_BLOCKS = 0,   // allocated blocks
    HV_CNT_OVERFLOW = 1, // pool/table overflow events
    HV_CNT_DROPPED = 2,  // points with out-of-range keys
    HV_CNT_OUT = 5,      // output row counter (compaction kernels)
    HV_CNT_OUT2 = 6,     // second output counter (triangles)
    HV_CNT_AUX = 7,      // scratch counter (association pending list); OUT, OUT2, AUX are cleared by one 12-byte memset
    HV_CNT_LABEL_OVERFLOW = 8, // probabilistic payload: label observations dropped (the overflow-node pool is exhausted / > 254 labels)
    HV_CNT_PROB_NODES = 9,     // probabilistic payload: overflow nodes handed out
    // TSDF counters that are hammered by atomics while other workgroups of the same launch READ their neighbours live on 128-byte
    // lines of their own
    HV_CNT_TOUCH0 = 32,  // touched-unit list length, scratch set 0 (online path: frame parity 0)
    HV_CNT_TOUCH1 = 64,  // ... set 1 (frame parity 1)
    HV_CNT_COUNT = 96
};
#define HV_CNT_TOUCH(set) (HV_CNT_TOUCH0 + 32 * (set))
static constexpr int HV_TSDF_SETS = 2; // scratch sets of the multi-frame paths (frame records, union list, frame masks, list counter)
static constexpr size_t HV_CNT_TOUCH_SPAN_BYTES = sizeof(int32_t) * (32 * (HV_TSDF_SETS - 1) + 1); // one memset clears every list length (nothing lives between them)

#ifdef __HIPCC__
// Lookup only.  Returns slot or -1.
__device__ inline int32_t hv_table_find(const HvTable &t, uint64_t key) {
    uint32_t s = hv_slot_hash(key) & t.mask;
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        const unsigned long long k = t.keys[s];
        if (k == key) return (int32_t)s;
        if (k == HV_EMPTY_KEY) return -1;
        s = (s + 1) & t.mask;
    }
    return -1;
}

// Find-or-claim.  Returns the slot (>= 0) or -1 on overflow.  A newly claimed slot gets a pool
// index from the block counter; pool memory is pre-zeroed, so no per-block initialisation runs.
__device__ inline int32_t hv_table_insert(const HvTable &t, uint64_t key) {
    uint32_t s = hv_slot_hash(key) & t.mask;
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        unsigned long long k = t.keys[s];
        if (k == key) return (int32_t)s;
        if (k == HV_EMPTY_KEY) {
            k = atomicCAS(&t.keys[s], HV_EMPTY_KEY, (unsigned long long)key);
            if (k == HV_EMPTY_KEY) {
                const int32_t idx = atomicAdd(&t.counters[HV_CNT_BLOCKS], 1);
                if (idx >= t.max_blocks) {
                    atomicAdd(&t.counters[HV_CNT_OVERFLOW], 1);
                    // leave vals[s] = -1: consumers skip it
                    return -1;
                }
                t.vals[s] = idx;
                t.block_keys[idx] = key;
                return (int32_t)s;
            }
            if (k == key) return (int32_t)s;
        }
        s = (s + 1) & t.mask;
    }
    atomicAdd(&t.counters[HV_CNT_OVERFLOW], 1);
    return -1;
}

// Find-or-claim WITHOUT a pool index: *is_new = this call put the key into the table; the caller gives it its block
// (hv_table_assign) - the bin pass does that for a whole workgroup's new keys with ONE atomic on the block counter (13 500 new blocks of
// a 2 mm keyframe, one returning atomic each on that one word, were most of the bin pass: ~10 ns apiece, serialised).
__device__ inline int32_t hv_table_claim(const HvTable &t, uint64_t key, bool *is_new) {
    *is_new = false;
    uint32_t s = hv_slot_hash(key) & t.mask;
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        unsigned long long k = t.keys[s];
        if (k == key) return (int32_t)s;
        if (k == HV_EMPTY_KEY) {
            k = atomicCAS(&t.keys[s], HV_EMPTY_KEY, (unsigned long long)key);
            if (k == HV_EMPTY_KEY) {
                *is_new = true;
                return (int32_t)s;
            }
            if (k == key) return (int32_t)s;
        }
        s = (s + 1) & t.mask;
    }
    atomicAdd(&t.counters[HV_CNT_OVERFLOW], 1);
    return -1;
}
// ... and the second half of hv_table_insert for a slot claimed that way: pool index idx (from the block counter)
__device__ inline void hv_table_assign(const HvTable &t, int32_t slot, uint64_t key, int32_t idx) {
    if (idx >= t.max_blocks) {
        atomicAdd(&t.counters[HV_CNT_OVERFLOW], 1); // vals[slot] stays -1: consumers skip it
        return;
    }
    t.vals[slot] = idx;
    t.block_keys[idx] = key;
}

__device__ inline int hv_lane_id() { return (int)(threadIdx.x & (HV_WAVE - 1)); }
// A wave that owns an LDS window: its lanes synchronise with wave barriers only.
__device__ __forceinline__ void hv_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Wave-aggregated append: every lane with `pred` gets a distinct index from *counter; one atomic
// per wave (ballot + popcount prefix).
__device__ inline int32_t hv_wave_append(int32_t *counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m == 0) return -1;
    const int lane = hv_lane_id();
    const int leader = __ffsll((long long)m) - 1;
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (int32_t)__popcll(m));
    base = __shfl(base, leader);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    return pred ? base + (int32_t)__popcll(m & lt) : -1;
}
#endif // __HIPCC__

// ------------------------------------------------------------------------------------------------
// TSDF unit layout in HBM.  R = 16: five planes of R^3 4-byte words, plane-major:
//   [tsdf f32][weight u32][sum_r u32][sum_g u32][sum_b u32]   = 5 * 16 KiB = 80 KiB / unit
// word index inside a plane: z*R*R + x*R + y  (y fastest) so that one wave (lane -> (x, 4 y's))
// reads/writes a z-slab as one contiguous 1 KiB dwordx4 burst per plane.
// Weight is an exact observation count; colour is kept as exact integer sums of the u8 samples
// (mean = sum / weight reproduces Open3D's running mean to double rounding).
// ------------------------------------------------------------------------------------------------
static constexpr int HV_TSDF_PLANES = 5;

// VOXEL_GRID voxel record: the reference's 28-byte VoxelData padded to 32 B so that one voxel is
// two aligned 16-byte accesses and never straddles a 64-B line.
struct __attribute__((aligned(16))) HvVoxel {
    int32_t count;
    float pos[3];
    float col[3];
    int32_t pad;
};
static_assert(sizeof(HvVoxel) == 32, "HvVoxel must be 32 bytes");

struct HvFrameParams { // per-frame constants of the TSDF kernels (passed by value)
    // the multi-frame sweep's per-frame constants, laid out the way its float2 arithmetic consumes them (one 16-dword scalar
    // load; pairs land in aligned scalar register pairs): {e0,e4, e1,e5, e2,e6, e3,e7, e8,e11, e9,e10, inc0,inc1, inc2,0}
    float sweep_k[16];
    float ext[12];          // T_cw.cast<float>() rows 0..2
    float ext_scaled_col2[3];
    float fx, fy, cx, cy;
    float ffl_inv_x, ffl_inv_y; // 1.0f / (float)fx ...
    float voxel_length_f, half_voxel_length_f;
    float sdf_trunc_f, sdf_trunc_inv_f;
    float safe_width_f, safe_height_f;
    double unit_length;
    double pose[12];        // inverse(T_cw) rows 0..2 (f64)
    double fx_d, fy_d, cx_d, cy_d;
    double sdf_trunc_d;
    float depth_scale_f;
    double depth_trunc_d;
    int32_t H, W, stride;
    int32_t depth_is_u16;
    int32_t frame_id;       // touched stamp value (> 0)
    int32_t tile_u0, tile_v0, tile_u1, tile_v1; // image-space tile owned by this GPU: [u0,u1) x [v0,v1)
    int32_t tiled;                              // 0: the tile is the whole image (the per-voxel tile test is skipped)
    int32_t owner_rank, owner_world;            // unit ownership sharding: this GPU fuses units with owner(key) == rank
    int32_t touch_box_bits;                     // touch pass: largest unit box enumerated through the LDS bitmap (0: never)
    int32_t bgr;                                // colour frames are B, G, R (pySLAM's keyframe.img as OpenCV hands it): swapped while packing
};

// ------------------------------------------------------------------------------------------------
// Pool occupancy as the last finished integrate call left it, in pinned host memory the device writes directly (no
// read-back, no synchronisation): the last kernel of every call that can allocate blocks publishes {blocks, overflow, seq}.
// hv_capacity_gate() reads it before the next call launches anything: grow in time, or fail BEFORE fusing on.
// ------------------------------------------------------------------------------------------------
struct HvStatus {
    int32_t blocks;
    int32_t overflow;
    int32_t seq;
    int32_t pad;         // VOXEL_GRID bucket path: the largest bucket of the last frame
    int32_t assoc_flags; // capacity flags of the semantic association, OR-ed in by k_sem_assoc_apply, reported and cleared by the host
    int32_t reserved[3];
};

#ifdef __HIPCC__
__device__ inline void hv_publish_status(const HvTable &t, HvStatus *status, int32_t seq) {
    volatile HvStatus *s = status;
    s->blocks = t.counters[HV_CNT_BLOCKS];
    s->overflow = t.counters[HV_CNT_OVERFLOW];
    __threadfence_system();
    s->seq = seq;
}
#endif

// ------------------------------------------------------------------------------------------------
// host-side volume object
// ------------------------------------------------------------------------------------------------
struct HvEventPair {
    hipEvent_t start, stop;
};

// the four semantic block grids (include/hipvol.h: hv_mode); the two probabilistic ones keep per-voxel label maps with overflow nodes
static inline bool hv_mode_is_semantic(int32_t mode) {
    return mode == HV_MODE_VOXEL_SEMANTIC_GRID || mode == HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID || mode == HV_MODE_VOXEL_SEMANTIC_GRID2 ||
           mode == HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID2;
}
static inline bool hv_mode_has_label_maps(int32_t mode) {
    return mode == HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID || mode == HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID2;
}

struct hv_volume {
    hv_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;

    // hash + pool
    HvTable table{};
    uint64_t table_capacity = 0;
    void *pool = nullptr; // max_blocks * bytes_per_block, zero-initialised
    int64_t bytes_per_block = 0;
    int32_t *h_counters = nullptr; // pinned mirror [HV_CNT_COUNT]
    HvStatus *h_status = nullptr;  // pinned + device-visible: written by the last kernel of each allocating call
    HvStatus *d_status = nullptr;  // the same memory as the device addresses it
    int32_t status_seq_issued = 0, status_seq_seen = 0;
    int64_t known_blocks = 0;      // blocks in use as of status_seq_seen
    int64_t max_new_per_call = 0;  // largest growth of `blocks` seen between two published states
    double avg_new_per_call = 0.0; // running mean of the growth per call (what the calls still in flight are expected to add)
    bool status_exact = false;     // known_blocks was read with the stream idle (after creation / reset / growth: true until the next launch)
    bool overflow_latched = false; // an unchecked call ran out of pool: every integrate call fails until the pool is rebuilt / reset

    // TSDF per-frame state
    int32_t *touched_stamp = nullptr; // [table_capacity] last frame id that touched the slot
    int32_t *touched_list = nullptr;  // [2][max_blocks] slots touched this frame / batch (second half: the batch pipeline's other set)
    uint64_t *touched_mask = nullptr; // [2][table_capacity] per-slot frame bitmask of the multi-frame sweep (ditto)
    void *frame_px = nullptr;         // [max_points] uint2 {depth f32 bits, packed rgb}: the gather target
    int color_bgr = 0;                // hv_tsdf_set_color_order
    int touch_box_bits = 2048;        // env HV_TSDF_TOUCH_BOX_BITS (0 forces the touch pass's general path; tests)
    int32_t frame_counter = 0;
    int32_t merge_stamp = 0;          // frame_counter at the last hv_tsdf_mark_merged: units stamped later are "dirty"
    int32_t last_touch_parity = 0;
    bool touch_counters_clean = true; // both touched-list counters are zero (false after an online frame)
    void *batch_buf = nullptr;        // multi-frame sweep scratch: B frame records + B HvFrameParams
    size_t batch_buf_bytes = 0;
    // batch pipeline (hv_tsdf_integrate_batch): the touch + pack launch of batch k+1 runs on stream_aux while batch k is swept
    // on `stream`; two sets of scratch (batch_buf / batch_buf2, the halves of touched_list / touched_mask, TOUCH0 / TOUCH1)
    void *batch_buf2 = nullptr;
    size_t batch_buf2_bytes = 0;
    hipEvent_t ev_h2d = nullptr;      // behind the H2D copies of a call whose sources are page-locked (hv_h2d_fence)
    hipStream_t stream_aux = nullptr;
    hipEvent_t ev_prep = nullptr;     // touch + pack of the current batch done (stream_aux -> stream)
    hipEvent_t ev_presweep = nullptr; // everything on `stream` up to the point just before the previous batch's sweep
    int batch_parity = 0;             // scratch set of the next batch
    bool pipe_armed = false;          // ev_presweep is recorded and ...
    uint64_t pipe_version = 0;        // ... content_version has this value iff nothing else touched the volume since that batch
    // undistort / rectify maps of the camera (hv_tsdf_set_rectify_maps): when set, every frame handed to hv_tsdf_integrate* is remapped
    // on the device first (colour bilinear, depth nearest: the reference's per-keyframe cv2.remap pair) into rect_buf
    float *rect_map_x = nullptr, *rect_map_y = nullptr;
    size_t rect_map_x_bytes = 0, rect_map_y_bytes = 0;
    int32_t rect_W = 0, rect_H = 0;   // 0: no rectification
    void *rect_buf = nullptr;         // [B frames depth][B frames rgb] of the batch being prepared
    size_t rect_buf_bytes = 0;
    float *mult_table = nullptr;      // per-pixel depth-to-distance multiplier of the current intrinsics (multi-frame sweep)
    size_t mult_table_bytes = 0;
    float mult_key[4] = {0.f, 0.f, 0.f, 0.f}; // cx, cy, 1/fx, 1/fy the table was built for
    int32_t mult_W = 0, mult_H = 0;
    // pinned ring of per-batch HvFrameParams (async H2D without a host sync per call)
    void *pinned_params[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t params_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int params_idx = 0;
    int32_t tile[4] = {0, 0, 0, 0}; // u0, v0, u1, v1; all zero = whole image
    int32_t owner_rank = 0, owner_world = 1; // hv_tsdf_set_owner
    float sem_depth_threshold = 10.0f;       // VoxelSemanticDataT::kDepthThreshold (hv_set_depth_threshold)
    float sem_depth_decay_rate = 0.07f;      // VoxelSemanticDataProbabilisticT::kDepthDecayRate (hv_set_depth_decay_rate)
    void *assoc_buf = nullptr;               // association vote table + pending list (hv_semantic_ops.hip)
    void *assoc_clean = nullptr;             // == assoc_buf (and assoc_clean_bytes == assoc_buf_bytes: a grown buffer may come back at the
    size_t assoc_clean_bytes = 0;            // same address) while its vote table is known to be empty - the compaction kernel clears what it reads
    size_t assoc_buf_bytes = 0;
    int32_t assoc_pending_cap = 0;           // pending-list capacity of the last hv_assoc_vote
    int assoc_state = 0;                     // 0: no association yet, 1: voted (pairs on the device), 2: decided (map on the device)
    void *segments_cache = nullptr;          // host-side result of hv_object_segments_compute (HvSegmentsCache*)
    // semantic grids: one bit per voxel of the pool, set when the voxel takes its first point (k_sem_reduce) and never cleared by a
    // voxel reset (carve / remove_*): "may be occupied".  A surface touches ~7 % of a block's voxels, and every per-voxel scan
    // (association vote, carve, get_voxels, segments, size) used to read all 64 / 128-byte records to find them; with the bit a
    // lane reads its record only when the bit is set.  [max_blocks * bs^3 / 64] words, pool order.
    unsigned long long *occ = nullptr;
    void *semb_tasks = nullptr;              // semantic bucket path: the big buckets' (block, voxel range) tasks
    size_t semb_tasks_bytes = 0;

    // staging for HV_HOST inputs
    void *stage_a = nullptr;
    void *stage_b = nullptr;
    size_t stage_a_bytes = 0, stage_b_bytes = 0;
    // pipelined staging of host-resident FRAMES (hv_stage_frames): caller memory -> one of two page-locked slots (worker
    // threads) -> DMA on a copy stream into one of two device sets, so that the next sub-chunk is copied by the CPU while
    // the previous one crosses PCIe, and batch k+1 crosses PCIe while batch k is swept
    void *hs_pinned[2] = {nullptr, nullptr};
    size_t hs_pinned_bytes[2] = {0, 0};
    hipEvent_t hs_pinned_done[2] = {nullptr, nullptr}; // the DMA that last read the slot has completed
    void *hs_dev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; // [set][depth | colour]
    size_t hs_dev_bytes[2][2] = {{0, 0}, {0, 0}};
    hipEvent_t hs_dev_ready[2] = {nullptr, nullptr};   // the set's frames have arrived (copy stream -> consumer stream)
    hipEvent_t hs_dev_free[2] = {nullptr, nullptr};    // the kernels that read the set are done (consumer -> copy stream)
    bool hs_dev_free_valid[2] = {false, false};
    hipStream_t hs_stream = nullptr;
    int hs_set = 0, hs_slot = 0;

    // VOXEL_GRID scratch
    uint32_t *sort_keys_in = nullptr, *sort_keys_out = nullptr;
    uint32_t *sort_vals_in = nullptr, *sort_vals_out = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    // per-call point bins of the grid modes (hv_bins.h): per-slot counts and inline entries sized by the table (re-made when it
    // moves), overflow pages and records sized by max_points
    HvBins bins{};
    uint64_t bins_cap = 0;           // table capacity the per-slot arrays were made for
    bool bins_clean = false;         // counts and list lengths are zero (false after a reset / rebuild / aborted claim pass)
    uint32_t bins_epoch = 0;
    void *bin_rec = nullptr;         // [max_points] packed per-point records the bin pass leaves for the fold (16 or 32 bytes each)
    size_t bin_rec_bytes = 0;
    float *scratch_points = nullptr; // [max_points*3]
    float *scratch_colors = nullptr; // [max_points*3]
    int local_bits = 9;

    // Surface-extraction cache (TSDF): the C ABI is "ask for the sizes, then fetch", i.e. two calls per extraction; the first
    // call does all the device work into out_a / out_b and the second only copies, as long as the volume did not change in
    // between (content_version is bumped by every call that changes voxels).
    uint64_t content_version = 1;
    uint64_t mesh_cache_version = 0, points_cache_version = 0; // content_version the cached results belong to (0: none)
    int64_t mesh_cache_nv = 0, mesh_cache_nt = 0, points_cache_n = 0;
    bool mesh_cache_f32 = false, points_cache_f32 = false; // out_a holds float32 rows (hv_tsdf_extract_mesh_f32 / _points_f32)
    // per-unit column masks both extractions start from (hv_extract.hip: k_unit_masks), valid for unit_masks_version
    void *unit_masks = nullptr;
    uint64_t unit_masks_version = 0;
    int unit_masks_units = 0;
    // Incremental extraction (round 6).  The masks, the marching-cubes classification and the point counts are kept PER UNIT between
    // extractions (indexed by pool slot, sized for unit_cache_cap units) and recomputed only where something changed: a unit whose
    // touched_stamp (the id of the last frame that wrote to it: every fuse path stamps its units) is later than the frame_counter of
    // the previous pass gets new masks and records that pass in mask_stamp[unit]; a unit is classified / counted again when a unit of
    // its neighbourhood has a mask_stamp later than the previous classification / count.  Writers that do NOT stamp (reset, import,
    // halo unpack, rebuild, roll-back) bump extract_epoch: a cache of another epoch is recomputed in full.
    uint64_t extract_epoch = 1;
    int unit_cache_cap = 0;                          // units the three caches are laid out for
    uint64_t unit_masks_epoch = 0, mc_epoch = 0, pc_epoch = 0;
    int32_t unit_masks_stamp = -1, mc_stamp = -1, pc_stamp = -1; // frame_counter at the last pass of each
    int mc_units = 0, pc_units = 0;
    void *mc_cache = nullptr; // [edge_mask cap*192 u64][counts cap+1 u64][word_prefix cap*192 u32][cases cap*4096 u8]
    void *pc_cache = nullptr; // [count cap+1 i32]

    // output scratch (grown on demand)
    void *out_a = nullptr;
    void *out_b = nullptr;
    void *out_c = nullptr;
    size_t out_a_bytes = 0, out_b_bytes = 0, out_c_bytes = 0;
    // histograms + state of hv_filter_shadow_points_on_stream (a caller's stream, beside the volume's: scratch of its own, four sets in turn)
    void *shadow_ring = nullptr;
    int shadow_ring_next = 0;
    // the halo merge's plan of hv_merge_halo_plan_device (hv_halo.hip): [shared keys n*3 i32][action n u8], device memory
    void *halo_plan = nullptr;
    size_t halo_plan_bytes = 0;
    int64_t halo_plan_n = 0;
    // hv_semantic_fuse_keyframe: the keyframe's filtered depth and object-id image ([npx f32][npx i32])
    void *kf_buf = nullptr;
    size_t kf_buf_bytes = 0;

    // profiling
    bool profiling = false;
    std::vector<HvEventPair> events;
    size_t events_used = 0;
    int64_t prof_units = 0;
};

int hv_ensure_buffer(hv_volume *v, void **buf, size_t *cur, size_t want);
// Before an integrate call launches anything.  HV_ERR_CAPACITY if an earlier call ran out of pool (nothing more is fused
// until hv_reserve_blocks / hv_reset); grows the pool when more than half of it is known to be in use; *checked = the
// caller must verify its claim pass synchronously (hv_claims_fit) because the headroom is not known to cover this call.
int hv_capacity_gate(hv_volume *v, bool *checked);
// After the kernel that claims blocks, in checked mode: synchronise, and if some claims did not fit, grow the pool (the table
// is rebuilt without the failed keys) and return HV_RETRY_CLAIM so that the caller relaunches its claim pass - nothing was
// written to voxels yet, so nothing is lost.  HV_ERR_CAPACITY if the pool cannot grow.
int hv_claims_fit(hv_volume *v);
static constexpr int HV_RETRY_CLAIM = 1000;
int32_t hv_next_status_seq(hv_volume *v); // sequence number for the call's publishing kernel
void hv_launch_publish_status(hv_volume *v); // modes whose last kernel does not publish by itself
int hv_read_counters(hv_volume *v); // D2H of the counter block (synchronises the stream)
// hv_prep.hip: n_frames device-resident frames through the volume's rectify maps, queued on `s`
int hv_rectify_frames_device(hv_volume *v, hipStream_t s, const void *d_depth, int32_t depth_dtype, const uint8_t *d_rgb, int n_frames,
                             int height, int width, void *d_depth_out, uint8_t *d_rgb_out);
int hv_stage_in(hv_volume *v, const void *src, size_t bytes, int32_t loc, int which, const void **dev);
// hipMemcpyAsync(H2D) of a caller's array on the volume's stream; waits for the copy when the source is page-locked (the DMA would
// otherwise read it after the call has returned: the ABI borrows host arrays for the duration of the call only)
int hv_h2d(hv_volume *v, void *dst, const void *src, size_t bytes);
// several arrays of one call: queue them with hv_h2d_lazy (sets *pending when a source is page-locked), then ONE hv_h2d_fence
int hv_h2d_lazy(hv_volume *v, void *dst, const void *src, size_t bytes, bool *pending);
int hv_h2d_fence(hv_volume *v, bool pending);
// Host-resident frames -> device (pipelined, see hv_volume::hs_*).  Frame f's depth is depth_ptrs[f] when depth_ptrs is given,
// else depth_base + f * depth_frame_bytes (same for colour).  Returns the device arrays (frames contiguous) and the set whose
// hs_dev_ready event the consuming stream has to wait for; hv_stage_frames_consumed records hs_dev_free on that stream after
// the last kernel that reads the set.  The caller's memory has been read completely when hv_stage_frames returns.
int hv_stage_frames(hv_volume *v, const void *const *depth_ptrs, const void *depth_base, size_t depth_frame_bytes,
                    const void *const *rgb_ptrs, const void *rgb_base, size_t rgb_frame_bytes, int n_frames,
                    const void **d_depth, const void **d_rgb, int *set);
int hv_stage_frames_consumed(hv_volume *v, int set, hipStream_t consumer);
void hv_profile_begin(hv_volume *v);
void hv_profile_end(hv_volume *v, int64_t units);
void hv_invert4x4(const double *m, double *out);
void hv_segments_cache_free(void *cache); // hv_semantic_ops.hip
