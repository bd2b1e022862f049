// Semantic voxel payloads of pySLAM's cpp/volumetric on the GPU block hash (gfx950).
//
//   HvSemVoxel   voting payload        VoxelSemanticData              voxel_data_semantic.h:106-202
//   HvProbVoxel  log-probability one   VoxelSemanticDataProbabilistic voxel_data_semantic.h:249-672
//
// Both keep `count`, float64 position sums and float32 colour sums; they differ in how the
// (object_id, class_id) label is fused.  The accessors below are the payload-independent interface the
// kernels use (get_object_id / get_class_id / get_confidence / set_object_id / reset of the reference).
#pragma once
#include <cmath>
#include <cstdint>

#include "hv_common.h"

// ---- voting payload: 64 B, ids stored +1 so the zeroed pool means object -1 / class -1 ------------
struct __attribute__((aligned(16))) HvSemVoxel {
    int32_t count;
    int32_t obj1; // object_id + 1
    int32_t cls1; // class_id + 1
    int32_t counter;
    double pos[3];
    float col[3];
    float pad[3];
};
static_assert(sizeof(HvSemVoxel) == 64, "HvSemVoxel must be 64 bytes");

// ---- probabilistic payload: 128 B.  The reference keeps a std::map<(object_id, class_id), float
// log_prob> per voxel ("typically 1-5 unique labels per voxel", voxel_data_semantic.h:239-242); here
// the map is HV_PROB_K inline slots in insertion order (sorted on the fly where the reference
// iterates the map).  A (K+1)-th distinct label on one voxel is dropped and counted
// (HV_CNT_LABEL_OVERFLOW, hv_label_overflows()) — never silently.
//   meta = nlab | (best + 1) << 8: best = slot of the cached most likely pair (most_likely_pair,
//   voxel_data_semantic.h:266-270; 0 = cache not valid / empty map).
static constexpr int HV_PROB_K = 7;
struct __attribute__((aligned(16))) HvProbVoxel {
    int32_t count;
    uint32_t meta;
    double pos[3];
    float col[3];
    int32_t obj[HV_PROB_K];
    int32_t cls[HV_PROB_K];
    float logp[HV_PROB_K];
};
static_assert(sizeof(HvProbVoxel) == 128, "HvProbVoxel must be 128 bytes");

// BASE_LOG_PROB_PER_OBSERVATION = -log(0.9), voxel_data_semantic.h:287
#define HV_BASE_LOG_PROB 0.10536051565782628f

struct HvSemParams {
    float inv_voxel_size;
    int32_t bs, nvox, local_bits;
    float depth_threshold;  // kDepthThreshold
    float depth_decay_rate; // kDepthDecayRate (probabilistic payload only)
    int32_t owner_rank, owner_world; // multi-GPU block ownership (hv_set_owner): points of blocks another GPU owns are skipped
};

// The reference calls std::exp / std::log on floats (glibc expf/logf, < 1 ulp and correctly rounded
// in all but ~1e-3 of the cases); evaluating in double and rounding once reproduces that on the
// device up to those rare last-bit cases (tests: labels exact, confidences <= 1e-6).
__host__ __device__ inline float hv_expf_cr(float x) { return (float)exp((double)x); }
__host__ __device__ inline float hv_logf_cr(float x) { return (float)log((double)x); }

// ---- voting accessors ------------------------------------------------------------------------------
__host__ __device__ inline int32_t sem_object_id(const HvSemVoxel *v) { return v->obj1 - 1; }
__host__ __device__ inline int32_t sem_class_id(const HvSemVoxel *v) { return v->cls1 - 1; }
// get_confidence(), voxel_data_semantic.h:116-133
__host__ __device__ inline float sem_confidence(const HvSemVoxel *v) {
    if (v->count == 0) return 0.0f;
    const float r = (float)v->counter / (float)v->count;
    return r < 1.0f ? r : 1.0f;
}
__host__ __device__ inline int32_t sem_confidence_counter(const HvSemVoxel *v) { return v->counter; }
__host__ __device__ inline void sem_set_object_id(HvSemVoxel *v, int32_t id) { v->obj1 = id + 1; }

// ---- probabilistic accessors -----------------------------------------------------------------------
__host__ __device__ inline int prob_nlab(uint32_t meta) { return (int)(meta & 0xffu); }
__host__ __device__ inline int prob_best(uint32_t meta) { return (int)((meta >> 8) & 0xffu) - 1; }
__host__ __device__ inline uint32_t prob_meta(int nlab, int best) { return (uint32_t)nlab | ((uint32_t)(best + 1) << 8); }
// std::pair<int,int> ordering as one signed 64-bit value
__host__ __device__ inline int64_t prob_key(int32_t obj, int32_t cls) {
    return (int64_t)obj * 4294967296ll + ((int64_t)cls + 2147483648ll);
}
// update_cache(), voxel_data_semantic.h:575-601: the first maximum in map (key) order
__host__ __device__ inline int prob_argmax(const HvProbVoxel *v, int nlab) {
    int best = -1;
    for (int i = 0; i < nlab; ++i) {
        if (best < 0 || v->logp[i] > v->logp[best] ||
            (v->logp[i] == v->logp[best] && prob_key(v->obj[i], v->cls[i]) < prob_key(v->obj[best], v->cls[best])))
            best = i;
    }
    return best;
}
__host__ __device__ inline int prob_best_slot(const HvProbVoxel *v) {
    const int nlab = prob_nlab(v->meta);
    if (nlab == 0) return -1;
    const int b = prob_best(v->meta);
    return b >= 0 ? b : prob_argmax(v, nlab);
}
__host__ __device__ inline int32_t sem_object_id(const HvProbVoxel *v) {
    const int b = prob_best_slot(v);
    return b < 0 ? -1 : v->obj[b];
}
__host__ __device__ inline int32_t sem_class_id(const HvProbVoxel *v) {
    const int b = prob_best_slot(v);
    return b < 0 ? -1 : v->cls[b];
}
// log_add_exp, voxel_data_semantic.h:639-648
__host__ __device__ inline float prob_log_add_exp(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float m = (a < b) ? b : a;
    return m + hv_logf_cr(hv_expf_cr(a - m) + hv_expf_cr(b - m));
}
// get_log_normalization(), voxel_data_semantic.h:620-637: incremental log-add-exp in map (key) order
__host__ __device__ inline float prob_log_normalization(const HvProbVoxel *v, int nlab) {
    float acc = -INFINITY;
    int64_t last = 0;
    for (int step = 0; step < nlab; ++step) {
        int pick = -1;
        int64_t pk = 0;
        for (int i = 0; i < nlab; ++i) {
            const int64_t k = prob_key(v->obj[i], v->cls[i]);
            if (step > 0 && k <= last) continue;
            if (pick < 0 || k < pk) {
                pick = i;
                pk = k;
            }
        }
        if (pick < 0) break;
        acc = prob_log_add_exp(acc, v->logp[pick]);
        last = pk;
    }
    return acc;
}
// compute_confidence(), voxel_data_semantic.h:562-572 (the cached confidence_ is refreshed on every update)
__host__ __device__ inline float sem_confidence(const HvProbVoxel *v) {
    const int nlab = prob_nlab(v->meta);
    if (nlab == 0) return 0.0f;
    const int b = prob_best_slot(v);
    if (v->obj[b] == -1 || v->cls[b] == -1) return 0.0f;
    return hv_expf_cr(v->logp[b] - prob_log_normalization(v, nlab));
}
// get_confidence_counter(), voxel_data_semantic.h:505-511
__host__ __device__ inline int32_t sem_confidence_counter(const HvProbVoxel *v) {
    return (int32_t)(sem_confidence(v) * (float)v->count);
}
// set_object_id() -> force_label_distribution(), voxel_data_semantic.h:476-481, 603-618: the label
// distribution collapses to the single pair (id, current class) with log-probability 0.
__host__ __device__ inline void sem_set_object_id(HvProbVoxel *v, int32_t id) {
    const int32_t cls = sem_class_id(v);
    if (id >= 0 && cls >= 0) {
        v->obj[0] = id;
        v->cls[0] = cls;
        v->logp[0] = 0.0f;
        v->meta = prob_meta(1, 0);
    } else {
        v->meta = prob_meta(0, -1);
    }
}

// One semantic observation folded into a probabilistic voxel: initialize_semantics_log_prob
// (count == 0, voxel_data_semantic.h:311-324) or update_semantics_log_prob (:358-417).  Returns false
// when the voxel already holds HV_PROB_K distinct labels and this one is new (observation dropped).
__host__ __device__ inline bool prob_fold(HvProbVoxel *v, bool first, int32_t obj, int32_t cls, float lp) {
    int nlab = prob_nlab(v->meta), best = prob_best(v->meta);
    int idx = -1;
    for (int i = 0; i < nlab; ++i)
        if (v->obj[i] == obj && v->cls[i] == cls) idx = i;
    if (idx < 0 && nlab == HV_PROB_K) return false;
    if (first) {
        if (idx < 0) {
            idx = nlab++;
            v->obj[idx] = obj;
            v->cls[idx] = cls;
        }
        v->logp[idx] = lp;
        best = idx;
    } else if (idx < 0) {
        idx = nlab++;
        v->obj[idx] = obj;
        v->cls[idx] = cls;
        v->logp[idx] = lp;
        if (best >= 0) {
            if (lp > v->logp[best]) best = idx;
        } else {
            best = prob_argmax(v, nlab);
        }
    } else {
        const float old = v->logp[idx];
        const float now = old + lp;
        v->logp[idx] = now;
        if (best >= 0) {
            if (idx == best) {
                if (now < old) best = prob_argmax(v, nlab);
            } else if (now > v->logp[best]) {
                best = idx;
            }
        } else {
            best = prob_argmax(v, nlab);
        }
    }
    v->meta = prob_meta(nlab, best);
    return true;
}

// log evidence of one observation: update_semantics / update_semantics_with_depth,
// voxel_data_semantic.h:419-447
__host__ __device__ inline float prob_observation_log_prob(bool has_depth, float depth, const HvSemParams &G) {
    if (!has_depth || depth <= G.depth_threshold) return HV_BASE_LOG_PROB;
    const float confidence = hv_expf_cr(-(depth - G.depth_threshold) * G.depth_decay_rate);
    return confidence * HV_BASE_LOG_PROB;
}

// The occupancy bit of voxel `gid` (pool order): false = the voxel never took a point, its record need not be read.  The lanes
// of a wave ask for 64 consecutive voxels: one 8-byte word.
__device__ __forceinline__ bool sem_maybe_occupied(const unsigned long long *__restrict__ occ, int64_t gid) {
    return occ == nullptr || ((occ[gid >> 6] >> (gid & 63)) & 1ull) != 0ull;
}

#ifdef __HIPCC__
// r-th (0-based) set bit of a 64-bit word that has more than r bits set
__device__ __forceinline__ int sem_select_bit(unsigned long long w, int r) {
    int pos = 0;
    uint32_t lo = (uint32_t)w;
    int c = __popc(lo);
    if (r >= c) {
        r -= c;
        pos = 32;
        lo = (uint32_t)(w >> 32);
    }
#pragma unroll
    for (int s = 16; s > 0; s >>= 1) {
        c = __popc(lo & ((1u << s) - 1u));
        if (r >= c) {
            r -= c;
            lo >>= s;
            pos += s;
        }
    }
    return pos;
}

// One WAVE visits the voxels of pool block b that may be occupied: body(gid, active) is called by all 64 lanes together (so that
// it may use ballots / wave-aggregated appends), `active` lanes hold a voxel whose occupancy bit is set (pool-order voxel index
// gid), the others hold nothing.  A surface touches ~34 of a block's 512 voxels: one call instead of eight, and no record of an
// empty voxel is read.  all_voxels (or occ == nullptr): every voxel of the block is visited, 64 at a time.
// The visit itself, for a block whose occupancy words are already in registers (lane w holds word w of the block; callers that
// walk many blocks request the next block's words before they work on this one).  Needs nvox % 64 == 0, nvox / 64 <= 64.
template <typename F>
__device__ __forceinline__ void sem_for_occupied_word(unsigned long long word, int64_t b, int nvox, F body) {
    const int lane = hv_lane_id();
    const int W = nvox >> 6;
    const int cnt = __popcll(word);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < HV_WAVE; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const int total = __shfl(incl, HV_WAVE - 1);
    for (int k0 = 0; k0 < total; k0 += HV_WAVE) { // wave-uniform trip count
        const int k = k0 + lane;
        const bool active = k < total;
        int wi = 0;
        for (int w = 0; w < W - 1; ++w) wi += (__shfl(incl, w) <= k) ? 1 : 0; // word that holds the k-th set bit
        const unsigned long long ww = __shfl(word, wi);
        const int before = __shfl(incl, wi) - __shfl(cnt, wi);
        const int pos = active ? sem_select_bit(ww, k - before) : 0;
        body(b * nvox + wi * 64 + pos, active);
    }
}
__device__ __forceinline__ bool sem_occ_words_usable(const unsigned long long *occ, int nvox) {
    return occ != nullptr && (nvox & 63) == 0 && (nvox >> 6) <= HV_WAVE;
}

template <typename F>
__device__ __forceinline__ void sem_for_occupied(const unsigned long long *__restrict__ occ, int64_t b, int nvox, bool all_voxels, F body) {
    const int lane = hv_lane_id();
    const int W = nvox >> 6;
    if (all_voxels || !sem_occ_words_usable(occ, nvox)) {
        for (int l0 = 0; l0 < nvox; l0 += HV_WAVE) {
            const int l = l0 + lane;
            const int64_t gid = b * nvox + l;
            body(gid, l < nvox && (occ == nullptr || all_voxels || sem_maybe_occupied(occ, gid)));
        }
        return;
    }
    sem_for_occupied_word(lane < W ? occ[b * W + lane] : 0ull, b, nvox, body);
}
#endif

static inline HvSemParams sem_params(const hv_volume *v) {
    HvSemParams G;
    G.inv_voxel_size = 1.0f / (float)v->cfg.voxel_size;
    G.bs = v->cfg.block_size;
    G.nvox = G.bs * G.bs * G.bs;
    G.local_bits = v->local_bits;
    G.depth_threshold = v->sem_depth_threshold;
    G.depth_decay_rate = v->sem_depth_decay_rate;
    G.owner_rank = v->owner_rank;
    G.owner_world = v->owner_world;
    return G;
}
static inline bool hv_is_semantic(const hv_volume *v) {
    return v->cfg.mode == HV_MODE_VOXEL_SEMANTIC_GRID || v->cfg.mode == HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID;
}
