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

static inline HvSemParams sem_params(const hv_volume *v) {
    HvSemParams G;
    G.inv_voxel_size = 1.0f / (float)v->cfg.voxel_size;
    G.bs = v->cfg.block_size;
    G.nvox = G.bs * G.bs * G.bs;
    G.local_bits = v->local_bits;
    G.depth_threshold = v->sem_depth_threshold;
    G.depth_decay_rate = v->sem_depth_decay_rate;
    return G;
}
static inline bool hv_is_semantic(const hv_volume *v) {
    return v->cfg.mode == HV_MODE_VOXEL_SEMANTIC_GRID || v->cfg.mode == HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID;
}
