// Semantic voxel payloads of pySLAM's cpp/volumetric on the GPU block hash (gfx950).
//
//   HvSemVoxel   voting payload        VoxelSemanticData              voxel_data_semantic.h:106-202
//   HvProbVoxel  log-probability one   VoxelSemanticDataProbabilistic voxel_data_semantic.h:249-672
//   HvSem2Voxel  two counters          VoxelSemanticData2             voxel_data_semantic2.h:46-196
//   HvProb2Voxel marginal label maps   VoxelSemanticDataProbabilistic2 voxel_data_semantic2.h:256-787
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

// ---- probabilistic payload: 128 B.  The reference keeps a std::map<(object_id, class_id), float log_prob> per voxel
// ("typically 1-5 unique labels per voxel", voxel_data_semantic.h:239-242); here the map is HV_PROB_K inline slots in insertion
// order (sorted on the fly where the reference iterates the map) and, beyond them, a chain of overflow nodes of HV_PROB_NK pairs
// from a per-volume node pool (`next` = node index + 1; only the voxel's own fold thread walks or extends its chain).  Rounds 1-3
// dropped the 8th distinct pair of a voxel (measured: 1-2 % of the occupied voxels under 5 % uniform label noise, largest
// reference map 23 pairs); now the map is as unbounded as the reference's up to 254 pairs or an exhausted node pool, both counted
// (HV_CNT_LABEL_OVERFLOW, hv_label_overflows()) - never silent.  A voxel that is reset (carve, remove_*: sem_reset) and a collapsed map
// (set_object_id) keep their chain and grow the next map into it: carve / re-observe cycles do not drain the pool
// (tests/test_gpu_semantic_ops.py::test_carve_and_reobserve_cycles_reuse_overflow_nodes); hv_reset takes every node back.
//   meta = nlab | (best + 1) << 8: best = index of the cached most likely pair (most_likely_pair, voxel_data_semantic.h:266-270;
//   0 = cache not valid / empty map).  Pair i lives inline for i < HV_PROB_K, else in node (i - K) / NK of the chain.
static constexpr int HV_PROB_K = 6;
static constexpr int HV_PROB_NK = 10;
static constexpr int HV_PROB_MAX = 254;
struct __attribute__((aligned(16))) HvProbVoxel {
    int32_t count;
    uint32_t meta;
    double pos[3];
    float col[3];
    int32_t obj[HV_PROB_K];
    int32_t cls[HV_PROB_K];
    float logp[HV_PROB_K];
    uint32_t next; // first overflow node + 1 (0: none)
    uint32_t pad[2];
};
static_assert(sizeof(HvProbVoxel) == 128, "HvProbVoxel must be 128 bytes");
struct __attribute__((aligned(16))) HvProbNode {
    int32_t obj[HV_PROB_NK];
    int32_t cls[HV_PROB_NK];
    float logp[HV_PROB_NK];
    uint32_t next;
    uint32_t pad;
};
static_assert(sizeof(HvProbNode) == 128, "HvProbNode must be 128 bytes");

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
__host__ __device__ inline int32_t sem_object_id(const HvSemVoxel *v, const void * = nullptr) { return v->obj1 - 1; }
__host__ __device__ inline int32_t sem_class_id(const HvSemVoxel *v, const void * = nullptr) { return v->cls1 - 1; }
// get_confidence(), voxel_data_semantic.h:116-133
__host__ __device__ inline float sem_confidence(const HvSemVoxel *v, const void * = nullptr) {
    if (v->count == 0) return 0.0f;
    const float r = (float)v->counter / (float)v->count;
    return r < 1.0f ? r : 1.0f;
}
__host__ __device__ inline int32_t sem_confidence_counter(const HvSemVoxel *v, const void * = nullptr) { return v->counter; }
__host__ __device__ inline void sem_set_object_id(HvSemVoxel *v, const HvTable &, int32_t id) { v->obj1 = id + 1; }

// ---- probabilistic accessors -----------------------------------------------------------------------
// `nodes` = the volume's overflow-node pool (HvTable::prob_nodes); the voting payload's accessors take and ignore it so that the
// payload-independent kernels read the same for both.
__host__ __device__ inline int prob_nlab(uint32_t meta) { return (int)(meta & 0xffu); }
__host__ __device__ inline int prob_best(uint32_t meta) { return (int)((meta >> 8) & 0xffu) - 1; }
__host__ __device__ inline uint32_t prob_meta(int nlab, int best) { return (uint32_t)nlab | ((uint32_t)(best + 1) << 8); }
// std::pair<int,int> ordering as one signed 64-bit value
__host__ __device__ inline int64_t prob_key(int32_t obj, int32_t cls) {
    return (int64_t)obj * 4294967296ll + ((int64_t)cls + 2147483648ll);
}
struct HvProbPair {
    int32_t obj, cls;
    float logp;
};
__host__ __device__ inline const HvProbNode *prob_node_of(const HvProbVoxel *v, const HvProbNode *nodes, int &i) {
    i -= HV_PROB_K;
    uint32_t n = v->next;
    while (i >= HV_PROB_NK) {
        n = nodes[n - 1].next;
        i -= HV_PROB_NK;
    }
    return &nodes[n - 1];
}
// CHAIN = false: the caller knows that the map has no pair beyond the inline slots (the common case: the node walk is compiled out)
template <bool CHAIN = true>
__host__ __device__ inline HvProbPair prob_get(const HvProbVoxel *v, const HvProbNode *nodes, int i) {
    if (!CHAIN || i < HV_PROB_K) return {v->obj[i], v->cls[i], v->logp[i]};
    const HvProbNode *nd = prob_node_of(v, nodes, i);
    return {nd->obj[i], nd->cls[i], nd->logp[i]};
}
template <bool CHAIN = true>
__host__ __device__ inline void prob_set_logp(HvProbVoxel *v, HvProbNode *nodes, int i, float lp) {
    if (!CHAIN || i < HV_PROB_K) {
        v->logp[i] = lp;
        return;
    }
    HvProbNode *nd = (HvProbNode *)prob_node_of(v, nodes, i);
    nd->logp[i] = lp;
}
// update_cache(), voxel_data_semantic.h:575-601: the first maximum in map (key) order
template <bool CHAIN = true>
__host__ __device__ inline int prob_argmax(const HvProbVoxel *v, const HvProbNode *nodes, int nlab) {
    int best = -1;
    HvProbPair bp{0, 0, 0.f};
    for (int i = 0; i < nlab; ++i) {
        const HvProbPair p = prob_get<CHAIN>(v, nodes, i);
        if (best < 0 || p.logp > bp.logp || (p.logp == bp.logp && prob_key(p.obj, p.cls) < prob_key(bp.obj, bp.cls))) {
            best = i;
            bp = p;
        }
    }
    return best;
}
__host__ __device__ inline int prob_best_slot(const HvProbVoxel *v, const HvProbNode *nodes) {
    const int nlab = prob_nlab(v->meta);
    if (nlab == 0) return -1;
    const int b = prob_best(v->meta);
    return b >= 0 ? b : prob_argmax(v, nodes, nlab);
}
__host__ __device__ inline int32_t sem_object_id(const HvProbVoxel *v, const void *nodes) {
    const int b = prob_best_slot(v, (const HvProbNode *)nodes);
    return b < 0 ? -1 : prob_get(v, (const HvProbNode *)nodes, b).obj;
}
__host__ __device__ inline int32_t sem_class_id(const HvProbVoxel *v, const void *nodes) {
    const int b = prob_best_slot(v, (const HvProbNode *)nodes);
    return b < 0 ? -1 : prob_get(v, (const HvProbNode *)nodes, b).cls;
}
// log_add_exp, voxel_data_semantic.h:639-648
__host__ __device__ inline float prob_log_add_exp(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float m = (a < b) ? b : a;
    return m + hv_logf_cr(hv_expf_cr(a - m) + hv_expf_cr(b - m));
}
// get_log_normalization(), voxel_data_semantic.h:620-637: incremental log-add-exp in map (key) order
__host__ __device__ inline float prob_log_normalization(const HvProbVoxel *v, const HvProbNode *nodes, int nlab) {
    float acc = -INFINITY;
    int64_t last = 0;
    for (int step = 0; step < nlab; ++step) {
        int pick = -1;
        int64_t pk = 0;
        float plp = 0.f;
        for (int i = 0; i < nlab; ++i) {
            const HvProbPair p = prob_get(v, nodes, i);
            const int64_t k = prob_key(p.obj, p.cls);
            if (step > 0 && k <= last) continue;
            if (pick < 0 || k < pk) {
                pick = i;
                pk = k;
                plp = p.logp;
            }
        }
        if (pick < 0) break;
        acc = prob_log_add_exp(acc, plp);
        last = pk;
    }
    return acc;
}
// compute_confidence(), voxel_data_semantic.h:562-572 (the cached confidence_ is refreshed on every update)
__host__ __device__ inline float sem_confidence(const HvProbVoxel *v, const void *nodes_) {
    const HvProbNode *nodes = (const HvProbNode *)nodes_;
    const int nlab = prob_nlab(v->meta);
    if (nlab == 0) return 0.0f;
    const int b = prob_best_slot(v, nodes);
    const HvProbPair bp = prob_get(v, nodes, b);
    if (bp.obj == -1 || bp.cls == -1) return 0.0f;
    return hv_expf_cr(bp.logp - prob_log_normalization(v, nodes, nlab));
}
// `confidence >= 0` - the visit predicate of iterate_voxels_in_camera_frustrum with min_confidence = 0 (association vote, carve) -
// without evaluating the confidence where its sign is known: the confidence of a map whose log-probabilities are all finite is
// exp(best - log sum exp) in (0, 1] (or 0 for an empty / unlabelled best pair), so only a map holding a NaN or an infinity (a NaN
// depth handed to the integrate) takes the exp / log chain; the answer is the reference's in every case.
__host__ __device__ inline bool sem_confidence_not_negative(const HvSemVoxel *v, const void * = nullptr) { return sem_confidence(v) >= 0.0f; }
__host__ __device__ inline bool sem_confidence_not_negative(const HvProbVoxel *v, const void *nodes_) {
    const HvProbNode *nodes = (const HvProbNode *)nodes_;
    const int nlab = prob_nlab(v->meta);
    bool finite = true;
    for (int i = 0; i < nlab; ++i) {
        const float lp = prob_get(v, nodes, i).logp;
        finite = finite && (lp - lp == 0.0f); // false for NaN and +-inf
    }
    return finite ? true : sem_confidence(v, nodes_) >= 0.0f;
}
// get_confidence_counter(), voxel_data_semantic.h:505-511
__host__ __device__ inline int32_t sem_confidence_counter(const HvProbVoxel *v, const void *nodes) {
    return (int32_t)(sem_confidence(v, nodes) * (float)v->count);
}
// set_object_id() -> force_label_distribution(), voxel_data_semantic.h:476-481, 603-618: the label
// distribution collapses to the single pair (id, current class) with log-probability 0.
__host__ __device__ inline void sem_set_object_id(HvProbVoxel *v, const HvTable &table, int32_t id) {
    const int32_t cls = sem_class_id(v, table.prob_nodes);
    if (id >= 0 && cls >= 0) {
        v->obj[0] = id;
        v->cls[0] = cls;
        v->logp[0] = 0.0f;
        v->meta = prob_meta(1, 0);
    } else {
        v->meta = prob_meta(0, -1);
    }
}

#ifdef __HIPCC__
// The fold keeps the voxel it works on as a LOCAL copy (sem_fold_run: read once, folded, written once).  An inline slot reached with
// a run-time index (`v->logp[i]`) forces that copy into scratch memory - the probabilistic fold kernels carried 144 bytes of it per
// lane and paid a memory round trip for every slot they looked at (round 5).  The helpers below reach the six inline slots through
// constant indices only (a select chain over the slots), so the copy lives in registers; pairs beyond the inline slots are in the
// node pool (global memory) either way.  Only the fold uses them: on a voxel in GLOBAL memory one indexed load beats six.
// (all six slots are read, unconditionally, and the VALUES are selected: loads in the arms of a branch are merged by the optimiser
// into one load through a selected address)
__device__ __forceinline__ HvProbPair prob_slot_get(const HvProbVoxel *v, int i) {
    HvProbPair p{v->obj[0], v->cls[0], v->logp[0]};
#pragma unroll
    for (int k = 1; k < HV_PROB_K; ++k) {
        const int32_t ok = v->obj[k], ck = v->cls[k];
        const float lk = v->logp[k];
        p.obj = i == k ? ok : p.obj;
        p.cls = i == k ? ck : p.cls;
        p.logp = i == k ? lk : p.logp;
    }
    return p;
}
// (every slot is stored, with its old or its new value: a conditional store per slot is merged by the optimiser into ONE store through
// a pointer picked from the six slots - a run-time address again)
__device__ __forceinline__ void prob_slot_set(HvProbVoxel *v, int i, int32_t obj, int32_t cls, float lp) {
#pragma unroll
    for (int k = 0; k < HV_PROB_K; ++k) {
        v->obj[k] = i == k ? obj : v->obj[k];
        v->cls[k] = i == k ? cls : v->cls[k];
        v->logp[k] = i == k ? lp : v->logp[k];
    }
}
__device__ __forceinline__ void prob_slot_set_logp(HvProbVoxel *v, int i, float lp) {
#pragma unroll
    for (int k = 0; k < HV_PROB_K; ++k) v->logp[k] = i == k ? lp : v->logp[k];
}
template <bool CHAIN>
__device__ __forceinline__ HvProbPair prob_get_r(const HvProbVoxel *v, const HvProbNode *nodes, int i) {
    HvProbPair p = prob_slot_get(v, i); // (for i >= HV_PROB_K: slot 0, replaced below)
    if (CHAIN && i >= HV_PROB_K) {
        const HvProbNode *nd = prob_node_of(v, nodes, i);
        p = HvProbPair{nd->obj[i], nd->cls[i], nd->logp[i]};
    }
    return p;
}
template <bool CHAIN>
__device__ __forceinline__ void prob_set_logp_r(HvProbVoxel *v, HvProbNode *nodes, int i, float lp) {
    if (!CHAIN || i < HV_PROB_K) {
        prob_slot_set_logp(v, i, lp);
        return;
    }
    HvProbNode *nd = (HvProbNode *)prob_node_of(v, nodes, i);
    nd->logp[i] = lp;
}
// prob_argmax for the local copy: the same scan in slot order (inline slots first, then the chain)
template <bool CHAIN>
__device__ __forceinline__ int prob_argmax_r(const HvProbVoxel *v, const HvProbNode *nodes, int nlab) {
    int best = -1;
    HvProbPair bp{0, 0, 0.f};
    auto take = [&](int i, const HvProbPair p) {
        if (best < 0 || p.logp > bp.logp || (p.logp == bp.logp && prob_key(p.obj, p.cls) < prob_key(bp.obj, bp.cls))) {
            best = i;
            bp = p;
        }
    };
#pragma unroll
    for (int i = 0; i < HV_PROB_K; ++i) {
        const HvProbPair p{v->obj[i], v->cls[i], v->logp[i]};
        if (i < nlab) take(i, p);
    }
    if (CHAIN)
        for (int i = HV_PROB_K; i < nlab; ++i) {
            int j = i;
            const HvProbNode *nd = prob_node_of(v, nodes, j);
            take(i, HvProbPair{nd->obj[j], nd->cls[j], nd->logp[j]});
        }
    return best;
}
// One semantic observation folded into a probabilistic voxel: initialize_semantics_log_prob
// (count == 0, voxel_data_semantic.h:311-324) or update_semantics_log_prob (:358-417).  Returns false when the pair is new and
// cannot be stored (254 pairs, or the node pool is exhausted): the observation is dropped and counted.
template <bool CHAIN>
__device__ inline bool prob_fold_t(HvProbVoxel *v, const HvTable &table, bool first, int32_t obj, int32_t cls, float lp) {
    HvProbNode *nodes = (HvProbNode *)table.prob_nodes;
    int nlab = prob_nlab(v->meta), best = prob_best(v->meta);
    int idx = -1;
#pragma unroll
    for (int i = 0; i < HV_PROB_K; ++i) {
        const bool hit = (int)(v->obj[i] == obj) & (int)(v->cls[i] == cls) & (int)(i < nlab);
        idx = hit ? i : idx;
    }
    if (CHAIN)
        for (int i = HV_PROB_K; i < nlab; ++i) {
            int j = i;
            const HvProbNode *nd = prob_node_of(v, nodes, j);
            if (nd->obj[j] == obj && nd->cls[j] == cls) idx = i;
        }
    float best_lp = best >= 0 ? prob_get_r<CHAIN>(v, nodes, best).logp : 0.f;
    if (idx < 0) {
        // a new pair goes to index nlab: inline, or in the chain's node (nlab - K) / NK, which may have to be linked in first
        if (nlab >= HV_PROB_MAX) return false;
        if (CHAIN && nlab >= HV_PROB_K && (nlab - HV_PROB_K) % HV_PROB_NK == 0) {
            // the link the new node hangs on: the voxel's own `next`, or the last node's (no pointer into the local voxel is formed:
            // one that may point there or into the node pool would pin the copy in memory)
            const int hops = (nlab - HV_PROB_K) / HV_PROB_NK;
            HvProbNode *tail = nullptr;
            if (hops > 0) {
                uint32_t n = v->next;
                for (int hop = hops - 1; hop > 0; --hop) n = nodes[n - 1].next;
                tail = &nodes[n - 1];
            }
            const uint32_t own_next = v->next;
            if ((tail ? tail->next : own_next) == 0u) { // (a chain left behind by a collapsed map is taken up again)
                if (nodes == nullptr) return false;
                const int32_t id = atomicAdd(&table.counters[HV_CNT_PROB_NODES], 1);
                if (id >= table.prob_node_cap) {
                    atomicSub(&table.counters[HV_CNT_PROB_NODES], 1); // (the counter stays the number of nodes handed out)
                    return false;
                }
                nodes[id].next = 0u;
                if (tail) tail->next = (uint32_t)id + 1u;
                v->next = tail ? own_next : (uint32_t)id + 1u;
            }
        }
        idx = nlab++;
        if (!CHAIN || idx < HV_PROB_K) {
            prob_slot_set(v, idx, obj, cls, lp);
        } else {
            int j = idx;
            HvProbNode *nd = (HvProbNode *)prob_node_of(v, nodes, j);
            nd->obj[j] = obj;
            nd->cls[j] = cls;
            nd->logp[j] = lp;
        }
        if (first) {
            best = idx;
        } else if (best >= 0) {
            if (lp > best_lp) best = idx;
        } else {
            best = prob_argmax_r<CHAIN>(v, nodes, nlab);
        }
    } else if (first) {
        prob_set_logp_r<CHAIN>(v, nodes, idx, lp);
        best = idx;
    } else {
        const float old = prob_get_r<CHAIN>(v, nodes, idx).logp;
        const float now = old + lp;
        prob_set_logp_r<CHAIN>(v, nodes, idx, now);
        if (best >= 0) {
            if (idx == best) {
                if (now < old) best = prob_argmax_r<CHAIN>(v, nodes, nlab);
            } else if (now > best_lp) {
                best = idx;
            }
        } else {
            best = prob_argmax_r<CHAIN>(v, nodes, nlab);
        }
    }
    v->meta = prob_meta(nlab, best);
    return true;
}
// (a map that stays inside the inline slots after this observation takes the version without the node walk)
__device__ inline bool prob_fold(HvProbVoxel *v, const HvTable &table, bool first, int32_t obj, int32_t cls, float lp) {
    return prob_nlab(v->meta) < HV_PROB_K ? prob_fold_t<false>(v, table, first, obj, cls, lp) : prob_fold_t<true>(v, table, first, obj, cls, lp);
}
#endif

// =====================================================================================================================
// The "*2" payloads of voxel_data_semantic2.h (bound by the reference's module as VoxelBlockSemanticGrid2 /
// VoxelBlockSemanticProbabilisticGrid2, volumetric_grid_module.h:1014-1032; documented there as the inferior variants).
// =====================================================================================================================

// ---- VoxelSemanticData2 (voxel_data_semantic2.h:46-196): one confidence counter for the object id, one for the class id; 64 B ------
struct __attribute__((aligned(16))) HvSem2Voxel {
    int32_t count;
    int32_t obj1;        // object_id + 1
    int32_t cls1;        // class_id + 1
    int32_t obj_counter; // object_confidence_counter_
    double pos[3];
    float col[3];
    int32_t cls_counter; // class_confidence_counter_
    float pad[2];
};
static_assert(sizeof(HvSem2Voxel) == 64, "HvSem2Voxel must be 64 bytes");

__host__ __device__ inline int32_t sem_object_id(const HvSem2Voxel *v, const void * = nullptr) { return v->obj1 - 1; }
__host__ __device__ inline int32_t sem_class_id(const HvSem2Voxel *v, const void * = nullptr) { return v->cls1 - 1; }
// std::min(1.0f, counter / count), voxel_data_semantic2.h:60-76
__host__ __device__ inline float sem2_ratio(int32_t counter, int32_t count) {
    const float r = (float)counter / (float)count;
    return r < 1.0f ? r : 1.0f;
}
__host__ __device__ inline float sem_object_confidence(const HvSem2Voxel *v, const void * = nullptr) {
    return v->count == 0 ? 0.0f : sem2_ratio(v->obj_counter, v->count);
}
__host__ __device__ inline float sem_class_confidence(const HvSem2Voxel *v, const void * = nullptr) {
    return v->count == 0 ? 0.0f : sem2_ratio(v->cls_counter, v->count);
}
// get_confidence(): std::min(object confidence, class confidence), :79-83
__host__ __device__ inline float sem_confidence(const HvSem2Voxel *v, const void * = nullptr) {
    if (v->count == 0) return 0.0f;
    const float o = sem2_ratio(v->obj_counter, v->count), c = sem2_ratio(v->cls_counter, v->count);
    return c < o ? c : o;
}
__host__ __device__ inline bool sem_confidence_not_negative(const HvSem2Voxel *v, const void * = nullptr) { return sem_confidence(v) >= 0.0f; }
// get_confidence_counter(): std::min of the two counters, :55-57
__host__ __device__ inline int32_t sem_confidence_counter(const HvSem2Voxel *v, const void * = nullptr) {
    return v->cls_counter < v->obj_counter ? v->cls_counter : v->obj_counter;
}
__host__ __device__ inline void sem_set_object_id(HvSem2Voxel *v, const HvTable &, int32_t id) { v->obj1 = id + 1; }
// one labelled observation: initialize_semantics[_with_depth] (count == 0) / update_semantics[_with_depth], :120-195
__host__ __device__ inline void sem2_fold(HvSem2Voxel *v, bool first, bool gate, int32_t obj, int32_t cls) {
    if (!gate) return; // (*_with_depth: depth >= kDepthThreshold leaves the label state alone)
    if (first) {
        v->obj1 = obj + 1;
        v->cls1 = cls + 1;
        v->obj_counter = 1;
        v->cls_counter = 1;
        return;
    }
    if (v->obj1 == obj + 1) {
        v->obj_counter++;
    } else if (--v->obj_counter <= 0) {
        v->obj1 = obj + 1;
        v->obj_counter = 1;
    }
    if (v->cls1 == cls + 1) {
        v->cls_counter++;
    } else if (--v->cls_counter <= 0) {
        v->cls1 = cls + 1;
        v->cls_counter = 1;
    }
}

// ---- VoxelSemanticDataProbabilistic2 (voxel_data_semantic2.h:256-787): two std::map<int, float> per voxel, object id -> log-probability
// and class id -> log-probability; every observation adds HALF its log-probability to its object's entry and half to its class's.
// Storage: the record, the inline slots and the overflow-node chain of HvProbVoxel, as ONE list of entries
//     obj[i] = the id, cls[i] = which map the entry belongs to (0: object_log_probabilities, 1: class_log_probabilities), logp[i]
// (a voxel that saw one object and one class holds two entries; 6 inline, 254 in all, both limits counted like HvProbVoxel's).
//   meta = n entries | (cached most likely OBJECT entry + 1) << 8 | (cached most likely CLASS entry + 1) << 16; 0 = that cache is not valid:
//   the reference's object_cache_valid / class_cache_valid (:281-287).  They are valid after the first observation (:311-326) and
//   after set_object_id (:424-452, which may name an entry that an argmax in key order would not pick among equal log-probabilities),
//   every update clears both (:366-368); a cache that is not valid is the argmax, a pure function of the map - computed where it is asked
//   for, never stored.
struct __attribute__((aligned(16))) HvProb2Voxel : HvProbVoxel {};
static_assert(sizeof(HvProb2Voxel) == 128, "HvProb2Voxel must be 128 bytes");
__host__ __device__ inline int prob2_best_obj(uint32_t meta) { return (int)((meta >> 8) & 0xffu) - 1; }
__host__ __device__ inline int prob2_best_cls(uint32_t meta) { return (int)((meta >> 16) & 0xffu) - 1; }
__host__ __device__ inline uint32_t prob2_meta(int n, int best_obj, int best_cls) {
    return (uint32_t)n | ((uint32_t)(best_obj + 1) << 8) | ((uint32_t)(best_cls + 1) << 16);
}
struct HvProb2Best {
    int32_t id; // most_likely_object_id / most_likely_class_id (-1: no entry wins)
    float lp;   // most_likely_*_log_prob (-inf then)
    bool any;   // the map has entries at all
};
// update_object_cache / update_class_cache, :601-645: the first entry in KEY order whose log-probability is greater than every one before
// it, starting from -inf (an entry at -inf or NaN never wins) = the largest log-probability, the smallest id among equals
__host__ __device__ inline HvProb2Best prob2_argmax(const HvProbVoxel *v, const HvProbNode *nodes, int n, int which) {
    HvProb2Best b{-1, -INFINITY, false};
    bool won = false;
    for (int i = 0; i < n; ++i) {
        const HvProbPair p = prob_get(v, nodes, i);
        if (p.cls != which) continue;
        b.any = true;
        if (p.logp > b.lp) {
            b.id = p.obj;
            b.lp = p.logp;
            won = true;
        } else if (won && p.logp == b.lp && p.obj < b.id) {
            b.id = p.obj;
        }
    }
    return b;
}
// the cached most likely entry of map `which`, or the argmax when the cache is not valid
__host__ __device__ inline HvProb2Best prob2_most_likely(const HvProbVoxel *v, const HvProbNode *nodes, int which) {
    const int n = prob_nlab(v->meta);
    const int c = which == 0 ? prob2_best_obj(v->meta) : prob2_best_cls(v->meta);
    if (c >= 0) {
        const HvProbPair p = prob_get(v, nodes, c);
        return {p.obj, p.logp, true};
    }
    return prob2_argmax(v, nodes, n, which);
}
// get_object_log_normalization / get_class_log_normalization, :647-691: max, then sum of exp(lp - max) in key order, max + log(sum)
__host__ __device__ inline float prob2_log_normalization(const HvProbVoxel *v, const HvProbNode *nodes, int n, int which) {
    float mx = -INFINITY;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const HvProbPair p = prob_get(v, nodes, i);
        if (p.cls != which) continue;
        ++m;
        if (p.logp > mx) mx = p.logp;
    }
    if (m == 0) return 0.0f;
    float sum = 0.0f;
    int32_t last = 0;
    for (int step = 0; step < m; ++step) { // entries of this map in ascending id order
        int pick = -1;
        int32_t pid = 0;
        float plp = 0.f;
        for (int i = 0; i < n; ++i) {
            const HvProbPair p = prob_get(v, nodes, i);
            if (p.cls != which || (step > 0 && p.obj <= last)) continue;
            if (pick < 0 || p.obj < pid) {
                pick = i;
                pid = p.obj;
                plp = p.logp;
            }
        }
        if (pick < 0) break;
        sum += hv_expf_cr(plp - mx);
        last = pid;
    }
    return mx + hv_logf_cr(sum);
}
__host__ __device__ inline int32_t sem_object_id(const HvProb2Voxel *v, const void *nodes) { return prob2_most_likely(v, (const HvProbNode *)nodes, 0).id; }
__host__ __device__ inline int32_t sem_class_id(const HvProb2Voxel *v, const void *nodes) { return prob2_most_likely(v, (const HvProbNode *)nodes, 1).id; }
// compute_confidence(), :579-598: the normalised joint probability of (most likely object, most likely class)
__host__ __device__ inline float sem_confidence(const HvProb2Voxel *v, const void *nodes_) {
    const HvProbNode *nodes = (const HvProbNode *)nodes_;
    const int n = prob_nlab(v->meta);
    const HvProb2Best o = prob2_most_likely(v, nodes, 0), c = prob2_most_likely(v, nodes, 1);
    if (o.id == -1 || c.id == -1) return 0.0f;
    if (!o.any || !c.any) return 0.0f;
    const float joint = o.lp + c.lp;
    const float norm = prob2_log_normalization(v, nodes, n, 0) + prob2_log_normalization(v, nodes, n, 1);
    return hv_expf_cr(joint - norm);
}
// get_object_confidence / get_class_confidence, :528-560: the marginal probability of the most likely id
__host__ __device__ inline float prob2_marginal_confidence(const HvProb2Voxel *v, const void *nodes_, int which) {
    const HvProbNode *nodes = (const HvProbNode *)nodes_;
    const HvProb2Best b = prob2_most_likely(v, nodes, which);
    if (!b.any || b.id == -1) return 0.0f;
    return hv_expf_cr(b.lp - prob2_log_normalization(v, nodes, prob_nlab(v->meta), which));
}
__host__ __device__ inline float sem_object_confidence(const HvProb2Voxel *v, const void *nodes) { return prob2_marginal_confidence(v, nodes, 0); }
__host__ __device__ inline float sem_class_confidence(const HvProb2Voxel *v, const void *nodes) { return prob2_marginal_confidence(v, nodes, 1); }
// (the two payloads without marginals answer -1: hv_dump_marginals_semantic)
__host__ __device__ inline float sem_object_confidence(const HvSemVoxel *, const void * = nullptr) { return -1.0f; }
__host__ __device__ inline float sem_class_confidence(const HvSemVoxel *, const void * = nullptr) { return -1.0f; }
__host__ __device__ inline float sem_object_confidence(const HvProbVoxel *, const void * = nullptr) { return -1.0f; }
__host__ __device__ inline float sem_class_confidence(const HvProbVoxel *, const void * = nullptr) { return -1.0f; }
__host__ __device__ inline bool sem_confidence_not_negative(const HvProb2Voxel *v, const void *nodes_) {
    const HvProbNode *nodes = (const HvProbNode *)nodes_;
    const int n = prob_nlab(v->meta);
    bool finite = true;
    for (int i = 0; i < n; ++i) {
        const float lp = prob_get(v, nodes, i).logp;
        finite = finite && (lp - lp == 0.0f);
    }
    return finite ? true : sem_confidence(v, nodes_) >= 0.0f; // (all finite: exp(finite) or the 0 of an unlabelled voxel)
}
// get_confidence_counter(), :506-511
__host__ __device__ inline int32_t sem_confidence_counter(const HvProb2Voxel *v, const void *nodes) {
    return (int32_t)(sem_confidence(v, nodes) * (float)v->count);
}

#ifdef __HIPCC__
// entry `n` (the next free index) of a label list gets (obj, cls, lp): inline, or in the chain's node (n - K) / NK, which may have to be
// linked in first (a chain left behind by a reset or collapsed map is taken up again).  false: 254 entries, or the node pool is exhausted.
// (the node pool comes as three scalars, not as `const HvTable &`: where one of these functions is not inlined, the address of the
// kernel's by-value table would escape into the call and the whole struct would be spilled to scratch memory at the kernel's entry)
struct HvNodePool {
    HvProbNode *nodes;
    int32_t *counters;
    int32_t cap;
};
__device__ __forceinline__ HvNodePool hv_node_pool(const HvTable &table) { return {(HvProbNode *)table.prob_nodes, table.counters, table.prob_node_cap}; }
template <bool CHAIN>
__device__ inline bool prob_append_t(HvProbVoxel *v, HvNodePool pool, int n, int32_t obj, int32_t cls, float lp) {
    HvProbNode *nodes = pool.nodes;
    if (n >= HV_PROB_MAX) return false;
    if (!CHAIN || n < HV_PROB_K) {
        prob_slot_set(v, n, obj, cls, lp);
        return true;
    }
    if ((n - HV_PROB_K) % HV_PROB_NK == 0) {
        const int hops = (n - HV_PROB_K) / HV_PROB_NK;
        HvProbNode *tail = nullptr;
        if (hops > 0) {
            uint32_t k = v->next;
            for (int hop = hops - 1; hop > 0; --hop) k = nodes[k - 1].next;
            tail = &nodes[k - 1];
        }
        const uint32_t own_next = v->next;
        if ((tail ? tail->next : own_next) == 0u) {
            if (nodes == nullptr) return false;
            const int32_t id = atomicAdd(&pool.counters[HV_CNT_PROB_NODES], 1);
            if (id >= pool.cap) {
                atomicSub(&pool.counters[HV_CNT_PROB_NODES], 1);
                return false;
            }
            nodes[id].next = 0u;
            if (tail) tail->next = (uint32_t)id + 1u;
            v->next = tail ? own_next : (uint32_t)id + 1u;
        }
    }
    int j = n;
    HvProbNode *nd = (HvProbNode *)prob_node_of(v, nodes, j);
    nd->obj[j] = obj;
    nd->cls[j] = cls;
    nd->logp[j] = lp;
    return true;
}
// entry (id, which) of the list, -1 if there is none
template <bool CHAIN>
__device__ __forceinline__ int prob2_find(const HvProbVoxel *v, const HvProbNode *nodes, int n, int32_t id, int32_t which) {
    int idx = -1;
#pragma unroll
    for (int i = 0; i < HV_PROB_K; ++i) {
        const bool hit = (int)(v->obj[i] == id) & (int)(v->cls[i] == which) & (int)(i < n);
        idx = hit ? i : idx;
    }
    if (CHAIN)
        for (int i = HV_PROB_K; i < n; ++i) {
            int j = i;
            const HvProbNode *nd = prob_node_of(v, nodes, j);
            if (nd->obj[j] == id && nd->cls[j] == which) idx = i;
        }
    return idx;
}
// initialize_semantics_log_prob (first: the entries are ASSIGNED half the log-probability and become the cached most likely ones,
// :311-326) / update_semantics_log_prob (found: += half, new: = half; both caches dropped, :339-369).  Returns the number of entries that
// could not be stored (0-2).
template <bool CHAIN>
__device__ inline int prob2_fold_t(HvProbVoxel *v, HvNodePool table, bool first, int32_t obj, int32_t cls, float lp) {
    HvProbNode *nodes = table.nodes;
    int n = prob_nlab(v->meta);
    const float half = lp * 0.5f;
    int lost = 0;
    int io = prob2_find<CHAIN>(v, nodes, n, obj, 0);
    if (io < 0) {
        if (prob_append_t<CHAIN>(v, table, n, obj, 0, half)) io = n++;
        else ++lost;
    } else {
        prob_set_logp_r<CHAIN>(v, nodes, io, first ? half : prob_get_r<CHAIN>(v, nodes, io).logp + half);
    }
    int ic = prob2_find<CHAIN>(v, nodes, n, cls, 1);
    if (ic < 0) {
        if (prob_append_t<CHAIN>(v, table, n, cls, 1, half)) ic = n++;
        else ++lost;
    } else {
        prob_set_logp_r<CHAIN>(v, nodes, ic, first ? half : prob_get_r<CHAIN>(v, nodes, ic).logp + half);
    }
    v->meta = first ? prob2_meta(n, io, ic) : prob2_meta(n, -1, -1);
    return lost;
}
__device__ __forceinline__ int prob2_fold(HvProbVoxel *v, const HvTable &table, bool first, int32_t obj, int32_t cls, float lp) {
    const HvNodePool pool = hv_node_pool(table);
    return prob_nlab(v->meta) + 2 <= HV_PROB_K ? prob2_fold_t<false>(v, pool, first, obj, cls, lp) : prob2_fold_t<true>(v, pool, first, obj, cls, lp);
}
// set_object_id(), :424-452 (merge_segments, the association's deferred assignment): the id's entry takes the log-probability of the
// current most likely object (0 when there is none) and becomes the cached most likely one; the class cache is recomputed.
__device__ inline void prob2_set_object_id(HvProb2Voxel *v, HvNodePool table, int32_t id) {
    HvProbNode *nodes = table.nodes;
    int n = prob_nlab(v->meta);
    const HvProb2Best b = prob2_most_likely(v, nodes, 0);
    const float target = (b.any && b.id != -1 && b.lp != -INFINITY) ? b.lp : 0.0f;
    int io = prob2_find<true>(v, nodes, n, id, 0);
    if (io < 0) {
        if (prob_append_t<true>(v, table, n, id, 0, target)) {
            io = n++;
        } else {
            atomicAdd(&table.counters[HV_CNT_LABEL_OVERFLOW], 1);
        }
    } else {
        prob_set_logp<true>(v, nodes, io, target);
    }
    v->meta = prob2_meta(n, io, -1);
}
__device__ __forceinline__ void sem_set_object_id(HvProb2Voxel *v, const HvTable &table, int32_t id) { prob2_set_object_id(v, hv_node_pool(table), id); }
#endif

// log evidence of one observation: update_semantics / update_semantics_with_depth,
// voxel_data_semantic.h:419-447
__host__ __device__ inline float prob_observation_log_prob(bool has_depth, float depth, const HvSemParams &G) {
    if (!has_depth || depth <= G.depth_threshold) return HV_BASE_LOG_PROB;
    const float confidence = hv_expf_cr(-(depth - G.depth_threshold) * G.depth_decay_rate);
    return confidence * HV_BASE_LOG_PROB;
}
// the "*2" probabilistic payload: update_semantics (confidence 1: log 1 = 0) / update_semantics_with_depth, voxel_data_semantic2.h:371-388
// (no BASE_LOG_PROB_PER_OBSERVATION there)
__host__ __device__ inline float prob2_observation_log_prob(bool has_depth, float depth, const HvSemParams &G) {
    if (!has_depth) return hv_logf_cr(1.0f);
    return depth <= G.depth_threshold ? 0.0f : -(depth - G.depth_threshold) * G.depth_decay_rate;
}

// which payload a voxel type is (the kernels are templates over the four; `maps`: the record is HvProbVoxel's, with an overflow chain)
template <typename VOX> struct HvPay;
template <> struct HvPay<HvSemVoxel> { static constexpr int kind = 0; static constexpr bool maps = false; };
template <> struct HvPay<HvProbVoxel> { static constexpr int kind = 1; static constexpr bool maps = true; };
template <> struct HvPay<HvSem2Voxel> { static constexpr int kind = 2; static constexpr bool maps = false; };
template <> struct HvPay<HvProb2Voxel> { static constexpr int kind = 3; static constexpr bool maps = true; };

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
    return hv_mode_is_semantic(v->cfg.mode);
}
// the statement(s) given, with VOX = the voxel type of the volume's mode
#define HV_SEM_DISPATCH(v, ...)                                                                                                            \
    switch ((v)->cfg.mode) {                                                                                                               \
    case HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID: { using VOX = HvProbVoxel; __VA_ARGS__; } break;                                       \
    case HV_MODE_VOXEL_SEMANTIC_GRID2: { using VOX = HvSem2Voxel; __VA_ARGS__; } break;                                                    \
    case HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID2: { using VOX = HvProb2Voxel; __VA_ARGS__; } break;                                     \
    default: { using VOX = HvSemVoxel; __VA_ARGS__; } break;                                                                               \
    }
