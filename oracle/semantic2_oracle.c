/*
 * TEST INFRASTRUCTURE ONLY — CPU restatement of pySLAM's *full* semantic block grids (never linked into
 * pyslam_amd/):
 *
 *   kind 0  VoxelBlockSemanticGrid              = VoxelBlockSemanticGridT<VoxelSemanticData>
 *   kind 1  VoxelBlockSemanticProbabilisticGrid = VoxelBlockSemanticGridT<VoxelSemanticDataProbabilistic>
 *   kind 2  VoxelBlockSemanticGrid2              = VoxelBlockSemanticGridT<VoxelSemanticData2>
 *   kind 3  VoxelBlockSemanticProbabilisticGrid2 = VoxelBlockSemanticGridT<VoxelSemanticDataProbabilistic2>
 *   (cpp/volumetric/voxel_block_semantic_grid.h:57-123)
 *
 * Restated, function by function:
 *   payloads                                voxel_data_semantic.h:106-202 (voting), :249-672 (log-probability);
 *                                           voxel_data_semantic2.h:46-196 (two counters), :256-787 (marginal label maps)
 *   integrate_raw -> update_voxel_direct    voxel_block_grid.hpp:221-287, 466-497, 524-614 (sequential branch)
 *   get_voxels                              voxel_block_grid.hpp:785-817
 *   iterate_voxels_in_camera_frustrum       voxel_block_grid.hpp:1335-1349, 1445-1537; CameraFrustrum::contains
 *                                           camera_frustrum.cpp:175-196, bbox :209-264
 *   carve                                   voxel_grid_carving.h:47-79
 *   assign_object_ids_to_instance_ids       voxel_semantic_data_association.h:70-373
 *   remap_instance_ids                      image_utils.h:69-163
 *   get_object_segments + PCA box           voxel_block_semantic_grid.hpp:217-267, bounding_boxes_3d.cpp:373-553
 *   merge/remove segment ops, get_ids       voxel_block_semantic_grid.hpp:119-213
 *
 * Parity: PINNED against the compiled reference (oracle/_ref, ref_sem2_*) in tests/test_semantic2_oracle.py and
 * against the golden fixture tests/golden/semantic_flow.npz generated from it (tools/make_golden.py).
 * Known, documented freedoms (the reference's results depend on its std::unordered_map iteration order there):
 *   - new object ids handed out inside one assign call may be permuted between the instances that need one;
 *   - the class id reported for an object = class of the first voxel met for it;
 *   - eigenvector signs of the PCA box (Eigen's solver vs the Jacobi sweep used here).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define S2_BASE_LOG_PROB 0.10536051565782628f /* BASE_LOG_PROB_PER_OBSERVATION, voxel_data_semantic.h:287 */

typedef struct {
    int32_t obj, cls;
    float logp;
} s2_label;

typedef struct {
    int32_t count;
    double position_sum[3];
    float color_sum[3];
    /* voting payload */
    int32_t object_id, class_id, confidence_counter;
    /* probabilistic payload: std::map<(obj,cls), float> kept sorted by key + the lazily cached arg-max */
    s2_label *labels;
    int32_t n_labels, cap_labels;
    int32_t ml_obj, ml_cls; /* most_likely_pair */
    float ml_logp;          /* most_likely_log_prob */
    int cache_valid;
    /* kind 2 (VoxelSemanticData2): confidence_counter above is object_confidence_counter_, this is class_confidence_counter_ */
    int32_t class_counter;
    /* kind 3 (VoxelSemanticDataProbabilistic2): `labels` holds BOTH maps, keyed (which map, id) - obj field = 0 for
     * object_log_probabilities / 1 for class_log_probabilities, cls field = the id - so that each map is contiguous and in id
     * order; the two caches of voxel_data_semantic2.h:281-287 */
    int32_t ml2_id[2];
    float ml2_logp[2];
    int ml2_valid[2];
} s2_voxel;

typedef struct {
    int32_t key[3];
    s2_voxel *data;
} s2_block;

typedef struct {
    int kind;
    float voxel_size, inv_voxel_size;
    int block_size, voxels_per_block;
    s2_block *blocks;
    int64_t num_blocks, cap_blocks;
    int64_t *table;
    int64_t table_size;
} s2_grid;

/* the reference keeps these as process-wide statics of the payload types */
static float s2_vote_depth_threshold = 10.0f; /* voxel_data_semantic.h:107-108 */
static float s2_prob_depth_threshold = 5.0f;  /* :251-252 */
static float s2_prob_depth_decay = 0.07f;     /* :253-254 */
static float s2_vote2_depth_threshold = 10.0f; /* voxel_data_semantic2.h:47-48 */
static float s2_prob2_depth_threshold = 5.0f;  /* :258-259 */
static float s2_prob2_depth_decay = 0.07f;     /* :260-261 */
static int32_t s2_next_object_id = 1;         /* voxel_semantic_shared_data.h:26-34 */

int32_t so2_peek_next_object_id(void) { return s2_next_object_id; }
void so2_set_next_object_id(int32_t v) { s2_next_object_id = v; }

/* ---- payload: probabilistic ------------------------------------------------------------------------ */
static int s2_key_less(int32_t o1, int32_t c1, int32_t o2, int32_t c2) { return o1 < o2 || (o1 == o2 && c1 < c2); }

static float s2_log_add_exp(float a, float b) { /* :639-648 */
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float m = (a < b) ? b : a;
    return m + logf(expf(a - m) + expf(b - m));
}
static float s2_log_normalization(const s2_voxel *v) { /* :620-637, map iteration = key order */
    if (v->n_labels == 0) return 0.0f;
    float acc = -INFINITY;
    for (int i = 0; i < v->n_labels; ++i) acc = s2_log_add_exp(acc, v->labels[i].logp);
    return acc;
}
static void s2_update_cache(s2_voxel *v) { /* :575-601 */
    if (v->n_labels == 0) {
        v->ml_obj = v->ml_cls = -1;
        v->ml_logp = -INFINITY;
        v->cache_valid = 1;
        return;
    }
    v->ml_logp = -INFINITY;
    for (int i = 0; i < v->n_labels; ++i) {
        if (v->labels[i].logp > v->ml_logp) {
            v->ml_logp = v->labels[i].logp;
            v->ml_obj = v->labels[i].obj;
            v->ml_cls = v->labels[i].cls;
        }
    }
    v->cache_valid = 1;
}
static int s2_find(const s2_voxel *v, int32_t obj, int32_t cls) {
    for (int i = 0; i < v->n_labels; ++i)
        if (v->labels[i].obj == obj && v->labels[i].cls == cls) return i;
    return -1;
}
static int s2_insert(s2_voxel *v, int32_t obj, int32_t cls, float logp) { /* sorted insert */
    if (v->n_labels == v->cap_labels) {
        v->cap_labels = v->cap_labels ? v->cap_labels * 2 : 4;
        v->labels = (s2_label *)realloc(v->labels, sizeof(s2_label) * (size_t)v->cap_labels);
    }
    int at = 0;
    while (at < v->n_labels && s2_key_less(v->labels[at].obj, v->labels[at].cls, obj, cls)) ++at;
    memmove(v->labels + at + 1, v->labels + at, sizeof(s2_label) * (size_t)(v->n_labels - at));
    v->labels[at].obj = obj;
    v->labels[at].cls = cls;
    v->labels[at].logp = logp;
    v->n_labels++;
    return at;
}
static void s2_prob_initialize(s2_voxel *v, int32_t obj, int32_t cls, float lp) { /* :311-324 */
    const int i = s2_find(v, obj, cls);
    if (i >= 0) v->labels[i].logp = lp;
    else s2_insert(v, obj, cls, lp);
    v->ml_obj = obj;
    v->ml_cls = cls;
    v->ml_logp = lp;
    v->cache_valid = 1;
}
static void s2_prob_update(s2_voxel *v, int32_t obj, int32_t cls, float lp) { /* :358-417 */
    const int i = s2_find(v, obj, cls);
    if (i < 0) {
        s2_insert(v, obj, cls, lp);
        if (v->cache_valid && lp > v->ml_logp) {
            v->ml_logp = lp;
            v->ml_obj = obj;
            v->ml_cls = cls;
        } else if (!v->cache_valid) {
            s2_update_cache(v);
        }
    } else {
        v->labels[i].logp += lp;
        const float now = v->labels[i].logp;
        if (v->cache_valid) {
            if (obj == v->ml_obj && cls == v->ml_cls) {
                const float old = v->ml_logp;
                v->ml_logp = now;
                if (now < old) s2_update_cache(v);
            } else if (now > v->ml_logp) {
                v->ml_logp = now;
                v->ml_obj = obj;
                v->ml_cls = cls;
            }
        } else {
            s2_update_cache(v);
        }
    }
}
static float s2_observation_log_prob(int has_depth, float depth) { /* :419-447 */
    if (!has_depth || depth <= s2_prob_depth_threshold) return S2_BASE_LOG_PROB;
    const float confidence = expf(-(depth - s2_prob_depth_threshold) * s2_prob_depth_decay);
    return confidence * S2_BASE_LOG_PROB;
}

/* ---- payload: marginal label maps (kind 3, voxel_data_semantic2.h:256-787) --------------------------------- */
static void s2_m_update_cache(s2_voxel *v, int which) { /* update_object_cache / update_class_cache, :601-645 */
    v->ml2_id[which] = -1;
    v->ml2_logp[which] = -INFINITY;
    for (int i = 0; i < v->n_labels; ++i) /* (this map's entries in id order) */
        if (v->labels[i].obj == which && v->labels[i].logp > v->ml2_logp[which]) {
            v->ml2_logp[which] = v->labels[i].logp;
            v->ml2_id[which] = v->labels[i].cls;
        }
    v->ml2_valid[which] = 1;
}
static int s2_m_size(const s2_voxel *v, int which) {
    int n = 0;
    for (int i = 0; i < v->n_labels; ++i) n += v->labels[i].obj == which;
    return n;
}
static float s2_m_log_normalization(const s2_voxel *v, int which) { /* :647-691: two passes, sum in id order */
    if (s2_m_size(v, which) == 0) return 0.0f;
    float mx = -INFINITY;
    for (int i = 0; i < v->n_labels; ++i)
        if (v->labels[i].obj == which && v->labels[i].logp > mx) mx = v->labels[i].logp;
    float sum = 0.0f;
    for (int i = 0; i < v->n_labels; ++i)
        if (v->labels[i].obj == which) sum += expf(v->labels[i].logp - mx);
    return mx + logf(sum);
}
static void s2_m_add(s2_voxel *v, int which, int32_t id, float half, int assign) {
    const int i = s2_find(v, which, id);
    if (i < 0) s2_insert(v, which, id, half);
    else v->labels[i].logp = assign ? half : v->labels[i].logp + half;
}
static void s2_m_initialize(s2_voxel *v, int32_t obj, int32_t cls, float lp) { /* initialize_semantics_log_prob, :311-326 */
    const float half = lp * 0.5f;
    s2_m_add(v, 0, obj, half, 1);
    s2_m_add(v, 1, cls, half, 1);
    v->ml2_id[0] = obj;
    v->ml2_id[1] = cls;
    v->ml2_logp[0] = v->ml2_logp[1] = half;
    v->ml2_valid[0] = v->ml2_valid[1] = 1;
}
static void s2_m_update(s2_voxel *v, int32_t obj, int32_t cls, float lp) { /* update_semantics_log_prob, :339-369 */
    const float half = lp * 0.5f;
    s2_m_add(v, 0, obj, half, 0);
    s2_m_add(v, 1, cls, half, 0);
    v->ml2_valid[0] = v->ml2_valid[1] = 0;
}
static float s2_m_observation_log_prob(int has_depth, float depth) { /* :371-388 (no base log-probability here) */
    if (!has_depth) return logf(1.0f);
    return depth <= s2_prob2_depth_threshold ? 0.0f : -(depth - s2_prob2_depth_threshold) * s2_prob2_depth_decay;
}
static float s2_m_confidence(s2_voxel *v) { /* ensure_cache_updated + compute_confidence, :562-598 */
    for (int w = 0; w < 2; ++w)
        if (!v->ml2_valid[w]) s2_m_update_cache(v, w);
    if (v->ml2_id[0] == -1 || v->ml2_id[1] == -1) return 0.0f;
    if (s2_m_size(v, 0) == 0 || s2_m_size(v, 1) == 0) return 0.0f;
    const float joint = v->ml2_logp[0] + v->ml2_logp[1];
    return expf(joint - (s2_m_log_normalization(v, 0) + s2_m_log_normalization(v, 1)));
}
static float s2_min1(float r) { return r < 1.0f ? r : 1.0f; } /* std::min(1.0f, r) */

/* ---- payload-independent accessors ----------------------------------------------------------------- */
static int32_t s2_object_id(const s2_grid *g, s2_voxel *v) {
    if (g->kind == 0 || g->kind == 2) return v->object_id;
    if (g->kind == 3) {
        if (!v->ml2_valid[0]) s2_m_update_cache(v, 0);
        return v->ml2_id[0];
    }
    if (!v->cache_valid) s2_update_cache(v);
    return v->ml_obj;
}
static int32_t s2_class_id(const s2_grid *g, s2_voxel *v) {
    if (g->kind == 0 || g->kind == 2) return v->class_id;
    if (g->kind == 3) {
        if (!v->ml2_valid[1]) s2_m_update_cache(v, 1);
        return v->ml2_id[1];
    }
    if (!v->cache_valid) s2_update_cache(v);
    return v->ml_cls;
}
static float s2_confidence(const s2_grid *g, s2_voxel *v) {
    if (g->kind == 0) { /* :116-133 */
        if (v->count == 0) return 0.0f;
        const float r = (float)v->confidence_counter / (float)v->count;
        return r < 1.0f ? r : 1.0f;
    }
    if (g->kind == 2) { /* voxel_data_semantic2.h:60-85: the smaller of the two clamped counter / count ratios */
        if (v->count == 0) return 0.0f;
        const float o = s2_min1((float)v->confidence_counter / (float)v->count), c = s2_min1((float)v->class_counter / (float)v->count);
        return c < o ? c : o;
    }
    if (g->kind == 3) return s2_m_confidence(v);
    if (!v->cache_valid) s2_update_cache(v);
    if (v->ml_obj == -1 || v->ml_cls == -1) return 0.0f; /* compute_confidence, :562-572 */
    if (v->n_labels == 0) return 0.0f;
    return expf(v->ml_logp - s2_log_normalization(v));
}
static int32_t s2_confidence_counter(const s2_grid *g, s2_voxel *v) {
    if (g->kind == 0) return v->confidence_counter;
    if (g->kind == 2) return v->class_counter < v->confidence_counter ? v->class_counter : v->confidence_counter; /* semantic2.h:55-57 */
    return (int32_t)(s2_confidence(g, v) * (float)v->count); /* :505-511; voxel_data_semantic2.h:506-511 */
}
static void s2_set_object_id(const s2_grid *g, s2_voxel *v, int32_t id) {
    if (g->kind == 0 || g->kind == 2) {
        v->object_id = id;
        return;
    }
    if (g->kind == 3) { /* voxel_data_semantic2.h:424-452 */
        if (!v->ml2_valid[0]) s2_m_update_cache(v, 0);
        const float target = (s2_m_size(v, 0) > 0 && v->ml2_id[0] != -1 && v->ml2_logp[0] != -INFINITY) ? v->ml2_logp[0] : 0.0f;
        s2_m_add(v, 0, id, target, 1);
        v->ml2_id[0] = id;
        v->ml2_logp[0] = target;
        v->ml2_valid[0] = 1;
        v->ml2_valid[1] = 0;
        (void)s2_m_confidence(v); /* ensure_cache_updated */
        return;
    }
    /* set_object_id -> force_label_distribution, :476-481, 603-618 */
    if (!v->cache_valid) s2_update_cache(v);
    v->ml_obj = id;
    v->n_labels = 0;
    if (v->ml_obj >= 0 && v->ml_cls >= 0) {
        s2_insert(v, v->ml_obj, v->ml_cls, 0.0f);
        v->ml_logp = 0.0f;
        v->cache_valid = 1;
    } else {
        v->ml_logp = -INFINITY;
        v->cache_valid = 0;
    }
}
static void s2_reset(s2_voxel *v) {
    s2_label *keep = v->labels;
    const int32_t cap = v->cap_labels;
    memset(v, 0, sizeof(*v));
    v->labels = keep;
    v->cap_labels = cap;
    v->object_id = v->class_id = -1;
    v->ml_obj = v->ml_cls = -1;
    v->ml_logp = -INFINITY;
    v->ml2_id[0] = v->ml2_id[1] = -1;
    v->ml2_logp[0] = v->ml2_logp[1] = -INFINITY;
}

/* ---- container -------------------------------------------------------------------------------------- */
static uint64_t s2_mix(int32_t x, int32_t y, int32_t z) {
    uint64_t h = (uint64_t)(uint32_t)x * 0x9E3779B97F4A7C15ull;
    h ^= (uint64_t)(uint32_t)y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    h ^= (uint64_t)(uint32_t)z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
    return h ^ (h >> 29);
}
static void s2_rebuild(s2_grid *g, int64_t n) {
    free(g->table);
    g->table_size = n;
    g->table = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) g->table[i] = -1;
    for (int64_t b = 0; b < g->num_blocks; ++b) {
        const int32_t *k = g->blocks[b].key;
        uint64_t s = s2_mix(k[0], k[1], k[2]) & (uint64_t)(n - 1);
        while (g->table[s] >= 0) s = (s + 1) & (uint64_t)(n - 1);
        g->table[s] = b;
    }
}
s2_grid *so2_create(int kind, double voxel_size, int block_size) {
    s2_grid *g = (s2_grid *)calloc(1, sizeof(s2_grid));
    g->kind = kind;
    g->voxel_size = (float)voxel_size; /* voxel_block_grid.hpp:4-9 */
    g->inv_voxel_size = 1.0f / g->voxel_size;
    g->block_size = block_size;
    g->voxels_per_block = block_size * block_size * block_size;
    s2_rebuild(g, 1024);
    return g;
}
void so2_clear(s2_grid *g) {
    for (int64_t b = 0; b < g->num_blocks; ++b) {
        for (int i = 0; i < g->voxels_per_block; ++i) free(g->blocks[b].data[i].labels);
        free(g->blocks[b].data);
    }
    g->num_blocks = 0;
    s2_rebuild(g, 1024);
}
void so2_destroy(s2_grid *g) {
    if (!g) return;
    so2_clear(g);
    free(g->blocks);
    free(g->table);
    free(g);
}
int64_t so2_num_blocks(const s2_grid *g) { return g->num_blocks; }
/* Test hook: hist[k] = occupied voxels whose label map holds k pairs (k >= cap - 1 collected in the last bin); returns the
 * largest map size.  What the product's 6 inline label slots + overflow nodes (hv_semantic.h) are sized against. */
int32_t so2_label_histogram(const s2_grid *g, int64_t *hist, int32_t cap) {
    int32_t most = 0;
    for (int32_t k = 0; k < cap; ++k) hist[k] = 0;
    for (int64_t b = 0; b < g->num_blocks; ++b)
        for (int i = 0; i < g->voxels_per_block; ++i) {
            const s2_voxel *v = &g->blocks[b].data[i];
            if (v->count <= 0) continue;
            const int32_t n = v->n_labels;
            if (n > most) most = n;
            hist[n < cap - 1 ? n : cap - 1] += 1;
        }
    return most;
}
void so2_set_depth_threshold(s2_grid *g, float t) {
    if (g->kind == 0) s2_vote_depth_threshold = t;
    else if (g->kind == 1) s2_prob_depth_threshold = t;
    else if (g->kind == 2) s2_vote2_depth_threshold = t;
    else s2_prob2_depth_threshold = t;
}
void so2_set_depth_decay_rate(s2_grid *g, float r) {
    if (g->kind == 1) s2_prob_depth_decay = r; /* only the probabilistic payloads, voxel_block_semantic_grid.hpp:32-37 */
    else if (g->kind == 3) s2_prob2_depth_decay = r;
}

static s2_block *s2_find_or_create(s2_grid *g, int32_t bx, int32_t by, int32_t bz) {
    uint64_t s = s2_mix(bx, by, bz) & (uint64_t)(g->table_size - 1);
    while (g->table[s] >= 0) {
        s2_block *b = &g->blocks[g->table[s]];
        if (b->key[0] == bx && b->key[1] == by && b->key[2] == bz) return b;
        s = (s + 1) & (uint64_t)(g->table_size - 1);
    }
    if (g->num_blocks == g->cap_blocks) {
        g->cap_blocks = g->cap_blocks ? g->cap_blocks * 2 : 256;
        g->blocks = (s2_block *)realloc(g->blocks, sizeof(s2_block) * (size_t)g->cap_blocks);
    }
    s2_block *b = &g->blocks[g->num_blocks];
    b->key[0] = bx; b->key[1] = by; b->key[2] = bz;
    b->data = (s2_voxel *)calloc((size_t)g->voxels_per_block, sizeof(s2_voxel));
    for (int i = 0; i < g->voxels_per_block; ++i) s2_reset(&b->data[i]);
    g->table[s] = g->num_blocks++;
    if (g->num_blocks * 2 > g->table_size) {
        s2_rebuild(g, g->table_size * 2);
        return &g->blocks[g->num_blocks - 1];
    }
    return b;
}

static inline int64_t s2_floor_div(int64_t a, int64_t b) { return (a >= 0) ? (a / b) : ((a - b + 1) / b); }

/* integrate_raw -> update_voxel -> update_voxel_direct.  pos_kind 0 float32 / 1 float64 points; color_kind 1 uint8,
 * 2 float32; class_ids == NULL: no semantic update; instance_ids == NULL: object id 0; depths may be NULL. */
void so2_integrate(s2_grid *g, const void *pts, int pos_kind, int64_t n, const void *cols, int color_kind,
                   const int32_t *class_ids, const int32_t *instance_ids, const float *depths) {
    const int bs = g->block_size;
    const float inv_255 = 1.0f / 255.0f;
    for (int64_t i = 0; i < n; ++i) {
        double xyz[3];
        int32_t vk[3], bk[3], lk[3];
        for (int k = 0; k < 3; ++k) {
            if (pos_kind == 0) {
                const float x = ((const float *)pts)[i * 3 + k];
                xyz[k] = (double)x;
                vk[k] = (int32_t)floorf(x * g->inv_voxel_size);
            } else {
                const double x = ((const double *)pts)[i * 3 + k];
                xyz[k] = x;
                vk[k] = (int32_t)floor(x * (double)g->inv_voxel_size);
            }
            bk[k] = (int32_t)s2_floor_div(vk[k], bs);
            lk[k] = (int32_t)((int64_t)vk[k] - (int64_t)bk[k] * bs);
        }
        s2_block *blk = s2_find_or_create(g, bk[0], bk[1], bk[2]);
        s2_voxel *v = &blk->data[lk[0] + lk[1] * bs + lk[2] * bs * bs];
        for (int k = 0; k < 3; ++k) v->position_sum[k] += xyz[k];
        if (color_kind == 1) {
            const uint8_t *c = (const uint8_t *)cols + i * 3;
            for (int k = 0; k < 3; ++k) v->color_sum[k] += (float)c[k] * inv_255;
        } else if (color_kind == 2) {
            const float *c = (const float *)cols + i * 3;
            for (int k = 0; k < 3; ++k) v->color_sum[k] += c[k];
        }
        if (class_ids != NULL) {
            const int32_t obj = instance_ids ? instance_ids[i] : 0;
            const int32_t cls = class_ids[i];
            if (g->kind == 0) {
                const int gate = depths ? (depths[i] < s2_vote_depth_threshold) : 1; /* :168-198 */
                if (v->count == 0) {
                    if (gate) { v->object_id = obj; v->class_id = cls; v->confidence_counter = 1; }
                } else if (gate) { /* update_semantics, :175-191 */
                    if (v->object_id == obj && v->class_id == cls) {
                        v->confidence_counter++;
                    } else {
                        v->confidence_counter--;
                        if (v->confidence_counter <= 0) { v->object_id = obj; v->class_id = cls; v->confidence_counter = 1; }
                    }
                }
            } else if (g->kind == 2) { /* voxel_data_semantic2.h:120-195: the object id and the class id vote on their own */
                const int gate = depths ? (depths[i] < s2_vote2_depth_threshold) : 1;
                if (v->count == 0) {
                    if (gate) { v->object_id = obj; v->class_id = cls; v->confidence_counter = 1; v->class_counter = 1; }
                } else if (gate) {
                    if (v->object_id == obj) v->confidence_counter++;
                    else if (--v->confidence_counter <= 0) { v->object_id = obj; v->confidence_counter = 1; }
                    if (v->class_id == cls) v->class_counter++;
                    else if (--v->class_counter <= 0) { v->class_id = cls; v->class_counter = 1; }
                }
            } else if (g->kind == 3) {
                const float lp = s2_m_observation_log_prob(depths != NULL, depths ? depths[i] : 0.0f);
                if (v->count == 0) s2_m_initialize(v, obj, cls, lp);
                else s2_m_update(v, obj, cls, lp);
            } else {
                const float lp = s2_observation_log_prob(depths != NULL, depths ? depths[i] : 0.0f);
                if (v->count == 0) s2_prob_initialize(v, obj, cls, lp);
                else s2_prob_update(v, obj, cls, lp);
            }
        }
        v->count = (v->count == 0) ? 1 : v->count + 1;
    }
}

static int s2_cmp(const void *pa, const void *pb) {
    const s2_block *a = *(const s2_block *const *)pa, *b = *(const s2_block *const *)pb;
    for (int k = 0; k < 3; ++k)
        if (a->key[k] != b->key[k]) return a->key[k] < b->key[k] ? -1 : 1;
    return 0;
}

int64_t so2_dump(s2_grid *g, int32_t *keys, int32_t *ints, float *conf, double *pos_sums, float *col_sums) {
    const int64_t nb = g->num_blocks;
    s2_block **order = (s2_block **)malloc(sizeof(void *) * (size_t)(nb ? nb : 1));
    for (int64_t b = 0; b < nb; ++b) order[b] = &g->blocks[b];
    qsort(order, (size_t)nb, sizeof(void *), s2_cmp);
    const int64_t nv = g->voxels_per_block;
    for (int64_t b = 0; b < nb; ++b) {
        if (keys) memcpy(keys + b * 3, order[b]->key, 12);
        for (int64_t i = 0; i < nv; ++i) {
            s2_voxel *v = &order[b]->data[i];
            if (ints) {
                int32_t *d = ints + (b * nv + i) * 4;
                d[0] = v->count; d[1] = s2_object_id(g, v); d[2] = s2_class_id(g, v); d[3] = s2_confidence_counter(g, v);
            }
            if (conf) conf[b * nv + i] = s2_confidence(g, v);
            if (pos_sums) memcpy(pos_sums + (b * nv + i) * 3, v->position_sum, 24);
            if (col_sums) memcpy(col_sums + (b * nv + i) * 3, v->color_sum, 12);
        }
    }
    free(order);
    return nb;
}

int64_t so2_get_voxels(s2_grid *g, int min_count, float min_confidence, double *pts, float *cols, int32_t *class_ids,
                       int32_t *object_ids, float *confidences, int64_t cap) {
    int64_t n = 0;
    for (int64_t b = 0; b < g->num_blocks; ++b)
        for (int i = 0; i < g->voxels_per_block; ++i) {
            s2_voxel *v = &g->blocks[b].data[i];
            const float conf = s2_confidence(g, v);
            if (v->count >= min_count && conf >= min_confidence) {
                if (pts && n < cap) {
                    for (int k = 0; k < 3; ++k) {
                        pts[n * 3 + k] = v->position_sum[k] / (double)v->count;
                        cols[n * 3 + k] = v->color_sum[k] / (float)v->count;
                    }
                    class_ids[n] = s2_class_id(g, v);
                    object_ids[n] = s2_object_id(g, v);
                    confidences[n] = conf;
                }
                ++n;
            }
        }
    return n;
}

int64_t so2_get_ids(s2_grid *g, int32_t *class_ids, int32_t *object_ids, int64_t cap) { /* :198-213 */
    int64_t n = 0;
    for (int64_t b = 0; b < g->num_blocks; ++b)
        for (int i = 0; i < g->voxels_per_block; ++i) {
            s2_voxel *v = &g->blocks[b].data[i];
            if (v->count > 0) {
                if (class_ids && n < cap) { class_ids[n] = s2_class_id(g, v); object_ids[n] = s2_object_id(g, v); }
                ++n;
            }
        }
    return n;
}

/* ---- frustum iteration ---------------------------------------------------------------------------- */
typedef struct {
    float fx, fy, cx, cy, depth_max, depth_min;
    int width, height;
    double R[9], t[3];
    int32_t vmin[3], vmax[3], bmin[3], bmax[3];
} s2_frustum;

static s2_frustum s2_make_frustum(const s2_grid *g, const float *intr, int width, int height, const double *T_cw,
                                  float depth_max, float depth_min) {
    s2_frustum f;
    f.fx = intr[0]; f.fy = intr[1]; f.cx = intr[2]; f.cy = intr[3];
    f.width = width; f.height = height;
    f.depth_max = depth_max; f.depth_min = depth_min;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) f.R[r * 3 + c] = T_cw[r * 4 + c];
        f.t[r] = T_cw[r * 4 + 3];
    }
    /* compute_frustum_corners_world_ + compute_bbox_, camera_frustrum.cpp:209-264 */
    double Rwc[9], twc[3], bb[6];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rwc[r * 3 + c] = f.R[c * 3 + r];
    for (int r = 0; r < 3; ++r) twc[r] = -(Rwc[r * 3 + 0] * f.t[0] + Rwc[r * 3 + 1] * f.t[1] + Rwc[r * 3 + 2] * f.t[2]);
    const double cu[4] = {0.0, (double)width, (double)width, 0.0};
    const double cv[4] = {0.0, 0.0, (double)height, (double)height};
    for (int k = 0; k < 3; ++k) { bb[k] = 1.7976931348623157e308; bb[3 + k] = -1.7976931348623157e308; }
    for (int i = 0; i < 4; ++i) {
        const double xn = (cu[i] - (double)f.cx) / (double)f.fx;
        const double yn = (cv[i] - (double)f.cy) / (double)f.fy;
        const double ds[2] = {(double)depth_min, (double)depth_max};
        for (int j = 0; j < 2; ++j) {
            const double pc[3] = {xn * ds[j], yn * ds[j], ds[j]};
            for (int r = 0; r < 3; ++r) {
                const double w = (Rwc[r * 3 + 0] * pc[0] + Rwc[r * 3 + 1] * pc[1] + Rwc[r * 3 + 2] * pc[2]) + twc[r];
                if (w < bb[r]) bb[r] = w;
                if (w > bb[3 + r]) bb[3 + r] = w;
            }
        }
    }
    for (int k = 0; k < 3; ++k) { /* voxel_block_grid.hpp:1340-1348 */
        f.vmin[k] = (int32_t)floor(bb[k] * (double)g->inv_voxel_size);
        f.vmax[k] = (int32_t)floor(bb[3 + k] * (double)g->inv_voxel_size);
        f.bmin[k] = (int32_t)s2_floor_div(f.vmin[k], g->block_size);
        f.bmax[k] = (int32_t)s2_floor_div(f.vmax[k], g->block_size);
    }
    return f;
}

/* visit predicate of iterate_voxels_in_camera_frustrum (min_count 1, min_confidence 0): fills uvd on success */
static int s2_visit(s2_grid *g, const s2_frustum *f, const s2_block *blk, int lx, int ly, int lz, s2_voxel *v, float *uvd) {
    const int bs = g->block_size;
    if (!(v->count >= 1 && s2_confidence(g, v) >= 0.0f)) return 0;
    const int32_t vk[3] = {blk->key[0] * bs + lx, blk->key[1] * bs + ly, blk->key[2] * bs + lz};
    for (int k = 0; k < 3; ++k)
        if (vk[k] < f->vmin[k] || vk[k] > f->vmax[k]) return 0;
    const double p[3] = {v->position_sum[0] / (double)v->count, v->position_sum[1] / (double)v->count,
                         v->position_sum[2] / (double)v->count};
    double pc[3];
    for (int r = 0; r < 3; ++r) pc[r] = (f->R[r * 3 + 0] * p[0] + f->R[r * 3 + 1] * p[1] + f->R[r * 3 + 2] * p[2]) + f->t[r];
    const float depth = (float)pc[2];
    if (!(depth >= f->depth_min && depth <= f->depth_max)) return 0;
    const float u = (float)((double)f->fx * (pc[0] / pc[2]) + (double)f->cx);
    const float vv = (float)((double)f->fy * (pc[1] / pc[2]) + (double)f->cy);
    uvd[0] = u; uvd[1] = vv; uvd[2] = depth;
    return u >= 0.0f && u < (float)f->width && vv >= 0.0f && vv < (float)f->height;
}

#define S2_FOR_VISITED(g, f, BODY)                                                                  \
    for (int64_t b_ = 0; b_ < (g)->num_blocks; ++b_) {                                              \
        s2_block *blk = &(g)->blocks[b_];                                                           \
        int skip_ = 0;                                                                              \
        for (int k_ = 0; k_ < 3; ++k_)                                                              \
            if (blk->key[k_] < (f).bmin[k_] || blk->key[k_] > (f).bmax[k_]) skip_ = 1;              \
        if (skip_) continue;                                                                        \
        for (int lx = 0; lx < (g)->block_size; ++lx)                                                \
            for (int ly = 0; ly < (g)->block_size; ++ly)                                            \
                for (int lz = 0; lz < (g)->block_size; ++lz) {                                      \
                    s2_voxel *v = &blk->data[lx + ly * (g)->block_size + lz * (g)->block_size * (g)->block_size]; \
                    float uvd[3];                                                                   \
                    if (!s2_visit((g), &(f), blk, lx, ly, lz, v, uvd)) continue;                    \
                    BODY                                                                            \
                }                                                                                   \
    }

void so2_carve(s2_grid *g, const float *intr, int width, int height, const double *T_cw, float depth_max, float depth_min,
               const float *depth, float depth_threshold) {
    const s2_frustum f = s2_make_frustum(g, intr, width, height, T_cw, depth_max, depth_min);
    S2_FOR_VISITED(g, f, {
        const float image_depth = depth[(int64_t)(int)uvd[1] * width + (int)uvd[0]];
        if (image_depth <= 0.0f || !isfinite(image_depth)) continue;
        if (uvd[2] < image_depth - depth_threshold) s2_reset(v);
    })
}

/* small ordered maps of the association (instance -> (object -> votes), instance -> new id, pending lists) */
typedef struct { int32_t inst, obj, count; } s2_vote;
typedef struct { int32_t inst; s2_voxel *v; } s2_pending;

static int s2_vote_cmp(const void *a, const void *b) {
    const s2_vote *x = (const s2_vote *)a, *y = (const s2_vote *)b;
    if (x->inst != y->inst) return x->inst < y->inst ? -1 : 1;
    if (x->obj != y->obj) return x->obj < y->obj ? -1 : 1;
    return 0;
}

int64_t so2_assign_object_ids(s2_grid *g, const float *intr, int width, int height, const double *T_cw, float depth_max,
                              float depth_min, const int32_t *class_img, const int32_t *inst_img, const float *depth,
                              float depth_threshold, int do_carving, float min_vote_ratio, int min_votes, int32_t *map_inst,
                              int32_t *map_obj, int64_t cap) {
    const s2_frustum f = s2_make_frustum(g, intr, width, height, T_cw, depth_max, depth_min);
    const int use_depth = depth != NULL;
    do_carving = do_carving && use_depth;
    s2_vote *votes = NULL;
    int64_t n_votes = 0, cap_votes = 0;
    s2_pending *pend = NULL;
    int64_t n_pend = 0, cap_pend = 0;
    int32_t *new_inst = NULL, *new_id = NULL; /* instance_id_to_new_object_id */
    int64_t n_new = 0, cap_new = 0;
    S2_FOR_VISITED(g, f, { /* process_point, :190-246 */
        const int64_t px = (int64_t)(int)uvd[1] * width + (int)uvd[0];
        const int32_t image_class = class_img[px];
        if (image_class < 0) continue;
        const int32_t point_class = s2_class_id(g, v);
        if (point_class < 0 || point_class != image_class) continue;
        const int32_t inst = inst_img[px];
        if (inst < 0) continue;
        int32_t obj = s2_object_id(g, v);
        if (use_depth) {
            const float image_depth = depth[px];
            if (image_depth <= 0.0f || !isfinite(image_depth)) continue;
            if (do_carving && uvd[2] < image_depth - depth_threshold) { s2_reset(v); continue; }
            if (uvd[2] > image_depth + depth_threshold) continue;
        }
        if (obj < 0) {
            if (inst == 0) {
                obj = 0;
                s2_set_object_id(g, v, 0);
            } else { /* assign_object_id, :151-188 */
                int64_t k = 0;
                while (k < n_new && new_inst[k] != inst) ++k;
                if (k == n_new) {
                    if (n_new == cap_new) {
                        cap_new = cap_new ? cap_new * 2 : 16;
                        new_inst = (int32_t *)realloc(new_inst, sizeof(int32_t) * (size_t)cap_new);
                        new_id = (int32_t *)realloc(new_id, sizeof(int32_t) * (size_t)cap_new);
                    }
                    new_inst[n_new] = inst;
                    new_id[n_new] = s2_next_object_id++;
                    ++n_new;
                }
                obj = new_id[k];
                if (n_pend == cap_pend) {
                    cap_pend = cap_pend ? cap_pend * 2 : 256;
                    pend = (s2_pending *)realloc(pend, sizeof(s2_pending) * (size_t)cap_pend);
                }
                pend[n_pend].inst = inst;
                pend[n_pend].v = v;
                ++n_pend;
            }
        }
        int64_t k = 0;
        while (k < n_votes && !(votes[k].inst == inst && votes[k].obj == obj)) ++k;
        if (k == n_votes) {
            if (n_votes == cap_votes) {
                cap_votes = cap_votes ? cap_votes * 2 : 64;
                votes = (s2_vote *)realloc(votes, sizeof(s2_vote) * (size_t)cap_votes);
            }
            votes[n_votes].inst = inst; votes[n_votes].obj = obj; votes[n_votes].count = 0;
            ++n_votes;
        }
        votes[k].count++;
    })
    /* voting, :268-311: per instance, objects in ascending id order (std::map), first strict maximum wins */
    qsort(votes, (size_t)n_votes, sizeof(s2_vote), s2_vote_cmp);
    int32_t *res_inst = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_votes + (int64_t)width * height / 16 + 16));
    int32_t *res_obj = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_votes + (int64_t)width * height / 16 + 16));
    int64_t n_res = 0, cap_res = n_votes + (int64_t)width * height / 16 + 16;
    for (int64_t i = 0; i < n_votes;) {
        int64_t j = i;
        int max_votes = 0, winning = -1, total = 0;
        while (j < n_votes && votes[j].inst == votes[i].inst) {
            total += votes[j].count;
            if (votes[j].count > max_votes) { max_votes = votes[j].count; winning = votes[j].obj; }
            ++j;
        }
        int32_t out = winning;
        if (total < min_votes) out = -1;
        else if ((float)max_votes / (float)total < min_vote_ratio) out = -1;
        res_inst[n_res] = votes[i].inst;
        res_obj[n_res] = out;
        ++n_res;
        i = j;
    }
    /* every instance id of the image gets a mapping, :313-341 */
    for (int64_t p = 0; p < (int64_t)width * height; ++p) {
        const int32_t inst = inst_img[p];
        if (inst < 0 || class_img[p] < 0) continue;
        int64_t k = 0;
        while (k < n_res && res_inst[k] != inst) ++k;
        if (inst == 0) {
            if (k == n_res) { res_inst[n_res] = 0; ++n_res; }
            res_obj[k] = 0;
        } else if (k == n_res) {
            if (n_res == cap_res) {
                cap_res *= 2;
                res_inst = (int32_t *)realloc(res_inst, sizeof(int32_t) * (size_t)cap_res);
                res_obj = (int32_t *)realloc(res_obj, sizeof(int32_t) * (size_t)cap_res);
            }
            res_inst[n_res] = inst;
            res_obj[n_res] = -1;
            ++n_res;
        }
    }
    /* deferred assignments, :344-361 */
    for (int64_t p = 0; p < n_pend; ++p) {
        int64_t k = 0;
        while (k < n_res && res_inst[k] != pend[p].inst) ++k;
        if (k < n_res && res_obj[k] >= 0) s2_set_object_id(g, pend[p].v, res_obj[k]);
    }
    /* output sorted by instance id */
    for (int64_t a = 1; a < n_res; ++a) { /* insertion sort: a handful of entries */
        const int32_t ki = res_inst[a], ko = res_obj[a];
        int64_t b = a - 1;
        while (b >= 0 && res_inst[b] > ki) { res_inst[b + 1] = res_inst[b]; res_obj[b + 1] = res_obj[b]; --b; }
        res_inst[b + 1] = ki; res_obj[b + 1] = ko;
    }
    for (int64_t a = 0; a < n_res && a < cap; ++a)
        if (map_inst) { map_inst[a] = res_inst[a]; map_obj[a] = res_obj[a]; }
    free(votes); free(pend); free(new_inst); free(new_id); free(res_inst); free(res_obj);
    return n_res;
}

void so2_remap_instance_ids(const int32_t *inst_img, int height, int width, const int32_t *map_inst, const int32_t *map_obj,
                            int64_t n_map, int32_t *out) {
    for (int64_t p = 0; p < (int64_t)height * width; ++p) {
        if (n_map == 0) { /* the binding hands the image back for an empty map, image_utils_module.h:52-58 */
            out[p] = inst_img[p];
            continue;
        }
        int32_t r = -1;
        for (int64_t k = 0; k < n_map; ++k)
            if (map_inst[k] == inst_img[p]) { r = map_obj[k]; break; }
        out[p] = r;
    }
}

/* ---- segment operations ---------------------------------------------------------------------------- */
void so2_merge_segments(s2_grid *g, int id1, int id2) {
    for (int64_t b = 0; b < g->num_blocks; ++b)
        for (int i = 0; i < g->voxels_per_block; ++i) {
            s2_voxel *v = &g->blocks[b].data[i];
            if (s2_object_id(g, v) == id2) s2_set_object_id(g, v, id1);
        }
}
void so2_remove_segment(s2_grid *g, int object_id) {
    for (int64_t b = 0; b < g->num_blocks; ++b)
        for (int i = 0; i < g->voxels_per_block; ++i) {
            s2_voxel *v = &g->blocks[b].data[i];
            if (s2_object_id(g, v) == object_id) s2_reset(v);
        }
}
void so2_remove_low_confidence_segments(s2_grid *g, int min_confidence) {
    for (int64_t b = 0; b < g->num_blocks; ++b)
        for (int i = 0; i < g->voxels_per_block; ++i) {
            s2_voxel *v = &g->blocks[b].data[i];
            if (s2_confidence(g, v) < (float)min_confidence) s2_reset(v);
        }
}

/* ---- PCA oriented bounding box, bounding_boxes_3d.cpp:373-553 ---------------------------------------- */
static void s2_cross(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static double s2_norm(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

static void s2_jacobi(const double A[9], double evals[3], double evecs[9]) {
    double a[3][3], q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) a[r][c] = A[r * 3 + c];
    for (int sweep = 0; sweep < 64; ++sweep) {
        if (a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2] < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int r = p + 1; r < 3; ++r) {
                if (a[p][r] == 0.0) continue;
                const double theta = (a[r][r] - a[p][p]) / (2.0 * a[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double x = a[k][p], y = a[k][r]; a[k][p] = c * x - s * y; a[k][r] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { const double x = a[p][k], y = a[r][k]; a[p][k] = c * x - s * y; a[r][k] = s * x + c * y; }
                for (int k = 0; k < 3; ++k) { const double x = q[k][p], y = q[k][r]; q[k][p] = c * x - s * y; q[k][r] = s * x + c * y; }
            }
    }
    for (int i = 0; i < 3; ++i) {
        evals[i] = a[i][i];
        for (int k = 0; k < 3; ++k) evecs[k * 3 + i] = q[k][i];
    }
}
static void s2_quat(const double R[9], double q[4]) { /* Eigen::Quaterniond(R) -> {w,x,y,z} */
    double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}
static void s2_obb_from_frame(const double *pts, int64_t n, const double c[3], double R[9], double *obb) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i) {
        const double d[3] = {pts[i * 3] - c[0], pts[i * 3 + 1] - c[1], pts[i * 3 + 2] - c[2]};
        for (int a = 0; a < 3; ++a) {
            const double l = R[0 * 3 + a] * d[0] + R[1 * 3 + a] * d[1] + R[2 * 3 + a] * d[2];
            if (l < mn[a]) mn[a] = l;
            if (l > mx[a]) mx[a] = l;
        }
    }
    double cl[3];
    for (int a = 0; a < 3; ++a) { obb[7 + a] = 2.0 * (0.5 * (mx[a] - mn[a])); cl[a] = 0.5 * (mx[a] + mn[a]); }
    for (int r = 0; r < 3; ++r) obb[r] = c[r] + (R[r * 3 + 0] * cl[0] + R[r * 3 + 1] * cl[1] + R[r * 3 + 2] * cl[2]);
    s2_quat(R, obb + 3);
}
void so2_compute_obb_pca(const double *pts, int64_t n, double *obb) {
    for (int i = 0; i < 10; ++i) obb[i] = 0.0;
    obb[3] = 1.0;
    if (n == 0) return;
    if (n == 1) { obb[0] = pts[0]; obb[1] = pts[1]; obb[2] = pts[2]; return; }
    if (n == 2) {
        double c[3], diff[3];
        for (int a = 0; a < 3; ++a) { c[a] = 0.5 * (pts[a] + pts[3 + a]); diff[a] = pts[3 + a] - pts[a]; }
        const double dn = s2_norm(diff);
        if (dn < 1e-10) { obb[0] = c[0]; obb[1] = c[1]; obb[2] = c[2]; return; }
        double a1[3] = {diff[0] / dn, diff[1] / dn, diff[2] / dn}, a2[3], a3[3];
        const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0};
        s2_cross(fabs(a1[0]) < 0.9 ? ex : ey, a1, a2);
        double nn = s2_norm(a2);
        for (int a = 0; a < 3; ++a) a2[a] /= nn;
        s2_cross(a1, a2, a3);
        nn = s2_norm(a3);
        for (int a = 0; a < 3; ++a) a3[a] /= nn;
        double R[9];
        for (int r = 0; r < 3; ++r) { R[r * 3 + 0] = a1[r]; R[r * 3 + 1] = a2[r]; R[r * 3 + 2] = a3[r]; }
        double c01[3];
        s2_cross(a1, a2, c01);
        if (c01[0] * a3[0] + c01[1] * a3[1] + c01[2] * a3[2] < 0.0) /* ensure right-handed */
            for (int r = 0; r < 3; ++r) R[r * 3 + 2] = -R[r * 3 + 2];
        s2_obb_from_frame(pts, n, c, R, obb);
        return;
    }
    double centroid[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t k = 0; k < n; ++k) { /* Welford */
        double d1[3], d2[3];
        for (int a = 0; a < 3; ++a) d1[a] = pts[k * 3 + a] - centroid[a];
        for (int a = 0; a < 3; ++a) centroid[a] += d1[a] / (double)(k + 1);
        for (int a = 0; a < 3; ++a) d2[a] = pts[k * 3 + a] - centroid[a];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) cov[r * 3 + c] += d1[r] * d2[c];
    }
    for (int i = 0; i < 9; ++i) cov[i] /= (double)n;
    const double sym[9] = {cov[0], cov[3], cov[6], cov[3], cov[4], cov[7], cov[6], cov[7], cov[8]};
    double evals[3], evecs[9];
    s2_jacobi(sym, evals, evecs);
    int order[3] = {0, 1, 2}; /* descending eigenvalues */
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (evals[order[j]] > evals[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    double R[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) R[r * 3 + c] = evecs[r * 3 + order[c]];
    const double c0[3] = {R[0], R[3], R[6]}, c1[3] = {R[1], R[4], R[7]}, c2[3] = {R[2], R[5], R[8]};
    double cr[3];
    s2_cross(c0, c1, cr);
    if (cr[0] * c2[0] + cr[1] * c2[1] + cr[2] * c2[2] < 0.0)
        for (int r = 0; r < 3; ++r) R[r * 3 + 2] = -R[r * 3 + 2];
    s2_obb_from_frame(pts, n, centroid, R, obb);
}

/* get_object_segments, voxel_block_semantic_grid.hpp:217-267.  Objects in ascending object-id order; per object
 * ids {object_id, class_id, n_points}, conf {min, max}, obb[10]; points/colors concatenated in that order. */
typedef struct { int32_t obj; int64_t block, vox; } s2_row;
static int s2_row_cmp(const void *a, const void *b) {
    const s2_row *x = (const s2_row *)a, *y = (const s2_row *)b;
    if (x->obj != y->obj) return x->obj < y->obj ? -1 : 1;
    if (x->block != y->block) return x->block < y->block ? -1 : 1;
    return x->vox < y->vox ? -1 : (x->vox > y->vox);
}
int64_t so2_get_object_segments(s2_grid *g, int min_count, float min_confidence, int32_t *ids, float *conf, double *obb,
                                double *pts, float *cols, int64_t cap_objects, int64_t cap_points, int64_t *n_points) {
    s2_row *rows = NULL;
    int64_t n = 0, cap = 0;
    for (int64_t b = 0; b < g->num_blocks; ++b)
        for (int i = 0; i < g->voxels_per_block; ++i) {
            s2_voxel *v = &g->blocks[b].data[i];
            if (v->count > min_count && s2_confidence(g, v) >= min_confidence) { /* strict '>', :224 */
                const int32_t obj = s2_object_id(g, v);
                if (obj < 0) continue;
                if (n == cap) { cap = cap ? cap * 2 : 1024; rows = (s2_row *)realloc(rows, sizeof(s2_row) * (size_t)cap); }
                rows[n].obj = obj; rows[n].block = b; rows[n].vox = i;
                ++n;
            }
        }
    qsort(rows, (size_t)n, sizeof(s2_row), s2_row_cmp);
    int64_t n_obj = 0;
    double *tmp = (double *)malloc(sizeof(double) * 3 * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n;) {
        int64_t j = i;
        float cmin = 0, cmax = 0;
        while (j < n && rows[j].obj == rows[i].obj) {
            s2_voxel *v = &g->blocks[rows[j].block].data[rows[j].vox];
            const float c = s2_confidence(g, v);
            if (j == i) cmin = cmax = c;
            if (c < cmin) cmin = c;
            if (c > cmax) cmax = c;
            for (int k = 0; k < 3; ++k) {
                tmp[(j - i) * 3 + k] = v->position_sum[k] / (double)v->count;
                if (pts && j < cap_points) {
                    pts[j * 3 + k] = tmp[(j - i) * 3 + k];
                    cols[j * 3 + k] = v->color_sum[k] / (float)v->count;
                }
            }
            ++j;
        }
        if (ids && n_obj < cap_objects) {
            s2_voxel *first = &g->blocks[rows[i].block].data[rows[i].vox];
            ids[n_obj * 3] = rows[i].obj; ids[n_obj * 3 + 1] = s2_class_id(g, first); ids[n_obj * 3 + 2] = (int32_t)(j - i);
            conf[n_obj * 2] = cmin; conf[n_obj * 2 + 1] = cmax;
            so2_compute_obb_pca(tmp, j - i, obb + n_obj * 10);
        }
        ++n_obj;
        i = j;
    }
    if (n_points) *n_points = n;
    free(tmp);
    free(rows);
    return n_obj;
}
