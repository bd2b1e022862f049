/*
 * TEST INFRASTRUCTURE ONLY — CPU restatement of the *voting* semantic voxel payload on the block grid:
 * VoxelBlockGridT<VoxelSemanticData> (VoxelSemanticData = VoxelSemanticDataT<double,float>,
 * cpp/volumetric/voxel_data_semantic.h:106-202) driven through integrate_raw / update_voxel_direct
 * (voxel_block_grid.hpp:115-136, 466-497, 524-614) and get_voxels (:785-817).
 *
 * Parity: PINNED against the compiled reference (oracle/_ref, ref_sgrid_*) in
 * tests/test_semantic_oracle.py, which also replays the reference's own voting known-answer tests
 * (cpp/test_volumetric_voxel_semantic.py:20-36, 79-97).  Segment operations, object association
 * (voxel_semantic_data_association.h) and the probabilistic payload are NOT restated.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t count;
    double position_sum[3];
    float color_sum[3];
    int32_t object_id, class_id, confidence_counter; /* -1, -1, 0 when empty */
} so_voxel;

typedef struct {
    int32_t key[3];
    so_voxel *data;
} so_block;

typedef struct {
    float voxel_size, inv_voxel_size;
    int block_size, voxels_per_block;
    so_block *blocks;
    int64_t num_blocks, cap_blocks;
    int64_t *table;
    int64_t table_size;
} so_grid;

static float so_depth_threshold = 10.0f; /* VoxelSemanticDataT::kDepthThreshold, voxel_data_semantic.h:107-108 */
void so_set_depth_threshold(float t) { so_depth_threshold = t; }

static uint64_t so_mix(int32_t x, int32_t y, int32_t z) {
    uint64_t h = (uint64_t)(uint32_t)x * 0x9E3779B97F4A7C15ull;
    h ^= (uint64_t)(uint32_t)y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    h ^= (uint64_t)(uint32_t)z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
    return h ^ (h >> 29);
}
static void so_rebuild(so_grid *g, int64_t n) {
    free(g->table);
    g->table_size = n;
    g->table = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) g->table[i] = -1;
    for (int64_t b = 0; b < g->num_blocks; ++b) {
        const int32_t *k = g->blocks[b].key;
        uint64_t s = so_mix(k[0], k[1], k[2]) & (uint64_t)(n - 1);
        while (g->table[s] >= 0) s = (s + 1) & (uint64_t)(n - 1);
        g->table[s] = b;
    }
}
so_grid *so_create(float voxel_size, int block_size) {
    so_grid *g = (so_grid *)calloc(1, sizeof(so_grid));
    g->voxel_size = voxel_size;
    g->inv_voxel_size = 1.0f / voxel_size;
    g->block_size = block_size;
    g->voxels_per_block = block_size * block_size * block_size;
    so_rebuild(g, 1024);
    return g;
}
void so_clear(so_grid *g) {
    for (int64_t b = 0; b < g->num_blocks; ++b) free(g->blocks[b].data);
    g->num_blocks = 0;
    so_rebuild(g, 1024);
}
void so_destroy(so_grid *g) {
    if (!g) return;
    so_clear(g);
    free(g->blocks);
    free(g->table);
    free(g);
}
int64_t so_num_blocks(const so_grid *g) { return g->num_blocks; }

static so_block *so_find_or_create(so_grid *g, int32_t bx, int32_t by, int32_t bz) {
    uint64_t s = so_mix(bx, by, bz) & (uint64_t)(g->table_size - 1);
    while (g->table[s] >= 0) {
        so_block *b = &g->blocks[g->table[s]];
        if (b->key[0] == bx && b->key[1] == by && b->key[2] == bz) return b;
        s = (s + 1) & (uint64_t)(g->table_size - 1);
    }
    if (g->num_blocks == g->cap_blocks) {
        g->cap_blocks = g->cap_blocks ? g->cap_blocks * 2 : 256;
        g->blocks = (so_block *)realloc(g->blocks, sizeof(so_block) * (size_t)g->cap_blocks);
    }
    so_block *b = &g->blocks[g->num_blocks];
    b->key[0] = bx; b->key[1] = by; b->key[2] = bz;
    b->data = (so_voxel *)calloc((size_t)g->voxels_per_block, sizeof(so_voxel));
    for (int i = 0; i < g->voxels_per_block; ++i) { b->data[i].object_id = -1; b->data[i].class_id = -1; }
    g->table[s] = g->num_blocks++;
    if (g->num_blocks * 2 > g->table_size) {
        so_rebuild(g, g->table_size * 2);
        return &g->blocks[g->num_blocks - 1];
    }
    return b;
}

static inline int64_t so_floor_div(int64_t a, int64_t b) { return (a >= 0) ? (a / b) : ((a - b + 1) / b); }

/* update_semantics, voxel_data_semantic.h:175-191 */
static inline void so_update_semantics(so_voxel *v, int32_t object_id, int32_t class_id) {
    if (v->object_id == object_id && v->class_id == class_id) {
        v->confidence_counter++;
    } else {
        v->confidence_counter--;
        if (v->confidence_counter <= 0) {
            v->object_id = object_id;
            v->class_id = class_id;
            v->confidence_counter = 1;
        }
    }
}

/* integrate_raw<Tpos,Tcolor,Tinstance,Tclass,Tdepth> -> sequential update_voxel -> update_voxel_direct.
 * pos_kind 0: float32 points (keys in float: get_voxel_key_inv<float,float>), 1: float64 points
 * (get_voxel_key_inv<double,double> with the float inv_voxel_size_ promoted, voxel_block_grid.hpp:473).
 * color_kind 1 uint8, 2 float32.  instance_ids / depths may be NULL (object id 0 / no depth gate). */
void so_integrate(so_grid *g, const void *pts, int pos_kind, int64_t n, const void *cols, int color_kind,
                  const int32_t *class_ids, const int32_t *instance_ids, const float *depths) {
    const int bs = g->block_size;
    const float inv_255 = 1.0f / 255.0f;
    for (int64_t i = 0; i < n; ++i) {
        double xyz[3];
        int32_t vk[3];
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
        }
        int32_t bk[3], lk[3];
        for (int k = 0; k < 3; ++k) {
            bk[k] = (int32_t)so_floor_div(vk[k], bs);
            lk[k] = (int32_t)((int64_t)vk[k] - (int64_t)bk[k] * bs);
        }
        so_block *blk = so_find_or_create(g, bk[0], bk[1], bk[2]);
        so_voxel *v = &blk->data[lk[0] + lk[1] * bs + lk[2] * bs * bs];
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
            const int gate = depths ? (depths[i] < so_depth_threshold) : 1; /* *_with_depth variants */
            if (v->count == 0) {
                if (gate) { v->object_id = obj; v->class_id = class_ids[i]; v->confidence_counter = 1; }
            } else if (gate) {
                so_update_semantics(v, obj, class_ids[i]);
            }
        }
        v->count = (v->count == 0) ? 1 : v->count + 1;
    }
}

static int so_cmp(const void *pa, const void *pb) {
    const so_block *a = *(const so_block *const *)pa, *b = *(const so_block *const *)pb;
    for (int k = 0; k < 3; ++k)
        if (a->key[k] != b->key[k]) return a->key[k] < b->key[k] ? -1 : 1;
    return 0;
}

int64_t so_dump(const so_grid *g, int32_t *keys, int32_t *ints, double *pos_sums, float *col_sums) {
    const int64_t nb = g->num_blocks;
    const so_block **order = (const so_block **)malloc(sizeof(void *) * (size_t)(nb ? nb : 1));
    for (int64_t b = 0; b < nb; ++b) order[b] = &g->blocks[b];
    qsort(order, (size_t)nb, sizeof(void *), so_cmp);
    const int64_t nv = g->voxels_per_block;
    for (int64_t b = 0; b < nb; ++b) {
        if (keys) memcpy(keys + b * 3, order[b]->key, 12);
        for (int64_t i = 0; i < nv; ++i) {
            const so_voxel *v = &order[b]->data[i];
            if (ints) {
                int32_t *d = ints + (b * nv + i) * 4;
                d[0] = v->count; d[1] = v->object_id; d[2] = v->class_id; d[3] = v->confidence_counter;
            }
            if (pos_sums) memcpy(pos_sums + (b * nv + i) * 3, v->position_sum, 24);
            if (col_sums) memcpy(col_sums + (b * nv + i) * 3, v->color_sum, 12);
        }
    }
    free(order);
    return nb;
}

/* get_confidence(), voxel_data_semantic.h:116-133 */
static inline float so_confidence(const so_voxel *v) {
    if (v->count == 0) return 0.0f;
    const float r = (float)v->confidence_counter / (float)v->count;
    return r < 1.0f ? r : 1.0f;
}

/* get_voxels(min_count, min_confidence), voxel_block_grid.hpp:785-817 (semantic branch) */
int64_t so_get_voxels(const so_grid *g, int min_count, float min_confidence, double *pts, float *cols, int32_t *class_ids,
                      int32_t *object_ids, float *confidences, int64_t cap) {
    int64_t n = 0;
    for (int64_t b = 0; b < g->num_blocks; ++b)
        for (int i = 0; i < g->voxels_per_block; ++i) {
            const so_voxel *v = &g->blocks[b].data[i];
            const float conf = so_confidence(v);
            if (v->count >= min_count && conf >= min_confidence) {
                if (pts && n < cap) {
                    const double c = (double)v->count;
                    const float cf = (float)v->count;
                    for (int k = 0; k < 3; ++k) {
                        pts[n * 3 + k] = v->position_sum[k] / c;
                        cols[n * 3 + k] = v->color_sum[k] / cf;
                    }
                    class_ids[n] = v->class_id;
                    object_ids[n] = v->object_id;
                    confidences[n] = conf;
                }
                ++n;
            }
        }
    return n;
}
