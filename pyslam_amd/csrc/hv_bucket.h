// Per-frame bucket machinery shared by the VOXEL_GRID integrate (hv_voxel_grid.hip) and the semantic grids' (hv_semantic.hip):
// a frame's points are grouped per BLOCK (a few thousand buckets of a few dozen to a few hundred points) and ordered inside a
// block by (voxel, point index) in LDS - the payload-independent part: wave grouping, bucket ranges, the scatter, the wave-level
// bitonic sort, the scratch arrays.  Moved here in round 4 (the semantic grids left the 18-launch device-wide radix sort).
#pragma once
#include "hv_common.h"

static constexpr int HV_VGB_IDX_BITS = 20;
static constexpr int HV_VGB_CAP = 4096; // entries of one bucket sorted in LDS at a time

// Lanes of a wave that hold the same slot form a group (neighbouring pixels fall into the same block: a wave of 64 points
// meets a handful of distinct slots).  One lane per group - the leader - talks to memory; every member learns the group's
// size and its own rank.  Ballot + shuffle only.
struct HvWaveGroup {
    bool leader;
    int leader_lane, size, rank;
};
__device__ __forceinline__ HvWaveGroup hv_wave_group_by(int32_t slot) {
    HvWaveGroup g{false, 0, 0, 0};
    const int lane = hv_lane_id();
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    unsigned long long remaining = __ballot(slot >= 0);
    while (remaining) {
        const int first = __ffsll((long long)remaining) - 1;
        const int32_t fslot = __shfl(slot, first);
        const unsigned long long same = __ballot(slot == fslot) & remaining;
        if (slot == fslot) {
            g.leader = lane == first;
            g.leader_lane = first;
            g.size = __popcll(same);
            g.rank = __popcll(same & lt);
        }
        remaining &= ~same;
    }
    return g;
}


// 1 thread / allocated block: blocks that received points this frame take their bucket range from the global cursor and
// enter the frame's touched list - both with one atomic per wave (prefix sums inside the wave).
static constexpr int HV_VGB_WCAP_DECL = 1024; // == HV_VGB_WCAP (defined with the wave fold below)
static __global__ __launch_bounds__(256) void k_vgb_offsets(HvTable table, int32_t *__restrict__ touched, unsigned long long *__restrict__ cursor_and_len,
                                                      const int32_t *__restrict__ cnt, int32_t *__restrict__ cur) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t n_blocks = table.counters[HV_CNT_BLOCKS];
    if (n_blocks > table.max_blocks) n_blocks = table.max_blocks;
    int32_t slot = -1, c = 0;
    if (b < n_blocks) {
        slot = hv_table_find(table, table.block_keys[b]);
        c = slot >= 0 ? cnt[slot] : 0;
    }
    const bool live = c > 0;
    const int lane = hv_lane_id();
    int32_t incl = c;
#pragma unroll
    for (int o = 1; o < HV_WAVE; o <<= 1) {
        const int32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const int32_t total = __shfl(incl, HV_WAVE - 1);
    const unsigned long long lm = __ballot(live);
    // bucket cursor (low 32 bits) and list length (high 32 bits) move together: ONE returning atomic per wave
    unsigned long long got = 0ull;
    if (lane == HV_WAVE - 1 && total > 0)
        got = atomicAdd(cursor_and_len, (unsigned long long)(uint32_t)total | ((unsigned long long)__popcll(lm) << 32));
    got = __shfl(got, HV_WAVE - 1);
    if (live) {
        cur[slot] = (int32_t)(uint32_t)got + incl - c;
        const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        touched[(int32_t)(got >> 32) + __popcll(lm & lt)] = slot;
    }
    // a bucket beyond a wave's LDS window: tell the host (it picks the fold kernel of the NEXT frame by it)
    if (c > HV_VGB_WCAP_DECL) atomicMax(&table.counters[HV_CNT_OUT2], c);
}

static __global__ __launch_bounds__(256) void k_vgb_scatter(const int32_t *__restrict__ pslot, const uint32_t *__restrict__ plidx, int64_t n,
                                                      int32_t *__restrict__ cur, uint32_t *__restrict__ entries) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t slot = i < n ? pslot[i] : -1;
    const HvWaveGroup g = hv_wave_group_by(slot);
    int32_t base = 0;
    if (g.leader) base = atomicAdd(&cur[slot], g.size);
    base = __shfl(base, g.leader_lane);
    if (slot >= 0) entries[base + g.rank] = (plidx[i] << HV_VGB_IDX_BITS) | (uint32_t)i;
}


// Wave-level counterparts of the two helpers above: a wave owns its LDS window, lanes synchronise with wave barriers only.
__device__ __forceinline__ void hv_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void hv_vgb_bitonic_wave(uint32_t *s, int m2) { // ascending, m2 a power of two >= 64
    const int lane = hv_lane_id();
    for (int k = 2; k <= m2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < m2; t += HV_WAVE) {
                const int x = t ^ j;
                if (x > t) {
                    const uint32_t a = s[t], b = s[x];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) {
                        s[t] = b;
                        s[x] = a;
                    }
                }
            }
            hv_wave_lds_sync();
        }
    }
}

static constexpr int HV_VGB_WCAP = HV_VGB_WCAP_DECL;

static int ensure_bucket_buffers(hv_volume *v) {
    if (v->vg_cap == v->table_capacity && v->vg_cnt != nullptr) return HV_OK;
    HV_HIP(hipStreamSynchronize(v->stream));
    for (int32_t **p : {&v->vg_cnt, &v->vg_cur, &v->vg_touched}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    if (v->vg_cursor == nullptr) HV_HIP(hipMalloc((void **)&v->vg_cursor, 2 * sizeof(unsigned long long)));
    HV_HIP(hipMemsetAsync(v->vg_cursor, 0, 2 * sizeof(unsigned long long), v->stream));
    if (v->semb_tasks) HV_HIP(hipMemsetAsync(v->semb_tasks, 0, 256, v->stream)); // the semantic path's task counters restart with the parity
    HV_HIP(hipMalloc((void **)&v->vg_cnt, sizeof(int32_t) * v->table_capacity));
    HV_HIP(hipMalloc((void **)&v->vg_cur, sizeof(int32_t) * v->table_capacity));
    HV_HIP(hipMalloc((void **)&v->vg_touched, sizeof(int32_t) * v->table_capacity));
    HV_HIP(hipMemsetAsync(v->vg_cnt, 0, sizeof(int32_t) * v->table_capacity, v->stream));
    HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT2], 0, sizeof(int32_t), v->stream));
    v->vg_cap = v->table_capacity;
    v->vg_parity = 0;
    return HV_OK;
}

