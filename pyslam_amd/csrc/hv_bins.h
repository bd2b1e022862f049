// Per-call point bins shared by the VOXEL_GRID integrate (hv_voxel_grid.hip) and the semantic grids' (hv_semantic.hip).
//
// The reference groups a call's points per block and updates each block's voxels serially in point order
// (integrate_raw_preorder_no_block_mutex, voxel_block_grid.hpp:292-462).  Here that is TWO launches (round 6; rounds 3-5 had four:
// count -> offsets over every ALLOCATED block -> scatter -> fold):
//   bin pass   1 thread / point: key, block claim, then ONE returning atomic per (wave, block) on the block's counter gives the
//              wave's points their places in the block's bin; the lane that finds the counter at zero is the block's first point of
//              this call and appends the block to the call's touched list.  Nothing is proportional to the size of the map.
//   fold pass  1 wave / touched block: bin -> LDS, (voxel, point index) order, run heads fold their points in point order.
// A bin is addressed by the block's hash-table SLOT (known the moment the key is found or claimed; the pool index of a block claimed
// by another lane of the same launch is not): HV_BIN_K0 inline entries per slot, and beyond them pages of HV_BIN_PG entries found
// through a small open-addressing table keyed (call epoch, slot, page number) whose storage is indexed by the page-table slot itself
// - a page is usable the moment its key is claimed, and a new call's epoch makes every old page key claimable: nothing is cleared.
// The touched list is eight lists, one per XCD (one returning atomic per workgroup on ONE word is 11 ns each, ~14 us per 640x480
// frame; MI355X_MICROARCH.md "dequeue": shard the head per XCD).
#pragma once
#include "hv_common.h"

static constexpr int HV_BIN_K0 = 256;         // inline entries per table slot (1 KiB; a 640x480 / 5 mm frame puts ~24 points in a block)
static constexpr int HV_BIN_PG = 256;         // entries per overflow page
static constexpr int HV_BIN_LISTS = 8;        // touched lists (XCDs)
static constexpr int HV_BIN_LEN_STRIDE = 32;  // int32 words between two list lengths: a 128-byte line each
static constexpr uint32_t HV_BIN_EPOCHS = 0xFFFFFEu;
static constexpr int HV_VGB_IDX_BITS = 20;    // VOXEL_GRID entries: local voxel index << 20 | point index
static constexpr int HV_VGB_CAP = 4096;       // entries of one bin sorted in a workgroup's LDS at a time (k_vgb_fold)
static constexpr int HV_VGB_WCAP = 1024;      // ... in a wave's LDS window (k_vgb_fold_wave)

#ifdef __HIPCC__
// Lanes of a wave that hold the same block key form a group (neighbouring pixels fall into the same block: a wave of 64 points
// meets a handful of distinct blocks).  One lane per group - the leader - talks to memory (hash probe, counter); every member
// learns the group's size and its own rank.  Ballot + readlane only.
struct HvWaveGroup {
    bool leader;
    int leader_lane, size, rank;
};
__device__ __forceinline__ HvWaveGroup hv_wave_group_by_key(bool has, unsigned long long key) {
    HvWaveGroup g{false, 0, 0, 0};
    const int lane = hv_lane_id();
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    unsigned long long remaining = __ballot(has);
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    while (remaining) {
        const int first = __ffsll((long long)remaining) - 1;
        const uint32_t flo = __builtin_amdgcn_readlane(klo, first), fhi = __builtin_amdgcn_readlane(khi, first);
        const bool mine = has && klo == flo && khi == fhi;
        const unsigned long long same = __ballot(mine);
        if (mine) {
            g.leader = lane == first;
            g.leader_lane = first;
            g.size = __popcll(same);
            g.rank = __popcll(same & lt);
        }
        remaining &= ~same;
    }
    return g;
}

__device__ __forceinline__ int hv_xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & (HV_BIN_LISTS - 1);
}

__device__ __forceinline__ unsigned long long hv_bins_page_key(const HvBins &B, int32_t slot, uint32_t q) {
    return ((unsigned long long)B.epoch << 40) | ((unsigned long long)(uint32_t)slot << 16) | (unsigned long long)q;
}
// Place `pos` of slot's bin, for writing (bin pass): inline, or in page (pos - K0) / PG, claimed here if this is its first entry.
// nullptr: the page table is full (cannot happen while n <= the max_points it was sized for; counted as an overflow).
__device__ __forceinline__ uint32_t *hv_bins_place(const HvTable &table, const HvBins &B, int32_t slot, int pos) {
    if (pos < HV_BIN_K0) return B.inl + (size_t)slot * HV_BIN_K0 + pos;
    const uint32_t q = (uint32_t)(pos - HV_BIN_K0) / HV_BIN_PG, o = (uint32_t)(pos - HV_BIN_K0) % HV_BIN_PG;
    const unsigned long long key = hv_bins_page_key(B, slot, q);
    uint32_t s = hv_slot_hash(key) & B.pg_mask;
    for (uint32_t probe = 0; probe <= B.pg_mask;) {
        const unsigned long long k = B.pg_keys[s];
        if (k == key) return B.pg_data + (size_t)s * HV_BIN_PG + o;
        if ((uint32_t)(k >> 40) != B.epoch) { // a page of an earlier call: free
            const unsigned long long prev = atomicCAS(&B.pg_keys[s], k, key);
            if (prev == k || prev == key) return B.pg_data + (size_t)s * HV_BIN_PG + o;
            if ((uint32_t)(prev >> 40) != B.epoch) continue; // (a stale line: look at the slot again)
        }
        s = (s + 1) & B.pg_mask; // taken by another page of this call
        ++probe;
    }
    atomicAdd(&table.counters[HV_CNT_OVERFLOW], 1);
    return nullptr;
}
// ... for reading (fold pass: the page keys of this call are all in place).  The lanes of a wave ask for 64 consecutive, 64-aligned
// places: one page, the same probe sequence - one request.
__device__ __forceinline__ uint32_t hv_bins_entry(const HvBins &B, int32_t slot, int pos) {
    if (pos < HV_BIN_K0) return B.inl[(size_t)slot * HV_BIN_K0 + pos];
    const uint32_t q = (uint32_t)(pos - HV_BIN_K0) / HV_BIN_PG, o = (uint32_t)(pos - HV_BIN_K0) % HV_BIN_PG;
    const unsigned long long key = hv_bins_page_key(B, slot, q);
    uint32_t s = hv_slot_hash(key) & B.pg_mask;
    for (uint32_t probe = 0; probe <= B.pg_mask; ++probe) {
        if (B.pg_keys[s] == key) return B.pg_data[(size_t)s * HV_BIN_PG + o];
        s = (s + 1) & B.pg_mask;
    }
    return 0xFFFFFFFFu; // (its write failed and was counted)
}

// The bin pass's common half, called by ALL threads of the workgroup (HV_BIN_THREADS; it synchronises): `has` = this thread holds a
// point of block `bkey` (packed key, in range, this GPU's), local voxel index lidx, point index i.
// Two single-word counters are touched ONCE per workgroup, both between the same two barriers: this XCD's touched-list length and
// the pool's block counter (a workgroup's new blocks take consecutive pool indices).  Workgroups of 512 threads (measured against 256
// and 1024, round 6: 512 and 256 are level, 1024 is 3-5 % slower at 640x480 - the two barriers hold 16 waves instead of 8 - and
// 2 450 workgroups' returning atomics per word for a 1296x968 keyframe stay spread over the launch).
static constexpr int HV_BIN_THREADS = 512;
__device__ __forceinline__ void hv_bins_push(const HvTable &table, const HvBins &B, bool has, unsigned long long bkey, uint32_t lidx, uint32_t i) {
    __shared__ int32_t s_first[HV_BIN_THREADS];
    __shared__ int32_t s_new_slot[HV_BIN_THREADS];
    __shared__ unsigned long long s_new_key[HV_BIN_THREADS];
    __shared__ int32_t s_n, s_nn, s_base, s_nbase;
    if (threadIdx.x == 0) s_n = s_nn = 0;
    __syncthreads();
    const int lane = hv_lane_id();
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const HvWaveGroup g = hv_wave_group_by_key(has, bkey);
    int32_t slot = -1, old = 0;
    bool is_new = false;
    if (g.leader) {
        slot = hv_table_claim(table, bkey, &is_new);
        if (slot >= 0) {
            old = atomicAdd(&B.cnt[slot], g.size);
            // a bin beyond a wave's LDS window: tell the host (it picks the fold kernel of the NEXT call by it)
            if (old + g.size > HV_VGB_WCAP) atomicMax(&table.counters[HV_CNT_OUT2], old + g.size);
        } else {
            atomicAdd(&table.counters[HV_CNT_DROPPED], g.size); // (the table is full: reported by the caller's capacity checks)
        }
    }
    slot = __shfl(slot, g.leader_lane);
    old = __shfl(old, g.leader_lane);
    if (has && slot >= 0) {
        uint32_t *at = hv_bins_place(table, B, slot, old + g.rank);
        if (at) *at = (lidx << B.idx_bits) | i;
    }
    // the block's first points of this call: its slot enters the touched list; a key this lane put into the table: it needs a block
    const bool first = g.leader && slot >= 0 && old == 0;
    const unsigned long long fm = __ballot(first), nm = __ballot(is_new);
    if (fm) {
        const int fl = __ffsll((long long)fm) - 1;
        int32_t wb = 0;
        if (lane == fl) wb = atomicAdd(&s_n, (int32_t)__popcll(fm));
        wb = __shfl(wb, fl);
        if (first) s_first[wb + __popcll(fm & lt)] = slot;
    }
    if (nm) {
        const int fl = __ffsll((long long)nm) - 1;
        int32_t wb = 0;
        if (lane == fl) wb = atomicAdd(&s_nn, (int32_t)__popcll(nm));
        wb = __shfl(wb, fl);
        if (is_new) {
            s_new_slot[wb + __popcll(nm & lt)] = slot;
            s_new_key[wb + __popcll(nm & lt)] = bkey;
        }
    }
    __syncthreads();
    const int32_t nf = s_n, nn = s_nn;
    if (nf == 0 && nn == 0) return; // (workgroup-uniform)
    const int xcc = hv_xcc_id();
    if (threadIdx.x == 0 && nf) s_base = atomicAdd(&B.len[(B.parity * HV_BIN_LISTS + xcc) * HV_BIN_LEN_STRIDE], nf);
    if (threadIdx.x == HV_WAVE && nn) s_nbase = atomicAdd(&table.counters[HV_CNT_BLOCKS], nn); // (another wave: both round trips in flight together)
    __syncthreads();
    const int32_t t = (int32_t)threadIdx.x;
    if (t < nf && s_base + t < B.touched_cap) B.touched[(size_t)xcc * B.touched_cap + s_base + t] = s_first[t];
    if (t < nn) hv_table_assign(table, s_new_slot[t], s_new_key[t], s_nbase + t);
}

// Fold side: the call's touched slots as one sequence 0 .. total) over the eight lists.
struct HvBinLists {
    int32_t len[HV_BIN_LISTS];
    int32_t total;
};
__device__ __forceinline__ HvBinLists hv_bins_lists(const HvBins &B) {
    HvBinLists L;
    L.total = 0;
#pragma unroll
    for (int x = 0; x < HV_BIN_LISTS; ++x) {
        int32_t n = B.len[(B.parity * HV_BIN_LISTS + x) * HV_BIN_LEN_STRIDE];
        n = n < B.touched_cap ? n : B.touched_cap;
        L.len[x] = n;
        L.total += n;
    }
    return L;
}
__device__ __forceinline__ int32_t hv_bins_touched(const HvBins &B, const HvBinLists &L, int32_t t) {
    int x = 0;
#pragma unroll
    for (int k = 0; k < HV_BIN_LISTS - 1; ++k) {
        const bool beyond = x == k && t >= L.len[k];
        t -= beyond ? L.len[k] : 0;
        x += beyond ? 1 : 0;
    }
    return B.touched[(size_t)x * B.touched_cap + t];
}
// The fold's walk over the touched bins: bins first, first + stride, ... of the call, one per call of body(header), by a WAVE (every
// lane calls; control flow is wave-uniform).  A bin costs its wave a chain of dependent round trips (list entry -> size, block and
// entries -> records / voxels), and a launch with one wave per bin is as long as the number of bins over the machine's wave slots
// times that chain.  Here a PERSISTENT wave requests the header of its next bin - size, pool block, the first 64 entries - before
// it works on the current one, and the list entry of the bin after that: the headers travel while the wave folds.
struct HvBinHeader {
    int32_t slot, nb, idx;
    uint32_t head; // entry [lane] of the bin (places beyond nb hold stale entries)
};
__device__ __forceinline__ HvBinHeader hv_bins_header(const HvTable &table, const HvBins &B, int32_t slot) {
    HvBinHeader h;
    h.slot = slot;
    h.head = B.inl[(size_t)slot * HV_BIN_K0 + hv_lane_id()];
    h.nb = B.cnt[slot];
    h.idx = table.vals[slot];
    return h;
}
template <typename F>
__device__ __forceinline__ void hv_bins_for_each(const HvTable &table, const HvBins &B, const HvBinLists &L, int first, int stride, F body) {
    int t = first;
    if (t >= L.total) return;
    HvBinHeader h = hv_bins_header(table, B, hv_bins_touched(B, L, t));
    int32_t slot_nn = t + stride < L.total ? hv_bins_touched(B, L, t + stride) : 0;
    while (true) {
        const HvBinHeader cur = h;
        const int t1 = t + stride;
        const bool more = t1 < L.total;
        if (more) {
            h = hv_bins_header(table, B, slot_nn);
            slot_nn = t1 + stride < L.total ? hv_bins_touched(B, L, t1 + stride) : 0;
        }
        body(cur);
        if (!more) break;
        t = t1;
    }
}

// the next call's list lengths (one thread of the fold)
__device__ __forceinline__ void hv_bins_clear_next(const HvBins &B) {
#pragma unroll
    for (int x = 0; x < HV_BIN_LISTS; ++x) B.len[((B.parity ^ 1) * HV_BIN_LISTS + x) * HV_BIN_LEN_STRIDE] = 0;
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
#endif // __HIPCC__

// Workgroups of a persistent fold launch: as many as are resident at once (registers and LDS of THIS kernel: 4 per CU for the
// probabilistic semantic fold at 128 registers, 8 for the VOXEL_GRID fold), asked once per kernel.
template <typename K>
static unsigned hv_bins_fold_grid(hv_volume *v, K kernel, size_t dyn_lds) {
    static std::vector<std::pair<const void *, unsigned>> cache; // (one process-wide answer per kernel: the library is gfx950 only)
    for (const auto &c : cache)
        if (c.first == (const void *)kernel) return c.second;
    int per_cu = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, v->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 4;
    const unsigned grid = (unsigned)(per_cu * cus);
    cache.emplace_back((const void *)kernel, grid);
    return grid;
}

// Host side: the bins of a volume (hv_volume::bins_*), made for the current table and max_points, re-made when the table moves.
static inline bool hv_bins_usable(const hv_volume *v, int64_t n, int idx_bits) {
    return n < (1ll << idx_bits) && n <= (int64_t)v->cfg.max_points && v->table_capacity <= (1ull << 24) &&
           (uint64_t)v->cfg.max_points / HV_BIN_PG < (1ull << 16);
}
static int hv_bins_ensure(hv_volume *v) {
    if (v->bins_cap == v->table_capacity && v->bins.cnt != nullptr && v->bins_clean) return HV_OK;
    const bool remake = v->bins_cap != v->table_capacity || v->bins.cnt == nullptr;
    if (remake) {
        HV_HIP(hipStreamSynchronize(v->stream));
        for (void **p : {(void **)&v->bins.cnt, (void **)&v->bins.inl, (void **)&v->bins.touched}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
    }
    const size_t len_bytes = sizeof(int32_t) * 2 * HV_BIN_LISTS * HV_BIN_LEN_STRIDE;
    if (v->bins.len == nullptr) HV_HIP(hipMalloc((void **)&v->bins.len, len_bytes));
    HV_HIP(hipMemsetAsync(v->bins.len, 0, len_bytes, v->stream));
    if (v->semb_tasks) HV_HIP(hipMemsetAsync(v->semb_tasks, 0, 256, v->stream)); // the semantic path's task counters restart with the parity
    if (v->bins.pg_keys == nullptr) {
        // pages in use <= n / PG + n / K0 (every bin beyond its inline part wastes at most one page): a table of max_points / 64 is at most half full
        uint64_t pages = 1024;
        while (pages < (uint64_t)v->cfg.max_points / 64) pages <<= 1;
        HV_HIP(hipMalloc((void **)&v->bins.pg_keys, sizeof(unsigned long long) * pages));
        HV_HIP(hipMalloc((void **)&v->bins.pg_data, sizeof(uint32_t) * HV_BIN_PG * pages));
        HV_HIP(hipMemsetAsync(v->bins.pg_keys, 0, sizeof(unsigned long long) * pages, v->stream)); // epoch 0: never used
        v->bins.pg_mask = (uint32_t)(pages - 1);
        v->bins_epoch = 0;
    }
    if (remake) {
        v->bins.touched_cap = (int32_t)std::min<int64_t>(std::min<int64_t>(v->cfg.max_blocks, v->cfg.max_points), (int64_t)v->table_capacity);
        HV_HIP(hipMalloc((void **)&v->bins.cnt, sizeof(int32_t) * v->table_capacity));
        HV_HIP(hipMalloc((void **)&v->bins.inl, sizeof(uint32_t) * HV_BIN_K0 * v->table_capacity));
        HV_HIP(hipMalloc((void **)&v->bins.touched, sizeof(int32_t) * HV_BIN_LISTS * (size_t)v->bins.touched_cap));
    }
    HV_HIP(hipMemsetAsync(v->bins.cnt, 0, sizeof(int32_t) * v->table_capacity, v->stream));
    HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT2], 0, sizeof(int32_t), v->stream));
    v->bins_cap = v->table_capacity;
    v->bins_clean = true;
    v->bins.parity = 0;
    return HV_OK;
}
// The device view for the next call: a fresh epoch (every page of earlier calls becomes claimable), this call's parity.
static inline HvBins hv_bins_begin(hv_volume *v, int idx_bits) {
    v->bins_epoch = v->bins_epoch % HV_BIN_EPOCHS + 1;
    v->bins.epoch = v->bins_epoch;
    v->bins.idx_bits = idx_bits;
    return v->bins;
}
