// libpyslam_hipvol.so - the halo merge's key lists and plan ON THE DEVICE (round 6).
//
// The halo merge of image-tile-sharded TSDF volumes (include/hipvol.h "halo merge", pyslam_amd/distributed.py::merge_halo) needs every
// rank's list of units written since its last merge and its list of units held, and from the all-gathered lists the plan: the keys some
// rank updated that two ranks or more hold, with this rank's action.  Rounds 3-5 downloaded the lists, gathered them through host
// arrays (two .cpu() per merge) and planned on the host (hv_merge_halo_plan_held: a std::sort of (key, rank) pairs).  Here the lists
// are written as packed 64-bit keys into DEVICE buffers of the caller (torch tensors: what RCCL's all_gather_into_tensor takes), the
// plan is a device radix sort + two small kernels, and the shared keys / actions stay in the volume for hv_merge_halo_pack_planned /
// _unpack_planned (hv_tsdf.hip).  What still crosses to the host is three integers per merge: the two list lengths (they size the
// gather) and the number of shared units (it sizes the payload).  Same plan as the host function, in packed-key order instead of
// (x, y, z) order - every rank of a group takes the same path, and tests/test_gpu_distributed.py holds the two to each other.
#include <algorithm>

#include "hv_common.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

// dirty units (stamp later than the last merge) and held units (every allocated unit) of one volume, as packed keys
__global__ __launch_bounds__(256) void k_halo_lists(HvTable table, const int32_t *__restrict__ stamp, int32_t merge_stamp, int32_t n_blocks,
                                                     unsigned long long *__restrict__ dirty, int64_t dirty_cap, unsigned long long *__restrict__ held,
                                                     int64_t held_cap) {
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool is_dirty = false;
    unsigned long long key = 0ull;
    if (idx < n_blocks) {
        key = table.block_keys[idx];
        const int32_t slot = hv_table_find(table, key);
        is_dirty = slot >= 0 && stamp[slot] > merge_stamp;
        if (idx < held_cap) held[idx] = key;
    }
    const int32_t at = hv_wave_append(&table.counters[HV_CNT_OUT], is_dirty);
    if (is_dirty && at < dirty_cap) dirty[at] = key;
}

struct HvHaloSpans {
    int32_t world;
    int64_t dirty_off[65], held_off[65]; // exclusive prefixes of the ranks' counts (world <= 64)
};
// the gathered, padded lists [world][stride] -> one entry array: every dirty entry (by rank), then every held entry (by rank);
// value = rank << 1 | held
__global__ __launch_bounds__(256) void k_halo_entries(HvHaloSpans S, const unsigned long long *__restrict__ dirty, int64_t dirty_stride,
                                                       const unsigned long long *__restrict__ held, int64_t held_stride,
                                                       unsigned long long *__restrict__ keys, int32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nd = S.dirty_off[S.world], nh = S.held_off[S.world];
    if (i >= nd + nh) return;
    const bool is_held = i >= nd;
    const int64_t j = is_held ? i - nd : i;
    const int64_t *off = is_held ? S.held_off : S.dirty_off;
    int r = 0;
    while (r + 1 < S.world && off[r + 1] <= j) ++r;
    const int64_t k = j - off[r];
    keys[i] = is_held ? held[(int64_t)r * held_stride + k] : dirty[(int64_t)r * dirty_stride + k];
    vals[i] = (r << 1) | (is_held ? 1 : 0);
}
// sorted by key (stable: a key's dirty entries come first, then its holders by ascending rank): the head of a key's run decides
__global__ __launch_bounds__(256) void k_halo_decide(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ vals, int64_t n, int32_t rank,
                                                      int32_t all_dirty_kept, int32_t *__restrict__ flag, uint8_t *__restrict__ act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t f = 0;
    uint8_t a = 0;
    const unsigned long long key = keys[i];
    if (i == 0 || keys[i - 1] != key) {
        bool any_dirty = false;
        int holders = 0, keeper = -1;
        for (int64_t j = i; j < n && keys[j] == key; ++j) {
            const int32_t v = vals[j];
            if (v & 1) {
                if (keeper < 0) keeper = v >> 1;
                ++holders;
            } else {
                any_dirty = true;
            }
        }
        if (all_dirty_kept) { // one rank taking its own dirty units through the collective path (force_collectives): it keeps them all
            f = any_dirty ? 1 : 0;
            a = 1;
        } else if (any_dirty && holders >= 2) {
            f = 1;
            a = keeper == rank ? 1 : 2;
        }
    }
    flag[i] = f;
    act[i] = a;
}
__global__ __launch_bounds__(256) void k_halo_emit(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ flag, const int32_t *__restrict__ pos,
                                                    const uint8_t *__restrict__ act, int64_t n, int32_t *__restrict__ out_keys, uint8_t *__restrict__ out_act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const int32_t at = pos[i];
    int32_t x, y, z;
    hv_unpack_key(keys[i], x, y, z);
    out_keys[at * 3 + 0] = x;
    out_keys[at * 3 + 1] = y;
    out_keys[at * 3 + 2] = z;
    out_act[at] = act[i];
}

extern "C" {

int hv_merge_halo_lists_device(hv_volume *v, int64_t *d_dirty_keys, int64_t dirty_cap, int64_t *d_held_keys, int64_t held_cap, int64_t *n_dirty,
                               int64_t *n_held) {
    HV_REQUIRE(v != nullptr && n_dirty != nullptr && n_held != nullptr, HV_ERR_INVALID, "hv_merge_halo_lists_device: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_merge_halo_lists_device: not a TSDF volume");
    HV_HIP(hipSetDevice(v->device));
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    *n_dirty = 0;
    *n_held = nb;
    if (nb == 0 || d_dirty_keys == nullptr || d_held_keys == nullptr) return HV_OK; // (size query: an upper bound of both lists is nb)
    HV_HIP(hipMemsetAsync(&v->table.counters[HV_CNT_OUT], 0, sizeof(int32_t), v->stream));
    hipLaunchKernelGGL(k_halo_lists, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, v->stream, v->table, (const int32_t *)v->touched_stamp,
                       v->merge_stamp, (int32_t)nb, (unsigned long long *)d_dirty_keys, dirty_cap, (unsigned long long *)d_held_keys, held_cap);
    HV_HIP(hipGetLastError());
    rc = hv_read_counters(v); // (synchronises: the one number the gather's size depends on)
    if (rc != HV_OK) return rc;
    *n_dirty = v->h_counters[HV_CNT_OUT];
    return HV_OK;
}

int hv_merge_halo_plan_device(hv_volume *v, const int64_t *d_dirty_all, const int64_t *dirty_counts, int64_t dirty_stride, const int64_t *d_held_all,
                              const int64_t *held_counts, int64_t held_stride, int32_t world_size, int32_t rank, int32_t all_dirty_kept,
                              int64_t *n_shared) {
    HV_REQUIRE(v != nullptr && dirty_counts != nullptr && held_counts != nullptr && n_shared != nullptr, HV_ERR_INVALID,
               "hv_merge_halo_plan_device: null argument");
    HV_REQUIRE(world_size >= 1 && world_size <= 64 && rank >= 0 && rank < world_size, HV_ERR_INVALID, "hv_merge_halo_plan_device: bad rank / world size");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_merge_halo_plan_device: not a TSDF volume");
    HV_HIP(hipSetDevice(v->device));
    HvHaloSpans S;
    S.world = world_size;
    S.dirty_off[0] = S.held_off[0] = 0;
    for (int r = 0; r < world_size; ++r) {
        HV_REQUIRE(dirty_counts[r] >= 0 && dirty_counts[r] <= dirty_stride && held_counts[r] >= 0 && held_counts[r] <= held_stride, HV_ERR_INVALID,
                   "hv_merge_halo_plan_device: a list is longer than its stride");
        S.dirty_off[r + 1] = S.dirty_off[r] + dirty_counts[r];
        S.held_off[r + 1] = S.held_off[r] + held_counts[r];
    }
    const int64_t n = S.dirty_off[world_size] + S.held_off[world_size];
    v->halo_plan_n = 0;
    *n_shared = 0;
    if (n == 0) return HV_OK;
    HV_REQUIRE(n < (1ll << 31) && d_dirty_all != nullptr && d_held_all != nullptr, HV_ERR_INVALID, "hv_merge_halo_plan_device: bad lists");
    // scratch: [keys in n u64][keys out n u64][vals in n i32][vals out n i32][flag n i32][pos n + 1 i32][act n u8] + the sort's own
    const size_t a8 = sizeof(uint64_t) * (size_t)n, a4 = sizeof(int32_t) * (size_t)(n + 1);
    int rc = hv_ensure_buffer(v, &v->out_c, &v->out_c_bytes, 2 * a8 + 4 * a4 + (size_t)n + 64);
    if (rc != HV_OK) return rc;
    char *p = (char *)v->out_c;
    unsigned long long *k_in = (unsigned long long *)p, *k_out = (unsigned long long *)(p + a8);
    int32_t *v_in = (int32_t *)(p + 2 * a8), *v_out = (int32_t *)(p + 2 * a8 + a4), *flag = (int32_t *)(p + 2 * a8 + 2 * a4),
            *pos = (int32_t *)(p + 2 * a8 + 3 * a4);
    uint8_t *act = (uint8_t *)(p + 2 * a8 + 4 * a4);
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_halo_entries, dim3(grid), dim3(256), 0, v->stream, S, (const unsigned long long *)d_dirty_all, dirty_stride,
                       (const unsigned long long *)d_held_all, held_stride, k_in, v_in);
    size_t tmp = 0;
    HV_HIP(rocprim::radix_sort_pairs(nullptr, tmp, k_in, k_out, v_in, v_out, (size_t)n, 0, 64, v->stream));
    rc = hv_ensure_buffer(v, &v->sort_tmp, &v->sort_tmp_bytes, tmp);
    if (rc != HV_OK) return rc;
    tmp = v->sort_tmp_bytes;
    HV_HIP(rocprim::radix_sort_pairs(v->sort_tmp, tmp, k_in, k_out, v_in, v_out, (size_t)n, 0, 64, v->stream)); // (stable: LSD radix)
    hipLaunchKernelGGL(k_halo_decide, dim3(grid), dim3(256), 0, v->stream, k_out, v_out, n, rank, all_dirty_kept, flag, act);
    HV_HIP(hipMemsetAsync(flag + n, 0, sizeof(int32_t), v->stream)); // the scan's extra element: its output there is the total
    size_t tmp2 = 0;
    HV_HIP(rocprim::exclusive_scan(nullptr, tmp2, flag, pos, 0, (size_t)(n + 1), rocprim::plus<int32_t>(), v->stream));
    rc = hv_ensure_buffer(v, &v->sort_tmp, &v->sort_tmp_bytes, tmp2);
    if (rc != HV_OK) return rc;
    tmp2 = v->sort_tmp_bytes;
    HV_HIP(rocprim::exclusive_scan(v->sort_tmp, tmp2, flag, pos, 0, (size_t)(n + 1), rocprim::plus<int32_t>(), v->stream));
    int32_t total = 0;
    HV_HIP(hipMemcpyAsync(&total, pos + n, sizeof(int32_t), hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream)); // (the one number the payload's size depends on)
    if (total > 0) {
        rc = hv_ensure_buffer(v, &v->halo_plan, &v->halo_plan_bytes, (size_t)total * 13 + 64);
        if (rc != HV_OK) return rc;
        hipLaunchKernelGGL(k_halo_emit, dim3(grid), dim3(256), 0, v->stream, k_out, flag, pos, act, n, (int32_t *)v->halo_plan,
                           (uint8_t *)v->halo_plan + (size_t)total * 12);
        HV_HIP(hipGetLastError());
    }
    v->halo_plan_n = total;
    *n_shared = total;
    return HV_OK;
}

// the plan as host arrays (inspection, tests): shared_keys [n, 3] i32, action [n] u8
int hv_merge_halo_plan_fetch(hv_volume *v, int32_t *shared_keys, uint8_t *action, int64_t cap, int64_t *n_shared) {
    HV_REQUIRE(v != nullptr && n_shared != nullptr, HV_ERR_INVALID, "hv_merge_halo_plan_fetch: null argument");
    *n_shared = v->halo_plan_n;
    const int64_t m = std::min<int64_t>(v->halo_plan_n, cap);
    if (shared_keys == nullptr || action == nullptr || m <= 0) return HV_OK;
    HV_HIP(hipSetDevice(v->device));
    HV_HIP(hipMemcpyAsync(shared_keys, v->halo_plan, sizeof(int32_t) * 3 * (size_t)m, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipMemcpyAsync(action, (const uint8_t *)v->halo_plan + (size_t)v->halo_plan_n * 12, (size_t)m, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}

} // extern "C"
