// libpyslam_hipvol.so — TSDF surface extraction on gfx950: marching cubes and point cloud
// (Open3D ScalableTSDFVolume::ExtractTriangleMesh / ExtractPointCloud semantics; reference call
// sites pyslam/dense/volumetric_integrator_tsdf.py:239-267).
//
// Both extractions start from per-unit COLUMN MASKS (k_unit_masks, one pass over the tsdf and weight planes, 256-byte runs,
// 32 KB read and 2 KB written per unit): marching cubes only asks "observed?" and "negative?" of a voxel, the point cloud
// "in [-0.98, 0.98)?" and the sign, so a voxel column is four 16-bit masks.  Everything a unit needs of its NEIGHBOURS (the
// 18^3 neighbourhood of marching cubes, the 17^3 one of the point cloud) is then read from their masks - a few KB - and not
// from their planes, where a y = 0 / y = 15 face costs one 64-byte line per voxel (round 2: 118 KB fetched per unit for a
// 32 KB unit, profiles/r02 and r03/pmc_summary.json before this form).
// Marching cubes, per allocated unit:
//   k_mc_classify   (workgroup) the 18 x 18 column masks of the unit's neighbourhood from the published masks; cube cases
//                   (kept, one byte each, for the triangle pass), the unit's own edge bitmask (3 axes x 4096 bits, keyed by the
//                   edge's owning voxel - the GPU analogue of Open3D's edgeindex_to_vertexindex map) as wave ballots, its
//                   popcount prefix (vertex rank inside the unit), per-unit vertex and triangle counts.  No atomics.
//   rocPRIM scan    unit bases for vertices and triangles
//   k_mc_vertices   (wave) lane j builds vertices j, j + 64, ... of the unit: interpolated vertex + colour (f64, as Open3D)
//   k_mc_triangles  (workgroup) reads the stored cube cases and emits triangles whose vertex indices are
//                   base[unit(edge)] + rank(edge) - no hash map, no atomics in the emit passes; like the vertices and the
//                   points, the unit's triangles are dealt out evenly over the threads (rank -> column by binary search).
// Point cloud: k_pc_extract<false> counts the unit's crossings from the masks, scan, k_pc_extract<true> finds them again and
// gathers the values of the crossing voxels only - no slab, no atomics.
// Vertex/triangle *order* differs from Open3D's unordered_map iteration order (so does Open3D's
// own from run to run); the vertex and triangle *sets* are identical to the CPU restatement.
// "Sizes first, data second" (the binding's protocol) costs one computation: the size query does all the device work and
// records it under the volume's content_version, the fetch only copies.
#include <algorithm>

#include "hv_common.h"
#include "mc_tables.h"
#include <rocprim/device/device_scan.hpp>

static constexpr int R = 16;
static constexpr int RR = R * R;
static constexpr int RRR = R * R * R;
static constexpr int PLANE_BYTES = RRR * 4;
static constexpr int UNIT_BYTES = PLANE_BYTES * HV_TSDF_PLANES;
static constexpr int MASK_WORDS = 3 * RRR / 64; // 192

__constant__ unsigned short c_edge_table[256];
__constant__ __attribute__((aligned(16))) signed char c_tri_table[256][16];
__constant__ unsigned char c_tri_count[256];

__device__ __forceinline__ int voxel_word(int x, int y, int z) { return z * RR + x * R + y; }
// A unit's 192 edge-mask words in z-MAJOR order: word = z * 12 + axis * 4 + quarter (quarter = x >> 2; the word's bit = voxel_word & 63).
// The vertex ranks of a unit follow the word order, and lane j of the vertex pass builds vertices j, j + 64, ...: with the three
// axes of a z slice next to each other a wave walks the unit's planes ONCE (round 6).  Axis-major words (rounds 2-5) walked them
// three times, microseconds apart - longer than a 4 MB L2 holds a line at this fetch rate: 1.35 GB fetched for 152 MB of values.
__device__ __forceinline__ int mc_word(int axis, int lin) { return (lin >> 8) * 12 + axis * 4 + ((lin >> 6) & 3); }

// pool indices of the 8 units {this, +x, +y, +x+y, +z, ...} (bit0 = x, bit1 = y, bit2 = z), -1 if absent
__device__ inline void load_neighbours(const HvTable &table, int idx, int *s_nbr) {
    if (threadIdx.x < 8) {
        int32_t ux, uy, uz;
        hv_unpack_key(table.block_keys[idx], ux, uy, uz);
        const int n = threadIdx.x;
        const int32_t kx = ux + (n & 1), ky = uy + ((n >> 1) & 1), kz = uz + ((n >> 2) & 1);
        int r = -1;
        if (n == 0) {
            r = idx;
        } else if (hv_key_in_range(kx, ky, kz)) {
            const int32_t slot = hv_table_find(table, hv_pack_key(kx, ky, kz));
            if (slot >= 0) r = table.vals[slot];
        }
        s_nbr[n] = r;
    }
}

// edge i of cube (x,y,z) -> (neighbour selector, axis, bit index inside that unit/axis)
__device__ __forceinline__ void edge_owner(int x, int y, int z, int i, int &n, int &axis, int &lin) {
    const int ox = x + hv_mc_edge_shift[i][0], oy = y + hv_mc_edge_shift[i][1], oz = z + hv_mc_edge_shift[i][2];
    axis = hv_mc_edge_shift[i][3];
    n = (ox >= R ? 1 : 0) | (oy >= R ? 2 : 0) | (oz >= R ? 4 : 0);
    lin = voxel_word(ox & (R - 1), oy & (R - 1), oz & (R - 1));
}

// Column masks of one unit (thread <-> column x * 16 + y; a wave load is one 256-byte run of a plane):
//   m_on[unit * 256 + column]  bits 0-15: weight != 0 at z;             bits 16-31: observed and tsdf < 0
//   m_ip[unit * 256 + column]  bits 0-15: observed, -0.98 <= tsdf < 0.98; bits 16-31: that and tsdf > 0
// (ExtractPointCloud's `f0 * f1 < 0` of two in-range values is "one negative, one positive": the product of two tsdf values
// cannot underflow to zero - a non-zero tsdf is a ratio of pixel-scale floats, never below 1e-23.)
//   unit_signs[unit]           bit 0: the unit holds an observed negative voxel, bit 1: an observed non-negative one
//   mask_stamp[unit]           `now` (the volume's frame counter) of the pass that computed the unit's masks last
// since >= 0: only units written after frame `since` are computed again (touched_stamp of the unit's table slot; the masks of the
// others are the previous pass's) - a tick of a running reconstruction reads the planes of the units its new keyframes touched,
// not of the whole map.
__global__ __launch_bounds__(256) void k_unit_masks(HvTable table, const char *__restrict__ pool, int n_units, uint32_t *__restrict__ m_on,
                                                     uint32_t *__restrict__ m_ip, uint32_t *__restrict__ unit_signs,
                                                     int32_t *__restrict__ mask_stamp, const int32_t *__restrict__ touched_stamp,
                                                     int32_t since, int32_t now) {
    __shared__ uint32_t s_signs;
    __shared__ int s_skip;
    const int idx = blockIdx.x;
    if (idx >= n_units) return;
    if (threadIdx.x == 0) {
        s_signs = 0u;
        int skip = 0;
        if (since >= 0) {
            const int32_t slot = hv_table_find(table, table.block_keys[idx]);
            skip = slot >= 0 && touched_stamp[slot] <= since;
        }
        s_skip = skip;
    }
    __syncthreads();
    if (s_skip) return;
    const char *unit = pool + (int64_t)idx * UNIT_BYTES;
    float t[R];
    uint32_t w[R];
#pragma unroll
    for (int z = 0; z < R; ++z) {
        t[z] = ((const float *)unit)[z * RR + threadIdx.x];
        w[z] = ((const uint32_t *)(unit + PLANE_BYTES))[z * RR + threadIdx.x];
    }
    uint32_t obs = 0u, neg = 0u, inr = 0u, pos = 0u;
#pragma unroll
    for (int z = 0; z < R; ++z) {
        const bool o = w[z] != 0u;
        const bool r = o && t[z] < 0.98f && t[z] >= -0.98f;
        obs |= (o ? 1u : 0u) << z;
        neg |= (o && t[z] < 0.0f ? 1u : 0u) << z;
        inr |= (r ? 1u : 0u) << z;
        pos |= (r && t[z] > 0.0f ? 1u : 0u) << z;
    }
    m_on[(int64_t)idx * RR + threadIdx.x] = obs | (neg << 16);
    m_ip[(int64_t)idx * RR + threadIdx.x] = inr | (pos << 16);
    const uint32_t wave_signs = (__any(neg != 0u) ? 1u : 0u) | (__any((obs & ~neg) != 0u) ? 2u : 0u);
    if (hv_lane_id() == 0 && wave_signs) atomicOr(&s_signs, wave_signs);
    __syncthreads();
    if (threadIdx.x == 0) {
        unit_signs[idx] = s_signs;
        mask_stamp[idx] = now;
    }
}

// Classification of one unit from per-COLUMN bit masks.  Marching cubes only asks two things of a voxel - observed (weight
// != 0) and negative - so a column (cx, cy) of the unit's 18^3 neighbourhood (-1 .. 16 in every direction: the cubes that
// share this unit's edges reach one voxel back, its own cubes one voxel forward) is two 18-bit masks along z.  A thread
// assembles the masks of its column from the published masks of the three units stacked along z (k_unit_masks), the
// 68 halo columns go to the first 68 threads, and 2.6 KB of LDS hold them all.  Then, per thread and with its 3 x 3
// neighbouring column masks in registers:
//   valid cubes of a cube column  V = AND over its four corner columns of (obs & obs >> 1)
//   cube case of (x, y, z)        8 bits picked from the four negative masks (0 if !V or all set) - 16 bytes per thread
//   vertex on the +z edge         (neg ^ neg >> 1) & obs-pair & (a valid cube among the four around the edge)
//   vertex on the +x / +y edge    (neg ^ neg of the next column) & both observed & (a valid cube among the four)
// which is Open3D's "for every valid cube with a mixed case, every crossing edge gets a vertex".  A unit's 192 mask words are
// wave ballots (mc_word: z * 12 + axis * 4 + wave) and are all written: no atomics, nothing to clear; the first wave then
// scans their popcounts (the rank of a vertex inside the unit; round 2 had a kernel of its own for that).  (First form: a 17^3
// float slab in LDS, 8 LDS reads per cube, one global atomicOr per crossing edge and cube - 0.45 of its 0.87 ms per 24 k
// units were those atomics, profiles/r02.)
static constexpr int H2 = 18;
__global__ __launch_bounds__(256) void k_mc_classify(HvTable table, const uint32_t *__restrict__ m_on,
                                                      const uint32_t *__restrict__ unit_signs, int n_units,
                                                      unsigned long long *__restrict__ edge_mask,
                                                      uint32_t *__restrict__ word_prefix, unsigned long long *__restrict__ counts,
                                                      uint8_t *__restrict__ cases, const int32_t *__restrict__ mask_stamp, int32_t since) {
    __shared__ uint32_t s_cnt[MASK_WORDS]; // popcounts of the unit's mask words
    __shared__ uint8_t s_tcnt[256];        // triangles per cube case (a per-lane index into constant memory is a vector load)
    s_tcnt[threadIdx.x] = c_tri_count[threadIdx.x];
    __shared__ int s_nbr[27]; // pool index of the unit at offset (dx, dy, dz) in {-1, 0, 1}^3: [(dx + 1) + 3 (dy + 1) + 9 (dz + 1)]
    __shared__ uint32_t s_obs[H2 * H2], s_neg[H2 * H2]; // [(cx + 1) * 18 + (cy + 1)], bit k <-> z = k - 1
    __shared__ int s_tris;
    __shared__ uint32_t s_signs; // bit 0: an observed negative voxel in the 18 x 18 x 18 neighbourhood, bit 1: an observed non-negative one
    __shared__ uint32_t s_coarse; // the same for the 27 units around (k_unit_masks' summaries)
    const int idx = blockIdx.x;
    if (idx >= n_units) return;
    if (threadIdx.x == 0) {
        s_tris = 0;
        s_signs = 0u;
    }
    if (idx == 0 && threadIdx.x == 1) counts[n_units] = 0ull; // the scan's extra element: its output there is the total
    if (threadIdx.x < 27) {
        int32_t ux, uy, uz;
        hv_unpack_key(table.block_keys[idx], ux, uy, uz);
        const int n = threadIdx.x;
        const int32_t kx = ux + n % 3 - 1, ky = uy + (n / 3) % 3 - 1, kz = uz + n / 9 - 1;
        int r = -1;
        if (n == 13) {
            r = idx;
        } else if (hv_key_in_range(kx, ky, kz)) {
            const int32_t slot = hv_table_find(table, hv_pack_key(kx, ky, kz));
            if (slot >= 0) r = table.vals[slot];
        }
        s_nbr[n] = r;
        // (coarse) the signs the 27 units hold between them
        const uint32_t sg = r >= 0 ? unit_signs[r] : 0u;
        const uint32_t all = (__any(sg & 1u) ? 1u : 0u) | (__any(sg & 2u) ? 2u : 0u);
        // nothing in the neighbourhood has new masks since this unit was classified last: its words, prefixes, cases and counts stand
        const bool fresh = __any(r >= 0 && mask_stamp[r] > since);
        if (n == 0) s_coarse = fresh ? all : 0xffu;
    }
    __syncthreads();
    if (s_coarse == 0xffu) return;
    // A crossing edge and a mixed cube both need an observed negative AND an observed non-negative voxel in the unit's 18 x 18 x 18
    // neighbourhood.  Most allocated units (free space in front of the surface, the far side of the band) see one kind only: no vertex,
    // no triangle, and nothing of their masks / prefixes / cases is ever read (the emit passes return on a zero count; a neighbour asks
    // for this unit's words only for an edge that crosses).  They end here - before the 324 column masks are fetched if the 27 units'
    // summaries say so already, after assembling them otherwise: counts = 0, 7 KB of scratch not written.
    if (s_coarse != 3u) {
        if (threadIdx.x == 0) counts[idx] = 0ull;
        return;
    }
    // ---- column masks ----
    for (int pass = 0; pass < 2; ++pass) {
        int cx, cy;
        if (pass == 0) {
            cx = threadIdx.x >> 4;
            cy = threadIdx.x & 15;
        } else {
            const int h = threadIdx.x; // halo columns: cx = -1 (18), cx = 16 (18), cy = -1 (16), cy = 16 (16)
            if (h >= 68) break;
            if (h < 18) { cx = -1; cy = h - 1; }
            else if (h < 36) { cx = 16; cy = h - 19; }
            else if (h < 52) { cx = h - 36; cy = -1; }
            else { cx = h - 52; cy = 16; }
        }
        const int nxy = (cx < 0 ? 0 : cx >= R ? 2 : 1) + 3 * (cy < 0 ? 0 : cy >= R ? 2 : 1);
        const int col = (cx & (R - 1)) * R + (cy & (R - 1));
        // bit k <-> z = k - 1: bit 15 of the unit below, the 16 bits of the unit at this level, bit 0 of the unit above
        const int nl = s_nbr[nxy], nm = s_nbr[nxy + 9], nh = s_nbr[nxy + 18];
        const uint32_t lo = nl >= 0 ? m_on[(int64_t)nl * RR + col] : 0u;
        const uint32_t mid = nm >= 0 ? m_on[(int64_t)nm * RR + col] : 0u;
        const uint32_t hi = nh >= 0 ? m_on[(int64_t)nh * RR + col] : 0u;
        const uint32_t obs = ((lo >> 15) & 1u) | ((mid & 0xffffu) << 1) | ((hi & 1u) << 17);
        const uint32_t neg = ((lo >> 31) & 1u) | ((mid >> 16) << 1) | (((hi >> 16) & 1u) << 17);
        s_obs[(cx + 1) * H2 + (cy + 1)] = obs;
        s_neg[(cx + 1) * H2 + (cy + 1)] = neg;
        const uint32_t signs = (neg ? 1u : 0u) | ((obs & ~neg) ? 2u : 0u);
        const uint32_t wave_signs = (__any(signs & 1u) ? 1u : 0u) | (__any(signs & 2u) ? 2u : 0u);
        if (hv_lane_id() == 0 && wave_signs) atomicOr(&s_signs, wave_signs);
    }
    __syncthreads();
    if (s_signs != 3u) { // (fine) the neighbourhood itself
        if (threadIdx.x == 0) counts[idx] = 0ull;
        return;
    }
    // ---- classification of column (x, y) ----
    const int x = threadIdx.x >> 4, y = threadIdx.x & 15;
    uint32_t o[3][3], g[3][3]; // [dx + 1][dy + 1]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            o[a][b] = s_obs[(x + a) * H2 + (y + b)];
            g[a][b] = s_neg[(x + a) * H2 + (y + b)];
        }
    auto pairz = [](uint32_t m) { return m & (m >> 1); };                                    // bit k: voxels k and k + 1
    auto cubes = [&](int a, int b) { return pairz(o[a][b]) & pairz(o[a + 1][b]) & pairz(o[a][b + 1]) & pairz(o[a + 1][b + 1]); }; // bit k: cube z = k - 1 of cube column (x + a - 1, y + b - 1)
    const uint32_t V11 = cubes(1, 1), V01 = cubes(0, 1), V10 = cubes(1, 0), V00 = cubes(0, 0);
    // edges owned by voxel (x, y, z): bit z + 1
    const uint32_t Wx = V11 | V10, Wy = V11 | V01;
    const uint32_t Ex = (g[1][1] ^ g[2][1]) & o[1][1] & o[2][1] & (Wx | (Wx << 1));
    const uint32_t Ey = (g[1][1] ^ g[1][2]) & o[1][1] & o[1][2] & (Wy | (Wy << 1));
    const uint32_t Ez = (g[1][1] ^ (g[1][1] >> 1)) & pairz(o[1][1]) & (V11 | V01 | V10 | V00);
    // this unit's mask words: mc_word = z * 12 + axis * 4 + wave, bit = lane (= voxel_word & 63)
    const int lane = hv_lane_id(), wave = threadIdx.x >> 6;
    unsigned long long mine = 0ull; // lane k keeps ballot k (k = axis * 16 + z)
#pragma unroll
    for (int z = 0; z < R; ++z) {
        const unsigned long long bx = __ballot((Ex >> (z + 1)) & 1u), by = __ballot((Ey >> (z + 1)) & 1u), bz = __ballot((Ez >> (z + 1)) & 1u);
        if (lane == z) mine = bx;
        if (lane == 16 + z) mine = by;
        if (lane == 32 + z) mine = bz;
    }
    if (lane < 48) {
        const int word = (lane & 15) * 12 + (lane >> 4) * 4 + wave; // mc_word(axis = lane >> 4, z = lane & 15, quarter = wave)
        edge_mask[(int64_t)idx * MASK_WORDS + word] = mine;
        s_cnt[word] = (uint32_t)__popcll(mine);
    }
    // cube cases of the column (corner i of hv_mc_shift: columns (x + sx, y + sy), voxel z + sz)
    const uint32_t c0 = g[1][1] >> 1, c1 = g[2][1] >> 1, c2 = g[2][2] >> 1, c3 = g[1][2] >> 1; // bit z <-> voxel z
    int tris = 0;
    uint32_t packed[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int z = 0; z < R; ++z) {
        int cube = (int)(((c0 >> z) & 1u) | (((c1 >> z) & 1u) << 1) | (((c2 >> z) & 1u) << 2) | (((c3 >> z) & 1u) << 3) |
                         (((c0 >> (z + 1)) & 1u) << 4) | (((c1 >> (z + 1)) & 1u) << 5) | (((c2 >> (z + 1)) & 1u) << 6) |
                         (((c3 >> (z + 1)) & 1u) << 7));
        if (!((V11 >> (z + 1)) & 1u) || cube == 255) cube = 0;
        packed[z >> 2] |= (uint32_t)cube << ((z & 3) * 8);
        tris += s_tcnt[cube];
    }
    ((uint4 *)(cases + (int64_t)idx * RRR))[threadIdx.x] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    if (tris) atomicAdd(&s_tris, tris);
    __syncthreads();
    if (wave == 0) { // exclusive popcount prefix over the 192 mask words: three words per lane + a wave scan
        const uint32_t c0 = s_cnt[lane * 3], c1 = s_cnt[lane * 3 + 1], c2 = s_cnt[lane * 3 + 2];
        uint32_t incl = c0 + c1 + c2;
#pragma unroll
        for (int o = 1; o < HV_WAVE; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const uint32_t before = incl - (c0 + c1 + c2);
        uint32_t *wp = word_prefix + (int64_t)idx * MASK_WORDS + lane * 3;
        wp[0] = before;
        wp[1] = before + c0;
        wp[2] = before + c0 + c1;
        // vertices in the low half, triangles in the high half: ONE 64-bit scan gives both bases
        if (lane == HV_WAVE - 1) counts[idx] = (unsigned long long)incl | ((unsigned long long)(uint32_t)s_tris << 32);
    }
}

struct HvMcParams {
    double voxel_length, half_voxel_length;
};

// One WAVE per unit.  A unit has ~120 vertices among 12 288 candidate edges, in ~40 of its 192 mask words: a thread per word
// (second version) left most lanes idle behind a few that walked two or three vertices, one dependent round trip after the
// other.  Here the unit's masks and prefixes go to LDS, vertex r of the unit is found by a binary search over the word
// prefixes plus a select-the-nth-set-bit, and lane j builds vertices j, j + 64, ... - every lane busy, all their loads in
// flight together, and four times as many units resident per CU.
__device__ __forceinline__ int hv_nth_set_bit(unsigned long long m, int n) { // position of the n-th (0-based) set bit of m
    int pos = 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const int c = __popcll((m >> pos) & ((1ull << s) - 1ull));
        if (n >= c) {
            n -= c;
            pos += s;
        }
    }
    return pos;
}

// OUT = double: Open3D's arrays; OUT = float: the same float64 values rounded once on the way out (hv_tsdf_extract_mesh_f32 - what
// pySLAM's viewer casts them to, config_parameters.py:290-291; half the bytes written here and half the bytes across PCIe).
template <typename OUT>
__global__ __launch_bounds__(64) void k_mc_vertices(HvTable table, const char *__restrict__ pool, int n_units,
                                                     const unsigned long long *__restrict__ edge_mask,
                                                     const uint32_t *__restrict__ word_prefix,
                                                     const unsigned long long *__restrict__ bases, HvMcParams M,
                                                     OUT *__restrict__ vertices, OUT *__restrict__ colors,
                                                     int64_t cap) {
    __shared__ unsigned long long s_mask[MASK_WORDS];
    __shared__ uint32_t s_prefix[MASK_WORDS + 1];
    const int idx = blockIdx.x;
    if (idx >= n_units) return;
    const int lane = threadIdx.x;
    const int64_t vbase = (int64_t)(uint32_t)bases[idx];
    const int total = (int)((uint32_t)bases[idx + 1] - (uint32_t)vbase);
    if (total == 0) return; // most allocated units hold no surface
    for (int w = lane; w < MASK_WORDS; w += HV_WAVE) {
        s_mask[w] = edge_mask[(int64_t)idx * MASK_WORDS + w];
        s_prefix[w] = word_prefix[(int64_t)idx * MASK_WORDS + w];
    }
    if (lane == 0) s_prefix[MASK_WORDS] = (uint32_t)total;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int32_t ux, uy, uz;
    hv_unpack_key(table.block_keys[idx], ux, uy, uz);
    const char *u0 = pool + (int64_t)idx * UNIT_BYTES;
    for (int r = lane; r < total; r += HV_WAVE) {
        const int64_t vi = vbase + r;
        if (vi >= cap) return;
        // the word holding vertex r: the last one whose exclusive prefix is <= r (empty words share their successor's prefix)
        int lo = 0, hi = MASK_WORDS; // prefix[lo] <= r < prefix[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_prefix[mid] <= (uint32_t)r) lo = mid; else hi = mid;
        }
        const int word = lo;
        const int axis = (word % 12) >> 2;
        const int lin = ((word / 12) * 4 + (word & 3)) * 64 + hv_nth_set_bit(s_mask[word], r - (int)s_prefix[word]);
        // owner voxel and its +axis neighbour
        const int z = lin / RR, x = (lin / R) % R, y = lin % R;
        int nx = x + (axis == 0), ny = y + (axis == 1), nz = z + (axis == 2);
        int nidx = idx;
        if (nx >= R || ny >= R || nz >= R) {
            const int32_t slot = hv_table_find(table, hv_pack_key(ux + (nx >= R), uy + (ny >= R), uz + (nz >= R)));
            nidx = slot >= 0 ? table.vals[slot] : -1;
            nx &= R - 1; ny &= R - 1; nz &= R - 1;
        }
        const double f0 = fabs((double)((const float *)u0)[lin]);
        const double w0 = (double)((const uint32_t *)(u0 + PLANE_BYTES))[lin];
        double c0[3], c1[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k) c0[k] = ((double)((const uint32_t *)(u0 + (2 + k) * PLANE_BYTES))[lin] / w0) / 255.0;
        double f1 = 0.0;
        if (nidx >= 0) {
            const char *u1 = pool + (int64_t)nidx * UNIT_BYTES;
            const int nl = voxel_word(nx, ny, nz);
            f1 = fabs((double)((const float *)u1)[nl]);
            const double w1 = (double)((const uint32_t *)(u1 + PLANE_BYTES))[nl];
#pragma unroll
            for (int k = 0; k < 3; ++k) c1[k] = ((double)((const uint32_t *)(u1 + (2 + k) * PLANE_BYTES))[nl] / w1) / 255.0;
        }
        double pt[3] = {M.half_voxel_length + M.voxel_length * (double)(ux * R + x),
                        M.half_voxel_length + M.voxel_length * (double)(uy * R + y),
                        M.half_voxel_length + M.voxel_length * (double)(uz * R + z)};
        pt[axis] += f0 * M.voxel_length / (f0 + f1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            vertices[vi * 3 + k] = (OUT)pt[k];
            colors[vi * 3 + k] = (OUT)((f1 * c0[k] + f0 * c1[k]) / (f0 + f1));
        }
    }
}

// Reads the cube cases k_mc_classify left (16 bytes per thread: its column) instead of staging the slab again.  Everything a
// triangle asks for repeatedly sits in LDS: the triangle table and the per-case counts (a per-lane index into constant memory
// is a vector load), each column's 16 counts as nibbles (the walk to a triangle's cube is register arithmetic), and the
// unit's own edge masks / prefixes (five of six vertex ids are the unit's own; the rest read the neighbour's words).
__global__ __launch_bounds__(256) void k_mc_triangles(HvTable table, int n_units, const uint8_t *__restrict__ cases,
                                                       const unsigned long long *__restrict__ edge_mask,
                                                       const uint32_t *__restrict__ word_prefix,
                                                       const unsigned long long *__restrict__ bases,
                                                       int32_t *__restrict__ triangles, int64_t cap) {
    __shared__ int s_nbr[8];
    __shared__ int32_t s_vbase[8];
    __shared__ int s_wave[4];
    __shared__ uint8_t s_tcnt[256];
    __shared__ signed char s_tt[256 * 16];
    __shared__ unsigned long long s_mask[MASK_WORDS];
    __shared__ uint32_t s_pref[MASK_WORDS];
    __shared__ uint4 s_cases[256];
    __shared__ unsigned long long s_nib[256];
    __shared__ int s_pre[257];
    const int idx = blockIdx.x;
    if (idx >= n_units) return;
    const int64_t tbase = (int64_t)(bases[idx] >> 32);
    const int total = (int)((bases[idx + 1] >> 32) - (unsigned long long)tbase);
    if (total == 0) return; // most allocated units hold no surface
    load_neighbours(table, idx, s_nbr);
    s_tcnt[threadIdx.x] = c_tri_count[threadIdx.x];
    ((uint4 *)s_tt)[threadIdx.x] = ((const uint4 *)&c_tri_table[0][0])[threadIdx.x];
    if (threadIdx.x < MASK_WORDS) {
        s_mask[threadIdx.x] = edge_mask[(int64_t)idx * MASK_WORDS + threadIdx.x];
        s_pref[threadIdx.x] = word_prefix[(int64_t)idx * MASK_WORDS + threadIdx.x];
    }
    const uint4 pk = ((const uint4 *)(cases + (int64_t)idx * RRR))[threadIdx.x];
    s_cases[threadIdx.x] = pk;
    __syncthreads();
    if (threadIdx.x < 8) s_vbase[threadIdx.x] = s_nbr[threadIdx.x] >= 0 ? (int32_t)(uint32_t)bases[s_nbr[threadIdx.x]] : 0;
    const uint32_t packed[4] = {pk.x, pk.y, pk.z, pk.w};
    int tris = 0;
    unsigned long long nib = 0ull; // nibble z: triangles of the column's cube z (at most 5)
#pragma unroll
    for (int z = 0; z < R; ++z) {
        const int c = s_tcnt[(packed[z >> 2] >> ((z & 3) * 8)) & 255u];
        tris += c;
        nib |= (unsigned long long)c << (4 * z);
    }
    s_nib[threadIdx.x] = nib;
    // rank of this thread's triangles inside the unit: wave prefix + the 4 wave totals
    const int lane = hv_lane_id(), wave = threadIdx.x >> 6;
    int incl = tris;
#pragma unroll
    for (int o = 1; o < HV_WAVE; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == HV_WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    // The unit's triangles are dealt out evenly (as the vertices and the points are): triangle r belongs to the column whose
    // exclusive count is the last one <= r; inside the column it is found by walking the 16 counts.
    s_pre[threadIdx.x] = before + incl - tris;
    if (threadIdx.x == 255) s_pre[256] = before + incl;
    __syncthreads();
    for (int r = threadIdx.x; r < total; r += 256) {
        const int64_t at = tbase + r;
        int lo = 0, hi = 256; // s_pre[lo] <= r < s_pre[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pre[mid] <= r) lo = mid; else hi = mid;
        }
        const int x = lo >> 4, y = lo & 15;
        unsigned long long cw = s_nib[lo];
        int j = r - s_pre[lo], z = 0;
        for (; z < R - 1; ++z) {
            const int c = (int)(cw & 15ull);
            if (j < c) break;
            j -= c;
            cw >>= 4;
        }
        const int cube = ((const uint8_t *)&s_cases[lo])[z];
        const int i = 3 * j;
        const int order[3] = {i, i + 2, i + 1}; // Open3D emits (e[i], e[i+2], e[i+1])
        int32_t vid[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int n, axis, lin;
            edge_owner(x, y, z, s_tt[cube * 16 + order[k]], n, axis, lin);
            const int word = mc_word(axis, lin);
            unsigned long long m;
            uint32_t pre;
            if (n == 0) {
                m = s_mask[word];
                pre = s_pref[word];
            } else {
                const int oidx = s_nbr[n];
                m = edge_mask[(int64_t)oidx * MASK_WORDS + word];
                pre = word_prefix[(int64_t)oidx * MASK_WORDS + word];
            }
            vid[k] = s_vbase[n] + (int32_t)pre + (int32_t)__popcll(m & ((1ull << (lin & 63)) - 1));
        }
        if (at < cap) {
            triangles[at * 3 + 0] = vid[0];
            triangles[at * 3 + 1] = vid[1];
            triangles[at * 3 + 2] = vid[2];
        }
    }
}

// ScalableTSDFVolume::ExtractPointCloud: a voxel (weight != 0, tsdf in [-0.98, 0.98)) and its +x / +y / +z neighbour of the
// same kind with the opposite sign give one interpolated point.  One workgroup per unit, one thread per (x, y) column: the
// crossings of a column follow from its masks and those of the columns (x + 1, y), (x, y + 1) and of the unit above
// (k_unit_masks) - 16 bits per axis.  Pass 1 (FILL = false) counts the unit's points; after an exclusive scan over the
// units pass 2 finds them again and writes every point at its final index, fetching tsdf / weight / colour of the two voxels
// of a crossing only - no atomics, and the output order is deterministic (unit, column, axis, z).  (Second form: both passes
// staged a 17^3 float slab of the planes, 0.38 + 0.53 ms and 5.6 GB of traffic per 32 k units, profiles/r03 before this
// form.  First form: one thread per voxel, a workgroup-aggregated append - 1.43 ms per pass, profiles/r02.)
template <bool FILL, typename OUT = double>
__global__ __launch_bounds__(256) void k_pc_extract(HvTable table, const char *__restrict__ pool, const uint32_t *__restrict__ m_on,
                                                     const uint32_t *__restrict__ m_ip, int n_units, HvMcParams M,
                                                     double unit_length, int32_t *__restrict__ count,
                                                     const int32_t *__restrict__ base, OUT *__restrict__ points,
                                                     OUT *__restrict__ colors, int64_t cap, const int32_t *__restrict__ mask_stamp, int32_t since) {
    __shared__ int s_nbr[8];
    __shared__ int s_wave[4];
    const int idx = blockIdx.x;
    if (idx >= n_units) return;
    if (FILL && base[idx + 1] == base[idx]) return; // most allocated units hold no surface
    load_neighbours(table, idx, s_nbr);
    __syncthreads();
    if (!FILL) { // the count of a unit whose own masks and those of its +x / +y / +z units are the previous pass's stands
        bool fresh = false;
#pragma unroll
        for (int n : {0, 1, 2, 4}) fresh = fresh || (s_nbr[n] >= 0 && mask_stamp[s_nbr[n]] > since);
        if (!fresh) return;
    }
    const int x = threadIdx.x >> 4, y = threadIdx.x & 15;
    // in-range / negative / positive masks (bit z) of column `col` of neighbour n (bit0 = +x, bit1 = +y, bit2 = +z)
    auto column = [&](int n, int col, uint32_t &inr, uint32_t &neg, uint32_t &pos) {
        const int nb = s_nbr[n];
        uint32_t ip = 0u, on = 0u;
        if (nb >= 0) {
            ip = m_ip[(int64_t)nb * RR + col];
            on = m_on[(int64_t)nb * RR + col];
        }
        inr = ip & 0xffffu;
        pos = ip >> 16;
        neg = inr & (on >> 16);
    };
    uint32_t ia, na, pa, ib, nb_, pb, ic, nc, pc, id, nd, pd;
    column(0, (int)threadIdx.x, ia, na, pa);
    column(x == R - 1 ? 1 : 0, ((x + 1) & (R - 1)) * R + y, ib, nb_, pb);
    column(y == R - 1 ? 2 : 0, x * R + ((y + 1) & (R - 1)), ic, nc, pc);
    column(4, (int)threadIdx.x, id, nd, pd); // z = 16: bit 0 of the unit above
    const uint32_t hx = ia & ib & ((na & pb) | (pa & nb_));
    const uint32_t hy = ia & ic & ((na & pc) | (pa & nc));
    const uint32_t i17 = ia | ((id & 1u) << 16), n17 = na | ((nd & 1u) << 16), p17 = pa | ((pd & 1u) << 16);
    const uint32_t hz = i17 & (i17 >> 1) & ((n17 & (p17 >> 1)) | (p17 & (n17 >> 1))) & 0xffffu;
    // hits of this column: bit axis * 16 + z
    const unsigned long long hits = (unsigned long long)hx | ((unsigned long long)hy << 16) | ((unsigned long long)hz << 32);
    const int mine = __popcll(hits);
    const int lane = hv_lane_id(), wave = threadIdx.x >> 6;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < HV_WAVE; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == HV_WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (!FILL) {
        if (threadIdx.x == 255) count[idx] = before + incl;
        return;
    }
    // The unit's points are dealt out evenly: thread r builds point r, r + 256, ... (a column holds one to six crossings, and
    // a thread that walks its own would chase six dependent round trips while most of the workgroup idles).  Point r belongs
    // to the column whose exclusive count is the last one <= r: binary search over the column prefixes, then select the
    // (r - prefix)-th set bit of that column's hit mask.
    __shared__ unsigned long long s_hits[256];
    __shared__ int s_pre[257];
    s_hits[threadIdx.x] = hits;
    s_pre[threadIdx.x] = before + incl - mine;
    if (threadIdx.x == 255) s_pre[256] = before + incl;
    __syncthreads();
    const int total = s_pre[256];
    if (total == 0) return;
    int32_t ux, uy, uz;
    hv_unpack_key(table.block_keys[idx], ux, uy, uz);
    const char *u0 = pool + (int64_t)idx * UNIT_BYTES;
    const int64_t vbase = base[idx];
    for (int r = threadIdx.x; r < total; r += 256) {
        const int64_t at = vbase + r;
        if (at >= cap) return;
        int lo = 0, hi = 256; // s_pre[lo] <= r < s_pre[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pre[mid] <= r) lo = mid; else hi = mid;
        }
        const int x = lo >> 4, y = lo & 15;
        const int bit = hv_nth_set_bit(s_hits[lo], r - s_pre[lo]);
        const int i = bit >> 4, z = bit & 15;
        const int lin = voxel_word(x, y, z);
        const float f0 = ((const float *)u0)[lin];
        const uint32_t w0 = ((const uint32_t *)(u0 + PLANE_BYTES))[lin];
        int nx = x + (i == 0), ny = y + (i == 1), nz = z + (i == 2);
        const int ni = s_nbr[(nx >= R ? 1 : 0) | (ny >= R ? 2 : 0) | (nz >= R ? 4 : 0)];
        nx &= R - 1; ny &= R - 1; nz &= R - 1;
        const char *u1 = pool + (int64_t)ni * UNIT_BYTES;
        const int l1 = voxel_word(nx, ny, nz);
        const float f1 = ((const float *)u1)[l1];
        const double w1 = (double)((const uint32_t *)(u1 + PLANE_BYTES))[l1];
        float c0[3], c1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { // color_.cast<float>() of the double running mean
            c0[k] = (float)((double)((const uint32_t *)(u0 + (2 + k) * PLANE_BYTES))[lin] / (double)w0);
            c1[k] = (float)((double)((const uint32_t *)(u1 + (2 + k) * PLANE_BYTES))[l1] / w1);
        }
        const double p0[3] = {(M.half_voxel_length + M.voxel_length * (double)x) + (double)ux * unit_length,
                              (M.half_voxel_length + M.voxel_length * (double)y) + (double)uy * unit_length,
                              (M.half_voxel_length + M.voxel_length * (double)z) + (double)uz * unit_length};
        const float r0 = fabsf(f0), r1 = fabsf(f1);
        double p[3] = {p0[0], p0[1], p0[2]};
        const double p1i = p0[i] + M.voxel_length;
        p[i] = (p0[i] * (double)r1 + p1i * (double)r0) / (double)(r0 + r1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            points[at * 3 + k] = (OUT)p[k];
            colors[at * 3 + k] = (OUT)(double)((c0[k] * r1 + c1[k] * r0) / (r0 + r1) / 255.0f);
        }
    }
}

// ScalableTSDFVolume::GetNormalAt for every extracted point (the normals Open3D's point cloud carries into dense_map.ply,
// volumetric_integrator_tsdf.py:246-247): central differences, at +/- 0.99 voxel along each axis, of GetTSDFAt = the trilinear
// interpolation of the eight voxels around a position.  GetTSDFAt does not look at weights (a voxel never observed holds its
// initial tsdf 0) and a missing unit contributes 0.  One thread per point; the unit of the previous fetch is remembered (most
// of the 48 fetches of a point fall into one or two units), the arithmetic is Open3D's, in double.
__device__ __forceinline__ float hv_tsdf_voxel(const HvTable &table, const char *__restrict__ pool, int32_t ux, int32_t uy, int32_t uz,
                                               int x, int y, int z, unsigned long long &cached_key, int32_t &cached_idx) {
    if (!hv_key_in_range(ux, uy, uz)) return 0.0f;
    const unsigned long long key = hv_pack_key(ux, uy, uz);
    if (key != cached_key) {
        const int32_t slot = hv_table_find(table, key);
        cached_key = key;
        cached_idx = slot >= 0 ? table.vals[slot] : -1;
    }
    if (cached_idx < 0) return 0.0f;
    return ((const float *)(pool + (int64_t)cached_idx * UNIT_BYTES))[voxel_word(x, y, z)];
}

__device__ inline double hv_tsdf_at(const HvTable &table, const char *__restrict__ pool, double voxel_length, double unit_length,
                                    const double *p, unsigned long long &ck, int32_t &ci) {
    int32_t index0[3];
    int idx0[3];
    double r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double p_locate = p[i] - 0.5 * voxel_length;
        index0[i] = (int32_t)floor(p_locate / unit_length);
        const double p_grid = (p_locate - (double)index0[i] * unit_length) / voxel_length;
        int q = (int)floor(p_grid);
        q = q < 0 ? 0 : (q >= R ? R - 1 : q);
        idx0[i] = q;
        r[i] = p_grid - (double)q;
    }
    {   // the unit of p itself decides "no such unit -> 0" (GetTSDFAt returns before looking at neighbours)
        unsigned long long k0 = HV_EMPTY_KEY;
        int32_t i0 = -1;
        (void)hv_tsdf_voxel(table, pool, index0[0], index0[1], index0[2], 0, 0, 0, k0, i0);
        if (i0 < 0) return 0.0;
        ck = k0;
        ci = i0;
    }
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int sx = (i == 1 || i == 2 || i == 5 || i == 6), sy = (i == 2 || i == 3 || i == 6 || i == 7), sz = i >= 4;
        int x = idx0[0] + sx, y = idx0[1] + sy, z = idx0[2] + sz;
        const int32_t ux = index0[0] + (x >= R), uy = index0[1] + (y >= R), uz = index0[2] + (z >= R);
        f[i] = hv_tsdf_voxel(table, pool, ux, uy, uz, x & (R - 1), y & (R - 1), z & (R - 1), ck, ci);
    }
    return (1 - r[0]) * ((1 - r[1]) * ((1 - r[2]) * f[0] + r[2] * f[4]) + r[1] * ((1 - r[2]) * f[3] + r[2] * f[7])) +
           r[0] * ((1 - r[1]) * ((1 - r[2]) * f[1] + r[2] * f[5]) + r[1] * ((1 - r[2]) * f[2] + r[2] * f[6]));
}

__global__ __launch_bounds__(256) void k_pc_normals(HvTable table, const char *__restrict__ pool, double voxel_length, double unit_length,
                                                     const double *__restrict__ points, int64_t n, double *__restrict__ normals) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double half_gap = 0.99 * voxel_length;
    unsigned long long ck = HV_EMPTY_KEY;
    int32_t ci = -1;
    double nn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double p0[3] = {points[k * 3], points[k * 3 + 1], points[k * 3 + 2]}, p1[3] = {p0[0], p0[1], p0[2]};
        p0[i] -= half_gap;
        p1[i] += half_gap;
        nn[i] = hv_tsdf_at(table, pool, voxel_length, unit_length, p1, ck, ci) - hv_tsdf_at(table, pool, voxel_length, unit_length, p0, ck, ci);
    }
    const double z = nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2];
    const double s = sqrt(z);
#pragma unroll
    for (int i = 0; i < 3; ++i) normals[k * 3 + i] = z > 0.0 ? nn[i] / s : nn[i]; // Eigen's normalized(): the zero vector stays zero
}

// ------------------------------------------------------------------------------------------------
static bool g_tables_uploaded[64] = {false};
// HV_EXTRACT_INCREMENTAL=0: every extraction recomputes masks, classification and counts of ALL units (rounds 2-5; A/B and tests)
static bool hv_extract_full() { return getenv("HV_EXTRACT_INCREMENTAL") && atoi(getenv("HV_EXTRACT_INCREMENTAL")) == 0; }

static int upload_tables(int device) {
    if (device < 64 && g_tables_uploaded[device]) return HV_OK;
    unsigned char counts[256];
    for (int c = 0; c < 256; ++c) {
        int n = 0;
        while (n < 16 && hv_mc_tri_table[c][n] != -1) n += 3;
        counts[c] = (unsigned char)(n / 3);
    }
    HV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_edge_table), hv_mc_edge_table, sizeof(hv_mc_edge_table)));
    HV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tri_table), hv_mc_tri_table, sizeof(hv_mc_tri_table)));
    HV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tri_count), counts, sizeof(counts)));
    if (device < 64) g_tables_uploaded[device] = true;
    return HV_OK;
}

static int exclusive_scan_i32(hv_volume *v, int32_t *in, int32_t *out, int n) {
    size_t bytes = 0;
    HV_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, (size_t)n, rocprim::plus<int32_t>(), v->stream));
    int rc = hv_ensure_buffer(v, &v->sort_tmp, &v->sort_tmp_bytes, bytes);
    if (rc != HV_OK) return rc;
    bytes = v->sort_tmp_bytes;
    HV_HIP(rocprim::exclusive_scan(v->sort_tmp, bytes, in, out, 0, (size_t)n, rocprim::plus<int32_t>(), v->stream));
    return HV_OK;
}

static int exclusive_scan_u64(hv_volume *v, unsigned long long *in, unsigned long long *out, int n) {
    size_t bytes = 0;
    HV_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), v->stream));
    int rc = hv_ensure_buffer(v, &v->sort_tmp, &v->sort_tmp_bytes, bytes);
    if (rc != HV_OK) return rc;
    bytes = v->sort_tmp_bytes;
    HV_HIP(rocprim::exclusive_scan(v->sort_tmp, bytes, in, out, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), v->stream));
    return HV_OK;
}

extern "C" {

// Both extraction entry points follow the "sizes first, data second" protocol of the reference binding (numpy arrays are
// allocated by the caller).  The call without output pointers does ALL the device work into out_a / out_b and records it
// under the volume's content_version; the call with pointers then only copies - unless the volume changed in between (or
// no size query preceded it), in which case it recomputes first.
// k_unit_masks for the volume's current contents (kept under content_version: a tick that extracts the mesh AND the point
// cloud of the same contents computes them once)
// The three per-unit caches (hv_common.h: incremental extraction) are laid out for unit_cache_cap units; a pool that outgrows the
// layout gets new buffers and one full pass.
static constexpr size_t MC_UNIT_BYTES = sizeof(uint64_t) * MASK_WORDS + sizeof(uint32_t) * MASK_WORDS + RRR;
static int unit_caches_ensure(hv_volume *v, int n) {
    if (n <= v->unit_cache_cap && v->unit_masks != nullptr) return HV_OK;
    int cap = 1024;
    while (cap < n) cap <<= 1;
    HV_HIP(hipStreamSynchronize(v->stream));
    for (void **p : {&v->unit_masks, &v->mc_cache, &v->pc_cache}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    v->unit_cache_cap = 0;
    v->unit_masks_version = 0;
    v->unit_masks_epoch = v->mc_epoch = v->pc_epoch = 0; // (no epoch is 0: everything is computed in full)
    // masks: [m_on cap*256][m_ip cap*256][unit_signs cap][mask_stamp cap]
    HV_HIP(hipMalloc(&v->unit_masks, sizeof(uint32_t) * (size_t)cap * (2 * RR + 2)));
    HV_HIP(hipMalloc(&v->mc_cache, MC_UNIT_BYTES * (size_t)cap + sizeof(uint64_t) * ((size_t)cap + 1) + 256));
    HV_HIP(hipMalloc(&v->pc_cache, sizeof(int32_t) * ((size_t)cap + 1)));
    v->unit_cache_cap = cap;
    return HV_OK;
}
struct HvUnitMasks {
    const uint32_t *m_on, *m_ip, *signs;
    const int32_t *stamp;
};
// k_unit_masks for the volume's current contents (kept under content_version: a tick that extracts the mesh AND the point cloud of
// the same contents computes them once), over the units written since the previous pass
static int unit_masks_compute(hv_volume *v, int n, HvUnitMasks *out) {
    int rc = unit_caches_ensure(v, n);
    if (rc != HV_OK) return rc;
    const size_t cap = (size_t)v->unit_cache_cap;
    uint32_t *on = (uint32_t *)v->unit_masks, *ip = on + (size_t)RR * cap, *signs = ip + (size_t)RR * cap;
    int32_t *stamp = (int32_t *)(signs + cap);
    if (v->unit_masks_version != v->content_version || v->unit_masks_units != n) {
        const bool full = v->unit_masks_epoch != v->extract_epoch || v->touched_stamp == nullptr || n < v->unit_masks_units || hv_extract_full();
        hipLaunchKernelGGL(k_unit_masks, dim3(n), dim3(256), 0, v->stream, v->table, (const char *)v->pool, n, on, ip, signs, stamp,
                           (const int32_t *)v->touched_stamp, full ? (int32_t)-1 : v->unit_masks_stamp, v->frame_counter);
        HV_HIP(hipGetLastError());
        v->unit_masks_version = v->content_version;
        v->unit_masks_epoch = v->extract_epoch;
        v->unit_masks_stamp = v->frame_counter;
        v->unit_masks_units = n;
    }
    *out = HvUnitMasks{on, ip, signs, stamp};
    return HV_OK;
}

static int mesh_compute(hv_volume *v, bool f32) {
    int rc = upload_tables(v->device);
    if (rc != HV_OK) return rc;
    int64_t nb = 0;
    rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    v->mesh_cache_version = 0;
    v->points_cache_version = 0; // shares out_a
    v->mesh_cache_nv = v->mesh_cache_nt = 0;
    if (nb == 0) {
        v->mesh_cache_version = v->content_version;
        return HV_OK;
    }
    const int n = (int)nb;
    hv_profile_begin(v); // measurement hook: column masks + classify + scan
    HvUnitMasks UM;
    rc = unit_masks_compute(v, n, &UM);
    if (rc != HV_OK) return rc;
    // per-unit cache: [edge_mask cap*192 u64][counts cap+1 u64][word_prefix cap*192 u32][cases cap*4096 u8]; scratch: [bases n+1 u64]
    // counts / bases: vertices in the low 32 bits, triangles in the high 32 bits of one word per unit
    const size_t cap = (size_t)v->unit_cache_cap;
    const size_t mask_bytes = sizeof(uint64_t) * MASK_WORDS * cap;
    const size_t cnt_bytes = sizeof(uint64_t) * (cap + 1);
    const size_t prefix_bytes = sizeof(uint32_t) * MASK_WORDS * cap;
    const size_t cases_off = (mask_bytes + cnt_bytes + prefix_bytes + 255) & ~(size_t)255;
    rc = hv_ensure_buffer(v, &v->out_c, &v->out_c_bytes, sizeof(uint64_t) * (size_t)(n + 1));
    if (rc != HV_OK) return rc;
    char *base = (char *)v->mc_cache;
    unsigned long long *edge_mask = (unsigned long long *)base;
    unsigned long long *counts = (unsigned long long *)(base + mask_bytes);
    unsigned long long *bases = (unsigned long long *)v->out_c;
    uint32_t *word_prefix = (uint32_t *)(base + mask_bytes + cnt_bytes);
    uint8_t *cases = (uint8_t *)(base + cases_off);
    const bool mc_full = v->mc_epoch != v->extract_epoch || n < v->mc_units || hv_extract_full();
    hipLaunchKernelGGL(k_mc_classify, dim3(n), dim3(256), 0, v->stream, v->table, UM.m_on, UM.signs, n, edge_mask, word_prefix, counts, cases,
                       UM.stamp, mc_full ? (int32_t)-1 : v->mc_stamp);
    HV_HIP(hipGetLastError());
    v->mc_epoch = v->extract_epoch;
    v->mc_stamp = v->frame_counter;
    v->mc_units = n;
    rc = exclusive_scan_u64(v, counts, bases, n + 1);
    if (rc != HV_OK) return rc;
    hv_profile_end(v, n);
    unsigned long long total64 = 0ull;
    HV_HIP(hipMemcpyAsync(&total64, bases + n, sizeof(total64), hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    const int64_t totals[2] = {(int64_t)(uint32_t)total64, (int64_t)(total64 >> 32)};
    const int64_t nv = totals[0], nt = totals[1];
    if (nv > 0 || nt > 0) {
        rc = hv_ensure_buffer(v, &v->out_a, &v->out_a_bytes, (f32 ? sizeof(float) : sizeof(double)) * 6 * (size_t)std::max<int64_t>(nv, 1));
        if (rc != HV_OK) return rc;
        rc = hv_ensure_buffer(v, &v->out_b, &v->out_b_bytes, sizeof(int32_t) * 3 * (size_t)std::max<int64_t>(nt, 1));
        if (rc != HV_OK) return rc;
        HvMcParams M{v->cfg.voxel_size, v->cfg.voxel_size * 0.5};
        hv_profile_begin(v); // vertices + triangles (D2H of the results is outside the bracket)
        if (f32) {
            float *d_vert = (float *)v->out_a, *d_col = d_vert + 3 * nv;
            hipLaunchKernelGGL(k_mc_vertices<float>, dim3(n), dim3(64), 0, v->stream, v->table,
                               (const char *)v->pool, n, edge_mask, word_prefix, bases, M, d_vert, d_col, nv);
        } else {
            double *d_vert = (double *)v->out_a, *d_col = d_vert + 3 * nv;
            hipLaunchKernelGGL(k_mc_vertices<double>, dim3(n), dim3(64), 0, v->stream, v->table,
                               (const char *)v->pool, n, edge_mask, word_prefix, bases, M, d_vert, d_col, nv);
        }
        hipLaunchKernelGGL(k_mc_triangles, dim3(n), dim3(256), 0, v->stream, v->table, n, (const uint8_t *)cases, edge_mask,
                           word_prefix, bases, (int32_t *)v->out_b, nt);
        hv_profile_end(v, n);
        HV_HIP(hipGetLastError());
    }
    v->mesh_cache_nv = nv;
    v->mesh_cache_nt = nt;
    v->mesh_cache_f32 = f32;
    v->mesh_cache_version = v->content_version;
    return HV_OK;
}

// OUT = double: hv_tsdf_extract_mesh; OUT = float: hv_tsdf_extract_mesh_f32.  A result cached in the other type is computed again
// (the per-unit caches stand: only the vertex and triangle passes run).
} // extern "C"
template <typename OUT>
static int extract_mesh_as(hv_volume *v, OUT *vertices, OUT *vertex_colors, int64_t cap_vertices, int32_t *triangles, int64_t cap_triangles,
                           int64_t *n_vertices, int64_t *n_triangles, const char *who) {
    HV_REQUIRE(v != nullptr && n_vertices != nullptr && n_triangles != nullptr, HV_ERR_INVALID, "%s: null argument", who);
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "%s: volume is not in TSDF mode", who);
    HV_HIP(hipSetDevice(v->device));
    constexpr bool f32 = sizeof(OUT) == sizeof(float);
    if (v->mesh_cache_version != v->content_version || v->mesh_cache_f32 != f32) {
        const int rc = mesh_compute(v, f32);
        if (rc != HV_OK) return rc;
    }
    *n_vertices = v->mesh_cache_nv;
    *n_triangles = v->mesh_cache_nt;
    if (vertices == nullptr || vertex_colors == nullptr || triangles == nullptr) return HV_OK;
    const int64_t nv = std::min<int64_t>(v->mesh_cache_nv, cap_vertices), nt = std::min<int64_t>(v->mesh_cache_nt, cap_triangles);
    const OUT *d_vert = (const OUT *)v->out_a, *d_col = d_vert + 3 * v->mesh_cache_nv;
    if (nv > 0) {
        HV_HIP(hipMemcpyAsync(vertices, d_vert, sizeof(OUT) * 3 * nv, hipMemcpyDefault /* the destination may be host or device memory (a GPU consumer) */, v->stream));
        HV_HIP(hipMemcpyAsync(vertex_colors, d_col, sizeof(OUT) * 3 * nv, hipMemcpyDefault, v->stream));
    }
    if (nt > 0) HV_HIP(hipMemcpyAsync(triangles, v->out_b, sizeof(int32_t) * 3 * nt, hipMemcpyDefault, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}

extern "C" {
int hv_tsdf_extract_mesh(hv_volume *v, double *vertices, double *vertex_colors, int64_t cap_vertices,
                         int32_t *triangles, int64_t cap_triangles, int64_t *n_vertices, int64_t *n_triangles) {
    return extract_mesh_as<double>(v, vertices, vertex_colors, cap_vertices, triangles, cap_triangles, n_vertices, n_triangles, "hv_tsdf_extract_mesh");
}

int hv_tsdf_extract_mesh_f32(hv_volume *v, float *vertices, float *vertex_colors, int64_t cap_vertices,
                             int32_t *triangles, int64_t cap_triangles, int64_t *n_vertices, int64_t *n_triangles) {
    return extract_mesh_as<float>(v, vertices, vertex_colors, cap_vertices, triangles, cap_triangles, n_vertices, n_triangles, "hv_tsdf_extract_mesh_f32");
}

static int points_compute(hv_volume *v, bool f32) {
    int64_t nb = 0;
    int rc = hv_num_blocks(v, &nb);
    if (rc != HV_OK) return rc;
    v->points_cache_version = 0;
    v->mesh_cache_version = 0; // shares out_a
    v->points_cache_n = 0;
    if (nb == 0) {
        v->points_cache_version = v->content_version;
        return HV_OK;
    }
    HvMcParams M{v->cfg.voxel_size, v->cfg.voxel_size * 0.5};
    const int nu = (int)nb;
    // pass 1 counts per unit (from the column masks; kept per unit between extractions), the scan places the units, pass 2 writes
    // into a buffer of exactly that size
    hv_profile_begin(v);
    HvUnitMasks UM;
    rc = unit_masks_compute(v, nu, &UM);
    if (rc != HV_OK) return rc;
    rc = hv_ensure_buffer(v, &v->out_c, &v->out_c_bytes, sizeof(int32_t) * (size_t)(nu + 1));
    if (rc != HV_OK) return rc;
    int32_t *count = (int32_t *)v->pc_cache, *base = (int32_t *)v->out_c;
    const uint32_t *m_on = UM.m_on, *m_ip = UM.m_ip;
    const bool pc_full = v->pc_epoch != v->extract_epoch || nu < v->pc_units || hv_extract_full();
    HV_HIP(hipMemsetAsync(count + nu, 0, sizeof(int32_t), v->stream)); // the scan's extra element
    hipLaunchKernelGGL((k_pc_extract<false, double>), dim3(nu), dim3(256), 0, v->stream, v->table, (const char *)v->pool, m_on, m_ip, nu, M,
                       v->cfg.voxel_size * (double)R, count, (const int32_t *)nullptr, (double *)nullptr, (double *)nullptr, (int64_t)0,
                       UM.stamp, pc_full ? (int32_t)-1 : v->pc_stamp);
    HV_HIP(hipGetLastError());
    v->pc_epoch = v->extract_epoch;
    v->pc_stamp = v->frame_counter;
    v->pc_units = nu;
    rc = exclusive_scan_i32(v, count, base, nu + 1);
    if (rc != HV_OK) return rc;
    hv_profile_end(v, nb);
    int32_t total = 0;
    HV_HIP(hipMemcpyAsync(&total, base + nu, sizeof(int32_t), hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    const int64_t n = total;
    if (n > 0) {
        rc = hv_ensure_buffer(v, &v->out_a, &v->out_a_bytes, (f32 ? sizeof(float) : sizeof(double)) * 6 * (size_t)n);
        if (rc != HV_OK) return rc;
        hv_profile_begin(v);
        if (f32) {
            float *d_pts = (float *)v->out_a, *d_cols = d_pts + 3 * n;
            hipLaunchKernelGGL((k_pc_extract<true, float>), dim3(nu), dim3(256), 0, v->stream, v->table, (const char *)v->pool, m_on, m_ip, nu, M,
                               v->cfg.voxel_size * (double)R, (int32_t *)nullptr, (const int32_t *)base, d_pts, d_cols, n, UM.stamp, (int32_t)-1);
        } else {
            double *d_pts = (double *)v->out_a, *d_cols = d_pts + 3 * n;
            hipLaunchKernelGGL((k_pc_extract<true, double>), dim3(nu), dim3(256), 0, v->stream, v->table, (const char *)v->pool, m_on, m_ip, nu, M,
                               v->cfg.voxel_size * (double)R, (int32_t *)nullptr, (const int32_t *)base, d_pts, d_cols, n, UM.stamp, (int32_t)-1);
        }
        hv_profile_end(v, nb);
        HV_HIP(hipGetLastError());
    }
    v->points_cache_n = n;
    v->points_cache_f32 = f32;
    v->points_cache_version = v->content_version;
    return HV_OK;
}

} // extern "C"
template <typename OUT>
static int extract_points_as(hv_volume *v, OUT *points, OUT *colors, int64_t cap, int64_t *n, const char *who) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "%s: null argument", who);
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "%s: volume is not in TSDF mode", who);
    HV_HIP(hipSetDevice(v->device));
    constexpr bool f32 = sizeof(OUT) == sizeof(float);
    if (v->points_cache_version != v->content_version || v->points_cache_f32 != f32) {
        const int rc = points_compute(v, f32);
        if (rc != HV_OK) return rc;
    }
    *n = v->points_cache_n;
    if (points == nullptr || colors == nullptr || cap <= 0) return HV_OK;
    const int64_t m = std::min<int64_t>(v->points_cache_n, cap);
    if (m > 0) {
        const OUT *d_pts = (const OUT *)v->out_a, *d_cols = d_pts + 3 * v->points_cache_n;
        HV_HIP(hipMemcpyAsync(points, d_pts, sizeof(OUT) * 3 * m, hipMemcpyDefault, v->stream));
        HV_HIP(hipMemcpyAsync(colors, d_cols, sizeof(OUT) * 3 * m, hipMemcpyDefault, v->stream));
        HV_HIP(hipStreamSynchronize(v->stream));
    }
    return HV_OK;
}

extern "C" {
int hv_tsdf_extract_points(hv_volume *v, double *points, double *colors, int64_t cap, int64_t *n) {
    return extract_points_as<double>(v, points, colors, cap, n, "hv_tsdf_extract_points");
}

int hv_tsdf_extract_points_f32(hv_volume *v, float *points, float *colors, int64_t cap, int64_t *n) {
    return extract_points_as<float>(v, points, colors, cap, n, "hv_tsdf_extract_points_f32");
}

int hv_tsdf_extract_point_normals(hv_volume *v, double *normals, int64_t cap, int64_t *n) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_tsdf_extract_point_normals: null argument");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_extract_point_normals: volume is not in TSDF mode");
    HV_HIP(hipSetDevice(v->device));
    if (v->points_cache_version != v->content_version || v->points_cache_f32) { // (the gradient is taken at the float64 points)
        const int rc = points_compute(v, false);
        if (rc != HV_OK) return rc;
    }
    *n = v->points_cache_n;
    if (normals == nullptr || cap <= 0 || v->points_cache_n == 0) return HV_OK;
    const int64_t m = std::min<int64_t>(v->points_cache_n, cap);
    int rc = hv_ensure_buffer(v, &v->out_b, &v->out_b_bytes, sizeof(double) * 3 * (size_t)m);
    if (rc != HV_OK) return rc;
    hv_profile_begin(v);
    hipLaunchKernelGGL(k_pc_normals, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, v->stream, v->table, (const char *)v->pool,
                       v->cfg.voxel_size, v->cfg.voxel_size * (double)R, (const double *)v->out_a, m, (double *)v->out_b);
    hv_profile_end(v, 0);
    HV_HIP(hipGetLastError());
    HV_HIP(hipMemcpyAsync(normals, v->out_b, sizeof(double) * 3 * m, hipMemcpyDefault, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}

} // extern "C"
