// CSR product on the "sliced pattern" structure: a set-up-time re-layout for matrices whose rows
// repeat a few column patterns (every linearised stencil: col = row + one of a few distance lists).
//
// At set-up each row gets the id of its pattern (the list of col - row over its entries, in entry
// order) and the patterns go into a small table.  Rows are taken in BLOCKS of 256 R consecutive rows,
// R = 16 / sizeof(T) (512 rows of fp64, 1024 of fp32): one workgroup of four waves per block, a lane
// owning R consecutive rows.  The entry values are copied block by block, for each of the block's four
// waves entry-major: value j of the R rows of lane l of wave w at
//     base + (w * maxlen + j) * 64 R + l * R,
// 16 bytes that one lane loads at once (1 KiB per wave-instruction), short rows padded with zeros.
// The product reads NO column array and no row offsets.  A block whose rows all follow one distance
// list (or ordered parts of one list: the row's 32-bit mask says which entries it has) is served
// through LDS: the list falls into runs of consecutive distances (27-point stencil: 9 runs of 3), the
// window of x each run reads, 256 R + length - 1 consecutive elements, is staged once, coalesced, and
// every lane then takes its x values from there -- 9 global reads of x per row instead of 27, which
// measured 1.3 ms of a 6.4 ms product on the 27-point 512^3 matrix.  y leaves with nontemporal
// 16-byte stores behind the block's barrier, 4 KiB per workgroup at a time: stores trickling out wave
// by wave, 512 bytes at a time, cost another 1.5 ms there.  Each row is summed by one lane in
// ascending entry order with separate multiply and add: bit for bit the loop of the reference's CPU
// task (src/CSRMatrixTasks.cpp:73-91 visits the entries in ascending k and adds val * x into the row's
// sum).  What the reference does on the GPU instead is hand the arrays to cuSPARSE on every call
// (src/CSRMatrixTasks.cu:117-150).
//
// Bytes per launch: padded entries * sizeof(T) + rows * 2 sizeof(T) (x once, y once) + 16 per block
// + 4 per row of a masked wave's rows (+ 1 per row of a block that fits no common list).
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <new>
#include <type_traits>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "lsk_common.hpp"

namespace {

using lsk::kBlock;
using lsk::kWave;
constexpr int kMaxPatterns = 64;
constexpr int kMaxLength = 32;
constexpr int kMaxRuns = 16;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr size_t kMaxStageBytes = 60 << 10; // of LDS per workgroup for the staged windows of x

// One pattern: its distance list and how a block stages the windows of x the list reads.
struct PatternInfo {
    int len;        // entries; -1: unused slot
    int nruns;      // runs of consecutive distances; 0: not staged (too many runs or too much LDS)
    int lds_elems;  // elements of LDS the staged windows take, for blocks of 256 R rows
    int w_diag_off; // where x at distance 0 (the row's own element) sits in a wave's windows; -1: the list has no such entry
    int dist[kMaxLength];     // distance of entry j
    int lds_off[kMaxLength];  // where the x value of entry j for the block's first row sits in LDS
    unsigned lds_off_packed[kMaxLength / 2]; // the same, two 16-bit offsets per word (entry 2k low, 2k + 1 high)
    int run_start[kMaxRuns];  // first distance of run q
    int run_len[kMaxRuns];
    int run_off[kMaxRuns];    // where run q's window starts in LDS
    // A run's window is the block's 256 R rows wide plus run_len - 1 elements; those extra elements
    // of all runs, as one list (at most kMaxLength - 1 of them): element t of it is x at distance
    // tail_dist[t] from the block's first row and goes to LDS position tail_off[t].
    int ntail;
    int tail_dist[kMaxLength];
    int tail_off[kMaxLength];
    // The same plan for ONE WAVE's 64 R rows (sliced_wave_kernel: every wave stages its own windows and
    // needs no barrier): window q is 64 R + run_len - 1 elements at w_run_off[q].
    int w_lds_elems;
    int w_ntail;
    int w_run_off[kMaxRuns];
    unsigned w_lds_off_packed[kMaxLength / 2];
    int w_tail_dist[kMaxLength];
    int w_tail_off[kMaxLength];
};

} // namespace

struct lsk_csr_sliced {
    int elem_bytes = 0;
    int64_t rows = 0, nnz = 0, k_base = 0, row_first = 0, x_lo = 0, x_hi = -1;
    int64_t blocks = 0, padded_entries = 0, mixed_blocks = 0, masked_waves = 0;
    int patterns = 0, max_len = 0;
    bool all_staged = true;        // every pattern's x windows fit the staging area
    size_t lds_bytes = 0;          // dynamic LDS of a launch: the largest staging area of any pattern
    size_t wave_lds_bytes = 0;     // ... of one wave's windows in the wave-staged form
    bool runs_of_three = true;     // every list is runs of exactly three consecutive distances (27-point stencils)
    void *val = nullptr;           // padded_entries values, block by block
    void *meta = nullptr;          // per block {int64 base; int32 maxlen; int32 pattern | flags, or -1}
    unsigned char *row_pattern = nullptr; // per row
    unsigned *row_mask = nullptr;  // per row: which entries of its block's list the row has
    PatternInfo *table = nullptr;  // kMaxPatterns entries
    // Uniform values: entry j of a block's list has ONE value in every row of every block on that list (a
    // constant-coefficient stencil).  uniform[list * kMaxLength + j] is that value; the product can then
    // run without reading the values at all (option sliced.uniform).
    void *uniform = nullptr;
    int *uniform_scratch = nullptr; // kMaxPatterns first-block slots + 1 mismatch flag
    // The verdict after update_values arrives without anyone waiting for it: the check is enqueued behind
    // the refill, its flag copied to pinned memory, and the products read the value array until the event
    // behind that copy has passed.
    int *uniform_host = nullptr;    // pinned: the mismatch flag as the last check left it
    hipEvent_t uniform_event = nullptr;
    mutable bool uniform_pending = false;
    mutable bool uniform_ok = false;
};

namespace {

template <typename T> struct Lane;
template <> struct Lane<double> {
    static constexpr int R = 2;
    typedef double V __attribute__((ext_vector_type(2)));
};
template <> struct Lane<float> {
    static constexpr int R = 4;
    typedef float V __attribute__((ext_vector_type(4)));
};

// A 16-byte nontemporal store, written out: given the builtin, the compiler merges this path with the
// row-by-row one beside it into two plain 8-byte stores.
typedef float Bits16 __attribute__((ext_vector_type(4)));
template <typename V>
__device__ __forceinline__ void store16_nontemporal(void *p, V value) {
    static_assert(sizeof(V) == 16, "16-byte vectors");
    const Bits16 bits = __builtin_bit_cast(Bits16, value);
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(bits) : "memory");
}

constexpr int kBlockMasked = 0x100;      // some row of the block has only part of the block's list
constexpr int kWaveMaskedShift = 12;     // bits 12..15: which of the block's four waves have such rows
struct BlockMeta {
    long long base;
    int maxlen;
    // list id (| flags): every row of the block has that distance list, or (kBlockMasked) an ordered
    // part of it given by its row_mask; -1: the rows fit no common list
    int pattern;
};
static_assert(sizeof(BlockMeta) == 16, "block record is 16 bytes");

// Flags the set-up kernels raise (any non-zero: the matrix does not get this structure).
enum { kFlagTooManyPatterns = 0, kFlagRowTooLong = 1, kFlagMismatch = 2, kFlagOutsideWindow = 3, kFlagBadRect = 4, kFlags = 8 };
// Per block, while it is being classified.
enum { kBlkNeedsMasks = 1, kBlkMixed = 2, kBlkWaveShift = 4 };

__device__ __forceinline__ unsigned long long mix64(unsigned long long h, unsigned long long v) {
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    return h ^ (h >> 32);
}

__global__ void sliced_init_kernel(unsigned long long *keys, int *flags, long long *counts, int *lists) {
    for (int i = threadIdx.x; i < kMaxPatterns * (kMaxLength + 1); i += blockDim.x) lists[i] = 0;
    if (threadIdx.x < kMaxPatterns) keys[threadIdx.x] = 0;
    if (threadIdx.x < kFlags) flags[threadIdx.x] = 0;
    if (threadIdx.x < 4) counts[threadIdx.x] = 0;
}

// Pass 1: hash of every row's (length, distances) list; first come, first served into the pattern
// table (keys[]: 0 = free), the inserting row recorded as the pattern's representative.  A hash
// only proposes: pass 3 compares every entry with the table.
__global__ __launch_bounds__(kBlock) void sliced_assign_kernel(
    int64_t rows, int64_t nnz, const long long *__restrict__ rects, long long k_base,
    const long long *__restrict__ col, long long row_first, unsigned long long *keys, long long *rep,
    unsigned char *__restrict__ row_pattern, int *flags) {
    const int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= rows) return;
    const long long lo = rects[2 * r] - k_base, hi = rects[2 * r + 1] - k_base;
    const long long len = hi >= lo ? hi - lo + 1 : 0; // hi < lo: an empty rect, whatever its bounds
    if (len > 0 && (lo < 0 || hi >= nnz)) {
        flags[kFlagBadRect] = 1;
        return;
    }
    if (len > kMaxLength) {
        flags[kFlagRowTooLong] = 1;
        return;
    }
    unsigned long long h = mix64(0x2545f4914f6cdd1dull, static_cast<unsigned long long>(len));
    for (long long k = lo; k < lo + len; ++k) h = mix64(h, static_cast<unsigned long long>(col[k] - (row_first + r)));
    if (h == 0) h = 1;
    for (int s = 0; s < kMaxPatterns; ++s) {
        unsigned long long cur = __hip_atomic_load(&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            cur = atomicCAS(&keys[s], 0ull, h);
            if (cur == 0) {
                rep[s] = r;
                cur = h;
            }
        }
        if (cur == h) {
            row_pattern[r] = static_cast<unsigned char>(s);
            return;
        }
    }
    flags[kFlagTooManyPatterns] = 1;
}

// Pass 2: the lists, from the representatives.  lists[s * 33] = length (-1: unused slot),
// lists[s * 33 + 1 + j] = distance of entry j.
__global__ void sliced_lists_kernel(const long long *__restrict__ rects, long long k_base,
                                    const long long *__restrict__ col, long long row_first,
                                    const unsigned long long *__restrict__ keys, const long long *__restrict__ rep,
                                    int *__restrict__ lists, int *flags) {
    const int s = threadIdx.x;
    if (s >= kMaxPatterns) return;
    int *mine = lists + s * (kMaxLength + 1);
    if (keys[s] == 0) {
        mine[0] = -1;
        return;
    }
    const long long r = rep[s];
    const long long lo = rects[2 * r] - k_base, hi = rects[2 * r + 1] - k_base;
    mine[0] = hi >= lo ? static_cast<int>(hi - lo + 1) : 0;
    for (long long k = lo; k <= hi; ++k) {
        const long long d = col[k] - (row_first + r);
        if (d > INT_MAX / 2 || d < INT_MIN / 2) flags[kFlagOutsideWindow] = 1;
        mine[1 + static_cast<int>(k - lo)] = static_cast<int>(d);
    }
}

// Pass 3: every entry against the table (so that a hash collision can never produce a wrong
// column) and against the window of x; per group of 64 rows, the longest row, the pattern of a
// longest row, and the one pattern all its rows share (-1: they differ).
__global__ __launch_bounds__(kBlock) void sliced_verify_kernel(
    int64_t rows, const long long *__restrict__ rects, long long k_base, const long long *__restrict__ col,
    long long row_first, long long x_lo, long long x_hi, const unsigned char *__restrict__ row_pattern,
    const PatternInfo *__restrict__ table, int *__restrict__ g_maxlen, int *__restrict__ g_holder,
    int *__restrict__ g_uniform, int *flags) {
    const int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const bool live = r < rows;
    int len = 0, pat = -1;
    if (live) {
        pat = row_pattern[r];
        const long long lo = rects[2 * r] - k_base;
        len = max(static_cast<int>(rects[2 * r + 1] - k_base - lo + 1), 0);
        bool ok = table[pat].len == len;
        for (int j = 0; ok && j < len; ++j) {
            const long long c = col[lo + j];
            ok = c == row_first + r + table[pat].dist[j];
            if (c < x_lo || c > x_hi) flags[kFlagOutsideWindow] = 1;
        }
        if (!ok) flags[kFlagMismatch] = 1;
    }
    int maxlen = len;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, kWave));
    const int first = __builtin_amdgcn_readfirstlane(pat); // lane 0 of a group is live whenever the group exists
    const bool uniform = __all(!live || pat == first);
    const unsigned long long holders = __ballot(live && len == maxlen);
    const int holder = holders ? __shfl(pat, __ffsll(static_cast<long long>(holders)) - 1, kWave) : -1;
    if (threadIdx.x % kWave == 0 && live) {
        const int64_t g = r / kWave;
        g_maxlen[g] = maxlen;
        g_holder[g] = holder;
        g_uniform[g] = uniform ? first : -1;
    }
}

// Pass 4: per block of `groups` groups of 64 rows: the longest row, the list of a longest row as
// the block's common list, and whether every row has exactly that list.
__global__ __launch_bounds__(kBlock) void sliced_combine_kernel(int64_t blocks, int64_t total_groups, int groups,
                                                                const int *__restrict__ g_maxlen,
                                                                const int *__restrict__ g_holder,
                                                                const int *__restrict__ g_uniform,
                                                                BlockMeta *__restrict__ meta, int *__restrict__ blk_flags) {
    const int64_t b = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (b >= blocks) return;
    int maxlen = -1, common = -1;
    bool uniform = true;
    for (int i = 0; i < groups; ++i) {
        const int64_t g = b * groups + i;
        if (g >= total_groups) break;
        if (g_maxlen[g] > maxlen) {
            maxlen = g_maxlen[g];
            common = g_holder[g];
        }
    }
    for (int i = 0; i < groups; ++i) {
        const int64_t g = b * groups + i;
        if (g >= total_groups) break;
        uniform = uniform && g_uniform[g] == common;
    }
    meta[b].maxlen = max(maxlen, 0);
    meta[b].pattern = common;
    blk_flags[b] = uniform ? 0 : kBlkNeedsMasks;
}

// Pass 5: every row's mask: which entries of its block's common list it has (in order).  A row whose
// list is no ordered part of the common list makes the block "mixed".
__global__ __launch_bounds__(kBlock) void sliced_masks_kernel(int64_t rows, int block_rows,
                                                              const unsigned char *__restrict__ row_pattern,
                                                              const PatternInfo *__restrict__ table,
                                                              const BlockMeta *__restrict__ meta, int *blk_flags,
                                                              unsigned *__restrict__ row_mask) {
    const int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= rows) return;
    const int64_t b = r / block_rows;
    const int pat = row_pattern[r];
    const int len = table[pat].len;
    const unsigned full = len >= 32 ? 0xffffffffu : (1u << len) - 1u;
    const int common = meta[b].pattern;
    if (pat == common || !(blk_flags[b] & kBlkNeedsMasks)) {
        row_mask[r] = full;
        return;
    }
    const int *mine = table[pat].dist;
    const int *list = table[common].dist;
    const int maxlen = table[common].len;
    unsigned mask = 0u;
    int pos = 0;
    bool fits = true;
    for (int j = 0; fits && j < len; ++j) {
        while (pos < maxlen && list[pos] != mine[j]) ++pos;
        fits = pos < maxlen;
        if (fits) mask |= 1u << pos++;
    }
    if (!fits) {
        atomicOr(&blk_flags[b], kBlkMixed);
        row_mask[r] = full; // in a mixed block entry j of a row sits at position j
        return;
    }
    row_mask[r] = mask;
    const int wave = static_cast<int>((r % block_rows) / (block_rows / kWavesPerBlock));
    atomicOr(&blk_flags[b], 1 << (kBlkWaveShift + wave));
}

// Pass 5b: a block whose rows fit no ordered part of its longest row's list (the last line of one grid
// plane and the first of the next: two lists of equal length, neither part of the other) looks for
// another list of the table that all of them are ordered parts of -- the interior row's, for a stencil --
// and, if there is one, runs on the shortest such list like any other masked block.  supersets[p]: the
// lists that list p is an ordered part of (bit c), from the host.  The masks pass runs again afterwards.
__global__ __launch_bounds__(kBlock) void sliced_rescue_kernel(int64_t rows, int block_rows,
                                                               const unsigned char *__restrict__ row_pattern,
                                                               const PatternInfo *__restrict__ table,
                                                               const unsigned long long *__restrict__ supersets,
                                                               BlockMeta *__restrict__ meta, int *__restrict__ blk_flags) {
    const int64_t b = blockIdx.x;
    if (!(blk_flags[b] & kBlkMixed)) return;
    __shared__ unsigned long long present;
    if (threadIdx.x == 0) present = 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < block_rows; i += kBlock) {
        const int64_t r = b * block_rows + i;
        if (r < rows) atomicOr(&present, 1ull << row_pattern[r]);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    unsigned long long candidates = ~0ull;
    for (int p = 0; p < kMaxPatterns; ++p)
        if ((present >> p) & 1ull) candidates &= supersets[p];
    int best = -1;
    for (int c = 0; c < kMaxPatterns; ++c)
        if (((candidates >> c) & 1ull) && (best < 0 || table[c].len < table[best].len)) best = c;
    if (best < 0) return; // stays a block without a common list
    meta[b].pattern = best;
    meta[b].maxlen = table[best].len;
    blk_flags[b] = kBlkNeedsMasks;
}

// Pass 6: the blocks' final records and sizes.
__global__ __launch_bounds__(kBlock) void sliced_finish_kernel(int64_t blocks, int64_t rows, int block_rows,
                                                               const int *__restrict__ blk_flags,
                                                               BlockMeta *__restrict__ meta, long long *__restrict__ entries,
                                                               long long *counts) {
    const int64_t b = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (b >= blocks) return;
    const int f = blk_flags[b];
    entries[b] = static_cast<long long>(meta[b].maxlen) * block_rows;
    // counts[2]: the padded entries of rows that exist (the last block's missing rows left out)
    const long long here = min(static_cast<long long>(block_rows), static_cast<long long>(rows - b * block_rows));
    atomicAdd(reinterpret_cast<unsigned long long *>(counts) + 2, static_cast<unsigned long long>(meta[b].maxlen * here));
    if (f & kBlkMixed) {
        meta[b].pattern = -1;
        atomicAdd(reinterpret_cast<unsigned long long *>(counts), 1ull);
        return;
    }
    // counts[3]: which lists serve some block (bit = list id)
    atomicOr(reinterpret_cast<unsigned long long *>(counts) + 3, 1ull << (meta[b].pattern & 63));
    const int waves = (f >> kBlkWaveShift) & 0xf;
    if (waves) {
        meta[b].pattern |= kBlockMasked | (waves << kWaveMaskedShift);
        atomicAdd(reinterpret_cast<unsigned long long *>(counts) + 1, static_cast<unsigned long long>(__popc(waves)));
    }
}

__global__ __launch_bounds__(kBlock) void sliced_bases_kernel(int64_t blocks, const long long *__restrict__ entries,
                                                              const long long *__restrict__ inclusive,
                                                              BlockMeta *__restrict__ meta) {
    const int64_t b = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (b < blocks) meta[b].base = inclusive[b] - entries[b];
}

// Pass 7: the values into their blocks -- entry j of a row at position j, or, in a masked block, at
// the position of its distance in the block's list -- zeros everywhere else (rows past the end too).
template <typename T>
__global__ __launch_bounds__(kBlock) void sliced_fill_kernel(int64_t rows, int64_t padded_rows, const long long *__restrict__ rects,
                                                             long long k_base, const T *__restrict__ val,
                                                             const BlockMeta *__restrict__ meta,
                                                             const unsigned *__restrict__ row_mask, T *__restrict__ out) {
    constexpr int R = Lane<T>::R;
    constexpr int kBlockRows = kBlock * R, kWaveRows = kWave * R;
    const int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (r >= padded_rows) return;
    const int64_t b = r / kBlockRows;
    const int in = static_cast<int>(r % kBlockRows);
    const BlockMeta m = meta[b];
    long long lo = 0;
    unsigned mask = 0u;
    if (r < rows) {
        lo = rects[2 * r] - k_base;
        const int len = max(static_cast<int>(rects[2 * r + 1] - k_base - lo + 1), 0);
        mask = (m.pattern >= 0 && (m.pattern & kBlockMasked)) ? row_mask[r] : (len >= 32 ? 0xffffffffu : (1u << len) - 1u);
    }
    T *dst = out + m.base + int64_t(in / kWaveRows) * m.maxlen * kWaveRows + in % kWaveRows;
    for (int j = 0; j < m.maxlen; ++j) {
        const bool has = (mask >> j) & 1u;
        dst[int64_t(j) * kWaveRows] = has ? val[lo + __popc(mask & ((1u << j) - 1u))] : T(0);
    }
}

__global__ void sliced_uniform_init_kernel(int *scratch) {
    scratch[threadIdx.x] = threadIdx.x < kMaxPatterns ? INT_MAX : 0;
}

// Uniform values, step 1: per list, the first block (lowest index) that runs on it and has a row with
// the whole list (a block that only borrowed the list -- sliced_rescue_kernel -- has none).  One
// workgroup per block.
__global__ __launch_bounds__(kBlock) void sliced_uniform_first_kernel(int64_t rows, int rows_per_lane,
                                                                      const BlockMeta *__restrict__ meta,
                                                                      const unsigned *__restrict__ row_mask,
                                                                      int *__restrict__ first) {
    const int64_t b = blockIdx.x;
    const BlockMeta m = meta[b];
    if (m.pattern < 0 || m.maxlen <= 0) return;
    const unsigned full = m.maxlen >= 32 ? 0xffffffffu : (1u << m.maxlen) - 1u;
    const int wave = threadIdx.x / kWave;
    const bool masked = (m.pattern & kBlockMasked) && ((m.pattern >> (kWaveMaskedShift + wave)) & 1);
    bool has = false;
    for (int h = 0; h < rows_per_lane; ++h) {
        const int64_t r = (b * kBlock + threadIdx.x) * rows_per_lane + h;
        has = has || (r < rows && (!masked || (row_mask[r] & full) == full));
    }
    if (__syncthreads_or(has) && threadIdx.x == 0) atomicMin(&first[m.pattern & 0xff], static_cast<int>(min(b, int64_t(INT_MAX))));
}

// Step 2: one workgroup per list: the values of the first row of that block which has the whole list
// (the row the list was taken from is one) become the list's candidate values.
template <typename T>
__global__ __launch_bounds__(kBlock) void sliced_uniform_pick_kernel(int64_t rows, const BlockMeta *__restrict__ meta,
                                                                     const unsigned *__restrict__ row_mask,
                                                                     const T *__restrict__ val, int *__restrict__ first,
                                                                     T *__restrict__ uniform) {
    constexpr int R = Lane<T>::R;
    constexpr int kBlockRows = kBlock * R, kWaveRows = kWave * R;
    __shared__ int chosen;
    const int list = blockIdx.x;
    const int b = first[list];
    if (b == INT_MAX) return; // no block runs on this list
    const BlockMeta m = meta[b];
    const unsigned full = m.maxlen >= 32 ? 0xffffffffu : (1u << m.maxlen) - 1u;
    if (threadIdx.x == 0) chosen = INT_MAX;
    __syncthreads();
    for (int h = 0; h < R; ++h) {
        const int in = threadIdx.x * R + h;
        const int64_t r = int64_t(b) * kBlockRows + in;
        if (r >= rows) continue;
        const bool masked = (m.pattern & kBlockMasked) && ((m.pattern >> (kWaveMaskedShift + in / kWaveRows)) & 1);
        if (!masked || (row_mask[r] & full) == full) atomicMin(&chosen, in);
    }
    __syncthreads();
    const int in = chosen;
    if (in == INT_MAX) { // (the unmasked wave's rows lie past the end of the piece: no candidate, no verdict)
        if (threadIdx.x == 0) first[list] = INT_MAX;
        return;
    }
    const T *src = val + m.base + int64_t(in / kWaveRows) * m.maxlen * kWaveRows + in % kWaveRows;
    for (int j = threadIdx.x; j < kMaxLength; j += kBlock)
        uniform[list * kMaxLength + j] = j < m.maxlen ? src[int64_t(j) * kWaveRows] : T(0);
}

// Step 3: every value of every row against its list's candidate, bit by bit.
template <typename T>
__global__ __launch_bounds__(kBlock) void sliced_uniform_check_kernel(int64_t rows, const BlockMeta *__restrict__ meta,
                                                                      const unsigned *__restrict__ row_mask,
                                                                      const T *__restrict__ val, const T *__restrict__ uniform,
                                                                      const int *__restrict__ first, int *__restrict__ mismatch) {
    constexpr int R = Lane<T>::R;
    using V = typename Lane<T>::V;
    using Bits = std::conditional_t<sizeof(T) == 8, unsigned long long, unsigned>;
    constexpr int kBlockRows = kBlock * R, kWaveRows = kWave * R;
    const int64_t b = blockIdx.x;
    const BlockMeta m = meta[b];
    if (m.pattern < 0 || (m.maxlen > 0 && first[m.pattern & 0xff] == INT_MAX)) {
        // a block without a common list, or on a list that no row of the piece has as a whole (no
        // candidate values): not this form's business
        if (threadIdx.x == 0) *mismatch = 1;
        return;
    }
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int64_t r = b * kBlockRows + int64_t(threadIdx.x) * R;
    const bool masked = (m.pattern & kBlockMasked) && ((m.pattern >> (kWaveMaskedShift + wave)) & 1);
    unsigned mask[R];
    for (int h = 0; h < R; ++h) mask[h] = r + h >= rows ? 0u : masked ? row_mask[r + h] : 0xffffffffu;
    const T *vp = val + m.base + int64_t(wave) * m.maxlen * kWaveRows + lane * R;
    const T *mine = uniform + (m.pattern & 0xff) * kMaxLength;
    bool differs = false;
    for (int j = 0; j < m.maxlen; ++j) {
        const V v = __builtin_nontemporal_load(reinterpret_cast<const V *>(vp + int64_t(j) * kWaveRows));
        const Bits want = __builtin_bit_cast(Bits, mine[j]);
        for (int h = 0; h < R; ++h) {
            const T got = v[h];
            differs = differs || (((mask[h] >> j) & 1u) && __builtin_bit_cast(Bits, got) != want);
        }
    }
    if (__any(differs) && lane == 0) *mismatch = 1;
}

// The product.  One workgroup per block of 256 R rows; lane t owns rows R t .. R t + R - 1 of the
// block.  CAP: the longest row the instance serves (all of a lane's value loads are in flight at
// once).  DOT: workgroup g (of the logical order, whatever its placement) also leaves partials[g] =
// the sum over its rows of y_new[r] * w[r], added up in a fixed order.
// Four waves per SIMD (128 registers) where the instance fits them without spilling -- a spill costs
// far more here than the fourth wave brings -- else three.
template <typename T, int CAP, bool GENERAL>
constexpr int sliced_waves_per_simd() {
    if (GENERAL && CAP > 27) return 2;
    return (CAP <= 20 || (CAP <= 27 && sizeof(T) == 8 && !GENERAL)) && !(GENERAL && sizeof(T) == 4 && CAP > 16) ? 4 : 3;
}
// GENERAL: also serves blocks whose list is not staged (x straight from memory) and blocks without a
// common list; the instance without it is what every stencil matrix runs on.
template <typename T, int CAP, bool DOT, bool GENERAL>
__global__ __launch_bounds__(kBlock, (sliced_waves_per_simd<T, CAP, GENERAL>())) void sliced_spmv_kernel(
    int64_t block_begin, int64_t rows, int64_t row_begin, int64_t row_end, const BlockMeta *__restrict__ meta,
    const T *__restrict__ val, const unsigned char *__restrict__ row_pattern, const unsigned *__restrict__ row_mask,
    const PatternInfo *__restrict__ table, const T *__restrict__ x, long long x_shift, long long x_extent,
    T *__restrict__ y, int beta, int remap, const T *__restrict__ w, T *__restrict__ partials, int yw_aligned) {
    constexpr int R = Lane<T>::R;
    using V = typename Lane<T>::V;
    // y and w are accessed 16 bytes at a time where they are 16-byte aligned (the lane's first row is a
    // multiple of R, so that is a property of the base pointers: yw_aligned), else element by element
    constexpr int kBlockRows = kBlock * R, kWaveRows = kWave * R;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *xs = reinterpret_cast<T *>(lds_raw);
    __shared__ T wave_share[kWavesPerBlock];

    const unsigned wg = lsk::xcd_remap(blockIdx.x, gridDim.x, remap);
    const int64_t block = block_begin + wg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x % kWave;
    const BlockMeta m = meta[block];
    const long long base = lsk::wave_first(m.base);
    const int maxlen = __builtin_amdgcn_readfirstlane(m.maxlen);
    const int pattern = __builtin_amdgcn_readfirstlane(m.pattern);
    const int64_t r0 = block * kBlockRows;
    const int64_t r = r0 + int64_t(threadIdx.x) * R;

    // all of the lane's values first: they are what comes from HBM (streamed past the caches: nothing
    // reads them twice)
    const T *vp = val + base + int64_t(wave) * maxlen * kWaveRows + lane * R;
    // All of a lane's values are requested at once, after the x windows have been staged and the
    // registers of that are free again (27 values of two doubles and everything else: 128 registers,
    // four waves per SIMD).  (Measured on the 27-point 512^3 matrix: everything in flight without a
    // spill 4.7 ms; 8 to 12 values in flight, the others requested as those are used, 5.4-5.7 ms; any
    // spill 5.5-6.8 ms.  LSK_SLICED_CHUNK builds the burst forms for such measurements.)
#ifdef LSK_SLICED_CHUNK
    constexpr int CHUNK = CAP < LSK_SLICED_CHUNK ? CAP : LSK_SLICED_CHUNK;
#else
    constexpr int CHUNK = CAP;
#endif
    constexpr int NCHUNKS = (CAP + CHUNK - 1) / CHUNK;
    V v[CAP];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int j = c * CHUNK; j < (c + 1) * CHUNK && j < CAP; ++j) {
            // (a value past the block's longest row is 0, not undefined: undefined ones cost spills)
            if (j < maxlen) v[j] = __builtin_nontemporal_load(reinterpret_cast<const V *>(vp + int64_t(j) * kWaveRows));
            else v[j] = V(0);
        }
    };
#ifndef LSK_SLICED_STAGE_FIRST
#define LSK_SLICED_STAGE_FIRST 1
#endif
    if (GENERAL || !LSK_SLICED_STAGE_FIRST) load_chunk(0);

    bool live[R];
    bool all_live = true;
    T acc[R];
#pragma unroll
    for (int h = 0; h < R; ++h) {
        live[h] = r + h >= row_begin && r + h < row_end;
        all_live = all_live && live[h];
        acc[h] = (beta && live[h]) ? y[r + h] : T(0);
    }

    if (!GENERAL || pattern >= 0) {
        const PatternInfo &info = table[pattern & 0xff]; // wave-uniform address
        unsigned mask[R];
        const bool masked = (pattern >> (kWaveMaskedShift + wave)) & 1;
#pragma unroll
        for (int h = 0; h < R; ++h) mask[h] = !live[h] ? 0u : masked ? row_mask[r + h] : 0xffffffffu;
        const int nruns = info.nruns;
        if (!GENERAL || nruns > 0) {
            // The windows of x the block's list reads, one per run of consecutive distances, staged
            // once: four runs' loads in flight at a time, never behind a branch (an element outside
            // the vector -- only rows that lack the entry point there -- is read from a clamped address
            // and replaced by 0), then the few elements by which the windows are wider than the block.
            const long long xbase = r0 + x_shift + threadIdx.x;
            const long long xlast = x_extent - 1;
            auto fetch = [&](long long xi) {
                const long long c = xi < 0 ? 0 : (xi > xlast ? xlast : xi);
                const T got = x[c];
                return c == xi ? got : T(0);
            };
            constexpr int QB = 4;
            for (int q0 = 0; q0 < nruns; q0 += QB) {
                T got[QB][R];
                // (run slots past the last one repeat it: staged again, harmlessly)
                int start[QB], off[QB];
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    start[qi] = info.run_start[q0 + qi];
                    off[qi] = info.run_off[q0 + qi];
                }
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    const long long first = xbase + start[qi];
#pragma unroll
                    for (int i = 0; i < R; ++i) got[qi][i] = fetch(first + i * kBlock);
                }
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    T *dst = xs + off[qi] + threadIdx.x;
#pragma unroll
                    for (int i = 0; i < R; ++i) dst[i * kBlock] = got[qi][i];
                }
            }
            if (static_cast<int>(threadIdx.x) < info.ntail)
                xs[info.tail_off[threadIdx.x]] = fetch(r0 + x_shift + info.tail_dist[threadIdx.x]);
            // every entry's place in LDS into scalar registers now: a scalar load between the LDS
            // reads below would make each of them wait for the one before
            unsigned packed[(CAP + 1) / 2];
#pragma unroll
            for (int k = 0; k < (CAP + 1) / 2; ++k) packed[k] = info.lds_off_packed[k];
            if (!GENERAL && LSK_SLICED_STAGE_FIRST) {
                __builtin_amdgcn_sched_barrier(0); // (the value loads stay behind the staging, whose registers they take)
                load_chunk(0);
            }
            __syncthreads();
            // A wave whose rows all have the whole list adds without looking at masks.
            const unsigned lenmask = maxlen >= 32 ? 0xffffffffu : (1u << maxlen) - 1u;
            const bool plain = !masked && __all(all_live);
            const T *lane_xs = xs + threadIdx.x * R;
            auto entry = [&](int j, auto plain_tag) {
                if (j < maxlen) { // (wave-uniform)
                    const T *src = lane_xs + ((packed[j / 2] >> (16 * (j % 2))) & 0xffffu);
                    T xv[R];
#pragma unroll
                    for (int h = 0; h < R; ++h) xv[h] = src[h];
                    if constexpr (decltype(plain_tag)::value) {
#pragma unroll
                        for (int h = 0; h < R; ++h) acc[h] = acc[h] + v[j][h] * xv[h];
                    } else {
                        // (the empty asm pins the LDS reads here: left alone, the compiler moves each
                        // one behind a branch on its row's mask bit, and the reads then wait one by one)
#pragma unroll
                        for (int h = 0; h < R; ++h) asm volatile("" : "+v"(xv[h]));
#pragma unroll
                        for (int h = 0; h < R; ++h) {
                            const T sum = acc[h] + v[j][h] * xv[h];
                            acc[h] = ((mask[h] & lenmask) >> j) & 1u ? sum : acc[h];
                        }
                    }
                }
            };
            if (plain) {
#pragma unroll
                for (int c = 0; c < NCHUNKS; ++c) {
                    if (c + 1 < NCHUNKS) load_chunk(c + 1);
#pragma unroll
                    for (int j = c * CHUNK; j < (c + 1) * CHUNK && j < CAP; ++j) entry(j, std::true_type{});
                    __builtin_amdgcn_sched_barrier(0); // (no load of a later burst moves up past this one's use)
                }
            } else {
#pragma unroll
                for (int c = 0; c < NCHUNKS; ++c) {
                    if (c + 1 < NCHUNKS) load_chunk(c + 1);
#pragma unroll
                    for (int j = c * CHUNK; j < (c + 1) * CHUNK && j < CAP; ++j) entry(j, std::false_type{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (GENERAL) {
            // a list with too many runs for the staging area: x straight from memory
#pragma unroll
            for (int c = 1; c < NCHUNKS; ++c) load_chunk(c);
            const T *xp = x + (r + x_shift);
#pragma unroll
            for (int j = 0; j < CAP; ++j) {
                if (j < maxlen) {
                    const int d = info.dist[j];
#pragma unroll
                    for (int h = 0; h < R; ++h)
                        if ((mask[h] >> j) & 1u) acc[h] = acc[h] + v[j][h] * xp[h + d];
                }
            }
        }
    } else if constexpr (GENERAL) {
        // Rows whose patterns fit no common list: every row looks its own distances up and adds the
        // entries it has (entry j at position j).
#pragma unroll
        for (int c = 1; c < NCHUNKS; ++c) load_chunk(c);
#pragma unroll
        for (int h = 0; h < R; ++h) {
            if (!live[h]) continue;
            const PatternInfo &mine = table[row_pattern[r + h]];
            const int len = mine.len;
            const T *xp = x + (r + h + x_shift);
#pragma unroll
            for (int j = 0; j < CAP; ++j)
                if (j < len) acc[h] = acc[h] + v[j][h] * xp[mine.dist[j]];
        }
    }

    if constexpr (DOT) {
        T share = T(0);
        if (all_live && yw_aligned) {
            const V wv = *reinterpret_cast<const V *>(w + r);
#pragma unroll
            for (int h = 0; h < R; ++h) share = share + acc[h] * wv[h];
        } else {
#pragma unroll
            for (int h = 0; h < R; ++h)
                if (live[h]) share = share + acc[h] * w[r + h];
        }
        share = lsk::wave_sum(share);
        if (lane == 0) wave_share[wave] = share;
    }
    // (the barrier of the staged path, or this one, also gathers the block's stores: 4 KiB at a time)
    __syncthreads();
    if (all_live && yw_aligned) {
        V out;
#pragma unroll
        for (int h = 0; h < R; ++h) out[h] = acc[h];
        store16_nontemporal(y + r, out);
    } else {
#pragma unroll
        for (int h = 0; h < R; ++h)
            if (live[h]) y[r + h] = acc[h];
    }
    if constexpr (DOT) {
        if (threadIdx.x == 0) {
            T total = wave_share[0];
#pragma unroll
            for (int i = 1; i < kWavesPerBlock; ++i) total = total + wave_share[i];
            partials[wg] = total;
        }
    }
}

// The wave-staged form: every wave stages the windows of x of its OWN 64 R rows (64 R + length - 1
// elements per run) in its own part of LDS and multiplies without waiting for the block's other
// waves; the block meets once, at the end, to write y together (and to add the waves' dot shares).
// K = 2: two right-hand sides, one after the other through the same LDS area, on values that were
// loaded once (two systems on one matrix, what the reference's BenchmarkStencil steps,
// test/BenchmarkStencil.cpp:181-199); each is summed in entry order, bit for bit what the one-vector
// product gives for it.  Staged lists only (every block on one staged list).
#ifndef LSK_WAVE_AHEAD
#define LSK_WAVE_AHEAD 0
#endif
#ifndef LSK_WAVE_K2_WAVES
#define LSK_WAVE_K2_WAVES 2
#endif
template <int K, int CAP> constexpr int sliced_wave_waves_per_simd() {
    return CAP > 27 ? 2 : (K == 1 && CAP <= 20) ? 4 : (K == 2 && CAP > 20) ? LSK_WAVE_K2_WAVES : 3; // (what each holds without spilling)
}
// RUN3: every list is runs of three consecutive distances.  The R rows of a lane then need R + 2
// consecutive elements of a run's window, taken with two aligned 16-byte LDS reads per run instead of
// three unaligned ones per entry -- a third of the LDS time -- and used in the same entry order.
// DMA: a window that lies wholly inside the vector goes from memory straight into LDS (one
// global_load_lds_dwordx4 per window and wave: no registers, nothing to wait for before the next
// request), the others through registers as before.
// UNIFORM: entry j of the block's list has one value for every row (a constant-coefficient stencil): the
// values come from `uniform` as scalars and the value array is not read at all.
template <typename T, int CAP, bool DOT, int K, bool RUN3, bool DMA, bool UNIFORM>
#ifndef LSK_UNIFORM_WAVES
#define LSK_UNIFORM_WAVES 4
#endif
__global__ __launch_bounds__(kBlock, (UNIFORM ? LSK_UNIFORM_WAVES : sliced_wave_waves_per_simd<K, CAP>())) void sliced_wave_kernel(
    int64_t block_begin, int64_t row_begin, int64_t row_end, const BlockMeta *__restrict__ meta,
    const T *__restrict__ val, const unsigned *__restrict__ row_mask, const PatternInfo *__restrict__ table,
    const T *__restrict__ xa, const T *__restrict__ xb, long long x_shift, long long x_extent, T *__restrict__ ya,
    T *__restrict__ yb, int beta, int remap, const T *__restrict__ wa, const T *__restrict__ wb, T *__restrict__ pa,
    T *__restrict__ pb, int lds_wave_elems, int yw_aligned, const T *__restrict__ uniform, int w_is_x) {
    // w_is_x (DOT): bit k: w of vector k is the very memory x's rows are (the product's own input vector, as
    // in CG's p . A p), so the shares take it from the window at distance 0, which is in LDS already,
    // instead of reading it a second time.
    constexpr int R = Lane<T>::R;
    using V = typename Lane<T>::V;
    constexpr int kBlockRows = kBlock * R, kWaveRows = kWave * R;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ T wave_share[K][kWavesPerBlock];

    const unsigned wg = lsk::xcd_remap(blockIdx.x, gridDim.x, remap);
    const int64_t block = block_begin + wg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x % kWave;
    T *xs = reinterpret_cast<T *>(lds_raw) + wave * lds_wave_elems;
    const BlockMeta m = meta[block];
    const long long base = lsk::wave_first(m.base);
    const int maxlen = __builtin_amdgcn_readfirstlane(m.maxlen);
    const int pattern = __builtin_amdgcn_readfirstlane(m.pattern);
    const int64_t r0 = block * kBlockRows + int64_t(wave) * kWaveRows; // the wave's first row
    const int64_t r = r0 + int64_t(lane) * R;
    const T *vp = val + base + int64_t(wave) * maxlen * kWaveRows + lane * R;

    bool live[R];
    bool all_live = true;
#pragma unroll
    for (int h = 0; h < R; ++h) {
        live[h] = r + h >= row_begin && r + h < row_end;
        all_live = all_live && live[h];
    }
    const PatternInfo &info = table[pattern & 0xff]; // wave-uniform address
    unsigned mask[R];
    const bool masked = (pattern >> (kWaveMaskedShift + wave)) & 1;
#pragma unroll
    for (int h = 0; h < R; ++h) mask[h] = !live[h] ? 0u : masked ? row_mask[r + h] : 0xffffffffu;
    const int nruns = info.nruns;
    const long long xbase = r0 + x_shift + lane;
    const long long xlast = x_extent - 1;
    const unsigned lenmask = maxlen >= 32 ? 0xffffffffu : (1u << maxlen) - 1u;
    const bool plain = !masked && __all(all_live);
    V v[UNIFORM ? 1 : CAP];
    T uv[UNIFORM ? CAP : 1]; // UNIFORM: the list's values (scalars)
    auto value = [&](int j, int h) -> T {
        if constexpr (UNIFORM) return uv[j];
        else return v[j][h];
    };
    unsigned packed[(CAP + 1) / 2];
    T result[K][R];
    // DIRECT (UNIFORM, lists of runs of three): x straight from memory into registers, no LDS at all.  With
    // no values to wait for, what a wave spends is the chain meta -> windows -> sums; without the LDS areas
    // (9.5 KiB per wave for the 27-point list) twice as many waves walk that chain at a time.
#ifndef LSK_UNIFORM_DIRECT
#define LSK_UNIFORM_DIRECT 1
#endif
    constexpr bool DIRECT = LSK_UNIFORM_DIRECT && UNIFORM && RUN3;
    // (27-point 512^3 fp64: 1.27 ms through LDS, 0.92 ms direct; one ALIGNED 16-byte load per lane and run
    // with the two outer elements taken from the neighbouring lanes' registers: 1.57 ms.)
    // AHEAD: both vectors' windows requested before anything is summed, each into its own LDS area; else
    // the vectors share one area, one after the other.  Measured with the windows going straight into
    // LDS: ahead 7.5 ms, one after the other 6.3 ms for the pair on the 27-point 512^3 fp64 matrix (the
    // second area takes the third workgroup from the CU), so it is off.
    constexpr bool AHEAD = LSK_WAVE_AHEAD && DMA && K == 2;
#ifndef LSK_WAVE_QB
#define LSK_WAVE_QB (CAP > 20 ? 9 : CAP > 8 ? 6 : 5)
#endif
    constexpr int QB = LSK_WAVE_QB;
    const bool has_tail = lane < info.w_ntail;
    T tail[K];
    auto fetch_from = [&](const T *x, long long xi) {
        const long long c = xi < 0 ? 0 : (xi > xlast ? xlast : xi);
        const T got = x[c];
        return c == xi ? got : T(0);
    };
    // Requests vector k's windows: all of a list's windows at once (a 27-point list has 9), and the few
    // elements by which they are wider than the wave's rows with them -- one trip to memory, not one per
    // batch.  Those few elements come back in a register and are written by finish_stage.
    // One window on its way: kWaveRows elements of x from distance `start` of the wave's first row to
    // `off` in the wave's LDS area.
    auto stage_window = [&](const T *x, T *area, int start, int off) {
        const long long first = r0 + x_shift + start; // (wave-uniform)
        T *dst = area + off;
        if (first >= 0 && first + kWaveRows <= x_extent) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + first + lane * R),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        } else { // a window that reaches past an end of the vector: element by element, clamped
#pragma unroll
            for (int i = 0; i < R; ++i) dst[lane + i * kWave] = fetch_from(x, first + lane + i * kWave);
        }
    };
    // (lists of runs of three: the windows' distances and places are few enough for scalar registers,
    // loaded together instead of one dependent scalar load per window)
    constexpr int NQ = RUN3 ? CAP / 3 : 1;
    int rstart[NQ], roff[NQ];
    if constexpr (RUN3 && DMA) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            rstart[q] = info.run_start[q];
            roff[q] = info.w_run_off[q];
        }
    }
    auto stage = [&](int k, T *area) {
        const T *x = k == 0 ? xa : xb;
        tail[k] = T(0);
        if (has_tail) tail[k] = fetch_from(x, r0 + x_shift + info.w_tail_dist[lane]);
        if constexpr (DMA && RUN3) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nruns) stage_window(x, area, rstart[q], roff[q]);
        } else if constexpr (DMA) {
            for (int q = 0; q < nruns; ++q) stage_window(x, area, info.run_start[q], info.w_run_off[q]);
        } else {
            for (int q0 = 0; q0 < nruns; q0 += QB) {
                T got[QB][R];
                int start[QB], off[QB];
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    const int q = q0 + qi < kMaxRuns ? q0 + qi : kMaxRuns - 1; // (slots past the last run repeat it)
                    start[qi] = info.run_start[q];
                    off[qi] = info.w_run_off[q];
                }
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    const long long first = xbase + start[qi];
#pragma unroll
                    for (int i = 0; i < R; ++i) got[qi][i] = fetch_from(x, first + i * kWave);
                }
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    T *dst = area + off[qi] + lane;
#pragma unroll
                    for (int i = 0; i < R; ++i) dst[i * kWave] = got[qi][i];
                }
            }
        }
    };
    auto load_values = [&]() {
        // every entry's place in LDS into scalar registers now: a scalar load between the LDS reads
        // below would make each of them wait for the one before
#pragma unroll
        for (int i = 0; i < (CAP + 1) / 2; ++i) packed[i] = info.w_lds_off_packed[i];
        if constexpr (UNIFORM) {
            const T *mine = uniform + (pattern & 0xff) * kMaxLength; // (wave-uniform address: scalar loads)
#pragma unroll
            for (int j = 0; j < CAP; ++j) uv[j] = mine[j]; // (zero past the list's end)
            return;
        }
        __builtin_amdgcn_sched_barrier(0); // (the value loads stay behind the staging, whose registers they take)
#pragma unroll
        for (int j = 0; j < CAP; ++j) {
            if (j < maxlen) v[j] = __builtin_nontemporal_load(reinterpret_cast<const V *>(vp + int64_t(j) * kWaveRows));
            else v[j] = V(0);
        }
    };
    if constexpr (AHEAD) {
        stage(0, xs);
        stage(1, xs + lds_wave_elems / 2);
        load_values();
    }
    // (Tried for K = 2 and dropped: requesting the second vector's window q into its place as soon as the
    // first vector's sums have read it, so that it travels while the remaining runs are summed.  The pair
    // on the 27-point 512^3 fp64 matrix: 5.17 -> 5.51 ms with both right-hand sides the same vector, 6.45 ->
    // 6.31 / 5.55 -> 5.92 ms with two vectors and either placement: the second trip is not what the pair
    // waits for; the second vector's share of L2 is, see lsk_csr_sliced_tune2_*.)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const T *yin = k == 0 ? ya : yb;
        T *area = AHEAD ? xs + k * (lds_wave_elems / 2) : xs;
        if constexpr (!AHEAD) {
            if constexpr (!DIRECT) stage(k, area);
            if (k == 0) load_values();
        }
        // A window that went straight into LDS has landed when the load counter says so.  The wave's own
        // LDS writes are in place before its reads: LDS serves a wave's accesses in order, and the fence
        // keeps the compiler from moving the reads up.
        constexpr int NV = (R + 2 + R - 1) / R; // 16-byte pieces that cover the R + 2 elements of a run of three
        V xg[DIRECT ? CAP / 3 : 1][NV];
        if constexpr (DIRECT) {
            const T *x = k == 0 ? xa : xb;
#pragma unroll
            for (int q = 0; q < CAP / 3; ++q) {
                if (3 * q >= maxlen) continue; // (wave-uniform)
                const long long first = r0 + x_shift + rstart[q];
                if (first >= 0 && first + kWaveRows + R <= x_extent) {
                    const T *src = x + first + lane * R;
#pragma unroll
                    for (int i = 0; i < NV; ++i) __builtin_memcpy(&xg[q][i], src + i * R, sizeof(V)); // (8-byte aligned)
                } else { // a window that reaches past an end of the vector: element by element, clamped
#pragma unroll
                    for (int i = 0; i < NV; ++i)
#pragma unroll
                        for (int c = 0; c < R; ++c) xg[q][i][c] = fetch_from(x, first + lane * R + i * R + c);
                }
            }
        } else {
            if constexpr (DMA) {
                if (!AHEAD || k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (has_tail) area[info.w_tail_off[lane]] = tail[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const T *lane_xs = area + lane * R;
        T acc[R];
#pragma unroll
        for (int h = 0; h < R; ++h) acc[h] = (beta && live[h]) ? yin[r + h] : T(0);
        auto entry = [&](int j, auto plain_tag) {
            if (j < maxlen) { // (wave-uniform)
                const T *src = lane_xs + ((packed[j / 2] >> (16 * (j % 2))) & 0xffffu);
                T xv[R];
#pragma unroll
                for (int h = 0; h < R; ++h) xv[h] = src[h];
                if constexpr (decltype(plain_tag)::value) {
#pragma unroll
                    for (int h = 0; h < R; ++h) acc[h] = acc[h] + value(j, h) * xv[h];
                } else {
#pragma unroll
                    for (int h = 0; h < R; ++h) asm volatile("" : "+v"(xv[h]));
#pragma unroll
                    for (int h = 0; h < R; ++h) {
                        const T sum = acc[h] + value(j, h) * xv[h];
                        acc[h] = ((mask[h] & lenmask) >> j) & 1u ? sum : acc[h];
                    }
                }
            }
        };
        auto run_of_three = [&](int q, auto plain_tag) {
            if (3 * q < maxlen) { // (wave-uniform; the list is whole runs)
                // packed[] holds the place of every entry: that of the run's first is the window's start
                const T *src = lane_xs + ((packed[(3 * q) / 2] >> (16 * ((3 * q) % 2))) & 0xffffu);
                V xw[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if constexpr (DIRECT) xw[i] = xg[q][i];
                    else xw[i] = *reinterpret_cast<const V *>(src + i * R);
                }
                if constexpr (!decltype(plain_tag)::value) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(xw[i]));
                }
                T xe[R + 2]; // the R + 2 elements of x the lane's R rows read in this run
#pragma unroll
                for (int i = 0; i < R + 2; ++i) xe[i] = xw[i / R][i % R];
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const int j = 3 * q + e;
#pragma unroll
                    for (int h = 0; h < R; ++h) {
                        const T xv = xe[e + h];
                        const T sum = acc[h] + value(j, h) * xv;
                        if constexpr (decltype(plain_tag)::value) acc[h] = sum;
                        else acc[h] = ((mask[h] & lenmask) >> j) & 1u ? sum : acc[h];
                    }
                }
            }
        };
        if constexpr (RUN3) {
            static_assert(CAP % 3 == 0, "lists of runs of three");
            if (plain) {
#pragma unroll
                for (int q = 0; q < CAP / 3; ++q) run_of_three(q, std::true_type{});
            } else {
#pragma unroll
                for (int q = 0; q < CAP / 3; ++q) run_of_three(q, std::false_type{});
            }
        } else if (plain) {
#pragma unroll
            for (int j = 0; j < CAP; ++j) entry(j, std::true_type{});
        } else {
#pragma unroll
            for (int j = 0; j < CAP; ++j) entry(j, std::false_type{});
        }
#pragma unroll
        for (int h = 0; h < R; ++h) result[k][h] = acc[h];
        if constexpr (DOT) {
            const T *w = k == 0 ? wa : wb;
            T share = T(0);
            const int own = info.w_diag_off; // (wave-uniform)
            if (!DIRECT && ((w_is_x >> k) & 1) && own >= 0) { // w is x: the rows' own elements are in the windows
                const T *mine = lane_xs + own;
#pragma unroll
                for (int h = 0; h < R; ++h)
                    if (live[h]) share = share + acc[h] * mine[h];
            } else if (all_live && yw_aligned) {
                const V wv = *reinterpret_cast<const V *>(w + r);
#pragma unroll
                for (int h = 0; h < R; ++h) share = share + acc[h] * wv[h];
            } else {
#pragma unroll
                for (int h = 0; h < R; ++h)
                    if (live[h]) share = share + acc[h] * w[r + h];
            }
            share = lsk::wave_sum(share);
            if (lane == 0) wave_share[k][wave] = share;
        }
        if (k + 1 < K) { // the reads are done before the next vector's windows overwrite the area
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the block's only barrier: its stores leave together, 4 KiB per vector
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        T *y = k == 0 ? ya : yb;
        if (all_live && yw_aligned) {
            V out;
#pragma unroll
            for (int h = 0; h < R; ++h) out[h] = result[k][h];
            store16_nontemporal(y + r, out);
        } else {
#pragma unroll
            for (int h = 0; h < R; ++h)
                if (live[h]) y[r + h] = result[k][h];
        }
    }
    if constexpr (DOT) {
        if (threadIdx.x < K) {
            T *pk = threadIdx.x == 0 ? pa : pb;
            T total = wave_share[threadIdx.x][0];
#pragma unroll
            for (int i = 1; i < kWavesPerBlock; ++i) total = total + wave_share[threadIdx.x][i];
            pk[wg] = total;
        }
    }
}

// Two products with one pass over the values: y_a <- A x_a, y_b <- A x_b (two systems on one matrix,
// what the reference's BenchmarkStencil steps, test/BenchmarkStencil.cpp:181-199).  The staged form
// only (every block on one staged list); both windows of x in LDS (twice the space, so two
// workgroups per CU and no register limit to speak of); each right-hand side summed in entry order,
// bit for bit what sliced_spmv_kernel gives for it alone.
template <typename T, int CAP, bool DOT>
__global__ __launch_bounds__(kBlock, 2) void sliced_spmv2_kernel(
    int64_t block_begin, int64_t row_begin, int64_t row_end, const BlockMeta *__restrict__ meta,
    const T *__restrict__ val, const unsigned *__restrict__ row_mask, const PatternInfo *__restrict__ table,
    const T *__restrict__ xa, const T *__restrict__ xb, long long x_shift, long long x_extent, T *__restrict__ ya,
    T *__restrict__ yb, int beta, int remap, const T *__restrict__ wa, const T *__restrict__ wb, T *__restrict__ pa,
    T *__restrict__ pb, int lds_stride, int yw_aligned) {
    constexpr int R = Lane<T>::R;
    constexpr int K = 2;
    using V = typename Lane<T>::V;
    constexpr int kBlockRows = kBlock * R, kWaveRows = kWave * R;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *xs = reinterpret_cast<T *>(lds_raw);
    __shared__ T wave_share[K][kWavesPerBlock];

    const unsigned wg = lsk::xcd_remap(blockIdx.x, gridDim.x, remap);
    const int64_t block = block_begin + wg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x % kWave;
    const BlockMeta m = meta[block];
    const long long base = lsk::wave_first(m.base);
    const int maxlen = __builtin_amdgcn_readfirstlane(m.maxlen);
    const int pattern = __builtin_amdgcn_readfirstlane(m.pattern);
    const int64_t r0 = block * kBlockRows;
    const int64_t r = r0 + int64_t(threadIdx.x) * R;
    const T *vp = val + base + int64_t(wave) * maxlen * kWaveRows + lane * R;
    const T *xk[K] = {xa, xb};
    T *yk[K] = {ya, yb};

    bool live[R];
    bool all_live = true;
    T acc[K][R];
#pragma unroll
    for (int h = 0; h < R; ++h) {
        live[h] = r + h >= row_begin && r + h < row_end;
        all_live = all_live && live[h];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k][h] = (beta && live[h]) ? yk[k][r + h] : T(0);
    }
    const PatternInfo &info = table[pattern & 0xff]; // wave-uniform address
    unsigned mask[R];
    const bool masked = (pattern >> (kWaveMaskedShift + wave)) & 1;
#pragma unroll
    for (int h = 0; h < R; ++h) mask[h] = !live[h] ? 0u : masked ? row_mask[r + h] : 0xffffffffu;
    const int nruns = info.nruns;
    const long long xbase = r0 + x_shift + threadIdx.x;
    const long long xlast = x_extent - 1;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const T *x = xk[k];
        T *xsk = xs + k * lds_stride;
        auto fetch = [&](long long xi) {
            const long long c = xi < 0 ? 0 : (xi > xlast ? xlast : xi);
            const T got = x[c];
            return c == xi ? got : T(0);
        };
        constexpr int QB = 4;
        for (int q0 = 0; q0 < nruns; q0 += QB) {
            T got[QB][R];
            int start[QB], off[QB];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                start[qi] = info.run_start[q0 + qi];
                off[qi] = info.run_off[q0 + qi];
            }
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                const long long first = xbase + start[qi];
#pragma unroll
                for (int i = 0; i < R; ++i) got[qi][i] = fetch(first + i * kBlock);
            }
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                T *dst = xsk + off[qi] + threadIdx.x;
#pragma unroll
                for (int i = 0; i < R; ++i) dst[i * kBlock] = got[qi][i];
            }
        }
        if (static_cast<int>(threadIdx.x) < info.ntail)
            xsk[info.tail_off[threadIdx.x]] = fetch(r0 + x_shift + info.tail_dist[threadIdx.x]);
    }
    unsigned packed[(CAP + 1) / 2];
#pragma unroll
    for (int k = 0; k < (CAP + 1) / 2; ++k) packed[k] = info.lds_off_packed[k];
    __builtin_amdgcn_sched_barrier(0);
    V v[CAP];
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
        if (j < maxlen) v[j] = __builtin_nontemporal_load(reinterpret_cast<const V *>(vp + int64_t(j) * kWaveRows));
        else v[j] = V(0);
    }
    __syncthreads();
    const unsigned lenmask = maxlen >= 32 ? 0xffffffffu : (1u << maxlen) - 1u;
    const bool plain = !masked && __all(all_live);
    const T *lane_xs = xs + threadIdx.x * R;
    auto entry = [&](int j, auto plain_tag) {
        if (j < maxlen) { // (wave-uniform)
            const unsigned off = (packed[j / 2] >> (16 * (j % 2))) & 0xffffu;
            T xv[K][R];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int h = 0; h < R; ++h) xv[k][h] = lane_xs[k * lds_stride + off + h];
            if constexpr (decltype(plain_tag)::value) {
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int h = 0; h < R; ++h) acc[k][h] = acc[k][h] + v[j][h] * xv[k][h];
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int h = 0; h < R; ++h) asm volatile("" : "+v"(xv[k][h]));
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int h = 0; h < R; ++h) {
                        const T sum = acc[k][h] + v[j][h] * xv[k][h];
                        acc[k][h] = ((mask[h] & lenmask) >> j) & 1u ? sum : acc[k][h];
                    }
            }
        }
    };
    if (plain) {
#pragma unroll
        for (int j = 0; j < CAP; ++j) entry(j, std::true_type{});
    } else {
#pragma unroll
        for (int j = 0; j < CAP; ++j) entry(j, std::false_type{});
    }

    if constexpr (DOT) {
        const T *wk[K] = {wa, wb};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            T share = T(0);
            if (all_live && yw_aligned) {
                const V wv = *reinterpret_cast<const V *>(wk[k] + r);
#pragma unroll
                for (int h = 0; h < R; ++h) share = share + acc[k][h] * wv[h];
            } else {
#pragma unroll
                for (int h = 0; h < R; ++h)
                    if (live[h]) share = share + acc[k][h] * wk[k][r + h];
            }
            share = lsk::wave_sum(share);
            if (lane == 0) wave_share[k][wave] = share;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (all_live && yw_aligned) {
            V out;
#pragma unroll
            for (int h = 0; h < R; ++h) out[h] = acc[k][h];
            store16_nontemporal(yk[k] + r, out);
        } else {
#pragma unroll
            for (int h = 0; h < R; ++h)
                if (live[h]) yk[k][r + h] = acc[k][h];
        }
    }
    if constexpr (DOT) {
        if (threadIdx.x < K) {
            T *pk = threadIdx.x == 0 ? pa : pb;
            T total = wave_share[threadIdx.x][0];
#pragma unroll
            for (int i = 1; i < kWavesPerBlock; ++i) total = total + wave_share[threadIdx.x][i];
            pk[wg] = total;
        }
    }
}

// Runs of consecutive distances of a list, and where each entry's x value sits in the staging area
// of a block of `block_rows` rows.  Returns false (nruns = 0) when the list needs more than
// kMaxRuns runs or more than kMaxStageBytes.
bool plan_staging(PatternInfo &p, int block_rows, size_t elem_bytes) {
    p.nruns = 0;
    p.lds_elems = 0;
    if (p.len <= 0) return false;
    int nruns = 0, off = 0;
    int run_of[kMaxLength];
    for (int j = 0; j < p.len; ++j) {
        if (j > 0 && p.dist[j] == p.dist[j - 1] + 1) {
            p.run_len[nruns - 1] += 1;
        } else {
            if (nruns == kMaxRuns) return false;
            p.run_start[nruns] = p.dist[j];
            p.run_len[nruns] = 1;
            ++nruns;
        }
        run_of[j] = nruns - 1;
    }
    for (int q = 0; q < nruns; ++q) {
        p.run_off[q] = off;
        off += (block_rows + p.run_len[q] - 1 + 3) & ~3; // keep the windows 16-byte aligned
    }
    if (static_cast<size_t>(off) * elem_bytes > kMaxStageBytes) return false;
    for (int j = 0; j < p.len; ++j) p.lds_off[j] = p.run_off[run_of[j]] + (p.dist[j] - p.run_start[run_of[j]]);
    for (int k = 0; k < kMaxLength / 2; ++k)
        p.lds_off_packed[k] = static_cast<unsigned>(p.lds_off[2 * k]) | (static_cast<unsigned>(p.lds_off[2 * k + 1]) << 16);
    // unused run slots repeat the last run, so that the kernel can take them four at a time
    for (int q = nruns; q < kMaxRuns; ++q) {
        p.run_start[q] = p.run_start[nruns - 1];
        p.run_len[q] = p.run_len[nruns - 1];
        p.run_off[q] = p.run_off[nruns - 1];
    }
    p.ntail = 0;
    for (int q = 0; q < nruns; ++q)
        for (int e = 0; e + 1 < p.run_len[q]; ++e) {
            p.tail_dist[p.ntail] = p.run_start[q] + block_rows + e;
            p.tail_off[p.ntail] = p.run_off[q] + block_rows + e;
            ++p.ntail;
        }
    p.nruns = nruns;
    p.lds_elems = off;
    return true;
}

// The wave-level plan of a list whose block-level plan exists (same runs, windows of wave_rows rows).
void plan_wave_staging(PatternInfo &p, int wave_rows) {
    int off = 0, run_of[kMaxLength], lds_off[kMaxLength];
    {
        int q = -1;
        for (int j = 0; j < p.len; ++j) {
            if (j == 0 || p.dist[j] != p.dist[j - 1] + 1) ++q;
            run_of[j] = q;
        }
    }
    for (int q = 0; q < p.nruns; ++q) {
        p.w_run_off[q] = off;
        off += (wave_rows + p.run_len[q] - 1 + 3) & ~3;
    }
    for (int q = p.nruns; q < kMaxRuns; ++q) p.w_run_off[q] = p.w_run_off[p.nruns - 1];
    for (int j = 0; j < kMaxLength; ++j)
        lds_off[j] = j < p.len ? p.w_run_off[run_of[j]] + (p.dist[j] - p.run_start[run_of[j]]) : 0;
    for (int k = 0; k < kMaxLength / 2; ++k)
        p.w_lds_off_packed[k] = static_cast<unsigned>(lds_off[2 * k]) | (static_cast<unsigned>(lds_off[2 * k + 1]) << 16);
    p.w_diag_off = -1;
    for (int j = 0; j < p.len; ++j)
        if (p.dist[j] == 0) p.w_diag_off = lds_off[j];
    p.w_ntail = 0;
    for (int q = 0; q < p.nruns; ++q)
        for (int e = 0; e + 1 < p.run_len[q]; ++e) {
            p.w_tail_dist[p.w_ntail] = p.run_start[q] + wave_rows + e;
            p.w_tail_off[p.w_ntail] = p.w_run_off[q] + wave_rows + e;
            ++p.w_ntail;
        }
    p.w_lds_elems = off;
}

template <typename T>
int uniform_refresh(lsk_csr_sliced *m, hipStream_t s, bool wait);

template <typename T>
int create(const char *name, int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, const int64_t *col,
           const T *val, int64_t row_first, int64_t x_lo, int64_t x_hi, lsk_stream_t stream, lsk_csr_sliced **out) {
    constexpr int R = Lane<T>::R;
    constexpr int kBlockRows = kBlock * R;
    if (!out) return lsk::fail(LSK_ERR_ARG, "%s: null output", name);
    *out = nullptr;
    if (rows < 0 || nnz < 0) return lsk::fail(LSK_ERR_ARG, "%s: negative size", name);
    if (rows == 0 || nnz == 0 || x_hi < x_lo) return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: empty piece", name);
    if (!rects || !col || !val) return lsk::fail(LSK_ERR_ARG, "%s: null pointer", name);
    if (rows > (int64_t(1) << 31) * kBlock - kBlockRows) return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: too many rows", name);
    hipStream_t s = lsk::as_stream(stream);
    const int64_t blocks = lsk::ceil_div(rows, kBlockRows);
    const int64_t groups = lsk::ceil_div(rows, kWave);
    const unsigned row_grid = static_cast<unsigned>(lsk::ceil_div(rows, kBlock));
    const unsigned block_grid = static_cast<unsigned>(lsk::ceil_div(blocks, kBlock));

    lsk_csr_sliced *m = new (std::nothrow) lsk_csr_sliced;
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: out of host memory", name);
    m->elem_bytes = sizeof(T);
    m->rows = rows;
    m->nnz = nnz;
    m->k_base = k_base;
    m->row_first = row_first;
    m->x_lo = x_lo;
    m->x_hi = x_hi;
    m->blocks = blocks;
    unsigned long long *keys = nullptr;
    long long *rep = nullptr, *entries = nullptr, *inclusive = nullptr, *counts = nullptr;
    std::vector<PatternInfo> infos; // the lists, host side
    int *flags = nullptr, *lists = nullptr, *g_maxlen = nullptr, *g_holder = nullptr, *g_uniform = nullptr, *blk_flags = nullptr;
    unsigned long long *supersets = nullptr;
    void *temp = nullptr;
    int status = LSK_OK;
    int host_flags[kFlags] = {0};
    auto release_scratch = [&] {
        for (void *p : {static_cast<void *>(keys), static_cast<void *>(rep), static_cast<void *>(entries),
                        static_cast<void *>(inclusive), static_cast<void *>(counts), static_cast<void *>(flags),
                        static_cast<void *>(lists), static_cast<void *>(g_maxlen), static_cast<void *>(g_holder),
                        static_cast<void *>(g_uniform), static_cast<void *>(blk_flags), static_cast<void *>(supersets), temp})
            (void)hipFree(p);
    };
#define LSK_TRY(expr)                                                                               \
    do {                                                                                            \
        const hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess && status == LSK_OK) status = lsk::hip_fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)
#define LSK_ALLOC(ptr, count) LSK_TRY(hipMalloc(reinterpret_cast<void **>(&(ptr)), static_cast<size_t>(count) * sizeof(*(ptr))))
    LSK_ALLOC(keys, kMaxPatterns);
    LSK_ALLOC(rep, kMaxPatterns);
    LSK_ALLOC(flags, kFlags);
    LSK_ALLOC(counts, 4);
    LSK_ALLOC(supersets, kMaxPatterns);
    LSK_ALLOC(lists, kMaxPatterns * (kMaxLength + 1));
    LSK_ALLOC(entries, blocks);
    LSK_ALLOC(inclusive, blocks);
    LSK_ALLOC(blk_flags, blocks);
    LSK_ALLOC(g_maxlen, groups);
    LSK_ALLOC(g_holder, groups);
    LSK_ALLOC(g_uniform, groups);
    LSK_ALLOC(m->row_pattern, rows);
    LSK_ALLOC(m->row_mask, rows);
    LSK_ALLOC(m->table, kMaxPatterns);
    LSK_TRY(hipMalloc(&m->meta, static_cast<size_t>(blocks) * sizeof(BlockMeta)));
#undef LSK_ALLOC
    const long long *rc = reinterpret_cast<const long long *>(rects);
    const long long *cl = reinterpret_cast<const long long *>(col);
    bool declined = false;
    if (status == LSK_OK) {
        sliced_init_kernel<<<1, kBlock, 0, s>>>(keys, flags, counts, lists);
        LSK_TRY(hipGetLastError());
        sliced_assign_kernel<<<row_grid, kBlock, 0, s>>>(rows, nnz, rc, k_base, cl, row_first, keys, rep, m->row_pattern, flags);
        LSK_TRY(hipGetLastError());
        LSK_TRY(hipMemcpyAsync(host_flags, flags, sizeof(host_flags), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipStreamSynchronize(s));
        declined = host_flags[kFlagTooManyPatterns] || host_flags[kFlagRowTooLong] || host_flags[kFlagBadRect];
    }
    if (status == LSK_OK && !declined) {
        // the distance lists, and on the host their runs and places in the staging area
        sliced_lists_kernel<<<1, kMaxPatterns, 0, s>>>(rc, k_base, cl, row_first, keys, rep, lists, flags);
        LSK_TRY(hipGetLastError());
        std::vector<int> host_lists(kMaxPatterns * (kMaxLength + 1));
        LSK_TRY(hipMemcpyAsync(host_lists.data(), lists, host_lists.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipMemcpyAsync(host_flags, flags, sizeof(host_flags), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipStreamSynchronize(s));
        declined = host_flags[kFlagOutsideWindow] != 0;
        if (status == LSK_OK && !declined) {
            infos.assign(kMaxPatterns, PatternInfo{});
            for (int p = 0; p < kMaxPatterns; ++p) {
                PatternInfo &info = infos[p];
                info = PatternInfo{};
                info.w_diag_off = -1;
                const int *mine = host_lists.data() + p * (kMaxLength + 1);
                info.len = mine[0];
                for (int j = 0; j < kMaxLength; ++j) info.dist[j] = j < info.len ? mine[1 + j] : 0;
                if (info.len >= 0) {
                    m->patterns += 1;
                    m->max_len = std::max(m->max_len, info.len);
                    if (plan_staging(info, kBlockRows, sizeof(T))) {
                        m->lds_bytes = std::max(m->lds_bytes, static_cast<size_t>(info.lds_elems) * sizeof(T));
                        plan_wave_staging(info, kBlockRows / kWavesPerBlock);
                        m->wave_lds_bytes = std::max(m->wave_lds_bytes, static_cast<size_t>(info.w_lds_elems) * sizeof(T));

                    } else if (info.len > 0)
                        m->all_staged = false;
                }
            }
            LSK_TRY(hipMemcpyAsync(m->table, infos.data(), kMaxPatterns * sizeof(PatternInfo), hipMemcpyHostToDevice, s));
            // which staged lists every list is an ordered part of (itself included): what a block whose
            // rows do not fit its longest row's list may run on instead (sliced_rescue_kernel)
            unsigned long long host_supersets[kMaxPatterns];
            for (int p = 0; p < kMaxPatterns; ++p) {
                host_supersets[p] = 0ull;
                if (infos[p].len < 0) continue;
                for (int c = 0; c < kMaxPatterns; ++c) {
                    if (infos[c].len < infos[p].len || infos[c].nruns <= 0) continue;
                    int pos = 0;
                    bool fits = true;
                    for (int j = 0; fits && j < infos[p].len; ++j) {
                        while (pos < infos[c].len && infos[c].dist[pos] != infos[p].dist[j]) ++pos;
                        fits = pos++ < infos[c].len;
                    }
                    if (fits) host_supersets[p] |= 1ull << c;
                }
            }
            LSK_TRY(hipMemcpyAsync(supersets, host_supersets, sizeof(host_supersets), hipMemcpyHostToDevice, s));
            LSK_TRY(hipStreamSynchronize(s)); // (infos and host_supersets leave scope)
        }
    }
    if (status == LSK_OK && !declined) {
        sliced_verify_kernel<<<row_grid, kBlock, 0, s>>>(rows, rc, k_base, cl, row_first, x_lo, x_hi, m->row_pattern, m->table,
                                                         g_maxlen, g_holder, g_uniform, flags);
        LSK_TRY(hipGetLastError());
        LSK_TRY(hipMemcpyAsync(host_flags, flags, sizeof(host_flags), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipStreamSynchronize(s));
        declined = host_flags[kFlagMismatch] || host_flags[kFlagOutsideWindow];
    }
    if (status == LSK_OK && !declined) {
        BlockMeta *meta = static_cast<BlockMeta *>(m->meta);
        sliced_combine_kernel<<<block_grid, kBlock, 0, s>>>(blocks, groups, kBlockRows / kWave, g_maxlen, g_holder, g_uniform,
                                                            meta, blk_flags);
        LSK_TRY(hipGetLastError());
        sliced_masks_kernel<<<row_grid, kBlock, 0, s>>>(rows, kBlockRows, m->row_pattern, m->table, meta, blk_flags, m->row_mask);
        LSK_TRY(hipGetLastError());
        sliced_rescue_kernel<<<static_cast<unsigned>(blocks), kBlock, 0, s>>>(rows, kBlockRows, m->row_pattern, m->table, supersets,
                                                                             meta, blk_flags);
        LSK_TRY(hipGetLastError());
        sliced_masks_kernel<<<row_grid, kBlock, 0, s>>>(rows, kBlockRows, m->row_pattern, m->table, meta, blk_flags, m->row_mask);
        LSK_TRY(hipGetLastError());
        sliced_finish_kernel<<<block_grid, kBlock, 0, s>>>(blocks, rows, kBlockRows, blk_flags, meta, entries, counts);
        LSK_TRY(hipGetLastError());
        size_t temp_bytes = 0;
        LSK_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, temp_bytes, entries, inclusive, static_cast<int>(blocks), s));
        LSK_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 1));
        if (status == LSK_OK)
            LSK_TRY(hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, entries, inclusive, static_cast<int>(blocks), s));
        sliced_bases_kernel<<<block_grid, kBlock, 0, s>>>(blocks, entries, inclusive, meta);
        LSK_TRY(hipGetLastError());
        long long total = 0, counts_host[4] = {0, 0, 0, 0};
        LSK_TRY(hipMemcpyAsync(&total, inclusive + (blocks - 1), sizeof(total), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipMemcpyAsync(counts_host, counts, sizeof(counts_host), hipMemcpyDeviceToHost, s));
        LSK_TRY(hipStreamSynchronize(s));
        m->padded_entries = total;
        m->mixed_blocks = counts_host[0];
        m->masked_waves = counts_host[1];
        // lists of runs of exactly three: judged on the lists that serve blocks (a row at the end of a grid
        // line has a list of its own with shorter runs, but its block runs on the line's list with a mask)
        m->runs_of_three = m->mixed_blocks == 0;
        for (int p = 0; p < kMaxPatterns && p < static_cast<int>(infos.size()); ++p)
            if ((static_cast<unsigned long long>(counts_host[3]) >> p) & 1ull) {
                if (infos[p].nruns <= 0) m->runs_of_three = false;
                for (int q = 0; q < infos[p].nruns; ++q)
                    if (infos[p].run_len[q] != 3) m->runs_of_three = false;
            }
        // not worth it when padding short rows to their block's longest would cost more than the column
        // array saves (the rows the last block lacks are left out: at most one block per piece)
        if (status == LSK_OK && static_cast<double>(counts_host[2]) * sizeof(T) > static_cast<double>(nnz) * (sizeof(T) + 3.0))
            declined = true;
        if (status == LSK_OK && !declined) {
            LSK_TRY(hipMalloc(&m->val, static_cast<size_t>(std::max<long long>(total, 1)) * sizeof(T)));
            if (status == LSK_OK) {
                const int64_t padded_rows = blocks * kBlockRows;
                sliced_fill_kernel<T><<<static_cast<unsigned>(lsk::ceil_div(padded_rows, kBlock)), kBlock, 0, s>>>(
                    rows, padded_rows, rc, k_base, val, meta, m->row_mask, static_cast<T *>(m->val));
                LSK_TRY(hipGetLastError());
                LSK_TRY(hipStreamSynchronize(s));
            }
            if (status == LSK_OK) status = uniform_refresh<T>(m, s, /*wait=*/true);
        }
    }
#undef LSK_TRY
    release_scratch();
    if (status != LSK_OK || declined) {
        (void)lsk_csr_sliced_destroy(m);
        if (status != LSK_OK) return status;
        return lsk::decline(LSK_ERR_UNSUPPORTED,
                            "%s: rows do not repeat a few column patterns (too many patterns %d, row too long %d, "
                            "rect outside the piece %d, mismatch %d, column outside the window %d, or too much padding)",
                            name, host_flags[kFlagTooManyPatterns], host_flags[kFlagRowTooLong], host_flags[kFlagBadRect],
                            host_flags[kFlagMismatch], host_flags[kFlagOutsideWindow]);
    }
    *out = m;
    return LSK_OK;
}

inline bool wave_form_serves(const lsk_csr_sliced *m);

// Finds out whether the copy's values are uniform per list position (see lsk_csr_sliced::uniform) and
// leaves the lists' values in m->uniform.  wait: the verdict is in m->uniform_ok on return (set-up time);
// else everything is enqueued on the stream and the verdict is picked up later (uniform_verdict).
template <typename T>
int uniform_refresh(lsk_csr_sliced *m, hipStream_t s, bool wait) {
    m->uniform_ok = false;
    m->uniform_pending = false;
    if (!wave_form_serves(m) || m->blocks <= 0 || m->blocks > INT_MAX) return LSK_OK;
    if (wait && !m->uniform) { // (the resources are made at set-up time only: update_values allocates nothing)
        LSK_HIP(hipMalloc(&m->uniform, sizeof(T) * kMaxPatterns * kMaxLength));
        LSK_HIP(hipMalloc(reinterpret_cast<void **>(&m->uniform_scratch), sizeof(int) * (kMaxPatterns + 1)));
        LSK_HIP(hipHostMalloc(reinterpret_cast<void **>(&m->uniform_host), sizeof(int)));
        LSK_HIP(hipEventCreateWithFlags(&m->uniform_event, hipEventDisableTiming));
    }
    if (!m->uniform || !m->uniform_scratch || !m->uniform_host || !m->uniform_event) return LSK_OK;
    const BlockMeta *meta = static_cast<const BlockMeta *>(m->meta);
    const T *val = static_cast<const T *>(m->val);
    T *uniform = static_cast<T *>(m->uniform);
    sliced_uniform_init_kernel<<<1, kMaxPatterns + 1, 0, s>>>(m->uniform_scratch);
    LSK_LAUNCH_CHECK();
    sliced_uniform_first_kernel<<<static_cast<unsigned>(m->blocks), kBlock, 0, s>>>(m->rows, Lane<T>::R, meta, m->row_mask,
                                                                                   m->uniform_scratch);
    LSK_LAUNCH_CHECK();
    sliced_uniform_pick_kernel<T><<<kMaxPatterns, kBlock, 0, s>>>(m->rows, meta, m->row_mask, val, m->uniform_scratch, uniform);
    LSK_LAUNCH_CHECK();
    sliced_uniform_check_kernel<T><<<static_cast<unsigned>(m->blocks), kBlock, 0, s>>>(m->rows, meta, m->row_mask, val, uniform,
                                                                                      m->uniform_scratch,
                                                                                      m->uniform_scratch + kMaxPatterns);
    LSK_LAUNCH_CHECK();
    LSK_HIP(hipMemcpyAsync(m->uniform_host, m->uniform_scratch + kMaxPatterns, sizeof(int), hipMemcpyDeviceToHost, s));
    LSK_HIP(hipEventRecord(m->uniform_event, s));
    m->uniform_pending = true;
    if (wait) {
        LSK_HIP(hipEventSynchronize(m->uniform_event));
        m->uniform_ok = *m->uniform_host == 0;
        m->uniform_pending = false;
    }
    return LSK_OK;
}

// Whether the structure's values are known to be uniform per list position.  While a check is still on
// its way (after update_values) the answer is no -- the product that reads the values is always right --
// unless the caller is willing to wait for it.
inline bool uniform_verdict(const lsk_csr_sliced *m, bool wait, hipStream_t launching_on = nullptr) {
    if (m->uniform_pending && !wait && launching_on) {
        // a product being recorded into a graph makes no runtime query: it reads the values
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(launching_on, &capturing) != hipSuccess) (void)hipGetLastError();
        if (capturing != hipStreamCaptureStatusNone) return false;
    }
    if (m->uniform_pending) {
        const hipError_t e = wait ? hipEventSynchronize(m->uniform_event) : hipEventQuery(m->uniform_event);
        if (e == hipSuccess) {
            m->uniform_ok = *m->uniform_host == 0;
            m->uniform_pending = false;
        } else if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            m->uniform_pending = false; // (an error: stay with the value array)
        }
    }
    return !m->uniform_pending && m->uniform_ok;
}

// The values again, from the caller's array into the blocks (the last set-up pass alone): for a caller
// that has rewritten the entries of the matrix in place.  The structure (rects, columns) must be the
// one the copy was made from; that is the caller's promise, only the sizes can be checked here.
template <typename T>
int update_values(const char *name, lsk_csr_sliced *m, int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                  const T *val, lsk_stream_t stream) {
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: null structure", name);
    if (m->elem_bytes != static_cast<int>(sizeof(T))) return lsk::fail(LSK_ERR_ARG, "%s: structure of another entry type", name);
    if (rows != m->rows || nnz != m->nnz || k_base != m->k_base)
        return lsk::fail(LSK_ERR_ARG, "%s: %lld rows, %lld entries from %lld; the structure was made for %lld, %lld from %lld",
                         name, (long long)rows, (long long)nnz, (long long)k_base, (long long)m->rows, (long long)m->nnz,
                         (long long)m->k_base);
    if (!rects || !val) return lsk::fail(LSK_ERR_ARG, "%s: null pointer", name);
    const int64_t padded_rows = m->blocks * kBlock * Lane<T>::R;
    sliced_fill_kernel<T><<<static_cast<unsigned>(lsk::ceil_div(padded_rows, kBlock)), kBlock, 0, lsk::as_stream(stream)>>>(
        rows, padded_rows, reinterpret_cast<const long long *>(rects), k_base, val, static_cast<const BlockMeta *>(m->meta),
        m->row_mask, static_cast<T *>(m->val));
    LSK_LAUNCH_CHECK();
    // (whether the new values are uniform per list position is found out again, behind the refill on the
    // same stream; nobody waits for it: until the verdict is in, the products read the value array)
    return uniform_refresh<T>(m, lsk::as_stream(stream), /*wait=*/false);
}

// Whether the wave-staged kernel serves the structure: every block on one staged list, and the four
// waves' windows within the 64 KiB a launch gets without asking.
inline bool wave_form_serves(const lsk_csr_sliced *m) {
    return m->all_staged && m->mixed_blocks == 0 && m->wave_lds_bytes > 0 &&
           m->wave_lds_bytes * kWavesPerBlock <= (size_t(64) << 10);
}

template <typename T, int K>
int launch_wave(const lsk_csr_sliced *m, int64_t block_begin, unsigned blocks, int64_t row_begin, int64_t row_end,
                const T *xa, const T *xb, T *ya, T *yb, int beta, const T *wa, const T *wb, T *pa, T *pb, bool dot,
                int yw_aligned, hipStream_t s) {
    const BlockMeta *meta = static_cast<const BlockMeta *>(m->meta);
    const T *val = static_cast<const T *>(m->val);
    const long long x_shift = m->row_first - m->x_lo, x_extent = m->x_hi - m->x_lo + 1;
    const int remap = lsk::xcd_remap_option();
    const bool dma = lsk::options().sliced_dma != 0;
    const int areas = (LSK_WAVE_AHEAD && dma && K == 2) ? 2 : 1; // LDS areas per wave (see AHEAD in the kernel)
    const int wave_elems = static_cast<int>(m->wave_lds_bytes / sizeof(T)) * areas;
    size_t lds = m->wave_lds_bytes * kWavesPerBlock * areas;
    // w the very rows of x (CG's p . A p): the shares take it from the staged windows (see the kernel)
    const int w_is_x = !dot ? 0 : (wa == xa + x_shift ? 1 : 0) | (K == 2 && wb == xb + x_shift ? 2 : 0);
#define LSK_WAVE_LAUNCH2(CAP, DOT, RUN3, DMA, UNIFORM)                                                             \
    do {                                                                                                           \
        auto kernel = sliced_wave_kernel<T, CAP, DOT, K, RUN3, DMA, UNIFORM>;                                      \
        if (lds > (size_t(64) << 10)) {                                                                            \
            static std::atomic<uint64_t> lds_allowed{0};                                                           \
            if (const int rc_ = lsk::allow_large_lds(reinterpret_cast<const void *>(kernel), lds_allowed, s); rc_ != LSK_OK) \
                return rc_;                                                                                        \
        }                                                                                                          \
        kernel<<<blocks, kBlock, lds, s>>>(block_begin, row_begin, row_end, meta, val, m->row_mask, m->table, xa, xb, \
                                           x_shift, x_extent, ya, yb, beta, remap, wa, wb, pa, pb, wave_elems,      \
                                           yw_aligned, static_cast<const T *>(m->uniform), w_is_x);                \
    } while (0)
#define LSK_WAVE_LAUNCH(CAP, DOT, RUN3)                                                                            \
    do {                                                                                                           \
        if (dma && uniform) LSK_WAVE_LAUNCH2(CAP, DOT, RUN3, true, true);                                          \
        else if (dma) LSK_WAVE_LAUNCH2(CAP, DOT, RUN3, true, false);                                               \
        else LSK_WAVE_LAUNCH2(CAP, DOT, RUN3, false, false);                                                       \
    } while (0)
#define LSK_WAVE_CAP(CAP, RUN3)                                                                                    \
    do {                                                                                                           \
        if (dot) LSK_WAVE_LAUNCH(CAP, true, RUN3);                                                                 \
        else LSK_WAVE_LAUNCH(CAP, false, RUN3);                                                                    \
    } while (0)
    const bool run3 = m->runs_of_three && lsk::options().sliced_runs != 0;
    // the values as scalars, where every list position has one value and the caller asked for it
    const bool uniform = lsk::options().sliced_uniform != 0 && m->uniform && uniform_verdict(m, /*wait=*/false, s);
    if (LSK_UNIFORM_DIRECT && uniform && dma && run3 && m->max_len > 20 && m->max_len <= 27) lds = 0; // (DIRECT in the kernel)
    if (m->max_len <= 8) LSK_WAVE_CAP(8, false);
    else if (m->max_len <= 16) LSK_WAVE_CAP(16, false);
    else if (m->max_len <= 20) LSK_WAVE_CAP(20, false);
    else if (m->max_len <= 27 && run3) LSK_WAVE_CAP(27, true);
    else if (m->max_len <= 27) LSK_WAVE_CAP(27, false);
    else LSK_WAVE_CAP(32, false);
#undef LSK_WAVE_CAP
#undef LSK_WAVE_LAUNCH
#undef LSK_WAVE_LAUNCH2
    LSK_LAUNCH_CHECK();
    return LSK_OK;
}

// Partial sums the fused product + dot leaves for rows row_begin .. row_end - 1: one per block touched.
template <typename T>
int64_t dot_partials(int64_t row_begin, int64_t row_end) {
    constexpr int64_t kBlockRows = int64_t(kBlock) * Lane<T>::R;
    if (row_end <= row_begin) return 0;
    return lsk::ceil_div(row_end, kBlockRows) - row_begin / kBlockRows;
}

template <typename T>
int spmv(const char *name, const lsk_csr_sliced *m, int64_t row_begin, int64_t row_end, const T *x, T *y, int beta,
         const T *w, T *partials, int64_t capacity, int64_t *count, lsk_stream_t stream) {
    constexpr int64_t kBlockRows = int64_t(kBlock) * Lane<T>::R;
    const bool dot = count != nullptr;
    if (dot) *count = 0;
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: null structure", name);
    if (m->elem_bytes != static_cast<int>(sizeof(T))) return lsk::fail(LSK_ERR_ARG, "%s: structure of another entry type", name);
    if (row_begin < 0 || row_end > m->rows || row_begin > row_end)
        return lsk::fail(LSK_ERR_ARG, "%s: rows %lld..%lld of %lld", name, (long long)row_begin, (long long)row_end,
                         (long long)m->rows);
    if (beta != 0 && beta != 1) return lsk::fail(LSK_ERR_ARG, "%s: beta must be 0 or 1", name);
    if (row_begin == row_end) return LSK_OK;
    if (!x || !y) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    const int64_t block_begin = row_begin / kBlockRows;
    const unsigned blocks = static_cast<unsigned>(dot_partials<T>(row_begin, row_end));
    if (dot) {
        if (!w || !partials) return lsk::fail(LSK_ERR_ARG, "%s: null w or partials", name);
        if (capacity < static_cast<int64_t>(blocks))
            return lsk::fail(LSK_ERR_ARG, "%s: room for %lld partial sums, %u needed", name, (long long)capacity, blocks);
    }
    const BlockMeta *meta = static_cast<const BlockMeta *>(m->meta);
    const T *val = static_cast<const T *>(m->val);
    const long long x_shift = m->row_first - m->x_lo, x_extent = m->x_hi - m->x_lo + 1;
    const int remap = lsk::xcd_remap_option();
    const int yw_aligned = lsk::aligned16(y) && (!dot || lsk::aligned16(w));
    hipStream_t s = lsk::as_stream(stream);
    const bool general = !m->all_staged || m->mixed_blocks > 0;
    if (lsk::options().sliced_form == 1 && wave_form_serves(m)) {
        const int rc = launch_wave<T, 1>(m, block_begin, blocks, row_begin, row_end, x, nullptr, y, nullptr, beta, w, nullptr,
                                         partials, nullptr, dot, yw_aligned, s);
        if (rc == LSK_OK && dot) *count = blocks;
        return rc;
    }
#define LSK_SLICED_LAUNCH(CAP, DOT, GENERAL)                                                                      \
    sliced_spmv_kernel<T, CAP, DOT, GENERAL><<<blocks, kBlock, m->lds_bytes, s>>>(                                \
        block_begin, m->rows, row_begin, row_end, meta, val, m->row_pattern, m->row_mask, m->table, x, x_shift,   \
        x_extent, y, beta, remap, w, partials, yw_aligned)
#define LSK_SLICED_CAP(CAP)                                                                                       \
    do {                                                                                                          \
        if (dot && general) LSK_SLICED_LAUNCH(CAP, true, true);                                                   \
        else if (dot) LSK_SLICED_LAUNCH(CAP, true, false);                                                        \
        else if (general) LSK_SLICED_LAUNCH(CAP, false, true);                                                    \
        else LSK_SLICED_LAUNCH(CAP, false, false);                                                                \
    } while (0)
    if (m->max_len <= 8) LSK_SLICED_CAP(8);
    else if (m->max_len <= 16) LSK_SLICED_CAP(16);
    else if (m->max_len <= 20) LSK_SLICED_CAP(20);
    else if (m->max_len <= 27) LSK_SLICED_CAP(27);
    else LSK_SLICED_CAP(32);
#undef LSK_SLICED_CAP
#undef LSK_SLICED_LAUNCH
    LSK_LAUNCH_CHECK();
    if (dot) *count = blocks;
    return LSK_OK;
}

// Whether the two-vector kernels serve the structure: every block on one staged list, both windows within a CU's LDS.
inline bool pair_served(const lsk_csr_sliced *m) {
    return m->all_staged && m->mixed_blocks == 0 && 2 * m->lds_bytes <= (size_t(150) << 10);
}

template <typename T>
int spmv2(const char *name, const lsk_csr_sliced *m, int64_t row_begin, int64_t row_end, const T *xa, const T *xb, T *ya,
          T *yb, int beta, const T *wa, const T *wb, T *pa, T *pb, int64_t capacity, int64_t *count, lsk_stream_t stream) {
    constexpr int64_t kBlockRows = int64_t(kBlock) * Lane<T>::R;
    const bool dot = count != nullptr;
    if (dot) *count = 0;
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: null structure", name);
    if (m->elem_bytes != static_cast<int>(sizeof(T))) return lsk::fail(LSK_ERR_ARG, "%s: structure of another entry type", name);
    if (row_begin < 0 || row_end > m->rows || row_begin > row_end)
        return lsk::fail(LSK_ERR_ARG, "%s: rows %lld..%lld of %lld", name, (long long)row_begin, (long long)row_end,
                         (long long)m->rows);
    if (beta != 0 && beta != 1) return lsk::fail(LSK_ERR_ARG, "%s: beta must be 0 or 1", name);
    if (!pair_served(m))
        return lsk::decline(LSK_ERR_UNSUPPORTED, "%s: not every block of this structure runs on one staged list", name);
    if (row_begin == row_end) return LSK_OK;
    if (!xa || !xb || !ya || !yb) return lsk::fail(LSK_ERR_ARG, "%s: null vector", name);
    if (ya == yb) return lsk::fail(LSK_ERR_ARG, "%s: both results in one vector", name);
    const int64_t block_begin = row_begin / kBlockRows;
    const unsigned blocks = static_cast<unsigned>(dot_partials<T>(row_begin, row_end));
    if (dot) {
        if (!wa || !wb || !pa || !pb) return lsk::fail(LSK_ERR_ARG, "%s: null w or partials", name);
        if (capacity < static_cast<int64_t>(blocks))
            return lsk::fail(LSK_ERR_ARG, "%s: room for %lld partial sums, %u needed", name, (long long)capacity, blocks);
    }
    const BlockMeta *meta = static_cast<const BlockMeta *>(m->meta);
    const T *val = static_cast<const T *>(m->val);
    const long long x_shift = m->row_first - m->x_lo, x_extent = m->x_hi - m->x_lo + 1;
    const int remap = lsk::xcd_remap_option();
    const int yw_aligned = lsk::aligned16(ya) && lsk::aligned16(yb) && (!dot || (lsk::aligned16(wa) && lsk::aligned16(wb)));
    const int lds_stride = static_cast<int>(m->lds_bytes / sizeof(T));
    const size_t lds = 2 * m->lds_bytes;
    hipStream_t s = lsk::as_stream(stream);
    const int pair_form = lsk::options().sliced_pair_form;
    if (pair_form != 0 && wave_form_serves(m)) {
        const int rc = launch_wave<T, 2>(m, block_begin, blocks, row_begin, row_end, xa, xb, ya, yb, beta, wa, wb, pa, pb, dot,
                                         yw_aligned, s);
        if (rc == LSK_OK && dot) *count = blocks;
        return rc;
    }
#define LSK_SLICED2_LAUNCH(CAP, DOT)                                                                                \
    do {                                                                                                            \
        auto kernel = sliced_spmv2_kernel<T, CAP, DOT>;                                                             \
        static std::atomic<uint64_t> lds_allowed{0};                                                                \
        if (const int rc_ = lsk::allow_large_lds(reinterpret_cast<const void *>(kernel), lds_allowed, s); rc_ != LSK_OK) \
            return rc_;                                                                                             \
        kernel<<<blocks, kBlock, lds, s>>>(block_begin, row_begin, row_end, meta, val, m->row_mask, m->table, xa, xb, \
                                           x_shift, x_extent, ya, yb, beta, remap, wa, wb, pa, pb, lds_stride,       \
                                           yw_aligned);                                                             \
    } while (0)
#define LSK_SLICED2_CAP(CAP)                                                                                        \
    do {                                                                                                            \
        if (dot) LSK_SLICED2_LAUNCH(CAP, true);                                                                     \
        else LSK_SLICED2_LAUNCH(CAP, false);                                                                        \
    } while (0)
    if (m->max_len <= 8) LSK_SLICED2_CAP(8);
    else if (m->max_len <= 16) LSK_SLICED2_CAP(16);
    else if (m->max_len <= 20) LSK_SLICED2_CAP(20);
    else if (m->max_len <= 27) LSK_SLICED2_CAP(27);
    else LSK_SLICED2_CAP(32);
#undef LSK_SLICED2_CAP
#undef LSK_SLICED2_LAUNCH
    LSK_LAUNCH_CHECK();
    if (dot) *count = blocks;
    return LSK_OK;
}

// Times the candidates of the workgroup -> XCD placement on this structure and reports the fastest.
template <typename T>
int tune(const char *name, const lsk_csr_sliced *m, const T *x, T *y, lsk_stream_t stream, int *remap_out, float *ms_out,
         const T *xb = nullptr, T *yb = nullptr) {
    // (xb, yb: the two-vector product is what is timed; it answers LSK_ERR_UNSUPPORTED where no two-vector
    // form serves the structure)
    // (short runs too: two blocks per XCD share six of their nine windows' lines in that XCD's L2 and still spread
    // the value stream over the XCDs as round robin does: 4.54 against 4.63 ms on the 27-point 512^3 fp64 matrix)
    static const int candidates[] = {0, 2, 4, 16, 64, 256, 1024, 4096};
    auto product = [&]() {
        return xb ? spmv2<T>(name, m, 0, m->rows, x, xb, y, yb, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, stream)
                  : spmv<T>(name, m, 0, m->rows, x, y, 0, nullptr, nullptr, 0, nullptr, stream);
    };
    constexpr int kWarm = 1, kTimed = 3;
    if (!m) return lsk::fail(LSK_ERR_ARG, "%s: null structure", name);
    LSK_NOT_WHILE_RECORDING(name, stream);
    int &placement = lsk::xcd_remap_option();
    const int saved = placement;
    hipStream_t s = lsk::as_stream(stream);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    LSK_HIP(hipEventCreate(&t0));
    if (const hipError_t e = hipEventCreate(&t1); e != hipSuccess) {
        (void)hipEventDestroy(t0);
        return lsk::hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
    }
    int best = saved, status = LSK_OK;
    float best_ms = -1.0f;
    for (int c : candidates) {
        placement = c;
        for (int i = 0; i < kWarm && status == LSK_OK; ++i) status = product();
        if (status != LSK_OK) break;
        if (hipEventRecord(t0, s) != hipSuccess) { status = lsk::fail(LSK_ERR_HIP, "%s: event record", name); break; }
        for (int i = 0; i < kTimed && status == LSK_OK; ++i) status = product();
        if (status != LSK_OK) break;
        float ms = 0.0f;
        if (hipEventRecord(t1, s) != hipSuccess || hipEventSynchronize(t1) != hipSuccess ||
            hipEventElapsedTime(&ms, t0, t1) != hipSuccess) {
            status = lsk::fail(LSK_ERR_HIP, "%s: event timing", name);
            break;
        }
        ms /= kTimed;
        if (std::getenv("LSK_TUNE_VERBOSE"))
            std::fprintf(stderr, "[lsk] %s: xcd_remap %d -> %.3f ms per launch\n", name, c, ms);
        if (best_ms < 0.0f || ms < best_ms) {
            best_ms = ms;
            best = c;
        }
    }
    placement = saved;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    if (status != LSK_OK) return status;
    if (remap_out) *remap_out = best;
    if (ms_out) *ms_out = best_ms;
    return LSK_OK;
}

} // namespace

extern "C" {

int lsk_csr_sliced_tune_f32(const lsk_csr_sliced_t *m, const float *x, float *y, lsk_stream_t stream, int *xcd_remap,
                            float *ms) {
    return tune<float>("lsk_csr_sliced_tune_f32", m, x, y, stream, xcd_remap, ms);
}
int lsk_csr_sliced_tune_f64(const lsk_csr_sliced_t *m, const double *x, double *y, lsk_stream_t stream, int *xcd_remap,
                            float *ms) {
    return tune<double>("lsk_csr_sliced_tune_f64", m, x, y, stream, xcd_remap, ms);
}

int lsk_csr_sliced_tune2_f32(const lsk_csr_sliced_t *m, const float *xa, const float *xb, float *ya, float *yb,
                             lsk_stream_t stream, int *xcd_remap, float *ms) {
    if (!xa || !xb || !ya || !yb) return lsk::fail(LSK_ERR_ARG, "lsk_csr_sliced_tune2_f32: null vector");
    return tune<float>("lsk_csr_sliced_tune2_f32", m, xa, ya, stream, xcd_remap, ms, xb, yb);
}
int lsk_csr_sliced_tune2_f64(const lsk_csr_sliced_t *m, const double *xa, const double *xb, double *ya, double *yb,
                             lsk_stream_t stream, int *xcd_remap, float *ms) {
    if (!xa || !xb || !ya || !yb) return lsk::fail(LSK_ERR_ARG, "lsk_csr_sliced_tune2_f64: null vector");
    return tune<double>("lsk_csr_sliced_tune2_f64", m, xa, ya, stream, xcd_remap, ms, xb, yb);
}

int lsk_csr_sliced_create_f32(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, const int64_t *col,
                              const float *val, int64_t row_first, int64_t x_lo, int64_t x_hi, lsk_stream_t stream,
                              lsk_csr_sliced_t **out) {
    return create<float>("lsk_csr_sliced_create_f32", rows, nnz, rects, k_base, col, val, row_first, x_lo, x_hi, stream, out);
}
int lsk_csr_sliced_create_f64(int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base, const int64_t *col,
                              const double *val, int64_t row_first, int64_t x_lo, int64_t x_hi, lsk_stream_t stream,
                              lsk_csr_sliced_t **out) {
    return create<double>("lsk_csr_sliced_create_f64", rows, nnz, rects, k_base, col, val, row_first, x_lo, x_hi, stream, out);
}

int lsk_csr_sliced_update_values_f32(lsk_csr_sliced_t *m, int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                                     const float *val, lsk_stream_t stream) {
    return update_values<float>("lsk_csr_sliced_update_values_f32", m, rows, nnz, rects, k_base, val, stream);
}
int lsk_csr_sliced_update_values_f64(lsk_csr_sliced_t *m, int64_t rows, int64_t nnz, const int64_t *rects, int64_t k_base,
                                     const double *val, lsk_stream_t stream) {
    return update_values<double>("lsk_csr_sliced_update_values_f64", m, rows, nnz, rects, k_base, val, stream);
}

int lsk_csr_sliced_spmv_f32(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const float *x, float *y,
                            int beta, lsk_stream_t stream) {
    return spmv<float>("lsk_csr_sliced_spmv_f32", m, row_begin, row_end, x, y, beta, nullptr, nullptr, 0, nullptr, stream);
}
int lsk_csr_sliced_spmv_f64(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const double *x, double *y,
                            int beta, lsk_stream_t stream) {
    return spmv<double>("lsk_csr_sliced_spmv_f64", m, row_begin, row_end, x, y, beta, nullptr, nullptr, 0, nullptr, stream);
}
int lsk_csr_sliced_spmv_dot_f32(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const float *x, float *y,
                                int beta, const float *w, float *partials, int64_t capacity, int64_t *count,
                                lsk_stream_t stream) {
    if (!count) return lsk::fail(LSK_ERR_ARG, "lsk_csr_sliced_spmv_dot_f32: null count");
    return spmv<float>("lsk_csr_sliced_spmv_dot_f32", m, row_begin, row_end, x, y, beta, w, partials, capacity, count, stream);
}
int lsk_csr_sliced_spmv_dot_f64(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const double *x, double *y,
                                int beta, const double *w, double *partials, int64_t capacity, int64_t *count,
                                lsk_stream_t stream) {
    if (!count) return lsk::fail(LSK_ERR_ARG, "lsk_csr_sliced_spmv_dot_f64: null count");
    return spmv<double>("lsk_csr_sliced_spmv_dot_f64", m, row_begin, row_end, x, y, beta, w, partials, capacity, count, stream);
}

#define LSK_SLICED2(SUF, T)                                                                                          \
    int lsk_csr_sliced_spmv2_##SUF(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const T *xa,       \
                                   const T *xb, T *ya, T *yb, int beta, lsk_stream_t stream) {                       \
        return spmv2<T>("lsk_csr_sliced_spmv2_" #SUF, m, row_begin, row_end, xa, xb, ya, yb, beta, nullptr, nullptr, \
                        nullptr, nullptr, 0, nullptr, stream);                                                      \
    }                                                                                                                \
    int lsk_csr_sliced_spmv2_dot_##SUF(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end, const T *xa,   \
                                       const T *xb, T *ya, T *yb, int beta, const T *wa, const T *wb, T *partials_a, \
                                       T *partials_b, int64_t capacity, int64_t *count, lsk_stream_t stream) {       \
        if (!count) return lsk::fail(LSK_ERR_ARG, "lsk_csr_sliced_spmv2_dot_" #SUF ": null count");                  \
        return spmv2<T>("lsk_csr_sliced_spmv2_dot_" #SUF, m, row_begin, row_end, xa, xb, ya, yb, beta, wa, wb,       \
                        partials_a, partials_b, capacity, count, stream);                                           \
    }
LSK_SLICED2(f32, float)
LSK_SLICED2(f64, double)
#undef LSK_SLICED2

int64_t lsk_csr_sliced_dot_partials(const lsk_csr_sliced_t *m, int64_t row_begin, int64_t row_end) {
    if (!m) return 0;
    return m->elem_bytes == 8 ? dot_partials<double>(row_begin, row_end) : dot_partials<float>(row_begin, row_end);
}

int lsk_csr_sliced_info(const lsk_csr_sliced_t *m, int64_t *out) {
    if (!m || !out) return lsk::fail(LSK_ERR_ARG, "lsk_csr_sliced_info: null pointer");
    out[0] = m->padded_entries;
    out[1] = m->blocks;
    out[2] = m->mixed_blocks;
    out[3] = m->patterns;
    out[4] = m->masked_waves;
    out[5] = int64_t(kBlock) * (16 / m->elem_bytes);
    out[6] = static_cast<int64_t>(m->lds_bytes);
    out[7] = pair_served(m) ? 1 : 0;
    out[8] = m->runs_of_three ? 1 : 0;
    return LSK_OK;
}

int lsk_csr_sliced_uniform_values(const lsk_csr_sliced_t *m) { return m && m->uniform && uniform_verdict(m, /*wait=*/true) ? 1 : 0; }

int lsk_csr_sliced_destroy(lsk_csr_sliced_t *m) {
    if (!m) return LSK_OK;
    (void)hipFree(m->val);
    (void)hipFree(m->meta);
    (void)hipFree(m->row_pattern);
    (void)hipFree(m->row_mask);
    (void)hipFree(m->table);
    (void)hipFree(m->uniform);
    (void)hipFree(m->uniform_scratch);
    if (m->uniform_host) (void)hipHostFree(m->uniform_host);
    if (m->uniform_event) (void)hipEventDestroy(m->uniform_event);
    delete m;
    return LSK_OK;
}

} // extern "C"
