// pfem_valdict.hpp -- the SpMV's copy of the matrix values as 16-bit codes into a dictionary of the DISTINCT values.
//
// WHY.  The CG iteration is the SpMV, and the SpMV is its bytes: the grouped forms stream 8 B of value + 1/2 B of column gap
// per stored slot at 0.89-0.93 of HBM peak -- nothing is left there.  But on a structured mesh (what genTetra.cpp writes, what
// north_star names) the element matrices repeat: the assembled K of config 3 holds 117 M entries and ~600 distinct bit
// patterns, the beam's 103 M entries ~2 600 (coordinates printed with 8 decimals: the cells differ by roundings, the sums by
// their order).  A dictionary of <= 4096 doubles sits in LDS (<= 32 KB a workgroup), a slot costs 2 B instead of 8, and the
// product is the SAME product: v = dict[code] is the very double the assembly wrote (lossless; same fma chain, same bits in
// y, in every dot and in every iterate -- tested).  Value indexing for sparse kernels is Kourtis, Goumas & Koziris 2008
// ("CSR-VI"); here it rides on the wave-sliced group forms.
//
// The values are re-assembled in every step, inside the reference's timers, so the codes are too:
//   steady state   k_vd_encode alone: every slot's value is looked up (bisection over the sorted dictionary in LDS), its code
//                  written; a value that is not in the dictionary raises `miss`
//   first time, or after a miss
//                  k_vd_collect (distinct bit patterns into an open-addressing table, a block-local filter in front of the
//                  global atomics) -> k_vd_finish (one workgroup: gather, bitonic sort, the dictionary) -> k_vd_encode
//   more than kVdMax distinct values (any unstructured mesh): the form is dropped for this pattern; the fp64 copy is what the
//   SpMV streams, as before.
// One read of the verdict (3 ints) by the host per assembly decides which kernel the solve launches.
#pragma once
#include "pfem_vdhash.hpp"

namespace pfem {

// distinct bit patterns of vals[0..n) into table (kVdTable slots, kVdEmpty = free); st->count of them, st->fail beyond kVdMax
__global__ void __launch_bounds__(kBlock) k_vd_collect(const double *__restrict__ vals, int64_t n, unsigned long long *table, VdState *st)
{
    __shared__ uint64_t seen[1024];             // values this block has already handed to the table (direct mapped)
    for (int i = threadIdx.x; i < 1024; i += kBlock) seen[i] = kVdEmpty;
    __syncthreads();
    const volatile int *failed = &st->fail;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const uint64_t b = static_cast<uint64_t>(__double_as_longlong(__builtin_nontemporal_load(vals + i)));
        // neighbouring lanes hold the same slot of neighbouring rows -- on a lattice mostly the same value: only the first lane of
        // a run looks it up (a lane that is active has an active left neighbour: its index is smaller)
        const uint64_t left = static_cast<uint64_t>(__shfl_up(static_cast<unsigned long long>(b), 1));
        if ((threadIdx.x & 63) != 0 && left == b) continue;
        if (b == kVdEmpty) { st->fail = 1; return; }    // (before the filter: `seen` starts out as kVdEmpty and would pass it for "seen")
        const uint32_t h = vd_hash(b);
        if (seen[h & 1023u] == b) continue;
        if (*failed) return;
        bool placed = false;
        for (uint32_t q = h & (kVdTable - 1), tries = 0; tries < kVdTable; q = (q + 1) & (kVdTable - 1), ++tries) {
            // (a plain look first: 600 values x 8192 workgroups would otherwise queue up on 600 words -- measured 20 ms)
            const unsigned long long there = __hip_atomic_load(table + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (there == b) { placed = true; break; }
            if (there != kVdEmpty) continue;
            const unsigned long long old = atomicCAS(table + q, static_cast<unsigned long long>(kVdEmpty), static_cast<unsigned long long>(b));
            if (old == kVdEmpty) {
                if (atomicAdd(&st->count, 1) + 1 > kVdMax) st->fail = 1;
                placed = true;
                break;
            }
            if (old == b) { placed = true; break; }
        }
        if (!placed) { st->fail = 1; return; }
        seen[h & 1023u] = b;                    // (racing writers store whole 8-byte words: a later reader sees one of them)
    }
}

// one workgroup: the table's entries sorted by bit pattern = the dictionary; st->count = their number
__global__ void __launch_bounds__(1024) k_vd_finish(const unsigned long long *__restrict__ table, double *__restrict__ dict, VdState *st)
{
    __shared__ uint64_t key[kVdMax];
    __shared__ int n_found;
    if (threadIdx.x == 0) n_found = 0;
    for (int i = threadIdx.x; i < kVdMax; i += 1024) key[i] = kVdEmpty;
    __syncthreads();
    if (st->fail) return;
    for (int q = threadIdx.x; q < kVdTable; q += 1024) {
        const uint64_t b = table[q];
        if (b != kVdEmpty) {
            const int at = atomicAdd(&n_found, 1);
            if (at < kVdMax) key[at] = b;
        }
    }
    __syncthreads();
    const int n = n_found;
    if (n > kVdMax) { if (threadIdx.x == 0) st->fail = 1; return; }
    int m = 2;                               // (the sort runs over the next power of two above n: 601 values -> 1024 slots, 55 stages instead of 78 over 4096)
    while (m < n) m <<= 1;
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < m; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const uint64_t a = key[i], c = key[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { key[i] = c; key[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 1024) dict[i] = __longlong_as_double(static_cast<long long>(key[i]));   // (kVdEmpty sorts last)
    if (threadIdx.x == 0) st->count = n;
}

// ---- the codes straight from the assembly kernel (round 6) -------------------------------------------------------------------
// In the steady state of a time loop the dictionary does not change: the gather kernel has every value of a row in its hands,
// so it looks the value up itself and writes the code where the encode pass would (k_gather_poisson_tet4): the pass over
// 118 M slots (0.31 ms at config 3) and the fp64 copy of the relative-group form it reads (945 MB written by the assembly) both
// go away.  The look-up is a hash table of the dictionary in global memory (L2-resident: kVdHashSlots x 16 B), open addressing,
// {bit pattern, code}; a value that is not there raises VdState::miss and the next solve falls back to the full path (re-pack
// of the fp64 copy, collection, sort, encode).  Slots the gather kernel never writes -- the explicit zeros of the group form --
// keep the codes the last full encode gave them: valid while the dictionary stands.
__global__ void __launch_bounds__(kBlock) k_vd_hash_build(const double *__restrict__ dict, const VdState *__restrict__ st, VdHashEntry *__restrict__ table)
{
    if (st->fail) return;
    const int n = min(st->count, kVdMax);
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(dict[i]));
    for (uint32_t h = vd_hash(b) & (kVdHashSlots - 1), tries = 0; tries < kVdHashSlots; h = (h + 1) & (kVdHashSlots - 1), ++tries) {
        const unsigned long long old = atomicCAS(&table[h].key, static_cast<unsigned long long>(kVdEmpty), b);
        if (old == kVdEmpty || old == b) { table[h].code = static_cast<unsigned long long>(i); return; }
    }
}
// codes of every slot of a group form with ROWS rows to the lane: slot (k, p) of lane l of a slice that starts at entry `off`
// lives at vals[ROWS * off + (ROWS * k + p) * 64 + l]; entry i = off + 64 k + l gets the word  code_0 | code_1 << 16 | ...
template <int ROWS>
__global__ void __launch_bounds__(kBlock) k_vd_encode(const double *__restrict__ vals, int64_t n_entries, const double *__restrict__ dict,
                                                       VdState *st, unsigned long long *__restrict__ codes)
{
    extern __shared__ uint64_t vd_keys[];
    if (st->fail) return;                        // (a collection that overflowed: nothing to encode against)
    const int nd = min(st->count, kVdMax);
    for (int i = threadIdx.x; i < nd; i += kBlock) vd_keys[i] = static_cast<uint64_t>(__double_as_longlong(dict[i]));
    __syncthreads();
    bool missed = false;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_entries; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int lane = static_cast<int>(i & 63);
        const double *vp = vals + ROWS * (i - lane) + lane;
        uint64_t b[ROWS];
#pragma unroll
        for (int p = 0; p < ROWS; ++p) b[p] = static_cast<uint64_t>(__double_as_longlong(__builtin_nontemporal_load(vp + 64 * p)));
        unsigned long long w = 0;
#pragma unroll
        for (int p = 0; p < ROWS; ++p) {
            int lo = 0, len = nd;                // first key >= b[p]
            while (len > 0) {
                const int half = len >> 1;
                const bool right = vd_keys[lo + half] < b[p];
                lo = right ? lo + half + 1 : lo;
                len = right ? len - half - 1 : half;
            }
            if (lo >= nd || vd_keys[lo] != b[p]) { missed = true; lo = 0; }
            w |= static_cast<unsigned long long>(lo) << (16 * p);
        }
        codes[i] = w;
    }
    if (missed) st->miss = 1;
}

// the same for a matrix in the row form (one row to the lane, slot k of lane l at off + 64 k + l): one 16-bit code per slot,
// in the slot's own place (coarse levels of a brick hierarchy: their Galerkin sums repeat like the element matrices do)
__global__ void __launch_bounds__(kBlock) k_vd_encode16(const double *__restrict__ vals, int64_t n_slots, const double *__restrict__ dict,
                                                         VdState *st, uint16_t *__restrict__ codes)
{
    extern __shared__ uint64_t vd_keys[];
    if (st->fail) return;
    const int nd = min(st->count, kVdMax);
    for (int i = threadIdx.x; i < nd; i += kBlock) vd_keys[i] = static_cast<uint64_t>(__double_as_longlong(dict[i]));
    __syncthreads();
    bool missed = false;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_slots; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const uint64_t b = static_cast<uint64_t>(__double_as_longlong(__builtin_nontemporal_load(vals + i)));
        int lo = 0, len = nd;
        while (len > 0) {
            const int half = len >> 1;
            const bool right = vd_keys[lo + half] < b;
            lo = right ? lo + half + 1 : lo;
            len = right ? len - half - 1 : half;
        }
        if (lo >= nd || vd_keys[lo] != b) { missed = true; lo = 0; }
        codes[i] = static_cast<uint16_t>(lo);
    }
    if (missed) st->miss = 1;
}

// ---- the SpMVs of pfem_kernels.hpp over the codes: same slices, same lanes, same order of the fma chain, same partials ----
template <bool WITH_DOT, bool DICT>
__global__ void __launch_bounds__(kBlock) k_spmvr_vd(SellRDev G, const unsigned long long *__restrict__ codes, const double *__restrict__ dict, int nd,
                                                      int64_t n_rows, const double *__restrict__ x, double *__restrict__ y, int64_t n_dot,
                                                      double *partial, const CgCtl *ctl, SliceSel sel)
{
    const unsigned vblock = blockIdx.x;
    extern __shared__ double vd[];
    __shared__ double sm[4];
    __shared__ int done_waves;
    __shared__ uint32_t gap_tbl[DICT ? kGapTable : 1];
    // (the CG's verdict: requested here, looked at behind the barrier below -- the wave's first loads go out without waiting for it;
    // an iteration enqueued after the verdict still leaves before its main loop)
    const int finished = WITH_DOT ? __builtin_nontemporal_load(&ctl->flag) : 0;
    if (WITH_DOT && threadIdx.x == 0) done_waves = 0;         // (before the barrier below)
    // A wave's life is a chain of memory latencies (rocprofv3 --pmc: 58 % of the wave cycles waiting; with an XCD-contiguous block
    // order the fabric traffic falls from 0.59 GB to its ideal 0.36 GB and the time does not move -- LAB_NOTES): slice header ->
    // first entry -> trips of four entries -> remainder.  So the header, the first entry's operands and the first gap words are requested BEFORE the dictionary is copied to LDS,
    // the first entry's product is formed under the first trip's loads, and the remainder (<= 3 entries) is ONE trip whose loads
    // go out together instead of a loop of dependent ones.  Same operands into the same fma chain: same bits.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gs = pick_slice(sel, (static_cast<int64_t>(vblock) << 2) + wave, G.n_gslices);
    const bool live = gs < G.n_gslices;
    int width = 0, nw = 0, c = 0;
    uint32_t w0 = 0, w1 = 0;
    const unsigned long long *__restrict__ qp = codes + lane;
    const uint32_t *__restrict__ wp = G.dwords + lane;
    double x0[kRelRows] = {0.0, 0.0, 0.0, 0.0}, xself[kRelRows] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long q0 = 0;
    if (live) {
        const int64_t off = G.gslice_off[gs];
        width = static_cast<int>((G.gslice_off[gs + 1] - off) >> 6);
        nw = width / 2;
        qp = codes + off + lane;
        wp = G.dwords + G.gslice_doff[gs] + lane;
        c = __builtin_nontemporal_load(G.col0 + (gs << 6) + lane);
        if (nw > 0) w0 = __builtin_nontemporal_load(wp);
        if (nw > 1) w1 = __builtin_nontemporal_load(wp + 64);
        if (width > 0) { load_x4(x, c, x0); q0 = __builtin_nontemporal_load(qp); }
        // (the rows' own x for the (p,Ap) partial: requested with the first loads, not as one more round trip at the end of the
        // wave's life -- the WITH_DOT variant cost 11 us over the plain one, profiles/r06)
        if (WITH_DOT) load_x4(x, static_cast<int>(((gs << 6) + lane) * kRelRows), xself);
    }
    for (int i = threadIdx.x; i < nd; i += kBlock) vd[i] = dict[i];
    if (DICT) gap_tbl[DICT ? threadIdx.x : 0] = G.gap_table[threadIdx.x];
    __syncthreads();
    if (WITH_DOT && finished != 0) return;
    const auto gap_of = [&](uint32_t code) -> int {
        return (DICT && (code & 0x8000u)) ? static_cast<int>(gap_tbl[DICT ? (code & (kGapTable - 1)) : 0]) : static_cast<int>(code);
    };
    double dot = 0.0;
    if (live) {
        double acc[kRelRows] = {0.0, 0.0, 0.0, 0.0};
        bool first_pending = width > 0;
        int j = 0;
        while (2 * (j + 2) < width) {
            const int c0 = c + gap_of(w0 & 0xffffu), c1 = c0 + gap_of(w0 >> 16);
            const int c2 = c1 + gap_of(w1 & 0xffffu), c3 = c2 + gap_of(w1 >> 16);
            unsigned long long q[4];
            double xv[4][kRelRows];
            load_x4(x, c0, xv[0]); load_x4(x, c1, xv[1]); load_x4(x, c2, xv[2]); load_x4(x, c3, xv[3]);
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] = __builtin_nontemporal_load(qp + 64 * (2 * j + 1 + t));
            c = c3;
            j += 2;
            w0 = j < nw ? __builtin_nontemporal_load(wp + 64 * j) : 0u;              // (the next trip's gaps, or the remainder's)
            w1 = j + 1 < nw ? __builtin_nontemporal_load(wp + 64 * (j + 1)) : 0u;
            __builtin_amdgcn_sched_barrier(0);
            if (first_pending) {
#pragma unroll
                for (int p = 0; p < kRelRows; ++p) acc[p] = vd[(q0 >> (16 * p)) & 0xffffu] * x0[p];
                first_pending = false;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < kRelRows; ++p) acc[p] = __builtin_fma(vd[(q[t] >> (16 * p)) & 0xffffu], xv[t][p], acc[p]);
        }
        const int rem = width - 1 - 2 * j;          // 0 .. 3 entries are left (the same for the whole wave)
        if (rem > 0) {
            const int c0 = c + gap_of(w0 & 0xffffu), c1 = c0 + gap_of(w0 >> 16), c2 = c1 + gap_of(w1 & 0xffffu);
            unsigned long long q[3] = {0, 0, 0};
            double xv[3][kRelRows];
            load_x4(x, c0, xv[0]);
            q[0] = __builtin_nontemporal_load(qp + 64 * (2 * j + 1));
            if (rem > 1) { load_x4(x, c1, xv[1]); q[1] = __builtin_nontemporal_load(qp + 64 * (2 * j + 2)); }
            if (rem > 2) { load_x4(x, c2, xv[2]); q[2] = __builtin_nontemporal_load(qp + 64 * (2 * j + 3)); }
            __builtin_amdgcn_sched_barrier(0);
            if (first_pending) {
#pragma unroll
                for (int p = 0; p < kRelRows; ++p) acc[p] = vd[(q0 >> (16 * p)) & 0xffffu] * x0[p];
                first_pending = false;
            }
#pragma unroll
            for (int p = 0; p < kRelRows; ++p) acc[p] = __builtin_fma(vd[(q[0] >> (16 * p)) & 0xffffu], xv[0][p], acc[p]);
            if (rem > 1) {
#pragma unroll
                for (int p = 0; p < kRelRows; ++p) acc[p] = __builtin_fma(vd[(q[1] >> (16 * p)) & 0xffffu], xv[1][p], acc[p]);
            }
            if (rem > 2) {
#pragma unroll
                for (int p = 0; p < kRelRows; ++p) acc[p] = __builtin_fma(vd[(q[2] >> (16 * p)) & 0xffffu], xv[2][p], acc[p]);
            }
        }
        if (first_pending) {
#pragma unroll
            for (int p = 0; p < kRelRows; ++p) acc[p] = vd[(q0 >> (16 * p)) & 0xffffu] * x0[p];
        }
        const int64_t r0 = ((gs << 6) + lane) * kRelRows;
#pragma unroll
        for (int p = 0; p < kRelRows; ++p)
            if (r0 + p < n_rows) {
                y[r0 + p] = acc[p];
                if (WITH_DOT && r0 + p < n_dot) dot = __builtin_fma(xself[p], acc[p], dot);
            }
    }
    if (WITH_DOT) block_sum_last_wave(dot, sm, &done_waves, partial + vblock);      // (no barrier at the end of a wave's life)
}

// W words = 2W consecutive entries of the node's rows (k_spmvg's trip over the codes)
template <int W, bool DICT>
__device__ __forceinline__ void spmvg_vd_trip(const unsigned long long *__restrict__ qp, const uint32_t *__restrict__ wp, const double *__restrict__ x,
                                              const double *vd, int &j, int &c, double (&acc)[kGroupRows], const uint32_t *tbl)
{
    uint32_t w[W];
    unsigned long long q[2 * W];
    double xv[2 * W];
#pragma unroll
    for (int t = 0; t < W; ++t) w[t] = __builtin_nontemporal_load(wp + 64 * (j + t));
#pragma unroll
    for (int t = 0; t < 2 * W; ++t) q[t] = __builtin_nontemporal_load(qp + 64 * (2 * j + 1 + t));
#pragma unroll
    for (int t = 0; t < W; ++t) {
        const int c0 = c + gap16<DICT>(w[t] & 0xffffu, tbl);
        const int c1 = c0 + gap16<DICT>(w[t] >> 16, tbl);
        xv[2 * t] = x[c0];
        xv[2 * t + 1] = x[c1];
        c = c1;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2 * W; ++t)
#pragma unroll
        for (int p = 0; p < kGroupRows; ++p) acc[p] = __builtin_fma(vd[(q[t] >> (16 * p)) & 0xffffu], xv[t], acc[p]);
    j += W;
}

template <bool WITH_DOT, bool DICT>
__global__ void __launch_bounds__(kBlock) k_spmvg_vd(SellGDev G, const unsigned long long *__restrict__ codes, const double *__restrict__ dict, int nd,
                                                      int64_t n_rows, const double *__restrict__ x, double *__restrict__ y, int64_t n_dot,
                                                      double *partial, const CgCtl *ctl, SliceSel sel)
{
    extern __shared__ double vd[];
    __shared__ double sm[4];
    __shared__ int done_waves;
    __shared__ uint32_t tbl[DICT ? kGapTable : 1];
    if (WITH_DOT && ctl->flag != 0) return;
    if (WITH_DOT && threadIdx.x == 0) done_waves = 0;         // (before the barrier below)
    for (int i = threadIdx.x; i < nd; i += kBlock) vd[i] = dict[i];
    if (DICT) tbl[DICT ? threadIdx.x : 0] = G.gap_table[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gs = pick_slice(sel, (static_cast<int64_t>(blockIdx.x) << 2) + wave, G.n_gslices);
    double dot = 0.0;
    if (gs < G.n_gslices) {
        const int64_t off = G.gslice_off[gs];
        const int width = static_cast<int>((G.gslice_off[gs + 1] - off) >> 6);
        const unsigned long long *__restrict__ qp = codes + off + lane;
        const uint32_t *__restrict__ wp = G.dwords + G.gslice_doff[gs] + lane;
        int c = __builtin_nontemporal_load(G.col0 + (gs << 6) + lane);
        double acc[kGroupRows];
#pragma unroll
        for (int p = 0; p < kGroupRows; ++p) acc[p] = 0.0;
        if (width > 0) {
            const double x0 = x[c];
            const unsigned long long q = __builtin_nontemporal_load(qp);
#pragma unroll
            for (int p = 0; p < kGroupRows; ++p) acc[p] = vd[(q >> (16 * p)) & 0xffffu] * x0;
        }
        const int nw = width / 2;
        int j = 0;
        while (2 * (j + 4) < width) spmvg_vd_trip<4, DICT>(qp, wp, x, vd, j, c, acc, tbl);
        if (2 * (j + 2) < width) spmvg_vd_trip<2, DICT>(qp, wp, x, vd, j, c, acc, tbl);
        for (; j < nw; ++j) {
            const uint32_t w0 = __builtin_nontemporal_load(wp + 64 * j);
            const int c0 = c + gap16<DICT>(w0 & 0xffffu, tbl);
            const int c1 = c0 + gap16<DICT>(w0 >> 16, tbl);
            const double x0 = x[c0];
            const unsigned long long qa = __builtin_nontemporal_load(qp + 64 * (2 * j + 1));
#pragma unroll
            for (int p = 0; p < kGroupRows; ++p) acc[p] = __builtin_fma(vd[(qa >> (16 * p)) & 0xffffu], x0, acc[p]);
            if (2 * j + 2 < width) {
                const double x1 = x[c1];
                const unsigned long long qb = __builtin_nontemporal_load(qp + 64 * (2 * j + 2));
#pragma unroll
                for (int p = 0; p < kGroupRows; ++p) acc[p] = __builtin_fma(vd[(qb >> (16 * p)) & 0xffffu], x1, acc[p]);
            }
            c = c1;
        }
        const int64_t g = (gs << 6) + lane;
        if (g < G.n_groups) {
            const int64_t r0 = G.group_row0[g];
            const int sz = G.group_row0[g + 1] - static_cast<int>(r0);
#pragma unroll
            for (int p = 0; p < kGroupRows; ++p)
                if (p < sz) {
                    y[r0 + p] = acc[p];
                    if (WITH_DOT && r0 + p < n_dot) dot = __builtin_fma(x[r0 + p], acc[p], dot);
                }
        }
    }
    if (WITH_DOT) block_sum_last_wave(dot, sm, &done_waves, partial + blockIdx.x);
}

}  // namespace pfem
