// pfem_vdhash.hpp -- the small pieces of the value dictionary (pfem_valdict.hpp) that the assembly kernels need too: the
// verdict block, the hash, the look-up table of a dictionary.  Included ahead of pfem_kernels.hpp.
#pragma once

namespace pfem {

constexpr int kVdMax = 4096;                    // distinct values a dictionary may hold: 32 KB of LDS in the SpMV
constexpr int kVdTable = 16384;                 // slots of the collection table
constexpr uint64_t kVdEmpty = ~0ull;            // (the bit pattern of a NaN no assembly produces; met as a VALUE it fails the form)
struct VdState { int count, fail, miss, pad; };

__device__ __forceinline__ uint32_t vd_hash(uint64_t b)
{
    b ^= b >> 33;
    b *= 0xff51afd7ed558ccdull;
    b ^= b >> 33;
    return static_cast<uint32_t>(b);
}

constexpr int kVdHashSlots = 8192;               // >= 2 x kVdMax
struct VdHashEntry { unsigned long long key, code; };
// code of the value with bit pattern b, or -1
__device__ __forceinline__ int vd_hash_find(const VdHashEntry *__restrict__ table, unsigned long long b)
{
    uint32_t h = vd_hash(b) & (kVdHashSlots - 1);
    for (int tries = 0; tries < 64; ++tries, h = (h + 1) & (kVdHashSlots - 1)) {
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(table + h);
        if (e.x == b) return static_cast<int>(e.y);
        if (e.x == kVdEmpty) return -1;
    }
    return -1;
}

}  // namespace pfem
