// Signed c-bit window digits of a scalar (shared by the digit-sort kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace blz {

template <int SW>
struct ScalarWords {
    uint32_t s[SW];
    __device__ __forceinline__ void load(const uint32_t* scalars, uint32_t p) {
        if constexpr (SW == 8) {
            const uint4* q = reinterpret_cast<const uint4*>(scalars) + 2 * (size_t)p;
            uint4 a = q[0], b = q[1];
            s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
        } else if constexpr (SW == 2) {
            // the 64-bit chunks of a precompute handle on the checked-table plan (arena_tables.hip resolve_arena_task)
            const uint2 a = reinterpret_cast<const uint2*>(scalars)[p];
            s[0] = a.x; s[1] = a.y;
        } else {
            s[0] = scalars[p];
        }
    }
    // pops the next c-bit window as a signed digit in (-2^(c-1), 2^(c-1)]
    __device__ __forceinline__ int next(int c, uint32_t mask, uint32_t half, uint32_t& carry) {
        uint32_t v = (s[0] & mask) + carry;
#pragma unroll
        for (int i = 0; i + 1 < SW; ++i) s[i] = __builtin_amdgcn_alignbit(s[i + 1], s[i], c);
        s[SW - 1] >>= c;
        if (v > half) { carry = 1; return (int)v - (int)(half << 1); }
        carry = 0;
        return (int)v;
    }
};


// Launch dispatch on the scalar width: SW = 32-bit words per scalar - 8 (256-bit scalars), 2 (the 64-bit chunks of a
// precompute handle on the checked-table plan) or 1 (its 32-bit chunks on the exact path)
#define BLZ_SW_DISPATCH(sbits, ...)                                   \
    do {                                                              \
        if ((sbits) == 256) { constexpr int SW = 8; __VA_ARGS__; }    \
        else if ((sbits) == 64) { constexpr int SW = 2; __VA_ARGS__; } \
        else { constexpr int SW = 1; __VA_ARGS__; }                   \
    } while (0)

}  // namespace blz
