// Fixed-point units of the hash-grid gradient (shared by the LDS owners, hashgrid_bwd.hip, and the data-parallel unit kernels,
// hashgrid_aux.hip): per-level power-of-two unit from max |dfeat| under a closed-loop headroom.
#pragma once
#include "common.hpp"

namespace perf {

constexpr int kHeadroomStartBias = 3;

// Scale of the fixed-point gradient fields of level l: one unit = 2^-sh.
__device__ __forceinline__ int fixed_point_shift(const float am, const int64_t n_live, const uint32_t size,
                                                 const int32_t* __restrict__ hr_state, const int l) {
    int e = 0;
    if (am > 0.f) (void)frexpf(am, &e);                         // am < 2^e
    if (e < -80) e = -80;                                        // (vanishing gradients: keep 2^sh finite)
    // Headroom of the fixed-point fields: an entry of level l sums 8 n / size_l contributions on average (1,700 at
    // the coarsest level of a 1 M-sample batch, 32 at a hashed one); 64x that average before the overflow flag is
    // raised, never less than 2^12, never more than 2^24 (which still leaves 2^-7 of the largest contribution as the
    // unit).  Derived from the LIVE sample count, so a capacity-sized launch keeps the resolution of an exact one.
    const unsigned long long fan = (8ull * (unsigned long long)n_live + size - 1ull) / size;     // ceil(8 n / size)
    int h = (fan <= 1ull ? 0 : 64 - __clzll((long long)(fan - 1ull))) + 6;                      // ceil(log2(fan)) + 6
    if (hr_state) {
        // Closed loop (caller-owned state, see perf_hashgrid_bwd): the static guess is corrected by what the fields of
        // the PREVIOUS calls really reached -- entries near a panorama's common ray origin sum 30x the average number
        // of contributions, hashed levels far fewer than the guess allows.  Starts 3 bits on the safe side.
        h += hr_state[l] + kHeadroomStartBias;
        h = h < 4 ? 4 : (h > 28 ? 28 : h);
    } else {
        h = h < 12 ? 12 : (h > 24 ? 24 : h);
    }
    return 31 - h - e;
}

// Headroom feedback: keep the largest field of a level between 2^21 and 2^25 units.  Above: add the excess bits at once
// (+1); below: give one bit back per call.  fm = the largest |field| the level's FINAL sums reached in the previous call
// (all replicas -- and, under data parallelism, all ranks -- added up), so that every partition of a batch follows the
// same sequence of units.  Deterministic: the state is a function of the call history only.
// The top of the band sits 16x below the level at which the overflow flag is raised (2^29) and the step gate drops the
// step: a soak of 25 episodes with the band at [2^23, 2^27] (4x) lost 8 of 112,500 steps to flags that were not
// overflows -- the colour table's largest sum quadrupling from one batch to the next (tools/soak_episodes.py).
constexpr int kHeadroomTopBit = 25, kHeadroomLowBit = 21;
constexpr int kLaggedMinHeadroom = 12;     // (dp_units_kernel, lagged units)
constexpr int kLaggedMaxFinerBits = 2;
__device__ __forceinline__ int headroom_feedback(int adj, int fm) {
    if (fm >= (1 << kHeadroomTopBit)) adj += (32 - __clz(fm)) - kHeadroomTopBit + 1;
    else if (fm < (1 << kHeadroomLowBit) && adj > -24) adj -= 1;
    return adj;
}

}  // namespace perf
