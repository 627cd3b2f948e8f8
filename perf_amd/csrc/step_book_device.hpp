// The bookkeeping of a sync-free training step (perf_step_bookkeeping in the header) as a device function of ONE thread, so that it
// can ride in another launch: perf_field_bwd_book runs it in the predicated repair launch of the grid backward.
#pragma once
#include "common.hpp"

namespace perf {

// clear_flag: consume the overflow flag here (the stand-alone launch); false when the launch this runs in still reads the flag in its
// other workgroups (the repair launch: perf_adam_step_dev's clear_flag consumes it afterwards)
__device__ __forceinline__ void step_bookkeeping_thread(const perf_step_book& b, bool clear_flag) {
    // every input is READ before anything is written (the pointers may alias as far as the compiler knows: interleaved, each
    // load waits for the store before it and the one thread walks a dozen round trips one after the other)
    const bool sched = b.schedule && b.iter_dev && b.n_schedule > 0;
    const int it = sched ? b.iter_dev[0] : 0;
    const int64_t marched = b.n_marched_dev ? b.n_marched_dev[0] : 0;
    const int64_t gate = b.gate_dev ? b.gate_dev[0] : 1;
    const int32_t own_flag = b.overflow_flag ? b.overflow_flag[0] : 0;
    const float remote_overflow = b.remote_flags ? b.remote_flags[0] : 0.f, remote_truncated = b.remote_flags ? b.remote_flags[1] : 0.f;
    const int32_t step_now = b.step_dev ? b.step_dev[0] : 0;
    const int64_t kept = (b.counters && b.n_kept_dev) ? b.n_kept_dev[0] : 0;
    int64_t cnt[6] = {0, 0, 0, 0, 0, 0};
    if (b.counters) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cnt[k] = b.counters[k];
    }
    // device-side schedule: row i = {learning rate of iteration i, distortion-loss ramp of iteration i}.  This sits
    // between the backward and Adam of iteration `it`: Adam reads lr(it) next, the loss head of iteration it + 1 reads
    // ratio(it + 1) -- a graph replay then needs no host-side scalar update at all
    const int cur = it < b.n_schedule ? it : b.n_schedule - 1, nxt = it + 1 < b.n_schedule ? it + 1 : b.n_schedule - 1;
    const float lr = sched ? b.schedule[2 * cur] : 0.f, ratio = sched ? b.schedule[2 * nxt + 1] : 0.f;

    const bool has_samples = gate > 0;
    const bool overflow = own_flag != 0 || remote_overflow > 0.f;
    // remote_flags: {overflow, truncated} summed over the ranks of a data-parallel job (this rank's own included): every rank
    // takes or skips the step alike
    const bool truncated = (b.capacity > 0 && marched > b.capacity) || remote_truncated > 0.f;
    // overflow_redone: the caller repaired a flagged gradient in place (perf_hashgrid_bwd's redo launch): the event is counted,
    // the step is taken
    const bool take = has_samples && (!overflow || b.overflow_redone) && !truncated;
    if (sched) {
        if (b.lr_out) b.lr_out[0] = lr;
        if (b.ratio_out) b.ratio_out[0] = ratio;
        b.iter_dev[0] = it + 1;
    }
    if (b.step_dev && take) b.step_dev[0] = step_now + 1;
    if (b.eff_gate_out) b.eff_gate_out[0] = take ? 1 : 0;
    if (clear_flag && own_flag != 0) b.overflow_flag[0] = 0;      // consumed: counted below
    if (b.counters) {
        b.counters[0] = cnt[0] + marched;
        if (b.n_kept_dev) b.counters[1] = cnt[1] + kept;
        b.counters[2] = cnt[2] + 1;
        if (marched > cnt[3]) b.counters[3] = marched;
        if (has_samples && overflow) b.counters[4] = cnt[4] + 1;
        if (has_samples && truncated) b.counters[5] = cnt[5] + 1;
    }
}

}  // namespace perf
