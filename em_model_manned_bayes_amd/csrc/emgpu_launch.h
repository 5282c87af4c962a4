// emgpu_launch.h -- host-callable launchers implemented in the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "emgpu_plan.h"

namespace emgpu {
hipError_t launch_dbn_generic(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name, const EmgpuPresets *presets = nullptr);
hipError_t launch_bn(const EmgpuPlan &P, const EmgpuBnRun &A, hipStream_t s, const char **name);
// Returns false when the (plan, run) pair is outside what the specialised kernel covers.
bool fast_uncor_eligible(const EmgpuPlan &P, const EmgpuRun &A);
hipError_t launch_uncor_fast(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name);
// Mixed-model batch in one launch: nb <= EMGPU_MAX_MIXED blocks whose plans are fast-eligible and share the instance
// uncor_fast_shape().  A model's plan lives in device memory (plan_f_bytes() bytes filled by plan_f_fill on the host, then uploaded).
int uncor_fast_shape(const EmgpuPlan &P);
size_t plan_f_bytes();
void plan_f_fill(const EmgpuPlan &P, void *host_buf);
hipError_t launch_uncor_fast_mixed(const EmgpuRun &A, int nb, const void *const *d_planf, const uint64_t *first, const int64_t *n, const int64_t *col,
                                   int shape, hipStream_t s, const char **name);
// The per-timestep DBN with dense output (dependent-branch models, EMGPU_TRANSITION_PER_STEP).
bool step_eligible(const EmgpuPlan &P, const EmgpuRun &A);
hipError_t launch_dbn_step(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name);
bool step2_eligible(const EmgpuPlan &P, const EmgpuRun &A);
// which dynamic variables are parents of which in the transition network, as the per-timestep kernel's instances see it:
// bit 4k+q of cur_mask: the time-t node of dynamic variable q is a parent of (t+1) node k; of new_mask: its (t+1) node is (q < k)
void step_parent_masks(const EmgpuPlan &P, uint32_t *cur_mask, uint32_t *new_mask);
hipError_t launch_dbn_step2(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name);
hipError_t launch_terminal_propagate(const EmgpuPlan &P, const EmgpuTermRun &A, hipStream_t s, const char **name);
int terminal_debug_counters(unsigned long long *out, int n);   // -DEMGPU_TERM_COUNTERS builds: the loop's path counters (0: not such a build)
// createEncounter.m:88-89 through the stand-in of EMGPU_FLAG_LOCAL_SMOOTH: v_ft_s (5 s) and z_ft (15 s) of n2 joined tracks, in place
hipError_t launch_terminal_smooth(float *traj, const int32_t *rows, int64_t n2, int32_t cap, hipStream_t s);
hipError_t launch_terminal_geo(const EmgpuTGeoRun &A, hipStream_t s);
hipError_t launch_terminal_filter(const EmgpuTFilterRun &A, hipStream_t s, const char **name);
hipError_t launch_uncor_track(const EmgpuUTrackRun &A, hipStream_t s, const char **name, int force_literal = 0);
// rejected lanes of a round -> the next round's index lists, in lane order; count: compact_scratch_words(n) words of device
// scratch, count[0] receives the number of rejected lanes
size_t compact_scratch_words(int64_t n);
hipError_t launch_compact_rejected(int64_t n, uint64_t first_index, const uint8_t *accepted, const uint64_t *gidx_in, const int64_t *slot_in,
                                   uint64_t *gidx_out, int64_t *slot_out, uint32_t *count, hipStream_t s);
// event lists [n][cap] -> one packed run of rows (emgpu_kernels_pack.hip): scratch = pack_scratch_words(n) words, of which the first two
// receive the total row count (u64) and word 2 + b the first packed row of workgroup b's 256 lists; packed: room for n x cap rows
size_t pack_scratch_words(int64_t n);
hipError_t launch_pack_events(int64_t n, uint32_t cap, const uint32_t *cnt, const uint64_t *ev, uint32_t *scratch, uint64_t *packed, hipStream_t s);
hipError_t launch_sample2track(const EmgpuTrackRun &A, bool dense, hipStream_t s, const char **name);
} // namespace emgpu
