// emgpu_launch.h -- host-callable launchers implemented in the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "emgpu_plan.h"

namespace emgpu {
hipError_t launch_dbn_generic(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name);
hipError_t launch_bn(const EmgpuPlan &P, const EmgpuBnRun &A, hipStream_t s, const char **name);
// Returns false when the (plan, run) pair is outside what the specialised kernel covers.
bool fast_uncor_eligible(const EmgpuPlan &P, const EmgpuRun &A);
hipError_t launch_uncor_fast(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name);
// The per-timestep DBN with dense output (dependent-branch models, EMGPU_TRANSITION_PER_STEP).
bool step_eligible(const EmgpuPlan &P, const EmgpuRun &A);
hipError_t launch_dbn_step(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name);
bool step2_eligible(const EmgpuPlan &P, const EmgpuRun &A);
hipError_t launch_dbn_step2(const EmgpuPlan &P, const EmgpuRun &A, hipStream_t s, const char **name);
hipError_t launch_terminal_propagate(const EmgpuPlan &P, const EmgpuTermRun &A, hipStream_t s, const char **name);
hipError_t launch_sample2track(const EmgpuTrackRun &A, bool dense, hipStream_t s, const char **name);
} // namespace emgpu
