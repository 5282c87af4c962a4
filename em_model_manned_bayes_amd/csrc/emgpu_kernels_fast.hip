// emgpu_kernels_fast.hip -- placeholder, replaced by the specialised uncor kernel.
#include "emgpu_launch.h"
namespace emgpu {
bool fast_uncor_eligible(const EmgpuPlan &, const EmgpuRun &) { return false; }
hipError_t launch_uncor_fast(const EmgpuPlan &, const EmgpuRun &, hipStream_t, const char **name) { *name = "none"; return hipErrorNotSupported; }
}
