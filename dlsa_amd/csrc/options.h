// Per-call options of the IRLS driver (include/dlsa_hip.h: dlsa_irls_options, dlsa_irls_set_options).  The driver's switches used to
// be process environment only; now a caller sets them per thread through the C ABI, the environment stays as the
// override of A/B runs for fields the caller left on automatic.
#pragma once
#include "dlsa_hip.h"
#include <hip/hip_runtime.h>

namespace dlsa {

// The value of a driver switch as TEXT (what the call sites parse): the calling thread's option field when it is set (>= 0), else
// the environment variable of that name, else nullptr (automatic).  Valid until the thread's next knob() call for the same name.
const char* knob(const char* env_name);

// The same for the kernel switches outside the IRLS driver (dlsa_kernel_options): the calling thread's field when set, else -- ONLY in
// builds made with -DDLSA_DEBUG_KNOBS -- the environment variable of that name, else nullptr.  The shipped library reads no environment
// variable here.
const char* kernel_knob(const char* env_name);
dlsa_kernel_options kernel_options_snapshot();
void kernel_options_adopt(const dlsa_kernel_options& o);

// A launch whose workgroups synchronise among themselves (clusters of the one-launch IRLS kernel, the LARS kernels' grid barriers):
// hipLaunchCooperativeKernel -- the runtime starts it only with every workgroup resident, or refuses cleanly -- when
// dlsa_kernel_options.cooperative = 1 (opt-in: see options.cpp for what it costs on this runtime).  Returns hipSuccess when the kernel was launched cooperatively; any other code (the grid does
// not fit, the device or stream cannot) means NOTHING was launched and the caller takes its plain launch, whose barriers are
// bounded and whose give-up path reruns on one workgroup.
hipError_t launch_cooperative(const void* func, dim3 grid, dim3 block, void** args, size_t shm, hipStream_t stream);

// worker threads of one call (partition chains) inherit the caller's options
dlsa_irls_options irls_options_snapshot();
void irls_options_adopt(const dlsa_irls_options& o);

}  // namespace dlsa
