// Per-call options of the IRLS driver (include/dlsa_hip.h: dlsa_irls_options, dlsa_irls_set_options).  The driver's switches used to
// be process environment only; now a caller sets them per thread through the C ABI, the environment stays as the
// override of A/B runs for fields the caller left on automatic.
#pragma once
#include "dlsa_hip.h"

namespace dlsa {

// The value of a driver switch as TEXT (what the call sites parse): the calling thread's option field when it is set (>= 0), else
// the environment variable of that name, else nullptr (automatic).  Valid until the thread's next knob() call for the same name.
const char* knob(const char* env_name);

// worker threads of one call (partition chains) inherit the caller's options
dlsa_irls_options irls_options_snapshot();
void irls_options_adopt(const dlsa_irls_options& o);

}  // namespace dlsa
