// Shared between irls_pass.hip (the fused kernel's batched form) and irls_batch.hip (the lock-step driver).
#pragma once
#include <stdint.h>

namespace dlsa {

// a workgroup's rows in the batched form of the fused Newton pass: rows of ONE partition
struct FusedSlab {
    int64_t xoff, yoff;       // element offsets of the slab's first row / label from X / y
    int nrows, part;
};

}  // namespace dlsa
