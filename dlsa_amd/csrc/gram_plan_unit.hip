// One translation unit of the plan-driven fp64 Gram kernel: the widths PLAN_LO .. PLAN_HI full tiles (Makefile PLAN_UNITS;
// the split exists only so that the generated plans compile in parallel).  PLAN_INC is the generated plan file of the unit.
#include "gram_plan.h"
#include <type_traits>

namespace dlsa {
#include "gram_plan_common.inc"
#include PLAN_INC
}  // namespace dlsa

#include "gram_plan_kernel.inc"

namespace dlsa {

template <int NT>
static int plan_launch_nt(const PlanArgs& a, int nt, int g, int nslab, hipStream_t stream) {
    if (nt == NT) return plan_launch_g<NT>(a, g, nslab, stream);
    if constexpr (NT < PLAN_HI) return plan_launch_nt<NT + 1>(a, nt, g, nslab, stream);
    set_error("gram: no plan for %d tiles in this unit", nt);
    return DLSA_ERR_INVALID;
}

#define PLAN_CAT2(a, b) a##b
#define PLAN_CAT(a, b) PLAN_CAT2(a, b)
int PLAN_CAT(gram_plan_launch_, PLAN_LO)(const PlanArgs& a, int nt, int g, int nslab, hipStream_t stream) {
    return plan_launch_nt<PLAN_LO>(a, nt, g, nslab, stream);
}

}  // namespace dlsa
