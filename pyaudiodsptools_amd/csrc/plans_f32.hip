// plans_f32.hip - kernel instantiations for float32 samples
// (this table also carries the dword-access kernels for chunk sizes that are not multiples of 4: the third argument of ADSP_PLAN_LIST)
#include "plan_table.hpp"

namespace {
using namespace adsp;
const PlanInfo kPlans[] = {ADSP_PLAN_LIST(false, false, true)};
}  // namespace

const adsp::PlanInfo* adsp::plans_f32(int* count) {
    *count = sizeof(kPlans) / sizeof(kPlans[0]);
    return kPlans;
}
