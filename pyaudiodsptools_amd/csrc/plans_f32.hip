// plans_f32.hip - kernel instantiations for float32 samples
#define ADSP_PLANS_WITH_UNALIGNED 1  // this table also carries the dword-access kernels for chunk sizes that are not multiples of 4
#include "plan_table.hpp"

namespace {
using namespace adsp;
const PlanInfo kPlans[] = {ADSP_PLAN_LIST(false, false)};
}  // namespace

const adsp::PlanInfo* adsp::plans_f32(int* count) {
    *count = sizeof(kPlans) / sizeof(kPlans[0]);
    return kPlans;
}
