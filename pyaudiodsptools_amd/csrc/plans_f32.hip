// plans_f32.hip - kernel instantiations for float32 samples
#include "plan_table.hpp"

namespace {
using namespace adsp;
const PlanInfo kPlans[] = {ADSP_PLAN_LIST(false, false)};
}  // namespace

const adsp::PlanInfo* adsp::plans_f32(int* count) {
    *count = sizeof(kPlans) / sizeof(kPlans[0]);
    return kPlans;
}
