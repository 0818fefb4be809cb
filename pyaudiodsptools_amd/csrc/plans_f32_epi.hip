// plans_f32_epi.hip - float32 kernel instantiations with the fused output effect (adsp_set_epilogue)
#include "plan_table.hpp"

namespace {
using namespace adsp;
const PlanInfo kPlans[] = {ADSP_PLAN_LIST(false, true, false)};
}  // namespace

const adsp::PlanInfo* adsp::plans_f32_epi(int* count) {
    *count = sizeof(kPlans) / sizeof(kPlans[0]);
    return kPlans;
}
