// plans_s16.hip - kernel instantiations for int16 PCM samples (fused WAV front end)
#include "plan_table.hpp"

namespace {
using namespace adsp;
const PlanInfo kPlans[] = {ADSP_PLAN_LIST(true, false, false)};
}  // namespace

const adsp::PlanInfo* adsp::plans_s16(int* count) {
    *count = sizeof(kPlans) / sizeof(kPlans[0]);
    return kPlans;
}
