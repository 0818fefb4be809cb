// plans_f32.hip - kernel instantiations for float32 samples
#include "plan_table.hpp"

namespace {
using namespace adsp;
const PlanInfo kPlans[] = {ADSP_PLAN_LIST(false, false)};
// alternative kept for A/B measurements, selected with ADSP_PLAN_VARIANT=0 (tuning only)
const PlanInfo kVariants[] = {
    make_plan<Plan<4096, 32, 3, 16, 16, 16, 1>, 1, 2, false, false>(),  // 0: in-register pairing, 2 waves/transform, ~190 VGPRs
    make_plan<Plan<16384, 32, 4, 32, 2, 16, 16>, 1, 4, false, false>(),  // 1: the former four-pass plan (-6 %)
};
}  // namespace

const adsp::PlanInfo* adsp::plans_f32(int* count) {
    *count = sizeof(kPlans) / sizeof(kPlans[0]);
    return kPlans;
}
const adsp::PlanInfo* adsp::variants_f32(int* count) {
    *count = sizeof(kVariants) / sizeof(kVariants[0]);
    return kVariants;
}
