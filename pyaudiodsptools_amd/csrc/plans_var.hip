// plans_var.hip - alternative float32 plans kept for A/B measurements, selected with ADSP_PLAN_VARIANT=<index> (tuning
// only: an engine whose (M, F/N) matches the variant runs it instead of the plan of plan_table.hpp)
#include "plan_table.hpp"

namespace {
using namespace adsp;
const PlanInfo kVariants[] = {
    make_plan<Plan<4096, 32, 3, 16, 16, 16, 1>, 1, 2, false, false>(),       // 0: in-register pairing, 2 waves/transform, ~190 VGPRs
    make_plan<Plan<16384, 32, 4, 32, 2, 16, 16>, 1, 4, false, false>(),      // 1: the former four-pass plan (-6 %)
    make_plan<Plan<16384, 16, 4, 4, 16, 16, 16, true>, 1, 4, false, false>(),  // 2: XL, 1024 threads, 16 points per thread
    make_plan<Plan<16384, 16, 4, 16, 4, 16, 16, true>, 1, 4, false, false>(),  // 3: same, radix 4 second
};
}  // namespace

const adsp::PlanInfo* adsp::variants_f32(int* count) {
    *count = sizeof(kVariants) / sizeof(kVariants[0]);
    return kVariants;
}
