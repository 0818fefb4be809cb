// plans_s16_f64.hip - kernel instantiations for int16 PCM samples with FLOAT64 arithmetic (namespace adsp::f64): the
// exact-FFT engines (ADSP_FORMAT_S16_F64).  Same plans, same index maps, twice the registers and LDS per point.
#define ADSP_WITH_F64 1
#include "plan_table.hpp"

namespace adsp {
namespace f64 {
namespace {
const PlanInfo kPlans[] = {ADSP_PLAN_LIST(true, false, false)};
}  // namespace
}  // namespace f64
}  // namespace adsp

const adsp::PlanInfo* adsp::plans_s16_f64(int* count) {
    *count = sizeof(adsp::f64::kPlans) / sizeof(adsp::f64::kPlans[0]);
    return adsp::f64::kPlans;
}
