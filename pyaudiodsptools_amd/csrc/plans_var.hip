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
    // half-buffer exchanges (32 KiB of LDS per M = 8192 transform) + register budget for 3 waves per SIMD: 3 workgroups per CU
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 3>, 1, 2, false, false>(),  // 4
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 3>, 1, 4, false, false>(),  // 5
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 2>, 1, 2, false, false>(),  // 6: half exchange alone (cost of the extra barriers)
    // 64 points per thread, 4 waves per transform, half exchange (64 KiB): two workgroups per CU for M = 16384
    make_plan<Plan<16384, 64, 3, 16, 32, 32, 1, false, true, 2>, 1, 4, false, false>(),  // 7
    make_plan<Plan<8192, 64, 3, 8, 32, 32, 1, false, true, 2>, 1, 2, false, false>(),    // 8: M = 8192, 2 waves per transform, 4 workgroups per CU
    make_plan<Plan<8192, 64, 3, 8, 32, 32, 1, false, true, 2>, 1, 4, false, false>(),    // 9
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true, true, 5>, 1, 2, false, false>(),    // 10: headline plan, 16 KiB of LDS, 5 workgroups per CU
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true, true, 6>, 1, 2, false, false>(),    // 11: 6 workgroups per CU
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 4>, 1, 2, false, false>(),   // 12: M = 8192 at 4 workgroups per CU (128 VGPRs)
    make_plan<Plan<16384, 32, 3, 32, 32, 16, 1, false, true, 3>, 1, 4, false, false>(),  // 13: M = 16384, 32 points per thread, 168 VGPRs (still one workgroup per CU: fewer registers alone)
};
}  // namespace

const adsp::PlanInfo* adsp::variants_f32(int* count) {
    *count = sizeof(kVariants) / sizeof(kVariants[0]);
    return kVariants;
}
