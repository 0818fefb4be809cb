// plans_var.hip - alternative float32 plans kept for A/B measurements, selected with ADSP_PLAN_VARIANT=<index> (tuning
// only: an engine whose (M, F/N) matches the variant runs it instead of the plan of plan_table.hpp).  Results of the
// round-2 A/Bs: profiles/r2_shapes_session1.txt (0-3), profiles/r2_shapes_session4.txt (the rest).
#include "plan_table.hpp"

namespace {
using namespace adsp;
const PlanInfo kVariants[] = {
    make_plan<Plan<4096, 32, 3, 16, 16, 16, 1>, 1, 2, false, false>(),       // 0: in-register pairing, 2 waves/transform, ~190 VGPRs (-8 %)
    make_plan<Plan<16384, 32, 4, 32, 2, 16, 16>, 1, 4, false, false>(),      // 1: four passes (-8 % against 3)
    make_plan<Plan<16384, 16, 4, 4, 16, 16, 16, true>, 1, 4, false, false>(),  // 2: XL, 1024 threads, 16 points per thread (-8 % against 3)
    make_plan<Plan<16384, 16, 4, 16, 4, 16, 16, true>, 1, 4, false, false>(),  // 3: same, radix 4 second (= 4)
    make_plan<Plan<16384, 32, 3, 32, 32, 16, 1>, 1, 4, false, false>(),      // 4: the round-1 plan: 32 points per thread, ONE workgroup per CU (-12 % against the default)
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1>, 1, 2, false, false>(),       // 5: the round-1 plan: full exchange, 186 VGPRs, TWO workgroups per CU (-8 %)
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1>, 1, 4, false, false>(),       // 6: same, F = 4N (-9 %)
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 2>, 1, 2, false, false>(),  // 7: half exchange alone, still two workgroups per CU: the price of its barriers (-6 % against 5)
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 4>, 1, 2, false, false>(),  // 8: four workgroups per CU at 128 VGPRs: spills (= 5)
    make_plan<Plan<8192, 64, 3, 8, 32, 32, 1, false, true, 2>, 1, 2, false, false>(),   // 9: 64 points per thread, 2 waves per transform (-3 % against 5)
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true, true, 5>, 1, 2, false, false>(),   // 10: headline plan, 16 KiB of LDS, FIVE workgroups per CU at 96 VGPRs (-9 %)
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true, true, 6>, 1, 2, false, false>(),   // 11: six at 80 VGPRs (-20 %)
    make_plan<Plan<16384, 64, 3, 32, 32, 16, 1, false, true, 2>, 1, 4, false, false>(),  // 12: 64 points per thread, paired passes of radix 16 (4 butterflies = 2 pairs per thread)
    make_plan<Plan<16384, 64, 3, 16, 32, 32, 1, false, true, 2>, 1, 4, false, false>(),  // 13: paired passes of radix 32 (120 bytes of scratch per lane)
    make_plan<Plan<8192, 32, 3, 16, 32, 16, 1, false, true, 3>, 1, 2, false, false>(),   // 14: M = 8192 with paired passes of radix 16
    make_plan<Plan<512, 16, 3, 16, 4, 8, 1>, 4, 2, false, false>(),                      // 15: M = 512 (config 3's stream transform), 4 channels per workgroup
    make_plan<Plan<512, 16, 3, 16, 4, 8, 1>, 8, 2, false, false>(),                      // 16: 8 channels per workgroup
    make_plan<Plan<512, 16, 3, 16, 4, 8, 1>, 1, 2, false, false>(),                      // 17: 1 channel (half a wave) per workgroup
    make_plan<Plan<512, 32, 2, 32, 16, 1, 1>, 4, 2, false, false>(),                     // 18: TWO passes (32 points per thread, 16 threads per transform), 4 channels per wave
    make_plan<Plan<512, 32, 2, 32, 16, 1, 1>, 8, 2, false, false>(),                     // 19: same, 8 channels per workgroup
};
}  // namespace

const adsp::PlanInfo* adsp::variants_f32(int* count) {
    *count = sizeof(kVariants) / sizeof(kVariants[0]);
    return kVariants;
}
