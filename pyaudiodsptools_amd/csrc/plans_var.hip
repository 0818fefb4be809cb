// plans_var.hip - alternative float32 plans kept for A/B measurements, selected with ADSP_PLAN_VARIANT=<index> (tuning
// only: an engine whose (M, F/N) matches the variant runs it instead of the plan of plan_table.hpp).  Every one is checked
// against the float64 direct convolution by tools/check_variant.py; the figures are A/Bs on one box against the default
// plan of the time (profiles/r2_shapes_session{1,4,5,7,10,11}.txt - those files number the variants as they were then).
#include "plan_table.hpp"

namespace {
using namespace adsp;
const PlanInfo kVariants[] = {
    make_plan<Plan<4096, 32, 3, 16, 16, 16, 1>, 1, 8, false, false>(),         // 0: headline size, in-register pairing, 2 waves/transform, ~190 VGPRs (-8 %)
    make_plan<Plan<16384, 32, 4, 32, 2, 16, 16>, 1, 16, false, false>(),        // 1: M = 16384 in four passes (-8 % against 4)
    make_plan<Plan<16384, 16, 4, 4, 16, 16, 16, true>, 1, 16, false, false>(),  // 2: XL, 1024 threads, 16 points per thread (-8 % against 4)
    make_plan<Plan<16384, 16, 4, 16, 4, 16, 16, true>, 1, 16, false, false>(),  // 3: same, radix 4 second (= 4)
    make_plan<Plan<16384, 32, 3, 32, 32, 16, 1>, 1, 16, false, false>(),        // 4: the round-1 plan: 32 points per thread, ONE workgroup per CU (-12 % against the default)
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1>, 1, 8, false, false>(),         // 5: the round-1 plan: full exchange, 186 VGPRs, TWO workgroups per CU (-8 %)
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1>, 1, 16, false, false>(),         // 6: same, F = 4N (-9 %)
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 2>, 1, 8, false, false>(),   // 7: half exchange alone, still two workgroups per CU: the price of its barriers (-6 % against 5)
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true, true, 5>, 1, 8, false, false>(),    // 8: headline plan, 16 KiB of LDS, FIVE workgroups per CU at 96 VGPRs (-9 %; six at 80 VGPRs: -20 %)
    make_plan<Plan<16384, 64, 3, 32, 32, 16, 1, false, true, 2>, 1, 16, false, false>(),  // 9: 64 points per thread with paired passes of radix 16 (2 pairs of butterflies per thread): 76 instead of 120 bytes of scratch, -4 %
    make_plan<Plan<512, 32, 2, 32, 16, 1, 1>, 4, 8, false, false>(),                     // 10: config 3's stream transform in TWO passes (32 points per thread, 16 threads per transform): stream 10.4 us per step instead of 7.8
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 3>, 1, 8, false, false>(),   // 11: the round-2 default for M = 8192: three workgroups per CU (round 3: FOUR at 128 VGPRs + 12 B of scratch, once the spectrum stage had lost its else branches: +5 %)
    make_plan<Plan<8192, 32, 3, 32, 16, 16, 1, false, true, 3>, 1, 16, false, false>(),   // 12: same, F = 4N (four: +3 %)
    make_plan<Plan<16384, 32, 3, 32, 32, 16, 1, false, true, 4>, 1, 16, false, false>(),  // 13: M = 16384 with 32 points per thread in 512 threads, half exchange: two workgroups of eight waves per CU at 128 VGPRs (48 B of scratch)
    make_plan<Plan<3072, 48, 3, 16, 16, 12, 1, false, true, 3>, 1, 6, false, false>(),    // 14: the 3 * 2^k plan at three waves per SIMD (168 VGPRs, 100 B of scratch)
    make_plan<Plan<3072, 24, 4, 8, 8, 4, 12, false, false, 4>, 1, 6, false, false>(),     // 15: M = 3072 with 24 points per thread in TWO waves, four passes 8 x 8 x 4 x 12, full exchange (24 KiB: six transforms per CU)
    make_plan<Plan<3072, 24, 4, 8, 8, 4, 12, false, true, 4>, 1, 6, false, false>(),      // 16: same, half exchange (12 KiB: eight transforms per CU, four waves per SIMD)
    make_plan<Plan<3072, 24, 4, 8, 8, 4, 12, false, true, 4>, 2, 6, false, false>(),      // 17: same, two transforms per 256-thread workgroup
    make_plan<Plan<512, 16, 3, 8, 8, 8, 1>, 2, 8, false, false>(),                        // 18: config 3's transform as 8 x 8 x 8 (resident launches: 5.96 - 6.01 us per step against 5.49 - 5.72 for the default)
    make_plan<Plan<512, 16, 3, 8, 8, 8, 1>, 4, 8, false, false>(),                        // 19: same, four transforms per workgroup (6.21 - 6.25)
    make_plan<Plan<512, 16, 3, 16, 4, 8, 1>, 4, 8, false, false>(),                       // 20: the default radices, four transforms per 128-thread workgroup (5.72 - 5.76)
    make_plan<Plan<512, 16, 3, 16, 4, 8, 1>, 1, 8, false, false>(),                       // 21: one transform per 32-thread workgroup (9.65)
    // round 5: config 2's per-chunk transform (M = 4096) WITHOUT the cross-lane pairing - 32 points per thread in TWO waves, half-buffer
    // exchange (16 KiB), 128 VGPRs: eight independent transforms per CU instead of the XL plan's four
    make_plan<Plan<4096, 32, 3, 32, 8, 16, 1, false, true, 4>, 1, 8, false, false>(),      // 22: radices 32 x 8 x 16 (F = 2N column; since round 5 the DEFAULT plan of this size in the F = 4N column, plan_table.hpp: ADSP_PLAN_4096)
    make_plan<Plan<4096, 32, 3, 16, 16, 16, 1, false, true, 4>, 1, 8, false, false>(),     // 23: radices 16 x 16 x 16 (two butterflies per thread and pass)
    make_plan<Plan<4096, 32, 3, 32, 8, 16, 1, false, true, 4>, 2, 8, false, false>(),      // 24: = 22, two transforms per 256-thread workgroup
    make_plan<Plan<4096, 32, 3, 32, 8, 16, 1, false, false, 4>, 1, 8, false, false>(),     // 25: = 22 with the full exchange (32 KiB: four workgroups of two waves per CU)
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true>, 1, 8, false, false>(),               // 26: the XL plan = the default of the F = 2N column (kept for A/B symmetry)
    make_plan<Plan<4096, 16, 3, 16, 16, 16, 1, true>, 1, 16, false, false>(),              // 27: the XL plan in the F = 4N column, its default until round 5 (N = 2048 batches: 500 800 against 543 500 Msamples/s)
    // M = 2048 as ONE wave per transform: 32 points per thread, radices 32 x 4 x 16, half exchange (8 KiB), no inter-wave barrier
    make_plan<Plan<2048, 32, 3, 32, 4, 16, 1, false, true, 4>, 1, 8, false, false>(),      // 28: F = 2N (N = 2048 per chunk)
    make_plan<Plan<2048, 32, 3, 32, 4, 16, 1, false, true, 4>, 1, 16, false, false>(),     // 29: F = 4N (N = 1024 batches): the DEFAULT of that column since round 5 (plan_table.hpp: ADSP_PLAN_2048_4N)
    make_plan<Plan<2048, 32, 3, 32, 4, 16, 1, false, true, 4>, 2, 8, false, false>(),      // 30: = 28, two transforms per 128-thread workgroup
    make_plan<Plan<2048, 16, 3, 16, 16, 8, 1>, 1, 16, false, false>(),                     // 31: the two-wave 16-point plan in the F = 4N column, its default until round 5
};
// also measured and dropped: M = 8192 at four workgroups per CU / 128 VGPRs (spills, = 5), 64 points per thread for M = 8192
// (-3 % against 5), M = 8192 with radix-16 paired passes (-7 %), one wave per transform with 8 points per thread for M = 512
// (-14 %) and XL for M = 1024 (-6 %), 1 / 4 / 8 channels per workgroup for M = 512 (stream 10.9 / 8.3 / 8.8 us against 7.8),
// M = 8192 as an XL plan with 32 points per thread (16 x 16 x 32, partner in lane ^ 32, 152 VGPRs: -4.5 %).
}  // namespace

const adsp::PlanInfo* adsp::variants_f32(int* count) {
    *count = sizeof(kVariants) / sizeof(kVariants[0]);
    return kVariants;
}
