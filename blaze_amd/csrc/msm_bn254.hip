// MSM kernels instantiated for Fq_BN254 (see msm_impl.hip.hpp).
// 136 VGPRs (the attribute counts in halves: msm_impl.hip.hpp): three accumulation waves per SIMD and one wave of the next
// task's digit sort are 3 x 136 + 72 = 480 registers - what a SIMD hands out (msm.hip run()); at the 140 the kernel
// compiles to on its own the sort only ran in the slots the accumulation left open
#define BLZ_ACC_VGPR_CAP 68
#include "msm_impl.hip.hpp"
namespace blz {
const MsmCurveOps& msm_ops_bn254() {
    static const MsmCurveOps ops = make_ops<Fq_BN254>();
    return ops;
}
}  // namespace blz
