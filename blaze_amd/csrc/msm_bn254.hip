// MSM kernels instantiated for Fq_BN254 (see msm_impl.cuh).
#include "msm_impl.cuh"
namespace blz {
const MsmCurveOps& msm_ops_bn254() {
    static const MsmCurveOps ops = make_ops<Fq_BN254>();
    return ops;
}
}  // namespace blz
