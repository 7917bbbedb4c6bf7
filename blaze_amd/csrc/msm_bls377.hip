// MSM kernels instantiated for Fq_BLS377 (see msm_impl.hip.hpp).
#include "msm_impl.hip.hpp"
namespace blz {
const MsmCurveOps& msm_ops_bls377() {
    static const MsmCurveOps ops = make_ops<Fq_BLS377>();
    return ops;
}
}  // namespace blz
