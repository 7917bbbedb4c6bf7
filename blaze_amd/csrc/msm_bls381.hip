// MSM kernels instantiated for Fq_BLS381 (see msm_impl.hip.hpp).
#include "msm_impl.hip.hpp"
namespace blz {
const MsmCurveOps& msm_ops_bls381() {
    static const MsmCurveOps ops = make_ops<Fq_BLS381>();
    return ops;
}
}  // namespace blz
