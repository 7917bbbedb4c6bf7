// MSM kernels instantiated for BN254's base field on 32-bit limbs (field.hip.hpp, ec.hip.hpp): the representation of the
// precompute shapes (msm_engine.hpp `repr`).  Same constants as Fq_BN254, no reduced-radix twin.
#include "msm_impl.hip.hpp"
namespace blz {
struct Fq_BN254_W32 : Fq_BN254 {
    using RR = void;
};
const MsmCurveOps& msm_ops_bn254_w32() {
    static const MsmCurveOps ops = make_ops<Fq_BN254_W32>();
    return ops;
}
}  // namespace blz
