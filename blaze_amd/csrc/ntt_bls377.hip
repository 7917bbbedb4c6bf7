// NTT kernels over the scalar field of BLS377 (one translation unit per field).
#include "ntt_impl.hip.hpp"

namespace blz {
const NttFieldOps& ntt_ops_bls377() {
    static const NttFieldOps ops = make_ntt_ops<Fr_BLS377>();
    return ops;
}
}  // namespace blz
