// NTT kernels over the scalar field of BN254 (one translation unit per field).
#include "ntt_impl.hip.hpp"

namespace blz {
const NttFieldOps& ntt_ops_bn254() {
    static const NttFieldOps ops = make_ntt_ops<Fr_BN254>();
    return ops;
}
}  // namespace blz
