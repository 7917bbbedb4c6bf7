// NTT kernels over the scalar field of BLS381 (one translation unit per field).
#include "ntt_impl.hip.hpp"

namespace blz {
const NttFieldOps& ntt_ops_bls381() {
    static const NttFieldOps ops = make_ntt_ops<Fr_BLS381>();
    return ops;
}
}  // namespace blz
