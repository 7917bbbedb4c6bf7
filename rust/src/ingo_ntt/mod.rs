//! NTT primitive: `NTTClient`, `NTTInput`, `NttInit`, `NTT` and the size constants.  The device buffer is flat, so
//! the host-side 16-bank permutation of the reference (`ntt_data`) has no counterpart in this crate; the permutation
//! exists as device kernels (`blz_ntt_banks_*`) for files already in bank order.
pub use self::ntt_api::*;

mod ntt_api;
