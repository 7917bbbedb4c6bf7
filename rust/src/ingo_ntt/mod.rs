//! `/root/reference/src/ingo_ntt/mod.rs`: `ntt_data` (the 16-bank wire permutation) and `ntt_hw_code` are not
//! needed on the host: the device buffer is flat (the permutation exists as device kernels for bank files).
mod ntt_api;

pub use ntt_api::*;
