//! `/root/reference/src/driver_client/mod.rs:1-7` with the register map (`dclient_code`) gone: the transport is
//! the C ABI of libblaze_hip, one function per `DriverPrimitive` method.
mod dclient;
mod dclient_cfg;
pub(crate) mod hip_ffi;

pub use dclient::*;
pub use dclient_cfg::{CardType, DriverConfig};
