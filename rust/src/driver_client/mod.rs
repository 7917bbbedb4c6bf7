//! The device handle (`DriverClient`), the `DriverPrimitive` trait every primitive implements, and the card
//! configuration type kept for source compatibility.  `hip_ffi` holds the `extern "C"` declarations of
//! `include/blaze_hip.h`: the transport that replaces the reference's XDMA register map.
pub use self::dclient::*;
pub use self::dclient_cfg::{CardType, DriverConfig};

pub(crate) mod hip_ffi;
mod dclient_cfg;
mod dclient;
