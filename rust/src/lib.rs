//! ingo-blaze on AMD MI355X: the reference crate's `DriverPrimitive` clients for MSM and NTT
//! (`/root/reference/src/lib.rs:8-13`), with `driver_client` re-implemented as a thin FFI shim over
//! `libblaze_hip` (hand-written HIP kernels for gfx950) instead of XDMA character devices.
pub mod driver_client;
pub mod error;
pub mod ingo_msm;
pub mod ingo_ntt;
