//! `/root/reference/src/ingo_msm/mod.rs`: same re-exports; the register map (`msm_hw_code`) is gone.
mod msm_api;
mod msm_cfg;

pub use msm_api::*;
pub use msm_cfg::{Curve, PointMemoryType};
