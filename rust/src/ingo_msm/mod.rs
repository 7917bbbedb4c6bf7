//! MSM primitive: `MSMClient` and its parameter / input / result types (`msm_api`), curve and memory-mode enums
//! (`msm_cfg`).  The reference's module of the same name also carries the FPGA register map; here the transport is
//! the C ABI, so there is nothing else to export.
pub use self::msm_api::*;
pub use self::msm_cfg::{Curve, PointMemoryType};

mod msm_cfg;
mod msm_api;
