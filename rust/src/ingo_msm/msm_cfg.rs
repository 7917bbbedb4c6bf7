//! Sizes per curve (`/root/reference/src/ingo_msm/msm_cfg.rs:3-92`).  The FIFO addresses of the FPGA flow have no
//! meaning here; `(BN254, HBM)`, a `todo!()` in the reference, is defined by analogy (32-byte coordinates).
#[derive(Debug, PartialEq, Eq, Clone, Copy)]
pub enum Curve {
    BLS377,
    BLS381,
    BN254,
}

#[derive(Debug, PartialEq, Eq, Clone, Copy)]
pub enum PointMemoryType {
    HBM,
    DMA,
}

#[derive(Debug, Copy, Clone)]
pub(super) struct MSMConfig {
    /// The size in bytes of result point. The point is expected to be in projective form.
    pub result_point_size: usize,
    /// The size of one point in bytes. Point is represented in affine form.
    pub point_size: Option<usize>,
    /// The size of scalar coordinate in bytes.
    pub scalar_size: usize,
}

impl MSMConfig {
    pub(super) fn msm_cfg(curve: Curve, _mem: PointMemoryType) -> Self {
        match curve {
            Curve::BLS377 | Curve::BLS381 => MSMConfig { result_point_size: 144, point_size: Some(96), scalar_size: 32 },
            Curve::BN254 => MSMConfig { result_point_size: 96, point_size: Some(64), scalar_size: 32 },
        }
    }
}

impl Curve {
    /// numbering of `enum blz_curve` (include/blaze_hip.h) = declaration order here
    pub(super) fn code(self) -> i32 {
        match self {
            Curve::BLS377 => 0,
            Curve::BLS381 => 1,
            Curve::BN254 => 2,
        }
    }
}
impl PointMemoryType {
    /// numbering of `enum blz_mem`
    pub(super) fn code(self) -> i32 {
        match self {
            PointMemoryType::HBM => 0,
            PointMemoryType::DMA => 1,
        }
    }
}
