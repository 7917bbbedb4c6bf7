//! Error type of the crate.  Variant names and payloads are the public surface the reference's callers match on
//! (`/root/reference/src/error.rs:6-32` lists them); everything else here is this crate's own: the C ABI reports a
//! small integer (`include/blaze_hip.h`, `BLZ_ERR_*`) and `DriverClientError::from_code` is the one place that
//! turns it back into a variant, with the library's last message as the payload.
use std::{error::Error as StdError, fmt, io};

pub type Result<T> = std::result::Result<T, DriverClientError>;

#[derive(Debug)]
pub enum DriverClientError {
    /// host -> device copy failed (`BLZ_ERR_WRITE` = 1); `offset` carries the library's message
    WriteError { offset: String, source: io::Error },
    /// device -> host copy failed (`BLZ_ERR_READ` = 2)
    ReadError { offset: String, source: io::Error },
    /// kept for source compatibility: there is no HBICAP on a GPU (`BLZ_ERR_HBICAP` = 3, never returned)
    HBICAPNotReady,
    /// a call out of order or an argument the primitive refuses (`BLZ_ERR_INVALID_PARAM` = 4)
    InvalidPrimitiveParam,
    /// kept for source compatibility with the Poseidon instruction loader (`BLZ_ERR_CSV` = 5)
    CsvError(csv::Error),
    /// the shared library or one of its dependencies could not be loaded (`BLZ_ERR_LOAD_FAILED` = 6)
    LoadFailed { path: String },
    /// no usable device behind the requested id (`BLZ_ERR_FILE` = 7: the reference fails to open `/dev/xdma*` here)
    FileError(io::Error),
    /// anything else, HIP runtime errors included (`BLZ_ERR_UNKNOWN` = 8)
    Unknown,
}

impl DriverClientError {
    /// The C ABI's return code for this variant (0 is success and has no variant).
    pub fn code(&self) -> i32 {
        match self {
            Self::WriteError { .. } => 1,
            Self::ReadError { .. } => 2,
            Self::HBICAPNotReady => 3,
            Self::InvalidPrimitiveParam => 4,
            Self::CsvError(_) => 5,
            Self::LoadFailed { .. } => 6,
            Self::FileError(_) => 7,
            Self::Unknown => 8,
        }
    }

    /// Variant for a non-zero return code; `detail` is `blz_last_error_message()`.
    pub fn from_code(code: i32, detail: String) -> Self {
        let as_io = |m: &str| io::Error::new(io::ErrorKind::Other, m.to_owned());
        match code {
            1 => Self::WriteError { source: as_io(&detail), offset: detail },
            2 => Self::ReadError { source: as_io(&detail), offset: detail },
            3 => Self::HBICAPNotReady,
            4 => Self::InvalidPrimitiveParam,
            5 => Self::CsvError(csv::Error::from(as_io(&detail))),
            6 => Self::LoadFailed { path: detail },
            7 => Self::FileError(as_io(&detail)),
            _ => Self::Unknown,
        }
    }
}

impl fmt::Display for DriverClientError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            Self::WriteError { offset, .. } => write!(f, "copy to the device failed: {offset}"),
            Self::ReadError { offset, .. } => write!(f, "copy from the device failed: {offset}"),
            Self::HBICAPNotReady => f.write_str("HBICAP not ready (not applicable to this device)"),
            Self::InvalidPrimitiveParam => f.write_str("call sequence or parameter refused by the primitive"),
            Self::CsvError(e) => write!(f, "instruction file: {e}"),
            Self::LoadFailed { path } => write!(f, "could not load {path}"),
            Self::FileError(e) => write!(f, "no usable device: {e}"),
            Self::Unknown => f.write_str("device runtime error"),
        }
    }
}

impl StdError for DriverClientError {
    fn source(&self) -> Option<&(dyn StdError + 'static)> {
        match self {
            Self::WriteError { source, .. } | Self::ReadError { source, .. } | Self::FileError(source) => Some(source),
            Self::CsvError(e) => Some(e),
            _ => None,
        }
    }
}

impl From<io::Error> for DriverClientError {
    fn from(e: io::Error) -> Self {
        Self::FileError(e)
    }
}

impl From<csv::Error> for DriverClientError {
    fn from(e: csv::Error) -> Self {
        Self::CsvError(e)
    }
}
