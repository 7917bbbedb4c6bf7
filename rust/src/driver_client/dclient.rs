//! `DriverPrimitive`, `ParametersAPI` and `DriverClient` of `/root/reference/src/driver_client/dclient.rs:16-93`.
//! `DriverClient` keeps its name and constructor; its three XDMA file handles become a HIP device ordinal.
use super::{dclient_cfg::*, hip_ffi};
use crate::error::*;

/// A trait for defining functions related to parameters of specific core image.
pub trait ParametersAPI {
    fn parse_image_params(params: u32) -> Self;
    fn debug_information(&self);
}

/// The operator contract every primitive client implements (`dclient.rs:28-46`), unchanged.
pub trait DriverPrimitive<T, P, I, O> {
    fn new(ptype: T, dclient: DriverClient) -> Self;
    fn loaded_binary_parameters(&self) -> Vec<u32>;
    fn initialize(&self, param: P) -> Result<()>;
    fn set_data(&self, input: I) -> Result<()>;
    fn start_process(&self, param: Option<usize>) -> Result<()>;
    fn wait_result(&self) -> Result<()>;
    fn result(&self, param: Option<usize>) -> Result<Option<O>>;
}

/// One client of one device: the reference's "FPGA slot" id is the HIP device ordinal.
pub struct DriverClient {
    #[allow(dead_code)]
    pub(crate) cfg: DriverConfig,
    pub id: i32,
}

impl DriverClient {
    /// `DriverClient::new("0", DriverConfig::driver_client_cfg(CardType::C1100))` as in the reference
    /// (`dclient.rs:79-86`).  Panics when the device does not exist, like the reference's `open().unwrap()`.
    pub fn new(id: &str, cfg: DriverConfig) -> Self {
        let id: i32 = id.parse().expect("device id must be a number");
        let n = unsafe { hip_ffi::blz_device_count() };
        assert!(id >= 0 && id < n, "no HIP device with ordinal {} ({} visible)", id, n);
        DriverClient { cfg, id }
    }

    /// `dclient.rs:88-93` toggles the DFX decoupler and sleeps 100 ms; a GPU client has nothing to reset here
    /// (`MSMClient::reset_engine` / `NTTClient::reset_engine` drain the primitive's own queues).
    pub fn reset(&self) -> Result<()> {
        Ok(())
    }

    // FPGA shell management (`dclient.rs:96-279`): accepted and ignored, so the reference's test prologues run.
    pub fn initialize_cms(&self) -> Result<()> {
        Ok(())
    }
    pub fn reset_sensor_data(&self) -> Result<()> {
        Ok(())
    }
    pub fn setup_before_load_binary(&self) -> Result<()> {
        Ok(())
    }
    pub fn load_binary(&self, _binary: &[u8]) -> Result<()> {
        Ok(())
    }
    pub fn unblock_firewalls(&self) -> Result<()> {
        Ok(())
    }
    pub fn firewalls_status(&self) {}

    /// The card's HBM outlives the process that wrote it; GPU memory needs a holder process: the holder exports
    /// its arena, other processes attach it (include/blaze_hip.h, `blz_arena_export` / `blz_arena_attach`).
    pub fn arena_export(&self, registry_path: &str) -> Result<()> {
        let p = std::ffi::CString::new(registry_path).map_err(|_| DriverClientError::InvalidPrimitiveParam)?;
        check(unsafe { hip_ffi::blz_arena_export(self.id, p.as_ptr()) })
    }
    pub fn arena_attach(&self, registry_path: &str) -> Result<()> {
        let p = std::ffi::CString::new(registry_path).map_err(|_| DriverClientError::InvalidPrimitiveParam)?;
        check(unsafe { hip_ffi::blz_arena_attach(self.id, p.as_ptr()) })
    }
}

/// Return code of the C ABI -> `Result`: the variant comes from `DriverClientError::from_code`, its payload from the
/// library's thread-local last message.
pub(crate) fn check(rc: std::os::raw::c_int) -> Result<()> {
    if rc == 0 {
        return Ok(());
    }
    let detail = unsafe {
        let p = hip_ffi::blz_last_error_message();
        if p.is_null() {
            String::new()
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    };
    Err(DriverClientError::from_code(rc as i32, detail))
}
