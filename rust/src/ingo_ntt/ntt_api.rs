//! `NTTClient` of `/root/reference/src/ingo_ntt/ntt_api.rs:8-125` over libblaze_hip.  The transform: size 2^27 over
//! the BLS12-381 scalar field, forward, natural order in and out, omega = 7^((r-1)/2^27) (the reference states none
//! of these).
use crate::{
    driver_client::{hip_ffi::*, *},
    error::*,
};

pub const NTT_LOG_SIZE: i32 = 27; // ntt_data.rs:65: NTT_SIZE = 2^27
pub const NTT_WORD_SIZE: usize = 32; // ntt_data.rs:66

pub enum NTT {
    Ntt,
}

pub struct NTTClient {
    nbytes: usize,
    pub driver_client: DriverClient,
    h: *mut BlzNtt,
}
unsafe impl Send for NTTClient {}

pub struct NttInit {}

#[derive(Debug, Clone)]
pub struct NTTInput {
    pub buf_host: usize,
    pub data: Vec<u8>,
}

impl DriverPrimitive<NTT, NttInit, NTTInput, Vec<u8>> for NTTClient {
    /// ntt_api.rs:26-31
    fn new(_ptype: NTT, dclient: DriverClient) -> Self {
        NTTClient::with_log_size(dclient, NTT_LOG_SIZE)
    }

    /// `todo!()` in the reference too (ntt_api.rs:33-35)
    fn loaded_binary_parameters(&self) -> Vec<u32> {
        todo!()
    }

    /// ntt_api.rs:37-56 writes the debug program; nothing to program here
    fn initialize(&self, _: NttInit) -> Result<()> {
        check(unsafe { blz_ntt_initialize(self.h) })
    }

    /// ntt_api.rs:58-70: select the buffer and AP_START
    fn start_process(&self, buf_kernel: Option<usize>) -> Result<()> {
        check(unsafe { blz_ntt_start_process(self.h, buf_kernel.unwrap()) })
    }

    /// ntt_api.rs:72-87: `NTTBanks::preprocess` and the 16 bank writes become one flat copy
    fn set_data(&self, input: NTTInput) -> Result<()> {
        check(unsafe { blz_ntt_set_data(self.h, input.buf_host, input.data.as_ptr(), input.data.len()) })
    }

    /// ntt_api.rs:89-108: the spin on AP_DONE
    fn wait_result(&self) -> Result<()> {
        check(unsafe { blz_ntt_wait_result(self.h) })
    }

    /// ntt_api.rs:110-124: 16 bank reads + `postprocess` become one flat copy
    fn result(&self, buf_num: Option<usize>) -> Result<Option<Vec<u8>>> {
        let mut res = vec![0u8; self.nbytes];
        check(unsafe { blz_ntt_result(self.h, buf_num.unwrap(), res.as_mut_ptr(), res.len()) })?;
        Ok(Some(res))
    }
}

impl Drop for NTTClient {
    fn drop(&mut self) {
        unsafe { blz_ntt_free(self.h) }
    }
}

impl NTTClient {
    /// Smaller transforms exist for tests (the reference has no such knob).
    pub fn with_log_size(dclient: DriverClient, log_size: i32) -> Self {
        let mut h: *mut BlzNtt = std::ptr::null_mut();
        check(unsafe { blz_ntt_new(dclient.id, log_size, &mut h) }).expect("blz_ntt_new failed");
        NTTClient { nbytes: NTT_WORD_SIZE << log_size, driver_client: dclient, h }
    }

    /// The transform's convention, which the reference leaves unstated (`NttInit {}` is empty, ntt_api.rs:8-23; its golden
    /// files are external, tests/integration_ntt.rs:15-18): `field` = a `Curve` as i32 (its scalar field), `flags` =
    /// BLZ_NTT_INVERSE (2) | BLZ_NTT_BITREV_INPUT (4) | BLZ_NTT_BITREV_OUTPUT (8) | BLZ_NTT_NO_FACTOR_TABLE (1), `root` = any
    /// primitive 2^log_size-th root of unity as 32 canonical little-endian bytes (checked on the device) or `None` for
    /// g^((r - 1) / 2^log_size).  A host that holds vectors made for the card says here what they assume.
    pub fn with_convention(dclient: DriverClient, field: i32, log_size: i32, flags: u32, root: Option<&[u8; 32]>) -> Result<Self> {
        let mut h: *mut BlzNtt = std::ptr::null_mut();
        let rp = root.map_or(std::ptr::null(), |r| r.as_ptr());
        check(unsafe { blz_ntt_new_ex3(dclient.id, field, log_size, flags, rp, &mut h) })?;
        Ok(NTTClient { nbytes: NTT_WORD_SIZE << log_size, driver_client: dclient, h })
    }

    /// Device time of the last transform in ms (HIP events around its three passes).
    pub fn last_kernel_ms(&self) -> Result<f32> {
        let mut v = 0f32;
        check(unsafe { blz_ntt_last_kernel_ms(self.h, &mut v) })?;
        Ok(v)
    }

    /// `result(Some(buf))` and `set_data(NTTInput { buf_host: buf, data })` of the double-buffered loop
    /// (tests/integration_ntt.rs:102-136) as one full-duplex call: the previous result leaves `buf` piece by piece into `out`
    /// while `data` lands in the places that have left.
    pub fn exchange(&self, buf: usize, data: &[u8], out: &mut [u8]) -> Result<()> {
        check(unsafe { blz_ntt_exchange(self.h, buf, data.as_ptr(), data.len(), out.as_mut_ptr(), out.len()) })
    }

    /// `[device bytes held, pass 2 reads its factor table (1) or steps (0), pass 1 boundary table, log_size]`.
    pub fn info(&self) -> Result<[u64; 4]> {
        let mut v = [0u64; 4];
        check(unsafe { blz_ntt_info(self.h, v.as_mut_ptr()) })?;
        Ok(v)
    }

    pub fn reset_engine(&self) -> Result<()> {
        check(unsafe { blz_ntt_reset(self.h) })
    }
}
