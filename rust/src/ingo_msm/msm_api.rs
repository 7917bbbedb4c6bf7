//! `MSMClient` of `/root/reference/src/ingo_msm/msm_api.rs:8-379` over libblaze_hip: the public types are the
//! reference's, every method forwards to the C-ABI function that replaces the register / DMA sequence it used to
//! drive (cited per method).
use super::msm_cfg::*;
use crate::{
    driver_client::{hip_ffi::*, *},
    error::*,
};

pub struct MSMClient {
    mem_type: PointMemoryType,
    // If precompute factor set to 1 is the basic MSM computation without optimization
    precompute_factor: u32,
    msm_cfg: MSMConfig,
    pub driver_client: DriverClient,
    h: *mut BlzMsm,
}
// the handle may move between threads but is not re-entrant: share it behind a Mutex, as the reference's
// Poseidon test shares its client (tests/integration_poseidon.rs:68)
unsafe impl Send for MSMClient {}

pub struct MSMInit {
    pub mem_type: PointMemoryType,
    pub is_precompute: bool,
    pub curve: Curve,
}

#[derive(Debug, Copy, Clone)]
pub struct MSMParams {
    pub nof_elements: u32,
    pub hbm_point_addr: Option<(u64, u64)>,
}

pub struct MSMInput {
    pub points: Option<Vec<u8>>,
    pub scalars: Vec<u8>,
    pub params: MSMParams,
}

#[derive(Debug, Clone)]
pub struct MSMResult {
    pub result: Vec<u8>,
    pub result_label: u32,
}

pub const PRECOMPUTE_FACTOR_BASE: u32 = 1;
pub const PRECOMPUTE_FACTOR: u32 = 8;

fn hbm_args(p: &MSMParams) -> (i32, u64, u64) {
    match p.hbm_point_addr {
        Some((addr, off)) => (1, addr, off),
        None => (0, 0, 0),
    }
}

impl DriverPrimitive<MSMInit, MSMParams, MSMInput, MSMResult> for MSMClient {
    /// msm_api.rs:44-55.  Panics when the device cannot be opened, like the reference's `open().unwrap()`.
    fn new(init: MSMInit, dclient: DriverClient) -> Self {
        let mut h: *mut BlzMsm = std::ptr::null_mut();
        check(unsafe { blz_msm_new(dclient.id, init.mem_type.code(), init.is_precompute as i32, init.curve.code(), &mut h) })
            .expect("blz_msm_new failed");
        MSMClient {
            mem_type: init.mem_type,
            precompute_factor: if init.is_precompute { PRECOMPUTE_FACTOR } else { PRECOMPUTE_FACTOR_BASE },
            msm_cfg: MSMConfig::msm_cfg(init.curve, init.mem_type),
            driver_client: dclient,
            h,
        }
    }

    /// msm_api.rs:57-70: [image id, image parameters]; the second word decodes with `MSMImageParametrs`.
    fn loaded_binary_parameters(&self) -> Vec<u32> {
        let mut v = [0u32; 2];
        check(unsafe { blz_msm_loaded_binary_parameters(self.h, v.as_mut_ptr()) }).expect("loaded_binary_parameters");
        v.to_vec()
    }

    /// msm_api.rs:72-111: bases source, HBM start address, number of elements.
    fn initialize(&self, params: MSMParams) -> Result<()> {
        let (has, addr, off) = hbm_args(&params);
        check(unsafe { blz_msm_initialize(self.h, params.nof_elements, has, addr, off) })
    }

    /// msm_api.rs:155-220.  The 2048-element chunk loop over the FIFO addresses is gone: the library stages the
    /// whole buffers (or takes the bases from the arena) and enqueues the task.
    fn set_data(&self, data: MSMInput) -> Result<()> {
        let (has, addr, off) = hbm_args(&data.params);
        let (pp, pl) = match data.points.as_ref() {
            Some(p) => (p.as_ptr(), p.len()),
            None => (std::ptr::null(), 0),
        };
        check(unsafe {
            blz_msm_set_data(self.h, pp, pl, data.scalars.as_ptr(), data.scalars.len(), data.params.nof_elements, has, addr, off)
        })
    }

    /// msm_api.rs:113-120: push the configured task onto the device task queue.
    fn start_process(&self, _: Option<usize>) -> Result<()> {
        check(unsafe { blz_msm_start_process(self.h) })
    }

    /// msm_api.rs:222-238: the spin on RESULT_VALID.
    fn wait_result(&self) -> Result<()> {
        check(unsafe { blz_msm_wait_result(self.h) })
    }

    /// msm_api.rs:240-274: result bytes `Z | Y | X`, result label, POP_RESULT.
    fn result(&self, _param: Option<usize>) -> Result<Option<MSMResult>> {
        let mut out = vec![0u8; self.msm_cfg.result_point_size];
        let mut n: usize = 0;
        let mut label: u32 = 0;
        check(unsafe { blz_msm_result(self.h, out.as_mut_ptr(), out.len(), &mut n, &mut label) })?;
        out.truncate(n);
        Ok(Some(MSMResult { result: out, result_label: label }))
    }
}

impl Drop for MSMClient {
    fn drop(&mut self) {
        unsafe { blz_msm_free(self.h) }
    }
}

impl MSMClient {
    /// msm_api.rs:278-283
    pub fn task_label(&self) -> Result<u32> {
        let mut v = 0u32;
        check(unsafe { blz_msm_task_label(self.h, &mut v) })?;
        Ok(v)
    }

    /// msm_api.rs:285-290
    pub fn nof_elements(&self) -> Result<u32> {
        let mut v = 0u32;
        check(unsafe { blz_msm_nof_elements(self.h, &mut v) })?;
        Ok(v)
    }

    /// msm_api.rs:292-297
    pub fn is_msm_engine_ready(&self) -> Result<u32> {
        let mut v = 0u32;
        check(unsafe { blz_msm_is_engine_ready(self.h, &mut v) })?;
        Ok(v)
    }

    /// A task fed by several `set_data` calls (the card counts what its FIFOs receive against NUMBER_OF_MSM_ELEMENTS,
    /// msm_api.rs:155-202 / msm_hw_code.rs:18-19; blaze_hip.h "STREAMED TASKS"): with a task queued, a `set_data` whose
    /// `params.nof_elements` is smaller than what the task still lacks is its next slice.  `(received, queued)`.
    pub fn stream_progress(&self) -> Result<(u32, u32)> {
        let mut v = [0u32; 2];
        check(unsafe { blz_msm_stream_progress(self.h, v.as_mut_ptr()) })?;
        Ok((v[0], v[1]))
    }

    /// msm_api.rs:299-313: raw bytes at arena offset `addr + offset`; bases then come from there.
    pub fn load_data_to_hbm(&self, points: &[u8], addr: u64, offset: u64) -> Result<()> {
        log::debug!("HBM adress: {:#X?}", &addr);
        check(unsafe { blz_msm_load_data_to_hbm(self.h, points.as_ptr(), points.len(), addr, offset) })
    }

    /// msm_api.rs:315-322
    pub fn get_data_from_hbm(&self, data_len: usize, addr: u64, offset: u64) -> Result<Vec<u8>> {
        let mut res = vec![0u8; data_len];
        check(unsafe { blz_msm_get_data_from_hbm(self.h, res.as_mut_ptr(), data_len, addr, offset) })?;
        Ok(res)
    }

    /// msm_api.rs:324-330 dumps every register; here: the HIP-event timers of the last finished task, in ms
    /// [total, accumulate kernel, sort, phase 1, phase 2, phase 3, window bits, windows].
    pub fn get_api(&self) -> [f32; 8] {
        let mut t = [0f32; 8];
        let _ = check(unsafe { blz_msm_last_timings(self.h, t.as_mut_ptr()) });
        log::debug!("MSM timers: {:?}", t);
        t
    }

    /// Drains the primitive's task and result queues (the reference resets the whole card through
    /// `driver_client.reset()`).
    pub fn reset_engine(&self) -> Result<()> {
        check(unsafe { blz_msm_reset(self.h) })
    }

    /// Opt in to the resident-base window table (`include/blaze_hip.h`): bases in the arena, `precompute_factor` 1.
    /// `mode`: 0 off, 1 where it pays (the BLS curves), 2 always.
    pub fn set_window_table(&self, mode: i32) -> Result<()> {
        check(unsafe { blz_msm_set_window_table(self.h, mode) })
    }
    /// Enqueue the table's build for the bases at `hbm_addr` (it runs beside the tasks, which take the plain path until it is
    /// there) and wait up to `wait_ms` for it (0: not at all, negative: the library's wait deadline); `true`: the table is in place.
    pub fn prepare_window_table(&self, nof_elements: u32, hbm_addr: (u64, u64), wait_ms: i32) -> Result<bool> {
        let mut ready: std::os::raw::c_int = 0;
        check(unsafe { blz_msm_prepare_window_table(self.h, nof_elements, hbm_addr.0, hbm_addr.1, wait_ms, &mut ready) })?;
        Ok(ready != 0)
    }
    /// Opt a precompute client in to the checked-table plan (`include/blaze_hip.h`): resident x8 tables are checked once per
    /// load against `precompute_base_*` and, if consistent, served as 4n even bases with 64-bit chunks; identical result bytes.
    pub fn set_precompute_plan(&self, enable: bool) -> Result<()> {
        check(unsafe { blz_msm_set_precompute_plan(self.h, enable as std::os::raw::c_int) })
    }
    /// Run the table check (and build the even-base copy) now; `true`: tasks over these bases take the plan.
    pub fn prepare_precompute_plan(&self, nof_elements: u32, hbm_addr: (u64, u64)) -> Result<bool> {
        let mut ok: std::os::raw::c_int = 0;
        check(unsafe { blz_msm_prepare_precompute_plan(self.h, nof_elements, hbm_addr.0, hbm_addr.1, &mut ok) })?;
        Ok(ok != 0)
    }
    /// `[took the plan, check state (0 unchecked, 1 consistent, 2 refuted), check microseconds, even-base copy bytes]`.
    pub fn precompute_plan_info(&self) -> Result<[u64; 4]> {
        let mut out = [0u64; 4];
        check(unsafe { blz_msm_precompute_plan_info(self.h, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// One shard of a job split by scalar chunk: only bits `[bit_lo, bit_hi)` of every scalar, result weighted `2^bit_lo`.
    pub fn set_scalar_range(&self, bit_lo: u32, bit_hi: u32) -> Result<()> {
        check(unsafe { blz_msm_set_scalar_range(self.h, bit_lo, bit_hi) })
    }
    /// `[first element, element count, bit_lo, bit_hi]` of `rank` in the split the library picks for `nranks` devices.
    pub fn shard_layout(curve: Curve, nof_elements: u32, nranks: i32, rank: i32) -> Result<[u32; 4]> {
        let mut out = [0u32; 4];
        check(unsafe { blz_msm_shard_layout(curve.code(), nof_elements, nranks, rank, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// The same choice with the flow's transfers priced in (`flags`: 1 = scalars from host memory with every task, 2 = bases
    /// too): `[first, count, bit_lo, bit_hi, ranges, compute us, link us, device MiB per rank]`.
    pub fn shard_layout_ex(curve: Curve, nof_elements: u32, nranks: i32, rank: i32, flags: u32) -> Result<[u32; 8]> {
        let mut out = [0u32; 8];
        check(unsafe { blz_msm_shard_layout_ex(curve.code(), nof_elements, nranks, rank, flags, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// Device bytes behind this client: `[workspace, staging, arena raw, arena Montgomery copies, arena window tables, total]`
    /// (the arena figures are per device).
    pub fn memory_info(&self) -> Result<[u64; 6]> {
        let mut out = [0u64; 6];
        check(unsafe { blz_msm_memory_info(self.h, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// `[table bytes, window bits, windows, build time in microseconds]` of the table the last HBM task used.
    pub fn window_table_info(&self) -> Result<[u64; 4]> {
        let mut out = [0u64; 4];
        check(unsafe { blz_msm_window_table_info(self.h, out.as_mut_ptr()) })?;
        Ok(out)
    }

    pub fn mem_type(&self) -> PointMemoryType {
        self.mem_type
    }
    pub fn precompute_factor(&self) -> u32 {
        self.precompute_factor
    }

    // ---- multi-GPU (one client per device; include/blaze_hip.h "Multi-GPU exchange")
    pub fn comm_unique_id() -> Result<[u8; BLZ_COMM_ID_BYTES]> {
        let mut id = [0u8; BLZ_COMM_ID_BYTES];
        check(unsafe { blz_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    pub fn comm_init(&self, rank: i32, nranks: i32, id: &[u8; BLZ_COMM_ID_BYTES]) -> Result<()> {
        check(unsafe { blz_msm_comm_init(self.h, rank, nranks, id.as_ptr()) })
    }
    /// all-gather of every rank's partial result + rank-ordered add: the full MSM, same bytes on every rank
    pub fn all_gather_combine(&self, partial: &[u8]) -> Result<Vec<u8>> {
        if partial.len() != self.msm_cfg.result_point_size {
            return Err(DriverClientError::InvalidPrimitiveParam);
        }
        let mut out = vec![0u8; self.msm_cfg.result_point_size];
        check(unsafe { blz_msm_all_gather_combine(self.h, partial.as_ptr(), out.as_mut_ptr(), out.len()) })?;
        Ok(out)
    }
    /// One process, one thread, one client per device: bring all communicator ranks up as one RCCL group
    /// (rank i = clients[i]); the per-rank `comm_init` is a blocking rendezvous and cannot be called in sequence
    /// from a single thread.
    pub fn comm_init_all(clients: &[&MSMClient]) -> Result<()> {
        let hs: Vec<*mut BlzMsm> = clients.iter().map(|c| c.h).collect();
        check(unsafe { blz_msm_comm_init_all(hs.as_ptr(), hs.len() as c_int) })
    }
    /// The exchange for the clients of `comm_init_all`: `partials[i]` is client i's result; returns every
    /// client's full sum (identical bytes).
    pub fn all_gather_combine_all(clients: &[&MSMClient], partials: &[Vec<u8>]) -> Result<Vec<Vec<u8>>> {
        if clients.is_empty() || partials.len() != clients.len() {
            return Err(DriverClientError::InvalidPrimitiveParam);
        }
        let rs = clients[0].msm_cfg.result_point_size;
        if partials.iter().any(|p| p.len() != rs) {
            return Err(DriverClientError::InvalidPrimitiveParam);
        }
        let hs: Vec<*mut BlzMsm> = clients.iter().map(|c| c.h).collect();
        let flat: Vec<u8> = partials.iter().flat_map(|p| p.iter().copied()).collect();
        let mut out = vec![0u8; rs * clients.len()];
        check(unsafe { blz_msm_all_gather_combine_all(hs.as_ptr(), hs.len() as c_int, flat.as_ptr(), out.as_mut_ptr(), out.len()) })?;
        Ok(out.chunks(rs).map(|c| c.to_vec()).collect())
    }
    pub fn combine_partials(&self, partials: &[u8], count: usize) -> Result<Vec<u8>> {
        if partials.len() != count * self.msm_cfg.result_point_size {
            return Err(DriverClientError::InvalidPrimitiveParam);
        }
        let mut out = vec![0u8; self.msm_cfg.result_point_size];
        check(unsafe { blz_msm_combine_partials(self.h, partials.as_ptr(), count, out.as_mut_ptr(), out.len()) })?;
        Ok(out)
    }
}

/// The image-parameter word of `loaded_binary_parameters()[1]` (msm_api.rs:333-347).  The reference decodes it
/// with `packed_struct` after a bit reversal; net effect on the original word `p` (bit 0 = least significant):
/// each field occupies the bit range below and is stored bit-reversed inside it.
///   is_stub [28..=31], curve [20..=27], number_of_ec_adders [16..=19], buckets_mem_addr_width [8..=15],
///   number_of_segments [4..=7], place_holder [0..=3]
#[derive(Debug, PartialEq, Eq)]
pub struct MSMImageParametrs {
    pub hif2cpu_c_is_stub: u8,
    pub hif2_cpu_c_curve: u8,
    pub hif2_cpu_c_number_of_ec_adders: u8,
    pub hif2_cpu_c_buckets_mem_addr_width: u8,
    pub hif2_cpu_c_number_of_segments: u8,
    pub hif2_cpu_c_place_holder: u8,
}

fn field_msb_first(p: u32, lo: u32, hi: u32) -> u8 {
    // bit `lo` of the word is the field's most significant bit
    let mut v = 0u8;
    for b in lo..=hi {
        v = (v << 1) | ((p >> b) & 1) as u8;
    }
    v
}

impl ParametersAPI for MSMImageParametrs {
    fn parse_image_params(params: u32) -> MSMImageParametrs {
        MSMImageParametrs {
            hif2cpu_c_is_stub: field_msb_first(params, 28, 31),
            hif2_cpu_c_curve: field_msb_first(params, 20, 27),
            hif2_cpu_c_number_of_ec_adders: field_msb_first(params, 16, 19),
            hif2_cpu_c_buckets_mem_addr_width: field_msb_first(params, 8, 15),
            hif2_cpu_c_number_of_segments: field_msb_first(params, 4, 7),
            hif2_cpu_c_place_holder: field_msb_first(params, 0, 3),
        }
    }

    fn debug_information(&self) {
        log::debug!("Is Stub: {:?}", self.hif2cpu_c_is_stub);
        // curve code in bits 2.. of the field (0 BLS12-377, 1 BN254, 2 BLS12-381), bit 0 = "complex" (G2) flag
        match self.hif2_cpu_c_curve >> 2 {
            0 => log::debug!("This is BLS12_377 curve"),
            1 => log::debug!("This is BN254 curve"),
            2 => log::debug!("This is BLS12_381 curve"),
            _ => log::debug!("This is UNKNOWN curve"),
        }
        log::debug!("Number of EC adders (x16 compute units): {:?}", self.hif2_cpu_c_number_of_ec_adders);
        log::debug!("Width of buckets memory adrreses: {:?}", self.hif2_cpu_c_buckets_mem_addr_width);
        log::debug!("Number of segmemts (XCDs): {:?}", self.hif2_cpu_c_number_of_segments);
        log::debug!("Place Holder: {:?}", self.hif2_cpu_c_place_holder);
    }
}
