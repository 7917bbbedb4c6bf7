//! `extern "C"` view of `include/blaze_hip.h` (libblaze_hip.so): what replaces the three XDMA file handles of the
//! reference's `DriverClient` (`/root/reference/src/driver_client/dclient.rs:50-59`).  One function per
//! `DriverPrimitive` method per primitive; plain pointers and sizes; return value 0 = Ok, else the ordinal of a
//! `DriverClientError` variant.  tests/test_rust_ffi.py compares every declaration below with the header.
#![allow(dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct BlzMsm {
    _private: [u8; 0],
}
#[repr(C)]
pub struct BlzNtt {
    _private: [u8; 0],
}

pub const BLZ_COMM_ID_BYTES: usize = 128;

extern "C" {
    pub fn blz_last_error_message() -> *const c_char;
    pub fn blz_device_count() -> c_int;
    pub fn blz_point_size(curve: c_int) -> usize;
    pub fn blz_result_size(curve: c_int) -> usize;

    // ---- MSM: MSMClient (msm_api.rs:42-331)
    pub fn blz_msm_new(device_id: c_int, mem_type: c_int, is_precompute: c_int, curve: c_int, out: *mut *mut BlzMsm) -> c_int;
    pub fn blz_msm_free(h: *mut BlzMsm);
    pub fn blz_msm_loaded_binary_parameters(h: *mut BlzMsm, out: *mut u32) -> c_int;
    pub fn blz_msm_initialize(h: *mut BlzMsm, nof_elements: u32, has_hbm: c_int, hbm_addr: u64, hbm_off: u64) -> c_int;
    pub fn blz_msm_start_process(h: *mut BlzMsm) -> c_int;
    pub fn blz_msm_set_data(
        h: *mut BlzMsm,
        points: *const u8,
        points_len: usize,
        scalars: *const u8,
        scalars_len: usize,
        nof_elements: u32,
        has_hbm: c_int,
        hbm_addr: u64,
        hbm_off: u64,
    ) -> c_int;
    pub fn blz_msm_wait_result(h: *mut BlzMsm) -> c_int;
    pub fn blz_msm_result(h: *mut BlzMsm, out: *mut u8, out_cap: usize, out_len: *mut usize, label: *mut u32) -> c_int;
    pub fn blz_msm_load_data_to_hbm(h: *mut BlzMsm, points: *const u8, len: usize, addr: u64, off: u64) -> c_int;
    pub fn blz_msm_get_data_from_hbm(h: *mut BlzMsm, out: *mut u8, len: usize, addr: u64, off: u64) -> c_int;
    pub fn blz_msm_task_label(h: *mut BlzMsm, out: *mut u32) -> c_int;
    pub fn blz_msm_nof_elements(h: *mut BlzMsm, out: *mut u32) -> c_int;
    pub fn blz_msm_is_engine_ready(h: *mut BlzMsm, out: *mut u32) -> c_int;
    pub fn blz_msm_stream_progress(h: *mut BlzMsm, out: *mut u32) -> c_int;
    pub fn blz_msm_reset(h: *mut BlzMsm) -> c_int;
    pub fn blz_msm_memory_info(h: *mut BlzMsm, out: *mut u64) -> c_int;
    pub fn blz_msm_set_window_table(h: *mut BlzMsm, enable: c_int) -> c_int;
    pub fn blz_msm_prepare_window_table(h: *mut BlzMsm, nof_elements: u32, hbm_addr: u64, hbm_off: u64, wait_ms: c_int, ready: *mut c_int) -> c_int;
    pub fn blz_msm_set_precompute_plan(h: *mut BlzMsm, enable: c_int) -> c_int;
    pub fn blz_msm_prepare_precompute_plan(h: *mut BlzMsm, nof_elements: u32, hbm_addr: u64, hbm_off: u64, consistent: *mut c_int) -> c_int;
    pub fn blz_msm_precompute_plan_info(h: *mut BlzMsm, out: *mut u64) -> c_int;
    pub fn blz_msm_set_scalar_range(h: *mut BlzMsm, bit_lo: u32, bit_hi: u32) -> c_int;
    pub fn blz_msm_shard_layout(curve: c_int, nof_elements: u32, nranks: c_int, rank: c_int, out: *mut u32) -> c_int;
    pub fn blz_msm_shard_layout_ex(curve: c_int, nof_elements: u32, nranks: c_int, rank: c_int, flags: u32, out: *mut u32) -> c_int;
    pub fn blz_msm_shard_layout_candidate(curve: c_int, nof_elements: u32, nranks: c_int, rank: c_int, flags: u32, r: c_int, out: *mut u32) -> c_int;
    pub fn blz_msm_window_table_info(h: *mut BlzMsm, out: *mut u64) -> c_int;
    pub fn blz_msm_last_timings(h: *mut BlzMsm, out: *mut f32) -> c_int;
    pub fn blz_msm_last_sort_hidden(h: *mut BlzMsm, out: *mut c_int) -> c_int;
    // multi-GPU exchange (no reference counterpart: README.md:20-22 leaves it to a "management layer")
    pub fn blz_msm_combine_partials(h: *mut BlzMsm, partials: *const u8, count: usize, out: *mut u8, out_cap: usize) -> c_int;
    pub fn blz_comm_unique_id(out: *mut u8) -> c_int;
    pub fn blz_msm_comm_init(h: *mut BlzMsm, rank: c_int, nranks: c_int, id: *const u8) -> c_int;
    pub fn blz_msm_all_gather_combine(h: *mut BlzMsm, partial: *const u8, out: *mut u8, out_cap: usize) -> c_int;
    pub fn blz_msm_comm_init_all(handles: *const *mut BlzMsm, n: c_int) -> c_int;
    pub fn blz_msm_all_gather_combine_all(handles: *const *mut BlzMsm, n: c_int, partials: *const u8, out: *mut u8, out_cap: usize) -> c_int;
    pub fn blz_msm_comm_free(h: *mut BlzMsm) -> c_int;
    // device arena
    pub fn blz_arena_release(device_id: c_int) -> c_int;
    pub fn blz_arena_export(device_id: c_int, registry_path: *const c_char) -> c_int;
    pub fn blz_arena_attach(device_id: c_int, registry_path: *const c_char) -> c_int;
    pub fn blz_msm_precompute_bases_device(device_id: c_int, curve: c_int, d_points_in: *const c_void, d_bases_out: *mut c_void, n: u64) -> c_int;

    // ---- NTT: NTTClient (ntt_api.rs:25-125)
    pub fn blz_ntt_new(device_id: c_int, log_size: c_int, out: *mut *mut BlzNtt) -> c_int;
    pub fn blz_ntt_new_ex(device_id: c_int, log_size: c_int, inverse: c_int, out: *mut *mut BlzNtt) -> c_int;
    pub fn blz_ntt_new_field(device_id: c_int, field: c_int, log_size: c_int, inverse: c_int, out: *mut *mut BlzNtt) -> c_int;
    pub fn blz_arena_set_policy(device_id: c_int, policy: u32) -> c_int;
    pub fn blz_host_malloc(device_id: c_int, bytes: usize, out: *mut *mut std::os::raw::c_void) -> c_int;
    pub fn blz_host_free(p: *mut std::os::raw::c_void) -> c_int;
    pub fn blz_ntt_new_ex2(device_id: c_int, field: c_int, log_size: c_int, inverse: c_int, flags: u32, out: *mut *mut BlzNtt) -> c_int;
    pub fn blz_ntt_new_ex3(device_id: c_int, field: c_int, log_size: c_int, flags: u32, root: *const u8, out: *mut *mut BlzNtt) -> c_int;
    pub fn blz_ntt_info(h: *mut BlzNtt, out: *mut u64) -> c_int;
    pub fn blz_ntt_exchange(h: *mut BlzNtt, buf: usize, next_in: *const u8, in_len: usize, prev_out: *mut u8, out_cap: usize) -> c_int;
    pub fn blz_ntt_free(h: *mut BlzNtt);
    pub fn blz_ntt_initialize(h: *mut BlzNtt) -> c_int;
    pub fn blz_ntt_set_data(h: *mut BlzNtt, buf_host: usize, data: *const u8, len: usize) -> c_int;
    pub fn blz_ntt_start_process(h: *mut BlzNtt, buf_kernel: usize) -> c_int;
    pub fn blz_ntt_wait_result(h: *mut BlzNtt) -> c_int;
    pub fn blz_ntt_result(h: *mut BlzNtt, buf: usize, out: *mut u8, out_cap: usize) -> c_int;
    pub fn blz_ntt_reset(h: *mut BlzNtt) -> c_int;
    pub fn blz_ntt_last_kernel_ms(h: *mut BlzNtt, out: *mut f32) -> c_int;
}
