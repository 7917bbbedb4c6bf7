//! `/root/reference/tests/integration_msm_hbm.rs:12-118`: bases resident in device memory
//! (`load_data_to_hbm`), `initialize` with `hbm_point_addr`, scalars-only `set_data`.
mod common;

use ingo_blaze::{driver_client::*, ingo_msm::*};
use std::env;

#[test]
fn hbm_msm_bls12_381_scalars_only() {
    let id = env::var("ID").unwrap_or_else(|_| 0.to_string());
    let v = common::msm_vectors()
        .into_iter()
        .filter(|v| v.curve == Curve::BLS381 && v.pf == 1)
        .max_by_key(|v| v.n)
        .unwrap();
    // the reference builds this client with mem_type DMA and still selects HBM bases through hbm_point_addr
    // (integration_msm_hbm.rs:41, msm_api.rs:82)
    let driver = MSMClient::new(
        MSMInit { mem_type: PointMemoryType::DMA, is_precompute: false, curve: Curve::BLS381 },
        DriverClient::new(&id, DriverConfig::driver_client_cfg(CardType::C1100)),
    );
    let (addr, offset) = (0x0u64, 0x0u64);
    driver.load_data_to_hbm(&v.points, addr, offset).unwrap();
    assert_eq!(driver.get_data_from_hbm(v.points.len(), addr, offset).unwrap(), v.points);

    let msm_params = MSMParams { nof_elements: v.n, hbm_point_addr: Some((addr, offset)) };
    driver.initialize(msm_params).unwrap();
    driver.start_process(None).unwrap();
    driver.set_data(MSMInput { points: None, scalars: v.scalars.clone(), params: msm_params }).unwrap();
    driver.wait_result().unwrap();
    assert_eq!(driver.result(None).unwrap().unwrap().result, v.result);

    // the bases persist for the next client on the same device
    let second = MSMClient::new(
        MSMInit { mem_type: PointMemoryType::HBM, is_precompute: false, curve: Curve::BLS381 },
        DriverClient::new(&id, DriverConfig::driver_client_cfg(CardType::C1100)),
    );
    second.initialize(msm_params).unwrap();
    second.start_process(None).unwrap();
    second.set_data(MSMInput { points: None, scalars: v.scalars.clone(), params: msm_params }).unwrap();
    second.wait_result().unwrap();
    assert_eq!(second.result(None).unwrap().unwrap().result, v.result);
}

/// A precompute client (`PRECOMPUTE_FACTOR` = 8) over a resident table on the checked-table plan: same bytes as the exact path.
#[test]
fn hbm_msm_precompute_checked_table_plan() {
    let id = env::var("ID").unwrap_or_else(|_| 0.to_string());
    for v in common::msm_vectors().into_iter().filter(|v| v.pf == 8) {
        let driver = MSMClient::new(
            MSMInit { mem_type: PointMemoryType::HBM, is_precompute: true, curve: v.curve },
            DriverClient::new(&id, DriverConfig::driver_client_cfg(CardType::C1100)),
        );
        driver.set_precompute_plan(true).unwrap();
        driver.load_data_to_hbm(&v.points, 0, 0).unwrap();
        assert!(driver.prepare_precompute_plan(v.n, (0, 0)).unwrap());
        let msm_params = MSMParams { nof_elements: v.n, hbm_point_addr: Some((0, 0)) };
        driver.initialize(msm_params).unwrap();
        driver.start_process(None).unwrap();
        driver.set_data(MSMInput { points: None, scalars: v.scalars.clone(), params: msm_params }).unwrap();
        driver.wait_result().unwrap();
        assert_eq!(driver.result(None).unwrap().unwrap().result, v.result);
        assert_eq!(driver.precompute_plan_info().unwrap()[0], 1);
    }
}
