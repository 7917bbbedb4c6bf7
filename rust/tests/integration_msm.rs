//! The call sequence of the reference's `msm_bls12_381_test` (`/root/reference/tests/integration_msm.rs:149-207`)
//! - new, reset, loaded_binary_parameters, is_msm_engine_ready, task_label, initialize, start_process, set_data,
//! wait_result, result - on every committed golden vector (three curves, precompute factor 1 and 8).
//! Needs an MI355X and libblaze_hip.so (`make -C ../blaze_amd/csrc`).
mod common;

use ingo_blaze::{driver_client::*, ingo_msm::*};
use std::env;

fn run_vector(id: &str, v: &common::MsmVector) -> MSMResult {
    let dclient = DriverClient::new(id, DriverConfig::driver_client_cfg(CardType::C1100));
    let driver = MSMClient::new(
        MSMInit { mem_type: PointMemoryType::DMA, is_precompute: v.pf == PRECOMPUTE_FACTOR, curve: v.curve },
        dclient,
    );
    driver.driver_client.reset().unwrap();

    let params = driver.loaded_binary_parameters();
    let params_parce = MSMImageParametrs::parse_image_params(params[1]);
    params_parce.debug_information();
    assert_eq!(params_parce.hif2cpu_c_is_stub, 0);
    driver.is_msm_engine_ready().unwrap();
    let label_before = driver.task_label().unwrap();
    driver.driver_client.firewalls_status();

    let msm_params = MSMParams { nof_elements: v.n, hbm_point_addr: None };
    driver.initialize(msm_params).unwrap();
    driver.start_process(None).unwrap();
    assert_eq!(driver.task_label().unwrap(), label_before + 1);
    assert_eq!(driver.nof_elements().unwrap(), v.n);

    driver
        .set_data(MSMInput { points: Some(v.points.clone()), scalars: v.scalars.clone(), params: msm_params })
        .unwrap();
    driver.wait_result().unwrap();
    let mres = driver.result(None).unwrap().unwrap();
    assert_eq!(mres.result_label, label_before + 1);
    mres
}

#[test]
fn msm_golden_vectors_all_curves() {
    let id = env::var("ID").unwrap_or_else(|_| 0.to_string());
    for v in common::msm_vectors() {
        let mres = run_vector(&id, &v);
        assert_eq!(mres.result, v.result, "{:?} pf={} {}", v.curve, v.pf, v.name);
    }
}

#[test]
fn msm_task_queue_two_in_flight() {
    // the card has a task queue and a result queue (msm_hw_code.rs:19-25): a second task may be submitted before
    // the first result is collected; results pop in submission order with consecutive labels
    let id = env::var("ID").unwrap_or_else(|_| 0.to_string());
    let vs: Vec<_> = common::msm_vectors().into_iter().filter(|v| v.curve == Curve::BLS381 && v.pf == 1).take(2).collect();
    assert_eq!(vs.len(), 2);
    let driver = MSMClient::new(
        MSMInit { mem_type: PointMemoryType::DMA, is_precompute: false, curve: Curve::BLS381 },
        DriverClient::new(&id, DriverConfig::driver_client_cfg(CardType::C1100)),
    );
    for v in &vs {
        let p = MSMParams { nof_elements: v.n, hbm_point_addr: None };
        driver.initialize(p).unwrap();
        driver.start_process(None).unwrap();
        driver.set_data(MSMInput { points: Some(v.points.clone()), scalars: v.scalars.clone(), params: p }).unwrap();
    }
    for (i, v) in vs.iter().enumerate() {
        driver.wait_result().unwrap();
        let r = driver.result(None).unwrap().unwrap();
        assert_eq!(r.result, v.result);
        assert_eq!(r.result_label, 1 + i as u32);
    }
    // nothing armed: the reference would spin forever, this build reports it
    assert!(driver.wait_result().is_err());
}

#[test]
fn msm_task_fed_by_several_set_data_calls() {
    // The reference's set_data walks its input in 2048-element chunks into the card's FIFOs and the card counts elements against
    // NUMBER_OF_MSM_ELEMENTS (/root/reference/src/ingo_msm/msm_api.rs:155-202, msm_hw_code.rs:18-19): a queued task may be fed by
    // any number of set_data calls; it is complete when the counts add up.
    let id = env::var("ID").unwrap_or_else(|_| 0.to_string());
    for v in common::msm_vectors().into_iter().filter(|v| v.n >= 3) {
        let driver = MSMClient::new(
            MSMInit { mem_type: PointMemoryType::DMA, is_precompute: v.pf == PRECOMPUTE_FACTOR, curve: v.curve },
            DriverClient::new(&id, DriverConfig::driver_client_cfg(CardType::C1100)),
        );
        let per_elem = v.points.len() / v.n as usize;
        driver.initialize(MSMParams { nof_elements: v.n, hbm_point_addr: None }).unwrap();
        driver.start_process(None).unwrap();
        let cuts = [0u32, 1, v.n / 2, v.n];
        for w in cuts.windows(2) {
            let (a, b) = (w[0] as usize, w[1] as usize);
            assert_eq!(driver.stream_progress().unwrap(), (if a == 0 { 0 } else { a as u32 }, v.n));
            driver
                .set_data(MSMInput {
                    points: Some(v.points[a * per_elem..b * per_elem].to_vec()),
                    scalars: v.scalars[a * 32..b * 32].to_vec(),
                    params: MSMParams { nof_elements: (b - a) as u32, hbm_point_addr: None },
                })
                .unwrap();
            if b < v.n as usize {
                // a half-fed task: start_process and wait_result are refused, over-feeding too
                assert!(driver.start_process(None).is_err());
                assert!(driver.wait_result().is_err());
            }
        }
        assert_eq!(driver.stream_progress().unwrap(), (0, 0));
        driver.wait_result().unwrap();
        assert_eq!(driver.result(None).unwrap().unwrap().result, v.result, "{:?} pf={} {}", v.curve, v.pf, v.name);
    }
}
