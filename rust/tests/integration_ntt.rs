//! `/root/reference/tests/integration_ntt.rs:62-146` (`ntt_parallel_test_correctness`): the double-buffer loop,
//! call for call, at the sizes of the committed golden vectors.
mod common;

use ingo_blaze::{driver_client::*, ingo_ntt::*};
use std::env;

#[test]
fn ntt_parallel_test_correctness() {
    let id = env::var("ID").unwrap_or_else(|_| 0.to_string());
    for (logn, input, output) in common::ntt_vectors() {
        let nof_vectors: usize = 3;
        let in_vecs: Vec<Vec<u8>> = (0..nof_vectors).map(|_| input.clone()).collect();
        let dclient = DriverClient::new(&id, DriverConfig::driver_client_cfg(CardType::C1100));
        let driver = NTTClient::with_log_size(dclient, logn);
        driver.initialize(NttInit {}).unwrap();

        let mut outputs: Vec<Vec<u8>> = Vec::new();
        for i in 0..(nof_vectors + 2) {
            let buf_host = i % 2;
            let buf_kernel = 1 - buf_host;
            driver.start_process(Some(buf_kernel)).unwrap();
            let res = driver.result(Some(buf_host)).unwrap().unwrap();
            if i >= 2 {
                outputs.push(res)
            }
            let host_wr_idx_adj = if i > nof_vectors - 1 { nof_vectors - 1 } else { i };
            driver.set_data(NTTInput { buf_host, data: in_vecs[host_wr_idx_adj].clone() }).unwrap();
            driver.wait_result().unwrap();
        }
        for out_vec in outputs.into_iter() {
            assert_eq!(out_vec, output, "log size {}", logn);
        }
    }
}

/// The same loop with `result` + `set_data` of a cycle fused into `exchange` (both directions of the link at once).
#[test]
fn ntt_parallel_test_correctness_with_exchange() {
    let id = env::var("ID").unwrap_or_else(|_| 0.to_string());
    for (logn, input, output) in common::ntt_vectors() {
        let nof_vectors: usize = 3;
        let dclient = DriverClient::new(&id, DriverConfig::driver_client_cfg(CardType::C1100));
        let driver = NTTClient::with_log_size(dclient, logn);
        driver.initialize(NttInit {}).unwrap();
        let mut res = vec![0u8; input.len()];
        for i in 0..(nof_vectors + 2) {
            let buf_host = i % 2;
            driver.start_process(Some(1 - buf_host)).unwrap();
            driver.exchange(buf_host, &input, &mut res).unwrap();
            if i >= 2 {
                assert_eq!(res, output, "log size {}", logn);
            }
            driver.wait_result().unwrap();
        }
    }
}
