//! `/root/reference/benches/ntt_bench.rs:7-47`: initialize + start_process + wait_result (+ reset) on a resident
//! buffer.  FNAME names a file of 2^27 x 32 bytes; without it a zero vector is transformed (the timing does not
//! depend on the values).
use criterion::*;
use ingo_blaze::{driver_client::*, ingo_ntt::*};
use std::{env, fs::File, io::Read};

fn bench_ntt_calc(c: &mut Criterion) {
    let _ = env_logger::try_init();
    let id = env::var("ID").unwrap_or_else(|_| 0.to_string());
    let mut in_vec: Vec<u8> = Default::default();
    match env::var("FNAME") {
        Ok(fname) => {
            let mut f = File::open(fname).expect("no file found");
            let _ = f.read_to_end(&mut in_vec);
        }
        Err(_) => in_vec = vec![0u8; NTT_WORD_SIZE << NTT_LOG_SIZE],
    }
    let buf_host = 0;
    let buf_kernel = 0;
    let dclient = DriverClient::new(&id, DriverConfig::driver_client_cfg(CardType::C1100));
    let driver = NTTClient::new(NTT::Ntt, dclient);
    driver.set_data(NTTInput { buf_host, data: in_vec }).unwrap();
    let _ = driver.driver_client.initialize_cms();
    let _ = driver.driver_client.reset_sensor_data();

    let mut group = c.benchmark_group("NTT computation");
    group.bench_function("NTT", |b| {
        b.iter(|| {
            let _ = driver.initialize(NttInit {});
            let _ = driver.start_process(Some(buf_kernel));
            let _ = driver.wait_result();
            let _ = driver.driver_client.reset();
        })
    });
    group.finish();
    let res = driver.result(Some(buf_kernel)).unwrap();
    log::info!("NTT result: {:?} bytes, last kernel {:?} ms", res.unwrap().len(), driver.last_kernel_ms());
}

criterion_group! {
    name = benches;
    config = Criterion::default().sample_size(10);
    targets = bench_ntt_calc
}
criterion_main!(benches);
