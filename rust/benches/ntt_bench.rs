//! 2^27 NTT latency on a resident buffer, the measurement `/root/reference/benches/ntt_bench.rs:7-47` makes:
//! per sample one `initialize` + `start_process` + `wait_result` (the reference's 100 ms `reset()` sleep has no
//! counterpart here).  The input comes from the file named by FNAME (2^27 x 32 bytes) or, without it, is a
//! counter pattern - the kernels' timing does not depend on the values.  Besides criterion's wall-clock figure
//! the device-side time of the last transform (HIP events on the kernel's stream) is logged.
use criterion::{criterion_group, criterion_main, Criterion};
use ingo_blaze::driver_client::{CardType, DriverClient, DriverConfig, DriverPrimitive};
use ingo_blaze::ingo_ntt::{NTTClient, NTTInput, NttInit, NTT, NTT_LOG_SIZE, NTT_WORD_SIZE};

const RESIDENT_BUFFER: usize = 0;

fn input_vector() -> Vec<u8> {
    if let Ok(path) = std::env::var("FNAME") {
        return std::fs::read(&path).unwrap_or_else(|e| panic!("cannot read {path}: {e}"));
    }
    let words = 1usize << NTT_LOG_SIZE;
    let mut v = vec![0u8; words * NTT_WORD_SIZE];
    for (i, w) in v.chunks_exact_mut(NTT_WORD_SIZE).enumerate() {
        w[..8].copy_from_slice(&(i as u64).to_le_bytes());   // canonical: far below r
    }
    v
}

fn ntt_latency(c: &mut Criterion) {
    let _ = env_logger::try_init();
    let card = std::env::var("ID").unwrap_or_else(|_| "0".into());
    let client = NTTClient::new(NTT::Ntt, DriverClient::new(&card, DriverConfig::driver_client_cfg(CardType::C1100)));
    client
        .set_data(NTTInput { buf_host: RESIDENT_BUFFER, data: input_vector() })
        .expect("set_data");

    c.benchmark_group("NTT computation").bench_function("NTT", |b| {
        b.iter(|| {
            client.initialize(NttInit {}).expect("initialize");
            client.start_process(Some(RESIDENT_BUFFER)).expect("start_process");
            client.wait_result().expect("wait_result");
        })
    });

    let out = client.result(Some(RESIDENT_BUFFER)).expect("result").expect("a transform has run");
    log::info!("transformed {} bytes; device time of the last transform: {:?} ms", out.len(), client.last_kernel_ms());
}

criterion_group! {
    name = benches;
    config = Criterion::default().sample_size(10);
    targets = ntt_latency
}
criterion_main!(benches);
