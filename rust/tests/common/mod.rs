//! Fixture access for the integration tests: the committed vectors of ../tests/golden (minted by the Python
//! oracle, tests/golden/make_golden.py) stand in for the arkworks generators of the reference's tests/msm/mod.rs.
#![allow(dead_code)]
use ingo_blaze::ingo_msm::Curve;
use std::path::PathBuf;

pub struct MsmVector {
    pub name: String,
    pub curve: Curve,
    pub pf: u32,
    pub n: u32,
    pub points: Vec<u8>,
    pub scalars: Vec<u8>,
    pub result: Vec<u8>,
}

pub fn unhex(s: &str) -> Vec<u8> {
    (0..s.len() / 2).map(|i| u8::from_str_radix(&s[2 * i..2 * i + 2], 16).expect("hex digit")).collect()
}

fn golden_dir() -> PathBuf {
    PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("..").join("tests").join("golden")
}

pub fn msm_vectors() -> Vec<MsmVector> {
    let text = std::fs::read_to_string(golden_dir().join("msm_vectors.json")).expect("msm_vectors.json");
    let v: serde_json::Value = serde_json::from_str(&text).expect("json");
    v.as_array()
        .unwrap()
        .iter()
        .map(|e| MsmVector {
            name: e["name"].as_str().unwrap().to_string(),
            curve: match e["curve"].as_str().unwrap() {
                "BLS377" => Curve::BLS377,
                "BLS381" => Curve::BLS381,
                _ => Curve::BN254,
            },
            pf: e["pf"].as_u64().unwrap() as u32,
            n: e["n"].as_u64().unwrap() as u32,
            points: unhex(e["points"].as_str().unwrap()),
            scalars: unhex(e["scalars"].as_str().unwrap()),
            result: unhex(e["result"].as_str().unwrap()),
        })
        .collect()
}

/// (logn, input, output) of ../tests/golden/ntt_vectors.json
pub fn ntt_vectors() -> Vec<(i32, Vec<u8>, Vec<u8>)> {
    let text = std::fs::read_to_string(golden_dir().join("ntt_vectors.json")).expect("ntt_vectors.json");
    let v: serde_json::Value = serde_json::from_str(&text).expect("json");
    v.as_array()
        .unwrap()
        .iter()
        .map(|e| (e["logn"].as_i64().unwrap() as i32, unhex(e["input"].as_str().unwrap()), unhex(e["output"].as_str().unwrap())))
        .collect()
}
