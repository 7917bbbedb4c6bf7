// Reads like tests/integration_msm.rs:149-207 (msm_bls12_381_test) in the C++ mirror.  Compiled (and,
// on the GPU box, run) by tests/test_abi.py / tests/test_gpu_msm.py::test_cpp_host_mirror.
#include <cstdio>
#include <fstream>
#include <iterator>
#include "blaze.hpp"
using namespace ingo_blaze;
static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}
// mode "plan": a precompute client (PRECOMPUTE_FACTOR = 8) over a RESIDENT table on the checked-table plan
// (tests/integration_msm_hbm.rs flow; points.bin holds 8 n bases), then the device-memory figures
static int run_plan(const std::vector<uint8_t>& points, const std::vector<uint8_t>& scalars, uint32_t n, const char* out_path) {
    DriverClient dclient(0, DriverConfig::driver_client_cfg(CardType::MI355X));
    MSMClient driver(MSMInit{PointMemoryType::HBM, true, Curve::BN254}, dclient);
    driver.set_precompute_plan(true);
    driver.load_data_to_hbm(points, 0, 0);
    const bool ok = driver.prepare_precompute_plan(n, {0, 0});
    MSMParams params{n, std::make_pair<uint64_t, uint64_t>(0, 0)};
    driver.initialize(params);
    driver.start_process();
    driver.set_data(MSMInput{std::nullopt, scalars, params});
    driver.wait_result();
    MSMResult r = *driver.result();
    const auto info = driver.precompute_plan_info();
    const auto mem = driver.memory_info();
    std::ofstream(out_path, std::ios::binary).write((const char*)r.result.data(), r.result.size());
    std::printf("plan consistent %d used %llu check_state %llu raw %llu montgomery %llu\n", ok ? 1 : 0, (unsigned long long)info[0],
                (unsigned long long)info[1], (unsigned long long)mem[2], (unsigned long long)mem[3]);
    return 0;
}
// mode "ntt": two cycles of the reference's double-buffered loop (tests/integration_ntt.rs:102-136) with result + set_data
// fused into exchange(); points.bin = the input vector, n = log_size; the transform of the input comes back in cycle 2
static int run_ntt(const std::vector<uint8_t>& input, int logn, const char* out_path) {
    DriverClient dclient(0, DriverConfig::driver_client_cfg(CardType::MI355X));
    NTTClient driver(NTT::Ntt, dclient, logn);
    driver.initialize(NttInit{});
    std::vector<uint8_t> res;
    for (int i = 0; i < 3; ++i) {
        const size_t buf_host = i % 2, buf_kernel = 1 - buf_host;
        driver.start_process(buf_kernel);
        driver.exchange(buf_host, input, res);
        driver.wait_result();
    }
    const auto info = driver.info();
    std::ofstream(out_path, std::ios::binary).write((const char*)res.data(), res.size());
    std::printf("ntt log_size %llu device_bytes %llu\n", (unsigned long long)info[3], (unsigned long long)info[0]);
    return 0;
}
// mode "stream": the same task fed by several set_data calls (blaze_hip.h "STREAMED TASKS": the reference's chunk loop,
// msm_api.rs:155-202, in the caller's hands): slices of 1, n / 3 and the rest; a half-fed task refuses wait_result
static int run_stream(const std::vector<uint8_t>& points, const std::vector<uint8_t>& scalars, uint32_t n, const char* out_path) {
    DriverClient dclient(0, DriverConfig::driver_client_cfg(CardType::MI355X));
    MSMClient driver(MSMInit{PointMemoryType::DMA, false, Curve::BLS381}, dclient);
    driver.initialize(MSMParams{n, std::nullopt});
    driver.start_process();
    const size_t ps = points.size() / n;
    const uint32_t cuts[4] = {0, 1, n / 3 > 1 ? n / 3 : 2, n};
    int refused = 0;
    for (int i = 0; i < 3; ++i) {
        const uint32_t a = cuts[i], b = cuts[i + 1];
        MSMInput in{std::vector<uint8_t>(points.begin() + a * ps, points.begin() + b * ps),
                    std::vector<uint8_t>(scalars.begin() + a * 32, scalars.begin() + b * 32), MSMParams{b - a, std::nullopt}};
        driver.set_data(in);
        if (b < n) {
            try { driver.wait_result(); } catch (const DriverClientError&) { ++refused; }
            if (driver.stream_progress() != std::make_pair(b, n)) return 3;
        }
    }
    driver.wait_result();
    MSMResult r = *driver.result();
    std::ofstream(out_path, std::ios::binary).write((const char*)r.result.data(), r.result.size());
    std::printf("streamed label %u bytes %zu refused %d\n", r.result_label, r.result.size(), refused);
    return 0;
}
int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s points.bin scalars.bin n out.bin [plan|ntt|stream]\n", argv[0]); return 2; }
    try {
        auto points = slurp(argv[1]);
        auto scalars = slurp(argv[2]);
        uint32_t msm_size = (uint32_t)std::stoul(argv[3]);
        if (argc > 5 && std::string(argv[5]) == "plan") return run_plan(points, scalars, msm_size, argv[4]);
        if (argc > 5 && std::string(argv[5]) == "ntt") return run_ntt(points, (int)msm_size, argv[4]);
        if (argc > 5 && std::string(argv[5]) == "stream") return run_stream(points, scalars, msm_size, argv[4]);
        DriverClient dclient(0, DriverConfig::driver_client_cfg(CardType::MI355X));
        MSMClient driver(MSMInit{PointMemoryType::DMA, false, Curve::BLS381}, dclient);
        MSMParams params{msm_size, std::nullopt};
        driver.initialize(params);
        driver.start_process();
        driver.set_data(MSMInput{points, scalars, params});
        driver.wait_result();
        MSMResult r = *driver.result();
        std::ofstream(argv[4], std::ios::binary).write((const char*)r.result.data(), r.result.size());
        std::printf("label %u bytes %zu\n", r.result_label, r.result.size());
    } catch (const DriverClientError& e) {
        std::fprintf(stderr, "DriverClientError kind %d: %s\n", (int)e.kind, e.what());
        return 1;
    }
    return 0;
}
