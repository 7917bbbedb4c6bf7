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
int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s points.bin scalars.bin n out.bin\n", argv[0]); return 2; }
    try {
        auto points = slurp(argv[1]);
        auto scalars = slurp(argv[2]);
        uint32_t msm_size = (uint32_t)std::stoul(argv[3]);
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
