// C++17 host-side mirror of the reference's operator surface above the C ABI (include/blaze_hip.h).
// Same names, argument meaning and call order as the Rust crate:
//   src/driver_client/dclient.rs:28-46   trait DriverPrimitive<T,P,I,O>
//   src/ingo_msm/msm_api.rs:8-331        MSMClient, MSMInit, MSMParams, MSMInput, MSMResult
//   src/ingo_ntt/ntt_api.rs:8-125        NTTClient, NTT, NttInit, NTTInput
//   src/error.rs:6-32                    DriverClientError
// Header-only; link with -lblaze_hip.  Errors are thrown as DriverClientError (the reference returns
// Result<_, DriverClientError>; its panics on bad mode combinations become InvalidPrimitiveParam).
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <array>
#include <string>
#include <utility>
#include <vector>

#include "blaze_hip.h"

namespace ingo_blaze {

struct DriverClientError : std::runtime_error {
    enum Kind { WriteError = 1, ReadError, HBICAPNotReady, InvalidPrimitiveParam, CsvError, LoadFailed, FileError, Unknown };
    Kind kind;
    DriverClientError(int code, const std::string& msg)
        : std::runtime_error(msg), kind(code >= 1 && code <= 8 ? static_cast<Kind>(code) : Unknown) {}
};
inline void check(int rc) {
    if (rc != BLZ_OK) throw DriverClientError(rc, blz_last_error_message());
}

enum class CardType { C1100, MI355X };  // dclient_cfg.rs:1-3 (+ this build's card)
struct DriverConfig {                   // dclient_cfg.rs:9-31: AXI base addresses have no GPU meaning
    CardType card = CardType::MI355X;
    static DriverConfig driver_client_cfg(CardType c) { return DriverConfig{c}; }
};

// dclient.rs:50-93: `id` is the slot id of /dev/xdma{id}_*; here the HIP device ordinal.
class DriverClient {
   public:
    int id;
    DriverConfig cfg;
    DriverClient(int id_, DriverConfig cfg_ = {}) : id(id_), cfg(cfg_) {
        if (id < 0 || id >= blz_device_count()) throw DriverClientError(BLZ_ERR_FILE, "no HIP device with this ordinal");
    }
    void reset() const {}  // DFX decouple toggle + sleep(100 ms) on the card (dclient.rs:88-93): nothing to do
};

// dclient.rs:28-46
template <class T, class P, class I, class O>
class DriverPrimitive {
   public:
    virtual ~DriverPrimitive() = default;
    virtual std::vector<uint32_t> loaded_binary_parameters() const = 0;
    virtual void initialize(const P& param) = 0;
    virtual void set_data(const I& input) = 0;
    virtual void start_process(std::optional<size_t> param = std::nullopt) = 0;
    virtual void wait_result() = 0;
    virtual std::optional<O> result(std::optional<size_t> param = std::nullopt) = 0;
};

// ---------------------------------------------------------------- MSM (src/ingo_msm)
enum class Curve { BLS377 = 0, BLS381 = 1, BN254 = 2 };   // msm_cfg.rs:4-8
enum class PointMemoryType { HBM = 0, DMA = 1 };          // msm_cfg.rs:11-14
constexpr uint32_t PRECOMPUTE_FACTOR_BASE = 1, PRECOMPUTE_FACTOR = 8;  // msm_api.rs:39-40

struct MSMInit { PointMemoryType mem_type; bool is_precompute; Curve curve; };                       // msm_api.rs:16-20
struct MSMParams { uint32_t nof_elements; std::optional<std::pair<uint64_t, uint64_t>> hbm_point_addr; };  // :23-26
struct MSMInput { std::optional<std::vector<uint8_t>> points; std::vector<uint8_t> scalars; MSMParams params; };  // :28-32
struct MSMResult { std::vector<uint8_t> result; uint32_t result_label; };                           // :33-37

class MSMClient : public DriverPrimitive<MSMInit, MSMParams, MSMInput, MSMResult> {
    blz_msm* h_ = nullptr;
    size_t result_size_;

   public:
    DriverClient driver_client;
    MSMClient(const MSMInit& init, DriverClient dclient) : driver_client(dclient) {  // msm_api.rs:44-55
        check(blz_msm_new(dclient.id, static_cast<int>(init.mem_type), init.is_precompute, static_cast<int>(init.curve), &h_));
        result_size_ = blz_result_size(static_cast<int>(init.curve));
    }
    ~MSMClient() override { blz_msm_free(h_); }
    MSMClient(const MSMClient&) = delete;
    MSMClient& operator=(const MSMClient&) = delete;

    std::vector<uint32_t> loaded_binary_parameters() const override {  // msm_api.rs:57-70
        uint32_t v[2];
        check(blz_msm_loaded_binary_parameters(h_, v));
        return {v[0], v[1]};
    }
    void initialize(const MSMParams& p) override {  // msm_api.rs:72-111
        auto a = p.hbm_point_addr.value_or(std::make_pair<uint64_t, uint64_t>(0, 0));
        check(blz_msm_initialize(h_, p.nof_elements, p.hbm_point_addr.has_value(), a.first, a.second));
    }
    void start_process(std::optional<size_t> = std::nullopt) override { check(blz_msm_start_process(h_)); }  // :113-120
    void set_data(const MSMInput& d) override {  // msm_api.rs:155-220
        auto a = d.params.hbm_point_addr.value_or(std::make_pair<uint64_t, uint64_t>(0, 0));
        check(blz_msm_set_data(h_, d.points ? d.points->data() : nullptr, d.points ? d.points->size() : 0, d.scalars.data(),
                               d.scalars.size(), d.params.nof_elements, d.params.hbm_point_addr.has_value(), a.first, a.second));
    }
    void wait_result() override { check(blz_msm_wait_result(h_)); }  // msm_api.rs:222-238
    std::optional<MSMResult> result(std::optional<size_t> = std::nullopt) override {  // msm_api.rs:240-274
        MSMResult r;
        r.result.resize(result_size_);
        size_t n = 0;
        check(blz_msm_result(h_, r.result.data(), r.result.size(), &n, &r.result_label));
        r.result.resize(n);
        return r;
    }
    // msm_api.rs:277-331
    uint32_t task_label() const { uint32_t v; check(blz_msm_task_label(h_, &v)); return v; }
    uint32_t nof_elements() const { uint32_t v; check(blz_msm_nof_elements(h_, &v)); return v; }
    uint32_t is_msm_engine_ready() const { uint32_t v; check(blz_msm_is_engine_ready(h_, &v)); return v; }
    // a task fed by several set_data calls (blaze_hip.h "STREAMED TASKS"): {elements received, elements of the queued task}
    std::pair<uint32_t, uint32_t> stream_progress() const { uint32_t v[2]; check(blz_msm_stream_progress(h_, v)); return {v[0], v[1]}; }
    void load_data_to_hbm(const std::vector<uint8_t>& points, uint64_t addr, uint64_t offset) {
        check(blz_msm_load_data_to_hbm(h_, points.data(), points.size(), addr, offset));
    }
    std::vector<uint8_t> get_data_from_hbm(size_t data_len, uint64_t addr, uint64_t offset) {
        std::vector<uint8_t> out(data_len);
        check(blz_msm_get_data_from_hbm(h_, out.data(), data_len, addr, offset));
        return out;
    }
    // multi-GPU (no reference counterpart: README.md:20-22 leaves it to a "management layer"): one communicator
    // rank per client; comm_unique_id() on rank 0, shipped to the others by the host
    static std::vector<uint8_t> comm_unique_id() {
        std::vector<uint8_t> id(BLZ_COMM_ID_BYTES);
        check(blz_comm_unique_id(id.data()));
        return id;
    }
    // resident-base window table (blaze_hip.h): opt-in, bases in the arena, precompute_factor 1
    void set_scalar_range(uint32_t bit_lo, uint32_t bit_hi) { check(blz_msm_set_scalar_range(h_, bit_lo, bit_hi)); }   // one shard of a job split by scalar chunk
    void set_window_table(int mode) { check(blz_msm_set_window_table(h_, mode)); }   // 0 off, 1 where it pays, 2 always
    // checked-table plan of a precompute client (blaze_hip.h blz_msm_set_precompute_plan): resident x8 tables are checked once per
    // load against precompute_base_* and, if consistent, served as 4n even bases with 64-bit chunks; identical result bytes
    void set_precompute_plan(bool enable) { check(blz_msm_set_precompute_plan(h_, enable ? 1 : 0)); }
    bool prepare_precompute_plan(uint32_t nof_elements, std::pair<uint64_t, uint64_t> hbm_addr = {0, 0}) {
        int ok = 0;
        check(blz_msm_prepare_precompute_plan(h_, nof_elements, hbm_addr.first, hbm_addr.second, &ok));
        return ok != 0;
    }
    // device bytes behind this client: {workspace, staging, arena raw, arena Montgomery copies, arena window tables, total}
    std::array<uint64_t, 6> memory_info() {
        std::array<uint64_t, 6> out{};
        check(blz_msm_memory_info(h_, out.data()));
        return out;
    }
    // {took the plan, check state (0 unchecked, 1 consistent, 2 refuted), check us, even-base copy bytes} of the last HBM task
    std::array<uint64_t, 4> precompute_plan_info() {
        std::array<uint64_t, 4> out{};
        check(blz_msm_precompute_plan_info(h_, out.data()));
        return out;
    }
    // enqueue the table's build for the bases at hbm_addr (it is paced by the tasks otherwise) and wait up to wait_ms for it
    // (0: not at all, < 0: the library's wait deadline); true: the table is in place
    bool prepare_window_table(uint32_t nof_elements, std::pair<uint64_t, uint64_t> hbm_addr = {0, 0}, int wait_ms = -1) {
        int ready = 0;
        check(blz_msm_prepare_window_table(h_, nof_elements, hbm_addr.first, hbm_addr.second, wait_ms, &ready));
        return ready != 0;
    }
    // {first element, count, bit_lo, bit_hi, ranges, compute us, link us, device MiB} of `rank`: the split priced with the flow's
    // transfers (flags: BLZ_SHARD_SCALARS_FROM_HOST | BLZ_SHARD_BASES_FROM_HOST)
    static std::array<uint32_t, 8> shard_layout_ex(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t flags) {
        std::array<uint32_t, 8> out{};
        check(blz_msm_shard_layout_ex(curve, nof_elements, nranks, rank, flags, out.data()));
        return out;
    }
    void comm_init(int rank, int nranks, const std::vector<uint8_t>& id) { check(blz_msm_comm_init(h_, rank, nranks, id.data())); }
    std::vector<uint8_t> all_gather_combine(const std::vector<uint8_t>& partial) {
        std::vector<uint8_t> out(result_size_);
        check(blz_msm_all_gather_combine(h_, partial.data(), out.data(), out.size()));
        return out;
    }
    // one thread driving one client per device: the group forms (the per-rank calls above are blocking rendezvous)
    static void comm_init_all(const std::vector<MSMClient*>& clients) {
        std::vector<blz_msm*> hs;
        for (MSMClient* c : clients) hs.push_back(c->h_);
        check(blz_msm_comm_init_all(hs.data(), (int)hs.size()));
    }
    static std::vector<std::vector<uint8_t>> all_gather_combine_all(const std::vector<MSMClient*>& clients,
                                                                    const std::vector<std::vector<uint8_t>>& partials) {
        std::vector<blz_msm*> hs;
        std::vector<uint8_t> flat;
        for (MSMClient* c : clients) hs.push_back(c->h_);
        for (const auto& p : partials) flat.insert(flat.end(), p.begin(), p.end());
        const size_t rs = clients.at(0)->result_size_;
        if (partials.size() != clients.size() || flat.size() != rs * clients.size()) throw DriverClientError(BLZ_ERR_INVALID_PARAM, "partials");
        std::vector<uint8_t> out(rs * clients.size());
        check(blz_msm_all_gather_combine_all(hs.data(), (int)hs.size(), flat.data(), out.data(), out.size()));
        std::vector<std::vector<uint8_t>> res;
        for (size_t i = 0; i < clients.size(); ++i) res.emplace_back(out.begin() + i * rs, out.begin() + (i + 1) * rs);
        return res;
    }
    std::vector<uint8_t> combine_partials(const std::vector<uint8_t>& partials, size_t count) {
        std::vector<uint8_t> out(result_size_);
        check(blz_msm_combine_partials(h_, partials.data(), count, out.data(), out.size()));
        return out;
    }
};

// ---------------------------------------------------------------- NTT (src/ingo_ntt)
enum class NTT { Ntt };                                             // ntt_api.rs:8-10
struct NttInit {};                                                  // ntt_api.rs:17
struct NTTInput { size_t buf_host; std::vector<uint8_t> data; };    // ntt_api.rs:19-23

class NTTClient : public DriverPrimitive<NTT, NttInit, NTTInput, std::vector<uint8_t>> {
    blz_ntt* h_ = nullptr;
    size_t nbytes_;

   public:
    DriverClient driver_client;
    // ntt_api.rs:26-31; 2^27 over BLS12-381 Fr, forward, is the reference shape.  `field` (a Curve: the
    // scalar field of that curve) and `inverse` have no reference counterpart.
    // flags: BLZ_NTT_NO_FACTOR_TABLE | BLZ_NTT_INVERSE | BLZ_NTT_BITREV_INPUT | BLZ_NTT_BITREV_OUTPUT; root: any primitive
    // 2^log_size-th root of unity (32 canonical little-endian bytes, checked on the device) instead of g^((r - 1) / 2^log_size) -
    // the transform's convention, which the reference leaves unstated (blaze_hip.h blz_ntt_new_ex3)
    NTTClient(NTT, DriverClient dclient, int log_size = 27, Curve field = Curve::BLS381, bool inverse = false, uint32_t flags = 0,
              const uint8_t* root = nullptr)
        : nbytes_(size_t(32) << log_size), driver_client(dclient) {
        check(blz_ntt_new_ex3(dclient.id, int(field), log_size, flags | (inverse ? BLZ_NTT_INVERSE : 0u), root, &h_));
    }
    ~NTTClient() override { blz_ntt_free(h_); }
    NTTClient(const NTTClient&) = delete;
    NTTClient& operator=(const NTTClient&) = delete;

    std::vector<uint32_t> loaded_binary_parameters() const override { throw std::logic_error("todo!() in the reference (ntt_api.rs:33-35)"); }
    void initialize(const NttInit&) override { check(blz_ntt_initialize(h_)); }                              // :37-56
    void set_data(const NTTInput& in) override { check(blz_ntt_set_data(h_, in.buf_host, in.data.data(), in.data.size())); }  // :72-87
    void start_process(std::optional<size_t> buf_kernel = std::nullopt) override { check(blz_ntt_start_process(h_, buf_kernel.value())); }  // :58-70
    void wait_result() override { check(blz_ntt_wait_result(h_)); }                                           // :89-108
    std::optional<std::vector<uint8_t>> result(std::optional<size_t> buf_num = std::nullopt) override {       // :110-124
        std::vector<uint8_t> out(nbytes_);
        check(blz_ntt_result(h_, buf_num.value(), out.data(), out.size()));
        return out;
    }
    // result(buf) into `out` and set_data({buf, next}) as one full-duplex call (blaze_hip.h blz_ntt_exchange): a cycle of the
    // reference's double-buffered loop (tests/integration_ntt.rs:102-136) on the buffer the kernel is not using
    void exchange(size_t buf, const std::vector<uint8_t>& next, std::vector<uint8_t>& out) {
        out.resize(nbytes_);
        check(blz_ntt_exchange(h_, buf, next.data(), next.size(), out.data(), out.size()));
    }
    // {device bytes held, pass 2 reads its factor table, pass 1 boundary table, log_size}
    std::array<uint64_t, 4> info() {
        std::array<uint64_t, 4> v{};
        check(blz_ntt_info(h_, v.data()));
        return v;
    }
};

}  // namespace ingo_blaze
