// Multi-GPU exchange of the MSM primitive: RCCL all-gather of the per-rank partial results + rank-ordered add.
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>

#include "msm_handle.hpp"

using namespace blz;

extern "C" {

#define BLZ_NCCL(api, call)                                                                                  \
    do {                                                                                                     \
        ncclResult_t r__ = (call);                                                                           \
        if (r__ != ncclSuccess) return fail(BLZ_ERR_UNKNOWN, "%s failed: %s", #call, (api)->GetErrorString(r__)); \
    } while (0)

int blz_comm_unique_id(uint8_t out[BLZ_COMM_ID_BYTES]) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    static_assert(BLZ_COMM_ID_BYTES == sizeof(ncclUniqueId), "id size");
    ncclUniqueId id;
    BLZ_NCCL(api, api->GetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return BLZ_OK;
}

// Communicator bring-up is a rendezvous: ncclCommInitRank returns when EVERY rank has called it, and for ever never
// if one of them died on the way.  It therefore runs on a helper thread and the caller waits for it against
// BLAZE_COMM_TIMEOUT_MS (default 60 000); on expiry the call fails with Unknown and the helper - parked inside RCCL - is
// abandoned (it owns its state through the shared_ptr and never touches the handle).
struct CommJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int rc = BLZ_OK;
    std::string err;
    std::vector<ncclComm_t> comms;
};
static int comm_timeout_ms() {
    const char* s = getenv("BLAZE_COMM_TIMEOUT_MS");
    int v = s && *s ? atoi(s) : 60000;
    return v > 0 ? v : 60000;
}
static int run_comm_job(std::shared_ptr<CommJob> job, std::function<int(CommJob&)> fn, const char* what) {
    std::thread([job, fn] {
        int rc = fn(*job);
        std::lock_guard<std::mutex> lk(job->mu);
        job->rc = rc;
        if (rc != BLZ_OK) job->err = blz_last_error_message();   // the message lives in the helper's thread-local buffer
        job->done = true;
        job->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(job->mu);
    const int limit = comm_timeout_ms();
    if (!job->cv.wait_for(lk, std::chrono::milliseconds(limit), [&] { return job->done; }))
        return fail(BLZ_ERR_UNKNOWN, "%s did not complete within %d ms (BLAZE_COMM_TIMEOUT_MS): a peer rank never arrived, or "
                    "RCCL cannot reach it; the bring-up thread is abandoned", what, limit);
    if (job->rc != BLZ_OK) return fail(job->rc, "%s", job->err.c_str());
    return BLZ_OK;
}

int blz_msm_comm_init(blz_msm* h, int rank, int nranks, const uint8_t id_bytes[BLZ_COMM_ID_BYTES]) {
    if (!h || !id_bytes) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(BLZ_ERR_INVALID_PARAM, "rank %d of %d", rank, nranks);
    if (h->comm) return fail(BLZ_ERR_INVALID_PARAM, "communicator already initialised on this handle");
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    BLZ_TRY(use_device(h->device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    auto job = std::make_shared<CommJob>();
    job->comms.assign(1, nullptr);
    const int dev = h->device;
    char what[96];
    snprintf(what, sizeof(what), "ncclCommInitRank (rank %d of %d)", rank, nranks);
    BLZ_TRY(run_comm_job(job, [api, dev, id, rank, nranks](CommJob& j) -> int {
        BLZ_HIP(hipSetDevice(dev), BLZ_ERR_FILE);
        BLZ_NCCL(api, api->CommInitRank(&j.comms[0], nranks, id, rank));   // collective: every rank calls it
        return BLZ_OK;
    }, what));
    h->comm = job->comms[0];
    h->comm_rank = rank;
    h->comm_size = nranks;
    return h->comm_buf.reserve((size_t)(nranks + 1) * result_size(h) + 64);
}

// One process driving several devices (the "management layer" of README.md:20-22 as a single host thread): one
// handle per device, rank i = handles[i].  The n bring-ups are one RCCL group (ncclGroupStart / End), because n
// sequential ncclCommInitRank calls from one thread would each wait for the ones that thread has not made yet.
int blz_msm_comm_init_all(blz_msm* const* handles, int n) {
    if (!handles || n < 1) return fail(BLZ_ERR_INVALID_PARAM, "no handles");
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return fail(BLZ_ERR_INVALID_PARAM, "null handle %d", i);
        if (handles[i]->comm) return fail(BLZ_ERR_INVALID_PARAM, "communicator already initialised on handle %d", i);
        if (handles[i]->curve != handles[0]->curve) return fail(BLZ_ERR_INVALID_PARAM, "handles of different curves");
        for (int k = 0; k < i; ++k)
            if (handles[k]->device == handles[i]->device)
                return fail(BLZ_ERR_INVALID_PARAM, "handles %d and %d share device %d (RCCL: one rank per device)", k, i, handles[i]->device);
    }
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    auto job = std::make_shared<CommJob>();
    job->comms.assign((size_t)n, nullptr);
    std::vector<int> devs;
    for (int i = 0; i < n; ++i) devs.push_back(handles[i]->device);
    BLZ_TRY(run_comm_job(job, [api, devs, n](CommJob& j) -> int {
        ncclUniqueId id;
        BLZ_NCCL(api, api->GetUniqueId(&id));
        BLZ_NCCL(api, api->GroupStart());
        for (int i = 0; i < n; ++i) {
            if (hipSetDevice(devs[i]) != hipSuccess) { (void)api->GroupEnd(); return fail_hip(BLZ_ERR_FILE, "hipSetDevice(%d) failed", devs[i]); }
            ncclResult_t r = api->CommInitRank(&j.comms[i], n, id, i);
            if (r != ncclSuccess) { (void)api->GroupEnd(); return fail(BLZ_ERR_UNKNOWN, "ncclCommInitRank(rank %d) failed: %s", i, api->GetErrorString(r)); }
        }
        BLZ_NCCL(api, api->GroupEnd());
        return BLZ_OK;
    }, "ncclCommInitRank group (single process)"));
    for (int i = 0; i < n; ++i) {
        handles[i]->comm = job->comms[i];
        handles[i]->comm_rank = i;
        handles[i]->comm_size = n;
        BLZ_TRY(use_device(handles[i]->device));
        BLZ_TRY(handles[i]->comm_buf.reserve((size_t)(n + 1) * result_size(handles[i]) + 64));
    }
    return BLZ_OK;
}

// enqueue this handle's half of the exchange on its exchange stream (no host wait)
static int enqueue_all_gather(blz_msm* h, const RcclApi* api, const uint8_t* partial, uint8_t** recv_out) {
    BLZ_TRY(use_device(h->device));
    // own stream: the exchange must not queue behind the next task's accumulation on the main stream
    hipStream_t st = h->eng.aux_stream;
    const size_t rs = result_size(h);
    uint8_t* send = h->comm_buf.as<uint8_t>();
    uint8_t* recv = send + ((rs + 63) / 64) * 64;
    BLZ_HIP(hipMemcpyAsync(send, partial, rs, hipMemcpyHostToDevice, st), BLZ_ERR_WRITE);
    BLZ_NCCL(api, api->AllGather(send, recv, rs, ncclUint8, h->comm, st));
    *recv_out = recv;
    return BLZ_OK;
}

int blz_msm_all_gather_combine(blz_msm* h, const uint8_t* partial, uint8_t* out, size_t out_cap) {
    if (!h || !partial || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    BLZ_LIVE(h);
    if (!h->comm) return fail(BLZ_ERR_INVALID_PARAM, "all_gather_combine before comm_init");
    if (out_cap < result_size(h)) return fail(BLZ_ERR_INVALID_PARAM, "result buffer too small");
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    uint8_t* recv = nullptr;
    BLZ_TRY(enqueue_all_gather(h, api, partial, &recv));
    // rank order = buffer order; the wait inside is bounded (a peer that never joins the all-gather: Unknown, wedged)
    BLZ_WAIT(h, h->eng.combine_partials(recv, (size_t)h->comm_size, out, true));
    return BLZ_OK;
}

// The exchange for the handles of blz_msm_comm_init_all, from the one thread that drives them: partials and out hold
// n x result_size bytes in handle order; every handle's sum is written (identical bytes).  All n all-gathers are
// enqueued as one RCCL group before any of them is waited for.
int blz_msm_all_gather_combine_all(blz_msm* const* handles, int n, const uint8_t* partials, uint8_t* out, size_t out_cap) {
    if (!handles || n < 1 || !partials || !out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    const RcclApi* api = rccl_api();
    if (!api) return BLZ_ERR_LOAD_FAILED;
    for (int i = 0; i < n; ++i) {
        if (!handles[i] || !handles[i]->comm || handles[i]->comm_size != n || handles[i]->comm_rank != i)
            return fail(BLZ_ERR_INVALID_PARAM, "handle %d is not rank %d of a %d-rank communicator (blz_msm_comm_init_all)", i, i, n);
        BLZ_LIVE(handles[i]);
    }
    const size_t rs = result_size(handles[0]);
    if (out_cap < rs * (size_t)n) return fail(BLZ_ERR_INVALID_PARAM, "result buffer too small: %zu < %zu", out_cap, rs * (size_t)n);
    std::vector<uint8_t*> recv((size_t)n, nullptr);
    BLZ_NCCL(api, api->GroupStart());
    for (int i = 0; i < n; ++i) {
        int rc = enqueue_all_gather(handles[i], api, partials + (size_t)i * rs, &recv[i]);
        if (rc != BLZ_OK) { (void)api->GroupEnd(); return rc; }
    }
    BLZ_NCCL(api, api->GroupEnd());
    for (int i = 0; i < n; ++i) BLZ_WAIT(handles[i], handles[i]->eng.combine_partials(recv[i], (size_t)n, out + (size_t)i * rs, true));
    return BLZ_OK;
}

int blz_msm_comm_free(blz_msm* h) {
    if (!h) return fail(BLZ_ERR_INVALID_PARAM, "null handle");
    if (!h->comm) return BLZ_OK;
    const RcclApi* api = rccl_api();
    if (api) {
        (void)hipSetDevice(h->device);
        // a communicator whose exchange never completed cannot be destroyed gracefully (ncclCommDestroy waits for it)
        if (sync_stream_bounded(h->eng.aux_stream, "comm_free: exchange stream") == BLZ_OK) (void)api->CommDestroy(h->comm);
        else if (api->CommAbort) (void)api->CommAbort(h->comm);
    }
    h->comm = nullptr;
    h->comm_size = 0;
    h->comm_buf.release();
    return BLZ_OK;
}

}  // extern "C"
