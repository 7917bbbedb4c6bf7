// Multi-GPU shard layouts (include/blaze_hip.h blz_msm_shard_layout*): host-side pricing of element chunks x scalar-bit ranges.
#include "msm_handle.hpp"

using namespace blz;

// One candidate of the shard layout: R scalar ranges of 256 / R bits x nranks / R element chunks; rank = chunk * R + range.
// Estimates for the most expensive rank of the layout (range 0 holds the most real bits).
static int shard_candidate(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t flags, int R, uint32_t out[8], double* cost_ms) {
    static const int r_bits[3] = {253, 255, 254};
    if (R < 1 || R > 8 || (R & (R - 1)) || nranks % R) return fail(BLZ_ERR_INVALID_PARAM, "%d scalar ranges do not divide %d ranks", R, nranks);
    const int PC = nranks / R;
    const uint64_t base = nof_elements / PC, rem = nof_elements % PC;
    const uint32_t per = (uint32_t)(base + (rem ? 1 : 0));   // the largest chunk
    if (per == 0 && R > 1) return fail(BLZ_ERR_INVALID_PARAM, "fewer elements than element chunks");
    const int vbits = 256 / R;
    const MsmPlan P = make_plan(per ? per : 1, vbits, vbits < r_bits[curve] ? vbits : r_bits[curve], 0);
    if (P.c == 0) return fail(BLZ_ERR_INVALID_PARAM, "no window plan for %u elements of %d bits", per, vbits);
    // the planner's cost is in ns, fitted to the round-1 kernels to RANK plans; as an absolute time it runs 13 % above what
    // the shards measure today (profiles/r03_shard_layouts.txt: element split 64.1 / 33.4 / 18.4 ms measured against 74.8 /
    // 39.3 / 20.7 estimated, scalar split 61.4 / 32.6 / 18.3 against 71.5 / 36.1 / 19.7) - and it is compared with a
    // transfer time here, so it is scaled
    const double compute_ms = P.cost * 1e-6 * 0.87;
    // measured host -> device rate of pageable buffers on this platform (56.3 GB/s)
    const double link_bytes = (flags & BLZ_SHARD_SCALARS_FROM_HOST ? (double)per * 32.0 : 0.0) +
                              (flags & BLZ_SHARD_BASES_FROM_HOST ? (double)per * (double)blz_point_size(curve) : 0.0);
    const double link_ms = link_bytes / 56.3e9 * 1e3;
    const double mem_bytes = (double)per * ((double)blz_point_size(curve) + (double)mont_point_bytes(curve) + 32.0);
    const int pc = rank / R, rg = rank % R;
    const uint64_t first = (uint64_t)pc * base + ((uint64_t)pc < rem ? pc : rem);
    out[0] = (uint32_t)first;
    out[1] = (uint32_t)(base + ((uint64_t)pc < rem ? 1 : 0));
    out[2] = (uint32_t)(rg * vbits);
    out[3] = (uint32_t)((rg + 1) * vbits);
    out[4] = (uint32_t)R;
    out[5] = (uint32_t)(compute_ms * 1e3);
    out[6] = (uint32_t)(link_ms * 1e3);
    out[7] = (uint32_t)(mem_bytes / 1048576.0);
    // A stream of tasks overlaps a task's transfer with its predecessor's compute - not for free: measured per rank of a 2^26
    // job (profiles/r04_shard_layouts.txt: every candidate with resident scalars and with scalars from host memory), a task
    // whose upload hides costs its compute + 13 - 20 % of the upload (the blocking set_data keeps the host from collecting and
    // submitting; the copy shares HBM with the accumulation), and one whose upload does not hide costs the upload + 3 - 4 ms.
    if (cost_ms) {
        const double hidden = compute_ms + 0.15 * link_ms, exposed = 1.1 * link_ms;
        *cost_ms = hidden > exposed ? hidden : exposed;
    }
    return BLZ_OK;
}

extern "C" {

int blz_msm_shard_layout_candidate(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t flags, int R, uint32_t out[8]) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (curve < 0 || curve > 2) return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(BLZ_ERR_INVALID_PARAM, "rank %d of %d", rank, nranks);
    return shard_candidate(curve, nof_elements, nranks, rank, flags, R, out, nullptr);
}

int blz_msm_shard_layout_ex(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t flags, uint32_t out[8]) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    if (curve < 0 || curve > 2) return fail(BLZ_ERR_INVALID_PARAM, "unknown curve %d", curve);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(BLZ_ERR_INVALID_PARAM, "rank %d of %d", rank, nranks);
    // BLAZE_SHARD = elements | bits forces R = 1 / the largest R that divides nranks
    const char* mode = getenv("BLAZE_SHARD");
    const bool force_elements = mode && strcmp(mode, "elements") == 0, force_bits = mode && strcmp(mode, "bits") == 0;
    // device memory a rank may spend on its bases (raw + Montgomery copy) and scalars: half of the 288 GB, the rest is
    // workspace (entries, sort intermediates, partial sums) and whatever else the host keeps there
    const double mem_budget_mib = 144.0 * 1024.0;
    int bestR = 0;
    double best = 1e300, cost1 = 1e300;
    uint32_t tmp[8];
    for (int R = 1; R <= 8; R *= 2) {
        if (nranks % R) continue;
        double cost = 0;
        if (shard_candidate(curve, nof_elements, nranks, rank, flags, R, tmp, &cost) != BLZ_OK) continue;
        if ((double)tmp[7] > mem_budget_mib && R > 1) continue;
        if (R == 1) cost1 = cost;
        if (force_elements) { if (R == 1) { bestR = 1; break; } continue; }
        if (force_bits) { bestR = R; continue; }
        // the element split is the simpler layout (no shared bases, the smallest upload per rank): a scalar split has to
        // beat it by more than 2 % of the planner's estimate
        const double eff = R == 1 ? cost : cost * 1.02;
        if (eff < best) { best = eff; bestR = R; }
    }
    (void)cost1;
    if (bestR == 0) return fail(BLZ_ERR_INVALID_PARAM, "no shard layout for %u elements on %d ranks", nof_elements, nranks);
    return shard_candidate(curve, nof_elements, nranks, rank, flags, bestR, out, nullptr);
}

int blz_msm_shard_layout(int curve, uint32_t nof_elements, int nranks, int rank, uint32_t out[4]) {
    if (!out) return fail(BLZ_ERR_INVALID_PARAM, "null argument");
    uint32_t o[8];
    BLZ_TRY(blz_msm_shard_layout_ex(curve, nof_elements, nranks, rank, 0u, o));
    for (int i = 0; i < 4; ++i) out[i] = o[i];
    return BLZ_OK;
}

}  // extern "C"
