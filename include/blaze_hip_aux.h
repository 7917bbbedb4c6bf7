/* libblaze_hip_aux - test, bench and calibration scaffolding around libblaze_hip (include/blaze_hip.h).
 *
 * NOT part of the product library: synthetic input generators, the multiply-add calibration kernel of bench.py, element-wise
 * test hooks for the device field / group arithmetic, and stall kernels for the bounded-wait tests.  Built as
 * blaze_amd/lib/libblaze_hip_aux.so (links against libblaze_hip.so); loaded by tests/, bench.py and tools/ only.
 * Same conventions as blaze_hip.h (return codes, blz_last_error_message).
 */
#ifndef BLAZE_HIP_AUX_H
#define BLAZE_HIP_AUX_H

#include "blaze_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ synthetic inputs (bench/tests) */

/* scalars: n x 32 B, uniform-ish in [0, r) from a counter-based generator */
int blz_synth_scalars(int device_id, int curve, void* d_out, uint64_t n, uint64_t seed);
/* same stream of values, elements [start, start+n): a shard of a larger synthetic set */
int blz_synth_scalars_at(int device_id, int curve, void* d_out, uint64_t n, uint64_t seed, uint64_t start);
/* points: element i gets pf bases B_{i,j} = 2^(32 j) * (start+i+1) * G, wire format x||y canonical */
int blz_synth_points(int device_id, int curve, void* d_out, uint64_t n, int pf, uint64_t start);
/* NTT input: n x 32 B uniform-ish in [0, r) of BLS12-381 Fr */
int blz_synth_field_elements(int device_id, void* d_out, uint64_t n, uint64_t seed);

/* The issue rate of v_mad_u64_u32 - the instruction the MSM / NTT kernels are bound by - measured on this device at
 * its clocks of this moment: a ~target_ms kernel of independent multiply-add chains on every SIMD.  out[0] lane-ops
 * per second, [1] kernel ms, [2] the device's nominal clock in MHz (hipDeviceAttributeClockRate), [3] ops executed.
 * Measurement aid of bench.py (roofline.integer_issue.peak); no reference counterpart. */
int blz_calib_mad_rate(int device_id, uint32_t target_ms, double out[4]);

/* Test hooks for the bounded waits: enqueue, on the handle's main stream, a one-lane kernel that spins until
 * blz_test_stall_release(token) or until max_ms (1..30000) have passed on the device clock, whichever comes first. */
int blz_test_msm_stall(blz_msm* h, uint32_t max_ms, void** token);
int blz_test_ntt_stall(blz_ntt* h, uint32_t max_ms, void** token);
int blz_test_stall_release(void* token);

/* ------------------------------------------------------------------ test hooks (element-wise kernels)
 * Run the device field / group primitives on arrays so tests can compare them one by one with the
 * CPU oracle.  Host pointers; canonical little-endian encodings.
 *   fq ops (field = 0: Fq, 1: Fr): 0 mul, 1 add, 2 sub, 3 inverse(a), 4 sqr(a), 5/6 a b +- (a + b)(a - b);
 *     reduced-radix twin (every base field and every scalar field has one):
 *     10 mul, 11 sqr, 12 a b + (a + b)(a - b), 13 (a - 3b) b, 14 [a == b], 15 a (a - 3b) through the
 *     product-free reduction of a lazy value
 *   ec ops: 0 P+Q (mixed, P as accumulator), 1 2P, 2 P+Q (full XYZZ add), 3 P-Q (mixed, negated),
 *     4 / 5 P+Q / P-Q through the reduced-radix mixed add; 6 / 7 P+Q / Q-P with both operands affine (the first
 *     addition of a bucket run); 8 P+Q through the reduced-radix full add, 9 2P through its doubling (bucket reduce)
 *     points x||y; inf_flags[i] bit0: P is infinity, bit1: Q is infinity; out_inf[i]=1 if result inf */
int blz_test_field_op(int device_id, int curve, int field, int op, const uint8_t* a, const uint8_t* b,
                      uint8_t* out, size_t n);
int blz_test_ec_op(int device_id, int curve, int op, const uint8_t* p, const uint8_t* q,
                   const uint8_t* inf_flags, uint8_t* out, uint8_t* out_inf, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* BLAZE_HIP_AUX_H */
