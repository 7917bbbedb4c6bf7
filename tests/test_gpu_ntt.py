"""NTT parity on the MI355X through the C ABI (bit-exact vs the CPU oracle), golden vectors, the
double-buffer contract of tests/integration_ntt.rs:62-146, and the full 2^27 shape by properties."""
import json
import os
import random

import pytest

import blaze_amd
from blaze_amd import DeviceBuffer, DriverClientError
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_ntt import NTT, NTTClient, NTTInput, NttInit
from oracle import pyref

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
R = pyref.CURVES["BLS381"]["r"]


def _ntt(cl, data, buf=0):
    # call order of benches/ntt_bench.rs:16-45: set_data, initialize, start_process, wait_result, result
    cl.set_data(NTTInput(buf, data))
    cl.initialize(NttInit())
    cl.start_process(buf)
    cl.wait_result()
    return bytes(cl.result(buf))


def test_golden_vectors(gpu):
    with open(os.path.join(HERE, "golden", "ntt_vectors.json")) as f:
        for v in json.load(f):
            cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=v["logn"])
            assert _ntt(cl, bytes.fromhex(v["input"])) == bytes.fromhex(v["output"]), v["logn"]
            cl.close()


@pytest.mark.parametrize("logn", [2, 5, 9, 10, 12, 13, 17, 18, 19, 20])
def test_against_oracle(gpu, orc, logn):
    """every pass structure: 1 pass (<=9), 2 passes (10..18), 3 passes (>=19)."""
    rng = random.Random(logn)
    n = 1 << logn
    data = b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(min(n, 4096)))
    data = (data * (n // min(n, 4096)))[: 32 * n]
    if n > 4096:  # de-periodise
        data = bytearray(data)
        for i in range(0, n, 97):
            data[32 * i: 32 * i + 8] = (i * 0x9E3779B97F4A7C15 % (1 << 64)).to_bytes(8, "little")
        data = bytes(data)
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    got = _ntt(cl, data)
    assert got == bytes(orc.ntt("BLS381", data, logn, threads=16))
    # edge values: 0, 1, r-1 everywhere
    for val in (0, 1, R - 1):
        d = val.to_bytes(32, "little") * n
        assert _ntt(cl, d, buf=1) == bytes(orc.ntt("BLS381", d, logn, threads=16))
    cl.close()


def test_double_buffer_contract(gpu, orc):
    """tests/integration_ntt.rs:102-136: compute on one buffer while the host loads / reads the other."""
    logn = 14
    n = 1 << logn
    rng = random.Random(5)
    ins = [b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(n)) for _ in range(3)]
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    cl.set_data(NTTInput(0, ins[0]))
    outs = []
    for i in range(3):
        b = i % 2
        cl.initialize(NttInit())
        cl.start_process(b)
        if i + 1 < 3:
            cl.set_data(NTTInput(1 - b, ins[i + 1]))     # host writes the other buffer meanwhile
        cl.wait_result()
        outs.append(bytes(cl.result(b)))
    for i in range(3):
        assert outs[i] == bytes(orc.ntt("BLS381", ins[i], logn, threads=8))
    with pytest.raises(DriverClientError):
        cl.wait_result()
    with pytest.raises(DriverClientError):
        cl.set_data(NTTInput(0, ins[0][:-32]))
    cl.close()


def test_parallel_correctness_reference_sequence(gpu, orc):
    """ntt_parallel_test_correctness replayed call for call (tests/integration_ntt.rs:99-143): every cycle starts
    the kernel on one buffer, READS the other buffer's result, writes the next input into it and only then waits.
    Cycle 0 starts a transform on a buffer nobody wrote and reads a buffer nobody wrote: both must succeed (the
    card's buffers simply exist); outputs are collected from cycle 2 on."""
    logn, nof_vectors = 14, 3
    n = 1 << logn
    rng = random.Random(77)
    in_vecs = [b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(n)) for _ in range(nof_vectors)]
    ref_vecs = [bytes(orc.ntt("BLS381", v, logn, threads=8)) for v in in_vecs]
    driver = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    driver.initialize(NttInit())
    outputs = []
    for i in range(nof_vectors + 2):
        buf_host = i % 2
        buf_kernel = 1 - buf_host
        driver.start_process(buf_kernel)
        res = driver.result(buf_host)
        assert len(res) == 32 * n
        if i == 0:
            assert bytes(res) == bytes(32 * n)          # never written: zero-filled, like freshly reset HBM
        if i >= 2:
            outputs.append(bytes(res))
        driver.set_data(NTTInput(buf_host, in_vecs[min(i, nof_vectors - 1)]))
        # the buffer under transform is off limits until wait_result
        with pytest.raises(DriverClientError):
            driver.result(buf_kernel)
        with pytest.raises(DriverClientError):
            driver.set_data(NTTInput(buf_kernel, in_vecs[0]))
        driver.wait_result()
    assert outputs == ref_vecs
    driver.close()


@pytest.mark.parametrize("logn", [14, 20])
def test_parallel_correctness_through_exchange(gpu, orc, logn):
    """The same loop with result + set_data of a cycle fused into blz_ntt_exchange (both directions of the link at once): the
    outputs are the reference loop's, byte for byte; the buffer under transform stays off limits."""
    nof_vectors = 3
    n = 1 << logn
    rng = random.Random(78)
    base = [b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(4096)) for _ in range(nof_vectors)]
    in_vecs = []
    for b in base:
        v = bytearray((b * (n // 4096 + 1))[: 32 * n])
        for i in range(0, n, 61):
            v[32 * i: 32 * i + 8] = (i * 0x9E3779B97F4A7C15 % (1 << 64)).to_bytes(8, "little")
        in_vecs.append(bytes(v))
    ref_vecs = [bytes(orc.ntt("BLS381", v, logn, threads=8)) for v in in_vecs]
    driver = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    driver.initialize(NttInit())
    outputs = []
    res = bytearray(32 * n)
    for i in range(nof_vectors + 2):
        buf_host = i % 2
        buf_kernel = 1 - buf_host
        driver.start_process(buf_kernel)
        with pytest.raises(DriverClientError):
            driver.exchange(buf_kernel, in_vecs[0], res)
        driver.exchange(buf_host, in_vecs[min(i, nof_vectors - 1)], res)
        if i == 0:
            assert bytes(res) == bytes(32 * n)
        if i >= 2:
            outputs.append(bytes(res))
        driver.wait_result()
    assert outputs == ref_vecs
    with pytest.raises(DriverClientError):
        driver.exchange(0, in_vecs[0][:-32], res)            # wrong input length
    with pytest.raises(ValueError):
        driver.exchange(0, in_vecs[0], bytearray(32 * n - 1))
    driver.close()


def test_exchange_with_page_locked_buffers(gpu, orc):
    """blz_host_malloc buffers take the exchange's other path (true asynchronous copies chained by events on the caller's
    thread, no helper thread): same bytes.  2^22 elements = 128 MiB: several pieces, the ramped ones included."""
    import numpy as np

    logn = 22
    n = 1 << logn
    rng = np.random.default_rng(9)
    x = rng.integers(0, 256, size=32 * n, dtype=np.uint8)
    x[31::32] &= 0x3F
    ref = bytes(orc.ntt("BLS381", x, logn, threads=16))
    hin, hout = blaze_amd.HostBuffer(0, 32 * n), blaze_amd.HostBuffer(0, 32 * n)
    a_in, a_out = hin.array(), hout.array()
    a_in[:] = x
    a_out[:] = 0xEE
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    cl.initialize(NttInit())
    for i in range(3):
        bh = i % 2
        cl.start_process(1 - bh)
        cl.exchange(bh, a_in, a_out)
        if i == 0:
            assert not a_out.any()                     # the buffer nobody wrote
        if i == 2:
            assert a_out.tobytes() == ref              # cycle 0's input, transformed in cycle 1
        cl.wait_result()
    # a pinned input with a pageable output takes the threaded path: still the same bytes
    out2 = bytearray(32 * n)
    cl.start_process(0)
    cl.exchange(1, a_in, out2)
    cl.wait_result()
    assert bytes(out2) == ref
    cl.close(); hin.free(); hout.free()


def test_exchange_full_size_2e27(gpu, orc):
    """The reference shape through the fused cycle: 4 GiB out and 4 GiB in at once, three cycles; the bytes that come back are
    the bytes result() returns (checked against a plain result of the same transform), and what went in is what set_data
    would have written (read back with result after the closing transform's inverse property: X of a delta)."""
    import numpy as np
    logn = 27
    n = 1 << logn
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    cl.initialize(NttInit())
    d_in = DeviceBuffer(0, 32 * n)
    blaze_amd._lib.check(blaze_amd.aux().blz_synth_field_elements(0, d_in.ptr, n, 777))
    x = np.frombuffer(d_in.download(), dtype=np.uint8).copy()
    d_in.free()
    delta = np.zeros(32 * n, dtype=np.uint8)
    delta[32] = 1
    out = np.empty(32 * n, dtype=np.uint8)
    out[:] = 7                                            # touched: no first-touch page faults inside the timed call
    cl.set_data(NTTInput(0, x))
    cl.start_process(0)
    cl.wait_result()
    y_plain = np.frombuffer(cl.result(0), dtype=np.uint8).copy()
    # cycle: kernel on buffer 1 (zeros), exchange on buffer 0: the result leaves, the delta lands
    import time
    cl.start_process(1)
    t0 = time.perf_counter()
    cl.exchange(0, delta, out)
    dt = time.perf_counter() - t0
    cl.wait_result()
    assert np.array_equal(out, y_plain), "exchange returned other bytes than result()"
    cl.start_process(0)
    cl.wait_result()
    z = np.frombuffer(cl.result(0), dtype=np.uint8)
    w = orc.omega("BLS381", logn)
    for k in (0, 1, 2, 511, 512, 513, 1 << 18, 99999999, n - 1):
        assert int.from_bytes(z[32 * k: 32 * k + 32].tobytes(), "little") == pow(w, k, R), k
    print("exchange of 2 x 4 GiB: %.1f ms" % (dt * 1e3))
    cl.close()


def test_full_size_2e27_properties(gpu, orc):
    """The reference shape (2^27 x 32 B = 4 GiB, ntt_data.rs:65-66).  Checked by: delta -> all ones,
    X[0] = sum x, sum_k X[k] = n x[0], spot coefficients against the O(n) Horner oracle."""
    logn = 27
    n = 1 << logn
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    d_in = DeviceBuffer(0, 32 * n)
    blaze_amd._lib.check(blaze_amd.aux().blz_synth_field_elements(0, d_in.ptr, n, 99))
    cl.set_data(NTTInput(0, d_in))
    cl.initialize(NttInit())
    cl.start_process(0)
    cl.wait_result()
    import numpy as np
    x = np.frombuffer(d_in.download(), dtype=np.uint8)
    d_in.free()
    out = cl.result(0)
    y = np.frombuffer(out, dtype=np.uint8)

    def elem(arr, i):
        return int.from_bytes(arr[32 * i: 32 * i + 32].tobytes(), "little")

    def field_sum(arr):  # sum of all elements mod r via 64-bit limb sums
        limbs = arr.view(np.uint64).reshape(-1, 4)
        tot = 0
        for j in range(4):
            lo = int((limbs[:, j] & np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64))
            hi = int((limbs[:, j] >> np.uint64(32)).sum(dtype=np.uint64))
            tot += (lo + (hi << 32)) << (64 * j)
        return tot % R

    assert elem(y, 0) == field_sum(x)
    assert field_sum(y) == (n * elem(x, 0)) % R
    for k in (1, 12345678, n - 1):
        assert elem(y, k) == orc.ntt_eval_at("BLS381", x, logn, k, threads=16), k
    # delta at index 1 -> X[k] = w^k
    delta = np.zeros(32 * n, dtype=np.uint8)
    delta[32] = 1
    cl.set_data(NTTInput(1, delta))
    cl.start_process(1)
    cl.wait_result()
    z = np.frombuffer(cl.result(1), dtype=np.uint8)
    w = orc.omega("BLS381", logn)
    for k in (0, 1, 2, 511, 512, 513, 1 << 18, (1 << 18) + 1, 99999999, n - 1):
        assert elem(z, k) == pow(w, k, R), k
    cl.close()


_ORACLE_2E27 = {}


@pytest.mark.parametrize("pass2", ["factor_table", "stepped"])
def test_full_size_2e27_every_output(gpu, orc, pass2):
    """Every one of the 2^27 outputs of the reference shape, byte for byte against the oracle's threaded radix-2
    transform of the same 4 GiB input, and inverse(forward(x)) == x at 2^27 (the 512^3 kernel k_ntt512 only
    ever runs at this size).  Both pass-2 kernels: the one that reads its boundary factors from the 4 GiB per-element
    table (what a handle gets when the table fits) and the one that steps them (what a memory-tight device gets, and what
    BLZ_NTT_NO_FACTOR_TABLE asks for); blz_ntt_info says which one a handle runs and what it holds."""
    import numpy as np
    logn = 27
    n = 1 << logn
    flags = NTTClient.NO_FACTOR_TABLE if pass2 == "stepped" else 0
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, flags=flags)
    info = cl.info()
    assert info["log_size"] == 27 and info["pass1_boundary_table"]
    assert info["pass2_factor_table"] == (pass2 == "factor_table")
    want = (3 + (1 if pass2 == "factor_table" else 0)) * 32 * n      # two buffers + scratch (+ the factor table)
    assert want <= info["device_bytes"] <= want * 1.05, info
    d_in = DeviceBuffer(0, 32 * n)
    blaze_amd._lib.check(blaze_amd.aux().blz_synth_field_elements(0, d_in.ptr, n, 4242))
    cl.set_data(NTTInput(0, d_in))
    cl.initialize(NttInit())
    cl.start_process(0)
    cl.wait_result()
    x = np.frombuffer(d_in.download(), dtype=np.uint8)
    y = np.frombuffer(cl.result(0), dtype=np.uint8)
    threads = max(1, min(64, (os.cpu_count() or 8)))
    # (the oracle's transform of this input - 20 s of host time - is computed by the first of the two variants and kept for
    # the second: same seed, same input)
    exp = _ORACLE_2E27.pop("exp", None)
    if exp is None:
        exp = np.frombuffer(orc.ntt("BLS381", x, logn, threads=threads), dtype=np.uint8)
        if pass2 == "factor_table":
            _ORACLE_2E27["exp"] = exp
    if not np.array_equal(y, exp):
        bad = np.flatnonzero(y.reshape(-1, 32) != exp.reshape(-1, 32))
        raise AssertionError(f"2^27 forward transform differs from the oracle, first at element {int(bad[0]) // 32}")
    del exp
    # round trip on the device: the forward result goes back through the inverse transform
    inv = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, inverse=True, flags=flags)
    assert inv.info()["pass2_factor_table"] == (pass2 == "factor_table")
    d_y = DeviceBuffer(0, 32 * n)
    cl.result_device(0, d_y)
    cl.close()
    inv.set_data(NTTInput(0, d_y))
    inv.initialize(NttInit())
    inv.start_process(0)
    inv.wait_result()
    z = np.frombuffer(inv.result(0), dtype=np.uint8)
    assert np.array_equal(z, x), "inverse(forward(x)) != x at 2^27"
    inv.close(); d_in.free(); d_y.free()


def test_random_sizes_fields_directions(gpu, orc):
    """A seeded walk over (field, size, direction, buffer) on long-lived clients: every combination of kernel
    families (radix-2 passes, 512-point reduced-radix passes) and table sets gets used more than once per client."""
    import numpy as np
    rng = random.Random(20260)
    clients = {}
    for it in range(36):
        field = rng.choice(["BLS381", "BLS377", "BN254"])
        logn = rng.choice([1, 4, 9, 10, 14, 17, 18, 19, 21, 22])
        inv = rng.random() < 0.35
        key = (field, logn, inv)
        if key not in clients:
            clients[key] = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, inverse=inv, field=field)
        cl = clients[key]
        n = 1 << logn
        x = np.random.default_rng(rng.randrange(1 << 30)).integers(0, 256, size=32 * n, dtype=np.uint8)
        x[31::32] &= 0x0F          # < 2^252: canonical in all three fields
        buf = rng.randrange(2)
        got = _ntt(cl, x.tobytes(), buf=buf)
        assert got == bytes(orc.ntt(field, x.tobytes(), logn, inverse=inv, threads=16)), (it, field, logn, inv, buf)
    for cl in clients.values():
        cl.close()


@pytest.mark.parametrize("field", ["BLS377", "BN254"])
def test_full_size_2e27_other_fields(gpu, orc, field):
    """2^27 over the other two scalar fields: the only size at which their pass 1 runs the 512-point reduced-radix
    kernel with its boundary table (and their wider input bound, 16 m / 8 m).  X[0] = sum x, sum_k X[k] = n x[0],
    spot coefficients against the O(n) evaluation oracle, inverse(forward(x)) == x on the device."""
    import numpy as np
    logn = 27
    n = 1 << logn
    r = pyref.CURVES[field]["r"]
    rng = np.random.default_rng(27 + len(field))
    # (a 2^20-element random block tiled 128 times, every element's low word then mixed with its index: no period, and
    # a tenth of the time 4 GiB of generator output takes)
    blk = rng.integers(0, 256, size=32 << 20, dtype=np.uint8)
    x = np.tile(blk, 128)
    x.view(np.uint64)[0::4] ^= np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    x[31::32] &= 0x0F          # < 2^252 < r: canonical in both fields
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field)
    cl.set_data(NTTInput(0, x))
    cl.initialize(NttInit())
    cl.start_process(0)
    cl.wait_result()
    y = np.frombuffer(cl.result(0), dtype=np.uint8)

    def elem(arr, i):
        return int.from_bytes(arr[32 * i: 32 * i + 32].tobytes(), "little")

    def field_sum(arr):
        limbs = arr.view(np.uint64).reshape(-1, 4)
        tot = 0
        for j in range(4):
            lo = int((limbs[:, j] & np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64))
            hi = int((limbs[:, j] >> np.uint64(32)).sum(dtype=np.uint64))
            tot += (lo + (hi << 32)) << (64 * j)
        return tot % r

    assert elem(y, 0) == field_sum(x)
    assert field_sum(y) == (n * elem(x, 0)) % r
    for k in (1, 87654321, n - 1):
        got = elem(y, k)
        assert got < r
        assert got == orc.ntt_eval_at(field, x, logn, k, threads=16), k
    inv = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, inverse=True, field=field)
    d_y = DeviceBuffer(0, 32 * n)
    cl.result_device(0, d_y)
    cl.close()
    inv.set_data(NTTInput(0, d_y))
    inv.initialize(NttInit())
    inv.start_process(0)
    inv.wait_result()
    z = np.frombuffer(inv.result(0), dtype=np.uint8)
    assert np.array_equal(z, x), "inverse(forward(x)) != x at 2^27"
    inv.close(); d_y.free()


@pytest.mark.parametrize("logn", [3, 11, 19])
def test_inverse_transform(gpu, orc, logn):
    """the inverse direction (omega^-1, scaled by n^-1), against the oracle and
    as a round trip."""
    rng = random.Random(100 + logn)
    n = 1 << logn
    base = b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(min(n, 2048)))
    data = bytearray((base * (n // min(n, 2048)))[: 32 * n])
    for i in range(0, n, 61):
        data[32 * i: 32 * i + 8] = (i * 0x9E3779B97F4A7C15 % (1 << 64)).to_bytes(8, "little")
    data = bytes(data)
    fwd = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    inv = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, inverse=True)
    y = _ntt(fwd, data)
    assert _ntt(inv, data) == bytes(orc.ntt("BLS381", data, logn, inverse=True, threads=8))
    assert _ntt(inv, y) == data
    fwd.close(); inv.close()


@pytest.mark.parametrize("logn", [10, 12, 19])
def test_bank_wire_permutation(gpu, orc, logn):
    """NTTBanks::preprocess / postprocess (ntt_data.rs:80-156) as device kernels == the oracle's
    restatement of the reference loops, at scaled shapes (2^27 itself is the same closed form)."""
    n = 1 << logn
    groups = max(1, n >> 18)
    data = bytearray(32 * n)
    for i in range(n):
        data[32 * i: 32 * i + 8] = (i * 0x9E3779B97F4A7C15 % (1 << 64)).to_bytes(8, "little")
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    d_in, d_banks, d_out = DeviceBuffer(0, 32 * n), DeviceBuffer(0, 32 * n), DeviceBuffer(0, 32 * n)
    d_in.upload(data)
    cl.banks_preprocess(d_in, d_banks)
    banks = d_banks.download()
    assert bytes(banks) == bytes(orc.ntt_preprocess(data, n))
    cl.banks_postprocess(d_banks, d_out)
    assert bytes(d_out.download()) == bytes(orc.ntt_postprocess(banks, n, groups))
    cl.close()
    for b in (d_in, d_banks, d_out):
        b.free()


@pytest.mark.parametrize("field", ["BLS377", "BN254"])
@pytest.mark.parametrize("logn", [3, 9, 12, 18, 20])
def test_other_scalar_fields(gpu, orc, field, logn):
    """the same passes over BLS12-377 Fr and BN254 Fr (lazy-range fields on the
    device, canonical on the wire), forward against the oracle and inverse(forward(x)) == x."""
    r = pyref.CURVES[field]["r"]
    rng = random.Random(1000 + logn)
    n = 1 << logn
    base = [rng.randrange(r) for _ in range(min(n, 2048))] + [0, 1, r - 1, r - 2]
    data = b"".join(base[(i * 7 + i // 2048) % len(base)].to_bytes(32, "little") for i in range(n))
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field)
    got = _ntt(cl, data)
    assert got == bytes(orc.ntt(field, data, logn, threads=16))
    inv = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, inverse=True, field=field)
    assert _ntt(inv, got) == data
    cl.close()
    inv.close()


@pytest.mark.parametrize("field", ["BLS381", "BLS377", "BN254"])
@pytest.mark.parametrize("logn", [9, 18, 20])
def test_non_canonical_words(gpu, orc, field, logn):
    """The wire format is canonical by contract, but any 256-bit word is a residue: x + k r (as large as 32 bytes
    hold, the all-ones word included) must transform like x - on the radix-2 kernel (2^9) and on the reduced-radix
    512-point kernel whose first step types its input as "< 2^256" (2^18: pass 2 reads the wire; 2^20: pass 1 generic)."""
    r = pyref.CURVES[field]["r"]
    rng = random.Random(77 + logn)
    n = 1 << logn
    kmax = ((1 << 256) - 1) // r
    base = []
    for i in range(1024):
        x = rng.randrange(r)
        k = rng.randrange(kmax + 1)
        if x + k * r >= 1 << 256:
            k -= 1
        base.append((x, x + k * r))
    base.append(((1 << 256) - 1 - kmax * r, (1 << 256) - 1))
    base.append((0, kmax * r))
    canon = b"".join(base[(i * 5 + i // 1024) % len(base)][0].to_bytes(32, "little") for i in range(n))
    stray = b"".join(base[(i * 5 + i // 1024) % len(base)][1].to_bytes(32, "little") for i in range(n))
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field)
    want = bytes(orc.ntt(field, canon, logn, threads=16))
    assert _ntt(cl, canon) == want
    assert _ntt(cl, stray, buf=1) == want
    cl.close()


def test_other_fields_large_and_limits(gpu, orc):
    """2^24 over BN254 Fr by properties + spot coefficients; BN254's two-adicity is 28, so 2^27 exists;
    sizes beyond a field's two-adicity are refused."""
    import numpy as np
    field, logn = "BN254", 24
    r = pyref.CURVES[field]["r"]
    n = 1 << logn
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, size=32 * n, dtype=np.uint8)
    x[31::32] &= 0x1F          # < 2^253 < r: canonical
    cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field)
    y = np.frombuffer(_ntt(cl, x.tobytes()), dtype=np.uint8)
    for k in (0, 1, 54321, n - 1):
        got = int.from_bytes(y[32 * k: 32 * k + 32].tobytes(), "little")
        assert got == orc.ntt_eval_at(field, x, logn, k, threads=16), k
        assert got < r
    cl.close()
    with pytest.raises(DriverClientError) as ei:
        NTTClient(NTT.Ntt, DriverClient(0), log_size=28, field="BN254")   # > 27: shape limit of the 3-pass plan
    assert ei.value.variant == "InvalidPrimitiveParam"


def test_wait_result_is_bounded(gpu, orc, monkeypatch):
    """The reference polls the NTT status register for ever (ntt_api.rs:89-108); here wait_result has a deadline
    (BLAZE_WAIT_TIMEOUT_MS): a stalled transform gives Unknown in bounded time, the handle is reset-only until the
    stall ends, and the transform that was queued behind it still produced the right output."""
    import ctypes as C
    import time

    logn = 12
    rng = random.Random(5)
    data = b"".join(rng.randrange(pyref.CURVES["BLS381"]["r"]).to_bytes(32, "little") for _ in range(1 << logn))
    nc = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn)
    nc.set_data(NTTInput(0, data))
    monkeypatch.setenv("BLAZE_WAIT_TIMEOUT_MS", "400")
    tok = C.c_void_p()
    blaze_amd._lib.check(blaze_amd.aux().blz_test_ntt_stall(nc._h, 20000, C.byref(tok)))
    nc.initialize(NttInit())
    nc.start_process(0)
    t0 = time.perf_counter()
    with pytest.raises(DriverClientError) as ei:
        nc.wait_result()
    dt = time.perf_counter() - t0
    assert ei.value.variant == "Unknown" and "timed out" in str(ei.value) and 0.3 < dt < 5.0, (str(ei.value), dt)
    with pytest.raises(DriverClientError) as ei:
        nc.start_process(1)
    assert "wedged" in str(ei.value)
    with pytest.raises(DriverClientError):
        nc.reset()
    blaze_amd._lib.check(blaze_amd.aux().blz_test_stall_release(tok))
    monkeypatch.setenv("BLAZE_WAIT_TIMEOUT_MS", "60000")
    nc.reset()
    assert bytes(nc.result(0)) == bytes(orc.ntt("BLS381", data, logn))   # the queued transform ran once the stall ended
    nc.close()


def test_random_call_sequences_never_break_a_client(gpu):
    """tests/probes/ntt_monkey.py: the reference's double-buffer cycle cut up at random - wrong buffer numbers and lengths, results read
    from the buffer under transform, start_process twice, exchange on either buffer - on clients of several sizes, fields and
    directions (both pass-2 kernels at 2^19 / 2^20).  Every call succeeds or fails with one of src/error.rs's variants; after every
    burst each client is reset and transforms a known vector into the oracle's bytes."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "probes", "ntt_monkey.py"), "40", "23"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout and "'ok':" in r.stdout and "InvalidPrimitiveParam" in r.stdout


def test_every_size_against_the_oracle(gpu):
    """tests/probes/ntt_sizes_probe.py: EVERY transform size 2^1 .. 2^24, forward and inverse over BLS12-381 Fr - and BLS12-377 / BN254 Fr at
    the sizes where the pass geometry changes - with every output compared to the CPU oracle (2^25 and 2^26 ran clean the same way;
    2^27 has its own tests)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "probes", "ntt_sizes_probe.py"), "24"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout and "MISMATCH" not in r.stdout and r.stdout.count(": ok") >= 2 * 24 + 2 * 7


def _nonres_root(orc, field, logn, t):
    """A primitive 2^logn-th root that is NOT this build's default: the default's t-th power, t odd."""
    r = pyref.CURVES[field]["r"]
    return pow(orc.omega(field, logn), t, r)


@pytest.mark.parametrize("field", ["BLS381", "BLS377", "BN254"])
@pytest.mark.parametrize("logn", [1, 4, 9, 12, 18, 20])
def test_convention_knobs_against_the_oracle(gpu, orc, field, logn):
    """blz_ntt_new_ex3: a caller-supplied primitive root and bit-reversed input / output order (the reference states neither root,
    direction nor order: NttInit {} is empty, ntt_api.rs:8-23; its goldens are external files, tests/integration_ntt.rs:15-18).
    Two non-default roots x {natural, bit-reversed} input x {natural, bit-reversed} output x {forward, inverse} against the
    oracle's transform under the same convention, every pass structure (1, 2, 3 passes; radix-2-in-LDS and 512-point kernels);
    default handles stay byte-identical to blz_ntt_new_ex2's; a root that is not primitive, or not canonical, is refused."""
    import numpy as np
    r = pyref.CURVES[field]["r"]
    n = 1 << logn
    x = np.random.default_rng(logn).integers(0, 256, size=32 * n, dtype=np.uint8).reshape(n, 32)
    x[:, 31] &= 0x0F
    data = x.tobytes()
    dflt = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field)
    base = _ntt(dflt, data)
    assert base == bytes(orc.ntt(field, data, logn, threads=16))
    dflt.close()
    assert _ntt(NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field, flags=0, root=None), data) == base
    assert _ntt(NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field, root=orc.omega(field, logn)), data) == base
    roots = [_nonres_root(orc, field, logn, t) for t in ((3, 2 * n - 1 if n > 2 else 3) if n > 2 else (1,))]
    for root in roots:
        for fl in range(8):
            inv, brin, brout = bool(fl & 1), bool(fl & 2), bool(fl & 4)
            flags = (NTTClient.INVERSE if inv else 0) | (NTTClient.BITREV_INPUT if brin else 0) | (NTTClient.BITREV_OUTPUT if brout else 0)
            cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field, flags=flags, root=root)
            got = _ntt(cl, data, buf=fl & 1)
            exp = bytes(orc.ntt(field, data, logn, inverse=inv, threads=16, root=root, bitrev_in=brin, bitrev_out=brout))
            assert got == exp, f"{field} 2^{logn} root={root:#x} inverse={inv} bitrev_in={brin} bitrev_out={brout}"
            cl.close()
    # refused roots: 1 (order 1), the default's square (order n / 2), a value >= r
    bad = [1, r + 1] + ([pow(orc.omega(field, logn), 2, r)] if n > 1 else [])
    for b in bad:
        with pytest.raises(DriverClientError) as ei:
            NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field, root=b)
        assert ei.value.variant == "InvalidPrimitiveParam", b
    with pytest.raises(DriverClientError):
        NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, field=field, flags=64)


def test_convention_knobs_full_size_2e27(gpu, orc):
    """The same knobs at the reference's size, where only the 512^3 kernels run: a non-default root with bit-reversed input and
    output, every one of the 2^27 outputs against the oracle's transform under that convention; the other root and the mixed
    orders through the identities that tie them to it (w -> w^t permutes the outputs: X_t[k] = X[t k mod n]; the orders are
    permutations of the buffers), on the device results; and the inverse handle of the same convention brings the input back."""
    import numpy as np
    field, logn = "BLS381", 27
    n = 1 << logn
    r = pyref.CURVES[field]["r"]
    threads = max(1, min(64, (os.cpu_count() or 8)))
    d_in = DeviceBuffer(0, 32 * n)
    blaze_amd._lib.check(blaze_amd.aux().blz_synth_field_elements(0, d_in.ptr, n, 2727))
    x = np.frombuffer(d_in.download(), dtype=np.uint8)
    root = _nonres_root(orc, field, logn, 5)
    both = NTTClient.BITREV_INPUT | NTTClient.BITREV_OUTPUT

    def run(flags, rt, src):
        cl = NTTClient(NTT.Ntt, DriverClient(0), log_size=logn, flags=flags | NTTClient.NO_FACTOR_TABLE, root=rt)
        cl.set_data(NTTInput(0, src)); cl.initialize(NttInit()); cl.start_process(0); cl.wait_result()
        out = np.frombuffer(cl.result(0), dtype=np.uint8)
        ms = cl.last_kernel_ms()
        cl.close()
        return out, ms

    y_bb, ms_bb = run(both, root, d_in)
    exp = np.frombuffer(orc.ntt(field, x, logn, threads=threads, root=root, bitrev_in=True, bitrev_out=True), dtype=np.uint8)
    assert np.array_equal(y_bb, exp), "2^27, caller's root, bit-reversed in and out: differs from the oracle"
    del exp
    # natural in / out under the same root: the same transform of the permuted buffer, permuted back
    d_xp = DeviceBuffer(0, 32 * n)
    d_xp.upload(orc.bitrev_permute(x, logn, threads))                                   # xp[p] = x[bitrev(p)]: read as a bit-reversed buffer, x IS the sequence xp
    y_nn, ms_nn = run(0, root, d_xp)                                                    # natural-order transform of xp ...
    y_bn, ms_bn = run(NTTClient.BITREV_INPUT, root, d_in)                               # ... = bit-reversed-input transform of x
    assert np.array_equal(y_nn, y_bn)
    del y_bn
    y_nb, ms_nb = run(NTTClient.BITREV_OUTPUT, root, d_xp)
    d_xp.free()
    assert np.array_equal(y_nb, y_bb)                                                   # (both flags on x = the transform of xp, written bit-reversed)
    del y_nb
    assert np.array_equal(np.frombuffer(orc.bitrev_permute(y_nn, logn, threads), dtype=np.uint8), y_bb)
    # the second root: w' = w^3: X'[k] = X[3 k mod n], checked on a strided sample of 2^20 outputs and on the default root's handle
    root2 = pow(root, 3, r)
    y2, _ = run(NTTClient.BITREV_INPUT, root2, d_in)
    ks = (np.arange(1 << 20, dtype=np.uint64) * 127 + 5) % n
    assert np.array_equal(y2.reshape(n, 32)[ks], y_nn.reshape(n, 32)[(3 * ks) % n])
    del y2
    # inverse of the same convention
    d_y = DeviceBuffer(0, 32 * n)
    d_y.upload(y_bb)
    z, _ = run(both | NTTClient.INVERSE, root, d_y)
    assert np.array_equal(z, x), "inverse(forward(x)) != x under the caller's convention"
    d_y.free(); d_in.free()
    print(f"[2^27 kernel ms] natural/natural {ms_nn:.2f}  bitrev-in {ms_bn:.2f}  bitrev-out {ms_nb:.2f}  both {ms_bb:.2f}")
