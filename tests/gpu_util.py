"""Helpers shared by the -m gpu tests: they read like the reference's integration tests
(tests/integration_msm.rs:149-207): initialize -> start_process -> set_data -> wait_result -> result."""
import blaze_amd
from blaze_amd import DeviceBuffer
from blaze_amd.driver_client import DriverClient
from blaze_amd.ingo_msm import Curve, MSMClient, MSMInit, MSMInput, MSMParams, PointMemoryType


def msm_client(curve: str, pf: int = 1, mem=PointMemoryType.DMA, device: int = 0) -> MSMClient:
    return MSMClient(MSMInit(mem, pf == 8, Curve[curve]), DriverClient(device))


def run_msm(client: MSMClient, points, scalars, n: int, hbm=None) -> bytes:
    params = MSMParams(n, hbm)
    client.initialize(params)
    client.start_process()
    client.set_data(MSMInput(points, scalars, params))
    client.wait_result()
    return client.result().result


def synth(curve: str, n: int, pf: int = 1, start: int = 0, seed: int = 7, device: int = 0):
    cid = int(Curve[curve])
    ps = int(blaze_amd.lib().blz_point_size(cid))
    dp = DeviceBuffer(device, n * pf * ps)
    ds = DeviceBuffer(device, n * 32)
    blaze_amd._lib.check(blaze_amd.aux().blz_synth_points(device, cid, dp.ptr, n, pf, start))
    blaze_amd._lib.check(blaze_amd.aux().blz_synth_scalars(device, cid, ds.ptr, n, seed))
    return dp, ds
