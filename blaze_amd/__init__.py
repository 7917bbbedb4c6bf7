"""blaze_amd: MI355X (gfx950) device path for blaze's MSM / NTT primitives behind the reference's
DriverPrimitive operator surface.  The product is blaze_amd/lib/libblaze_hip.so (hand-written HIP,
C ABI in include/blaze_hip.h); the modules here mirror the reference's host-side types:

    driver_client  <- src/driver_client   (DriverClient, DriverPrimitive, DriverConfig)
    ingo_msm       <- src/ingo_msm        (MSMClient, MSMInit, MSMParams, MSMInput, MSMResult, Curve, ...)
    ingo_ntt       <- src/ingo_ntt        (NTTClient, NTT, NttInit, NTTInput)
"""
from . import driver_client, ingo_msm, ingo_ntt  # noqa: F401
from ._lib import DeviceBuffer, DriverClientError, HostBuffer, aux, lib  # noqa: F401

__all__ = ["driver_client", "ingo_msm", "ingo_ntt", "DeviceBuffer", "DriverClientError", "HostBuffer", "aux", "lib"]
