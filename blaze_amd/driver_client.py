"""Host-side mirror of src/driver_client (dclient.rs, dclient_cfg.rs): the operator trait and the
device handle.  In the reference `DriverClient` owns three XDMA character devices
(dclient.rs:50-59); here it names a HIP device ordinal - the transport is the C ABI of
include/blaze_hip.h, and the FPGA-shell management calls (DFX decouple, HBICAP, firewalls, CMS)
have nothing to manage, so they are accepted and do nothing."""
from __future__ import annotations

import abc
import enum
from dataclasses import dataclass
from typing import Generic, Optional, TypeVar

from ._lib import check, DriverClientError, lib  # noqa: F401  (re-exported)

T = TypeVar("T")
P = TypeVar("P")
I = TypeVar("I")
O = TypeVar("O")


class CardType(enum.Enum):
    """dclient_cfg.rs:1-3 has the single variant C1100; this build adds the GPU."""

    C1100 = 0
    MI355X = 1


@dataclass(frozen=True)
class DriverConfig:
    """dclient_cfg.rs:9-31.  The AXI base addresses of the FPGA shell have no meaning on a GPU;
    only the card type is kept so call sites read the same."""

    card: CardType = CardType.MI355X

    @staticmethod
    def driver_client_cfg(card: CardType) -> "DriverConfig":  # dclient_cfg.rs:26-31
        return DriverConfig(card)

    @staticmethod
    def driver_client_mi355x_cfg() -> "DriverConfig":  # analogue of driver_client_c1100_cfg (:34-47)
        return DriverConfig(CardType.MI355X)


class DriverClient:
    """dclient.rs:50-93.  `id` is the reference's slot id (/dev/xdma{id}_*): here the HIP ordinal."""

    def __init__(self, id: int | str = 0, cfg: Optional[DriverConfig] = None):
        self.id = int(id)
        self.cfg = cfg or DriverConfig.driver_client_mi355x_cfg()
        n = lib().blz_device_count()
        if not (0 <= self.id < n):
            # the reference unwraps the open() of the char device (utils.rs:74)
            raise DriverClientError(7, f"no HIP device with ordinal {self.id} ({n} visible)")

    # the card's HBM outlives the process that loaded it; GPU memory needs a holder process (include/blaze_hip.h)
    def arena_export(self, registry_path: str) -> None:
        check(lib().blz_arena_export(self.id, registry_path.encode()))

    def arena_attach(self, registry_path: str) -> None:
        check(lib().blz_arena_attach(self.id, registry_path.encode()))

    def arena_set_policy(self, drop_raw: bool) -> None:
        """include/blaze_hip.h blz_arena_set_policy: free an extent's raw bytes once its Montgomery copy is complete."""
        check(lib().blz_arena_set_policy(self.id, 1 if drop_raw else 0))

    # FPGA-shell management (dclient.rs:88-279): accepted, nothing to do on a GPU
    def reset(self) -> None:
        return None

    def initialize_cms(self) -> None:
        return None

    def reset_sensor_data(self) -> None:
        return None

    def setup_before_load_binary(self) -> None:
        return None

    def load_binary(self, _binary: bytes) -> None:
        return None

    def unblock_firewalls(self) -> None:
        return None


class DriverPrimitive(abc.ABC, Generic[T, P, I, O]):
    """dclient.rs:28-46: the seven-method operator contract every primitive client implements."""

    @abc.abstractmethod
    def __init__(self, ptype: T, dclient: DriverClient): ...

    @abc.abstractmethod
    def loaded_binary_parameters(self) -> list[int]: ...

    @abc.abstractmethod
    def initialize(self, param: P) -> None: ...

    @abc.abstractmethod
    def set_data(self, input: I) -> None: ...

    @abc.abstractmethod
    def start_process(self, param: Optional[int] = None) -> None: ...

    @abc.abstractmethod
    def wait_result(self) -> None: ...

    @abc.abstractmethod
    def result(self, param: Optional[int] = None) -> Optional[O]: ...
