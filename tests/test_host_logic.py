"""Host-side mirror of the reference's types (no GPU needed)."""
import dataclasses

from blaze_amd.driver_client import CardType, DriverConfig, DriverPrimitive
from blaze_amd.ingo_msm import (PRECOMPUTE_FACTOR, PRECOMPUTE_FACTOR_BASE, Curve, MSMClient, MSMConfig, MSMInit,
                                MSMInput, MSMParams, MSMResult, PointMemoryType)
from blaze_amd.ingo_ntt import NTT, NTT_LOG_SIZE, NTTClient, NTTInput, NttInit


def test_enum_orders_match_reference():
    assert [c.name for c in Curve] == ["BLS377", "BLS381", "BN254"]          # msm_cfg.rs:4-8
    assert [m.name for m in PointMemoryType] == ["HBM", "DMA"]               # msm_cfg.rs:11-14
    assert (PRECOMPUTE_FACTOR_BASE, PRECOMPUTE_FACTOR) == (1, 8)             # msm_api.rs:39-40
    assert NTT_LOG_SIZE == 27                                                # ntt_data.rs:65


def test_type_fields_match_reference():
    assert [f.name for f in dataclasses.fields(MSMInit)] == ["mem_type", "is_precompute", "curve"]
    assert [f.name for f in dataclasses.fields(MSMParams)] == ["nof_elements", "hbm_point_addr"]
    assert [f.name for f in dataclasses.fields(MSMInput)] == ["points", "scalars", "params"]
    assert [f.name for f in dataclasses.fields(MSMResult)] == ["result", "result_label"]
    assert [f.name for f in dataclasses.fields(NTTInput)] == ["buf_host", "data"]
    assert dataclasses.fields(NttInit) == ()
    assert NTT.Ntt is not None


def test_clients_implement_the_seven_method_trait():
    methods = {"__init__", "loaded_binary_parameters", "initialize", "set_data", "start_process", "wait_result", "result"}
    assert methods <= set(DriverPrimitive.__abstractmethods__)                # dclient.rs:28-46
    for cls in (MSMClient, NTTClient):
        assert issubclass(cls, DriverPrimitive)
        assert not getattr(cls, "__abstractmethods__", None)
    for extra in ("task_label", "nof_elements", "is_msm_engine_ready", "load_data_to_hbm", "get_data_from_hbm", "get_api"):
        assert hasattr(MSMClient, extra)                                      # msm_api.rs:277-331


def test_msm_config_sizes():
    assert MSMConfig.msm_cfg(Curve.BLS381, PointMemoryType.DMA) == MSMConfig(144, 96, 32)
    assert MSMConfig.msm_cfg(Curve.BN254, PointMemoryType.HBM) == MSMConfig(96, 64, 32)
    assert DriverConfig.driver_client_cfg(CardType.MI355X).card is CardType.MI355X
