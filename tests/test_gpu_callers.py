"""Boundary objects and callers with the device work done by the real HIP kernels,
against outputs recorded from the reference."""
import pytest

from . import _cases

pytestmark = pytest.mark.gpu


def test_match_to(hip_backend):
    assert _cases.check_match_to_golden() > 3000


def test_linked_adapters_c4(hip_backend):
    assert _cases.check_linked_c4() == 512


def test_adapter_cutter(hip_backend):
    assert _cases.check_cutter_golden() > 2500


def test_insert_adapter_cutter(hip_backend):
    assert _cases.check_insert_cutter_golden() > 1000


def test_reference_caller_kats(hip_backend):
    _cases.check_caller_kats()


def test_device_resident_adapters(hip_backend):
    assert _cases.check_device_resident_adapters() > 5000


def test_merge_overlapping(hip_backend):
    assert _cases.check_merge_golden(batch=True) == 1000
