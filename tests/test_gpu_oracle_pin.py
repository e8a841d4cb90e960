"""The checker re-pinned ON THE GPU BOX: the `-m gpu` tier compiles oracle/align_oracle.c with that box's gcc and
uses it to judge the HIP kernels, so the same tier re-runs the oracle against the fixtures the reference generated
(tests/golden/make_*golden.py) -- tests/test_oracle_golden.py, which the CPU tier runs, under the gpu mark.
(Round-3 verdict, evidence hygiene: "the driver's box re-pins the checker it uses".)"""
import pytest

from . import test_oracle_golden as G

pytestmark = pytest.mark.gpu

_PINS = [name for name in dir(G) if name.startswith("test_")]


def test_pins_exist():
    assert len(_PINS) >= 8, _PINS


@pytest.mark.parametrize("name", _PINS)
def test_oracle_pin(name, oracle):
    fn = getattr(G, name)
    args = fn.__code__.co_varnames[:fn.__code__.co_argcount]
    assert set(args) <= {"oracle"}, "pin %s needs fixtures this wrapper does not pass: %r" % (name, args)
    fn(oracle) if args else fn()
