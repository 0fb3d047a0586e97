"""CPU (emulator): every planted fault of tests/sensitivity_cases.py is caught by a named golden of the REFERENCE, the clean run passes."""
import pytest

from aero_amd import _lib
import sensitivity_cases as S


@pytest.fixture(scope='module')
def emu():
    from emu.build_emu import build
    return _lib.load(build())


def test_clean_run_passes_the_bar(emu, meta):
    e_spec, e_wav = S.run(meta, emu, None)
    assert e_spec < S.BAR and e_wav < 5e-3, (e_spec, e_wav)


@pytest.mark.parametrize('fault', [f for f in S.FAULTS if f not in S.SURVIVORS])
def test_fault_is_caught(emu, meta, fault):
    """the stress golden (or, for the two one-step index slips, the reference's op-level module vector) FAILS its 1e-3 bar with the fault in"""
    e_spec, _ = S.run(meta, emu, fault)
    assert e_spec > S.BAR, f'{fault} ({S.FAULTS[fault][2]}) survived: {e_spec:.3e} <= {S.BAR}'


@pytest.mark.parametrize('fault', sorted(S.SURVIVORS))
def test_documented_blind_spots_of_the_end_to_end_golden(emu, meta, fault):
    """one-step index slips inside the DConv branch move the end-to-end figure by < 1e-4: below what a 1e-3 bar can see.  Recorded, not
    hidden -- and their op-level guards (the same fault against the reference's module vector) are asserted by test_fault_is_caught."""
    e_spec, _ = S.run(meta, emu, fault)
    print(f'{fault}: end-to-end {e_spec:.3e} (bar {S.BAR}); op-level guard: {S.SURVIVORS[fault]}')
    assert S.SURVIVORS[fault] in S.FAULTS and isinstance(S.FAULTS[S.SURVIVORS[fault]][1], str)
    assert e_spec == e_spec                                      # (a finite figure; when a tightened golden starts to catch it, move it up)
