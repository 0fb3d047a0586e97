"""GPU: run-to-run determinism of the kernels behind the two hardware-only wrong-result reports of round 1 (DESIGN.md
section 5b): the collapsed encoder-0 FTB (`aero_ftb_first_fwd`, now also its fused successor `aero_enc0_fwd`) and the
8-wave conv tiles with GroupNorm statistics in the epilogue (k_conv.h `aero_conv_glds8_kernel<., 32, true>` and the ring
kernel's statistics epilogue).  Every kernel runs 50 times on the same inputs: outputs bit-identical to the first run,
fp64 statistics equal to 1e-12, conv outputs within 2e-3 of fp32 torch.  (tools/dbg/determinism.py is the driver.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, cases):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dbg', 'determinism.py'), '--n', '50', '--cases', cases],
                       env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'ALL OK' in r.stdout
    return r.stdout


def test_fifty_runs_bit_identical_default_kernels():
    out = _run({}, 'enc0,ftb_first,conv_stats,conv')
    assert 'aero_conv_ring_kernel' in out


def test_fifty_runs_bit_identical_8wave_tiles_with_statistics():
    """AERO_CONV_RING=0 routes the wide convs to the 8-wave k_conv.h tiles, STATS instantiation included."""
    out = _run({'AERO_CONV_RING': '0'}, 'conv_stats,conv')
    assert 'aero_conv_glds8_kernel<4, 32, true>' in out and 'aero_conv_glds8_kernel<3, 32, true>' in out
