"""CPU: the gfx950 C-ABI library loads and exports every symbol include/aero_hip.h declares (no compute)."""
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'aero_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(aero_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_are_exported_by_the_gfx950_library():
    import __graft_entry__ as g
    from aero_amd import _lib
    if not os.path.exists(_lib.DEFAULT_LIB):
        g.build()
    lib = _lib.load()
    assert 'gfx950' in lib.version
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib.cdll, n), f'{n} declared in include/aero_hip.h but not exported'
    assert set(_lib.EXPORTS) == set(names)


def test_argument_errors_are_reported_not_thrown():
    """Error convention: negative return code + thread-local message; no compute, no GPU needed."""
    import ctypes as C
    from aero_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    rc = lib.cdll.aero_conv_fwd(C.byref(d), None)
    assert rc == -1 and b'conv' in lib.cdll.aero_last_error()
    rc = lib.cdll.aero_stft_fwd(None, 1, 10, 10, 512, 16, None, 256, None, 1, None, 1, None)
    assert rc < 0 and b'stft' in lib.cdll.aero_last_error()
    mp, kp = C.c_int32(), C.c_int32()
    assert lib.cdll.aero_lstm_geometry(96, C.byref(mp), C.byref(kp)) == 0 and (mp.value, kp.value) == (384, 96)
    assert lib.cdll.aero_lstm_geometry(200, C.byref(mp), C.byref(kp)) == -3
    # the LSTM kernel keeps per-thread offsets in 32 bits: a tensor beyond 2^31 output elements is refused up front (no launch, fake pointers)
    ld = _lib.LstmDesc()
    ld.xproj = ld.xbias = ld.whh = ld.out = 16
    ld.H, ld.nseq, ld.W = 48, 1 << 20, 400                       # 2^20 x 400 rows x 96 = 4e10 elements
    rc = lib.cdll.aero_lstm_fwd(C.byref(ld), None)
    assert rc == -3 and b'32-bit' in lib.cdll.aero_last_error()


def test_round6_entry_points_without_a_device():
    """the pitch query is host arithmetic; the stream / pitched entry points validate their arguments before touching the device"""
    import ctypes as C
    from aero_amd import _lib
    lib = _lib.load()
    p, t = C.c_int32(), C.c_int32()
    assert lib.cdll.aero_istft_pitch(512, 64, 501, C.byref(p), C.byref(t)) == 0 and (p.value, t.value) == (528, 12)   # headline geometry
    assert p.value * 8 % 128 == 0 and (64 * 0 + 4 + t.value) % 16 == 0            # rows are whole lines; a block's 16-frame groups (first at frame 4) start on one
    assert lib.cdll.aero_istft_pitch(1024, 256, 376, C.byref(p), C.byref(t)) == 0 and p.value % 16 == 0 and p.value >= t.value + 376
    assert lib.cdll.aero_istft_pitch(512, 50, 100, C.byref(p), C.byref(t)) == 0 and (p.value, t.value) == (100, 0)    # not the frame-run kernel: plain layout
    assert lib.cdll.aero_istft_pitch(512, 64, 501, None, C.byref(t)) == -1
    assert lib.cdll.aero_stream_create(0, None, 0, None) == -1 and b'stream_create' in lib.cdll.aero_last_error()
    assert lib.cdll.aero_stream_create(0, None, 4, C.byref(C.c_void_p())) == -1        # a mask length without a mask
    assert lib.cdll.aero_stream_destroy(None) == -1
    rc = lib.cdll.aero_istft_pitched_fwd(16, 1, 256, 501, 400, 0, 512, 64, 16, 16, 16, 100, None)
    assert rc == -1 and b'pitch' in lib.cdll.aero_last_error()                        # pitch < t_off + T
    rc = lib.cdll.aero_convtr_tail_finish_pitched(16, 16, None, None, None, 16, 1, 4, 501, 16, 2, 500, 12, None)
    assert rc == -1 and b'pitch' in lib.cdll.aero_last_error()
