"""tools/isa_lint.py rule W3: the MFMA -> VALU read hazard on a branch target that hipcc left unpadded in the first unrolled
LSTM step of round 5.  The rule is exercised on hand-written disassembly text: the pattern as it appeared in that build, the padded
fall-through path hipcc does emit, and a write (not a read) of the matrix destination at the target."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import isa_lint  # noqa: E402

HEAD = '0000000000001000 <_Z6kernelv>:\n'


def _asm(lines):
    out, addr = [HEAD], 0x1000
    for text in lines:
        out.append(f'\t{text:60s}// {addr:012X}: 00000000\n')
        addr += 4
    return ''.join(out)


def test_unpadded_taken_path_is_flagged():
    # v_mfma -> s_cbranch (taken, +3 dwords) -> v_mov reads the destination at once; the fall-through path carries the nops
    txt = _asm(['v_mfma_f32_16x16x32_f16 v[32:35], v[22:25], v[66:69], v[32:35]',
                's_cbranch_execnz 3',
                's_nop 7',
                's_nop 7',
                'v_mov_b64_e32 v[34:35], v[30:31]',
                'v_mov_b64_e32 v[28:29], v[32:33]',
                's_endpgm'])
    assert isa_lint.mfma_branch_hazards(txt) == {'_Z6kernelv': 1}


def test_padded_target_and_overwrites_are_not_flagged():
    padded = _asm(['v_mfma_f32_16x16x32_f16 v[32:35], v[22:25], v[66:69], v[32:35]',
                   's_cbranch_execnz 1',
                   's_nop 0',
                   's_nop 7',
                   'v_mov_b64_e32 v[28:29], v[32:33]',
                   's_endpgm'])
    assert isa_lint.mfma_branch_hazards(padded) == {}
    # the target first OVERWRITES the destination (a select), later instructions read the new value
    overwrite = _asm(['v_mfma_f32_16x16x16_f16 v[10:13], v[30:31], v[28:29], 0',
                      's_cbranch_execz 1',
                      's_nop 0',
                      'v_cndmask_b32_e64 v10, 0, 1, s[58:59]',
                      'v_cmp_ne_u32_e64 s[22:23], 1, v10',
                      's_endpgm'])
    assert isa_lint.mfma_branch_hazards(overwrite) == {}
    # no matrix instruction in front of the branch: nothing to flag
    plain = _asm(['v_add_f32_e32 v1, v2, v3', 's_cbranch_scc1 1', 's_nop 0', 'v_mov_b32_e32 v4, v1', 's_endpgm'])
    assert isa_lint.mfma_branch_hazards(plain) == {}


def test_packed_fp32_counter():
    txt = _asm(['v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]', 'v_pk_mul_f32 v[0:1], v[2:3], v[4:5]', 'v_pk_fma_f16 v0, v1, v2, v3', 's_endpgm'])
    assert len(isa_lint.PACKED_FP32.findall(txt)) == 2


def test_private_segment_rule():
    # kernel metadata as `llvm-readelf --notes` prints it: one kernel with a private segment, one without
    notes = ['''amdhsa.kernels:
  - .agpr_count:     0
    .name:           _Z1av
    .private_segment_fixed_size: 16
    .vgpr_count:     128
  - .agpr_count:     0
    .name:           _Z1bv
    .private_segment_fixed_size: 0
    .vgpr_count:     64
''']
    assert isa_lint.private_segments(notes) == {'_Z1av': 16}
    import re
    assert any(re.match(a, 'aero_conv_skinny_kernel<8>') for a in isa_lint.SCRATCH_ALLOWED)
    # the kernels of the measured paths are NOT on the allowed list
    for k in ('aero_conv_glds8_kernel<3, 32, true>', 'aero_lstm_ring_kernel<12, 1, 2, 3, 4, false>', 'aero_conv_ring_kernel<2, 4, 4, 3, 0>'):
        assert not any(re.match(a, k) for a in isa_lint.SCRATCH_ALLOWED)
