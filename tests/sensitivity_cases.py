"""Do the goldens SEE every branch of the network?  ("what's weak" 1 / "next round" 3.)

At a fresh init the whole DConv branch (BLSTM, LocalState, Snake, both Conv1d) re-enters the trunk through LayerScale = 1e-3
(modules.py:138), and round 5 shipped -- for a day -- an LSTM kernel with 0.4 relative error at the op under which every tolerance-based
model test passed.  The `stress_*` goldens (oracle/stress.py: LayerScale ~ U(0.2, 1), live attention decay, wide Snake
spread; outputs of the REFERENCE on those weights) exist to un-hide that branch.  This file demonstrates that they do: one deliberate
fault at a time is planted in the product's launch sequence -- at the boundary between the engine and a kernel, so the fault stands for
"this kernel is wrong in this way" -- and the stress golden must then FAIL its 1e-3 bar on the complex spectrogram, while the unfaulted
run passes it.  The same table runs on the CPU emulator (tests/test_sensitivity.py, `-m "not gpu"`) and once on the MI355X.

A fault is `install(model, engine)`; it edits `engine.ops` (tensor-level wrappers over the C ABI: reference modules.py:32-65 BLSTM,
:94-127 LocalState, :221-249 DConv, :258-276 ScaledEmbedding, :304-325 FTB, snake.py:67) or the packed layer tables -- never the kernels.
"""
import torch

from aero_amd._lib import ACT_GLU, ACT_NONE, ACT_SNAKE

BAR = 1e-3            # north_star: relative L2 on the complex spectrogram


def _wrap(ops, name, fn):
    """shadow the bound method `ops.<name>` with fn(orig, *args, **kw)"""
    orig = getattr(ops, name)
    setattr(ops, name, lambda *a, **k: fn(orig, *a, **k))


# ---- BLSTM (k_lstm.h) ----------------------------------------------------------------------------------------------------------------
def lstm_output_zero(m, eng):
    """the recurrent kernel writes nothing (both layers)"""
    def f(orig, *a, **k):
        orig(*a, **k)
        a[11].zero_()
    _wrap(eng.ops, 'lstm', f)


def lstm_directions_swapped(m, eng):
    """layer 2 writes the backward direction's h into the forward half and vice versa"""
    def f(orig, *a, **k):
        orig(*a, **k)
        H, out_mode, out = a[3], a[7], a[11]
        if k.get('x') is not None and k['x'].shape[-1] == 2 * H:          # (layer 2: its input is layer 1's 2H-wide output)
            out.copy_(torch.cat([out[..., H:], out[..., :H]], -1))
    _wrap(eng.ops, 'lstm', f)


def lstm_stitch_off_by_one(m, eng):
    """the stitch map (models/utils.py:22-35, modules.py:52-62) places the frames with a stride of S - 1 while they were cut with S.
    (S - 1 for BOTH the cut and the stitch is not a fault: every kept step has >= 50 steps of warm-up, after which the recurrent state has
    forgotten where its frame began to below fp16 resolution -- the stitched output is BIT-identical, measured on the emulator.)"""
    def f(orig, *a, **k):
        a = list(a)
        if a[8] > 1 and a[7] == 1:                                         # nframes > 1 (T > 200), out_mode 1 = the stitching layer
            a[9] -= 1
        orig(*a, **k)
    _wrap(eng.ops, 'lstm', f)


def lstm_skip_dropped(m, eng):
    """the BLSTM's Linear without `+ x` (modules.py:64)"""
    lin = {id(L.get('lstm_lin_pw')) for P in eng.P.values() if isinstance(P, dict) for L in P.get('dconv', []) if 'lstm' in L}
    lin |= {id(L.get('lstm_lin')) for P in eng.P.values() if isinstance(P, dict) for L in P.get('dconv', []) if 'lstm' in L}

    def f(orig, spec, *a, **k):
        if id(spec) in lin:
            k['res'] = None
        return orig(spec, *a, **k)
    _wrap(eng.ops, 'pw', f)
    _wrap(eng.ops, 'conv', f)


# ---- LocalState (k_attn.h) -----------------------------------------------------------------------------------------------------------
def localstate_decay_dropped(m, eng):
    """the learned distance decay (modules.py:112-117) contributes nothing: sigmoid(query_decay) = 0"""
    def f(orig, qkvd, R, T, Cc, heads, ndecay):
        qkvd = qkvd.clone()
        qkvd[..., 3 * Cc:] = -30.0
        return orig(qkvd, R, T, Cc, heads, ndecay)
    _wrap(eng.ops, 'localstate', f)


def localstate_self_kill_dropped(m, eng):
    """`dots[t, t] = -100` (modules.py:119) missing is not expressible at this boundary; the nearest kernel fault that is: keys shifted by
    one time step (the diagonal lands on a neighbour)"""
    def f(orig, qkvd, R, T, Cc, heads, ndecay):
        qkvd = qkvd.clone()
        k = qkvd[..., Cc:2 * Cc].clone()
        qkvd[..., 1:, Cc:2 * Cc] = k[..., :-1, :]
        return orig(qkvd, R, T, Cc, heads, ndecay)
    _wrap(eng.ops, 'localstate', f)


def localstate_heads_rotated(m, eng):
    """the output of head h stored in the channels of head h + 1"""
    def f(orig, qkvd, R, T, Cc, heads, ndecay):
        out = orig(qkvd, R, T, Cc, heads, ndecay)
        return torch.roll(out, Cc // heads, dims=-1)
    _wrap(eng.ops, 'localstate', f)


# ---- Snake / DConv (k_norm.h, k_dconv.h, k_pw.h) -------------------------------------------------------------------------------------
def snake_skipped(m, eng):
    """x + sin^2(a x) / a evaluated as x -- in the GroupNorm + activation kernel (encoder 2-3) and in the row kernel (encoder 0-1)"""
    def f(orig, x, G, per_row, gamma, beta, act, **k):
        return orig(x, G, per_row, gamma, beta, ACT_NONE if act == ACT_SNAKE else act, **k)
    _wrap(eng.ops, 'norm_act', f)

    def g(orig, x, layers, act, F, **k):
        return orig(x, layers, ACT_NONE, F, **k)
    _wrap(eng.ops, 'dconv_row', g)


def snake_skipped_deep_layers_only(m, eng):
    """... only where the branch also carries BLSTM + LocalState (encoder 2-3)"""
    def f(orig, x, G, per_row, gamma, beta, act, **k):
        return orig(x, G, per_row, gamma, beta, ACT_NONE if act == ACT_SNAKE else act, **k)
    _wrap(eng.ops, 'norm_act', f)


def dconv_dilations_swapped(m, eng):
    """layer 0 runs with dilation 2 and layer 1 with dilation 1 (modules.py:200: dilation = 2 ** d)"""
    for P in eng.P.values():
        if not isinstance(P, dict) or len(P.get('dconv', [])) != 2:
            continue
        a, b = P['dconv']
        a['conv1'].dt, b['conv1'].dt = b['conv1'].dt, a['conv1'].dt
        if 'row' in a and 'row' in b:
            a['row']['dilation'], b['row']['dilation'] = b['row']['dilation'], a['row']['dilation']


def dconv_residual_dropped(m, eng):
    """`x = x + layer(x)` (modules.py:247) as `x = layer(x)` in the tails behind BLSTM / LocalState"""
    tails = {id(L.get(n)) for P in eng.P.values() if isinstance(P, dict) for L in P.get('dconv', []) for n in ('pw2', 'conv2_glu', 'conv2_glu_pad')}
    tails.discard(id(None))

    def f(orig, spec, *a, **k):
        if id(spec) in tails and k.get('res') is not None:
            k['res'] = torch.zeros_like(k['res'])
        return orig(spec, *a, **k)
    _wrap(eng.ops, 'pw', f)
    _wrap(eng.ops, 'conv', f)


def layer_scale_ignored(m, eng):
    """LayerScale (modules.py:130-142) applied as 1"""
    def f(orig, *a, **k):
        if k.get('layer_scale') is not None:
            k['layer_scale'] = torch.ones_like(k['layer_scale'])
        return orig(*a, **k)
    _wrap(eng.ops, 'pw', f)

    def g(orig, spec, *a, **k):
        if k.get('stat') and k['stat'].get('layer_scale') is not None:
            k['stat'] = dict(k['stat'], layer_scale=torch.ones_like(k['stat']['layer_scale']))
        return orig(spec, *a, **k)
    _wrap(eng.ops, 'conv', g)
    _wrap(eng.ops, 'norm_act', f)
    # (the row kernel of encoder 0-1 keeps LayerScale inside its constant table; the deep layers are what this fault covers)


def dconv_glu_halves_swapped(m, eng):
    """GLU of the DConv tail (modules.py:213) gating with the wrong half: a * sigmoid(b) computed as b * sigmoid(a)"""
    with torch.no_grad():
        for name, p in m.named_parameters():
            if '.dconv.layers.' in name and ('.conv2.0.' in name or '.conv2.1.' in name) and 'encoder.3' in name:
                h = p.shape[0] // 2
                p.copy_(torch.cat([p[h:], p[:h]], 0))


# ---- ScaledEmbedding, FTB -----------------------------------------------------------------------------------------------------------
def freq_embedding_dropped(m, eng):
    """`x = x + freq_emb_scale * emb` (aero.py:475-480) omitted"""
    def f(orig, *a, **k):
        k['post_add'] = None
        return orig(*a, **k)
    _wrap(eng.ops, 'pw', f)
    _wrap(eng.ops, 'conv', f)


def ftb_gate_skipped(m, eng):
    """FTB (modules.py:311-316): the attention multiply `c2 * x` with c2 = 1"""
    gates = {id(P.get(n)) for P in eng.P.values() if isinstance(P, dict) for n in ('ftb0_g',)}
    gates.discard(id(None))
    assert gates, 'the model has no collapsed first-layer FTB'

    def f(orig, spec, src0, *a, **k):
        if id(spec) in gates:
            src0 = torch.ones_like(src0)
        return orig(spec, src0, *a, **k)
    _wrap(eng.ops, 'conv', f)


def ftb_freq_fc_transposed(m, eng):
    """freq_fc (modules.py:318) applied with W instead of W^T"""
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith('freq_attn_block.freq_fc.weight'):
                p.copy_(p.t().clone())


# ---- trunk (sanity: the obvious ones must of course be seen too) ---------------------------------------------------------------------
def encoder_glu_halves_swapped(m, eng):
    """GLU behind an encoder's rewrite conv (aero.py:133) with the halves exchanged"""
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.startswith('encoder.2.rewrite.') or name.startswith('encoder.2.norm2.'):
                h = p.shape[0] // 2
                p.copy_(torch.cat([p[h:], p[:h]], 0))


# name -> (install, clip length of small_io.npz to run -- or the modules.npz vector 'tag:case' whose op-level bar has to catch it --, what it guards)
FAULTS = {
    'lstm_output_zero': (lstm_output_zero, 800, 'k_lstm.h: aero_lstm_ring_kernel writes h'),
    'lstm_directions_swapped': (lstm_directions_swapped, 800, 'k_lstm.h: direction -> output half'),
    'lstm_stitch_off_by_one': (lstm_stitch_off_by_one, 'blstm:framed', 'k_lstm.h: frame / stitch index map (T > 200)'),
    'lstm_stitch_off_by_one_end_to_end': (lstm_stitch_off_by_one, 2003, 'k_lstm.h: frame / stitch index map (T > 200), end to end'),
    'lstm_skip_dropped': (lstm_skip_dropped, 800, 'k_pw.h / k_conv.h: Linear + skip behind the BLSTM'),
    'localstate_decay_dropped': (localstate_decay_dropped, 800, 'k_attn.h: decay slope from query_decay'),
    'localstate_keys_shifted': (localstate_self_kill_dropped, 'localstate:y', 'k_attn.h: key index / diagonal'),
    'localstate_keys_shifted_end_to_end': (localstate_self_kill_dropped, 800, 'EXPECTED TO SURVIVE the end-to-end golden (documented blind spot; guarded at the op level)'),
    'localstate_heads_rotated': (localstate_heads_rotated, 800, 'k_attn.h: head -> channel map of the output'),
    'snake_skipped': (snake_skipped, 800, 'k_norm.h (AERO_ACT_SNAKE), k_dconv.h'),
    'snake_skipped_deep_layers_only': (snake_skipped_deep_layers_only, 800, 'k_norm.h (AERO_ACT_SNAKE) in encoder 2-3'),
    'dconv_dilations_swapped': (dconv_dilations_swapped, 800, 'k_conv.h / k_dconv.h: dilated Conv1d taps'),
    'dconv_residual_dropped': (dconv_residual_dropped, 800, 'k_pw.h: DConv tail + skip'),
    'layer_scale_ignored': (layer_scale_ignored, 800, 'k_pw.h / k_norm.h: LayerScale in the DConv tail'),
    'dconv_glu_halves_swapped': (dconv_glu_halves_swapped, 800, 'k_pw.h: GLU half selection in the DConv tail (deepest encoder)'),
    'freq_embedding_dropped': (freq_embedding_dropped, 800, 'k_pw.h / k_conv.h: post_add epilogue'),
    'ftb_gate_skipped': (ftb_gate_skipped, 800, 'k_enc0.h / k_ftb.h: FTB attention gate'),
    'ftb_freq_fc_transposed': (ftb_freq_fc_transposed, 800, 'k_ftb.h: freq_fc orientation'),
    'encoder_glu_halves_swapped': (encoder_glu_halves_swapped, 800, 'k_pw.h / k_norm.h: GLU behind the rewrite conv (trunk)'),
}


# Faults the END-TO-END stress golden does not see (measured: 3.2e-4 with the fault against 2.5e-4 clean, bar 1e-3): one-step index slips
# inside the branch, whose effect on the spectrogram is below the bar even at LayerScale O(1).  Their guards are the reference's own
# module vectors (tests/golden/modules.npz, bar 1e-3, clean 3e-4): 8.1e-3 and 6.4e-2 with the fault.
SURVIVORS = {'lstm_stitch_off_by_one_end_to_end': 'lstm_stitch_off_by_one', 'localstate_keys_shifted_end_to_end': 'localstate_keys_shifted'}


class _OpsProxy:
    """what a fault sees as `engine.ops` when the patch has to reach every Ops object a test case builds: attribute writes go to the CLASS"""

    def __init__(self, cls, saved):
        object.__setattr__(self, '_cls', cls)
        object.__setattr__(self, '_saved', saved)

    def __getattr__(self, name):
        f = getattr(self._cls, name)
        return lambda *a, **k: f(self._self, *a, **k)

    def __setattr__(self, name, fn):
        self._saved.setdefault(name, getattr(self._cls, name))
        proxy = self

        def method(ops_self, *a, **k):
            object.__setattr__(proxy, '_self', ops_self)
            return fn(*a, **k)
        setattr(self._cls, name, method)


def run_module(lib, fault, device='cpu'):
    """-> rel-L2 of the module-level vector named by the fault (tests/golden/modules.npz, the REFERENCE's module on seeded inputs), with the
    fault patched into the Ops class for the duration of the case (None as `fault[0]`: clean)"""
    import op_cases as oc
    from aero_amd.engine import Ops
    install, where = fault
    tag, case = where.split(':')
    saved = {}
    try:
        if install is not None:
            class _E:
                ops = _OpsProxy(Ops, saved)
                P = {}
            install(None, _E)
        return oc.case_module_golden(lib, device, tag)[case]
    finally:
        for name, f in saved.items():
            setattr(Ops, name, f)


def run(meta, lib, fault, device='cpu'):
    """-> (rel-L2 of the complex spectrogram against the reference's stress golden, of the waveform) with `fault` installed (None: clean)"""
    from aero_amd.engine import HipEngine
    from conftest import build_model, load_npz, rel_l2
    if fault and isinstance(FAULTS[fault][1], str):
        e = run_module(lib, FAULTS[fault][:2], device)
        return e, e
    L = FAULTS[fault][1] if fault else 800
    m = build_model(meta, 'stress_small')
    if device != 'cpu':
        m = m.to(device)
    eng = HipEngine(m, lib=lib)
    object.__setattr__(m, '_engine', eng)
    x = torch.from_numpy(load_npz('small_io.npz')[f'x_{L}'])[:1].to(device)
    io = load_npz('stress_small_io.npz')
    if fault:
        install = FAULTS[fault][0]
        if install in (dconv_glu_halves_swapped, ftb_freq_fc_transposed, encoder_glu_halves_swapped):
            install(m, eng)                                    # weight-level: before the pack
            eng._prepare(x.device)
        else:
            eng._prepare(x.device)                             # table / wrapper level: after the pack
            install(m, eng)
    with torch.no_grad():
        y, s = m(x, return_spec=True)
    return rel_l2(s.cpu(), io[f'spec_{L}'][:1]), rel_l2(y.cpu(), io[f'y_{L}'][:1])
