"""FlatAdam (aero_amd/optim.py, aero_adam_step) against torch.optim.Adam -- the optimizer train.py:83 builds for the generator --
step by step on the same gradients.  CPU: through the kernel emulator (test double); GPU: the real library."""
import pytest
import torch

from aero_amd import _lib
from aero_amd.optim import FlatAdam


def _run(lib, dev, steps=5):
    g = torch.Generator().manual_seed(7)
    shapes = [(48, 2, 1, 1), (48,), (5, 48, 1, 1), (13,), (96, 48, 8, 1), (3,)]           # sizes not multiples of 4 on purpose
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    opt_ref = torch.optim.Adam(ref, lr=3e-4, betas=(0.9, 0.999))
    opt = FlatAdam(mine, lr=3e-4, betas=(0.9, 0.999), lib=lib)
    for it in range(steps):
        opt_ref.zero_grad()
        opt.zero_grad()
        for p, q in zip(ref, mine):
            gr = torch.randn(*p.shape, generator=g) * (10.0 ** (it - 2))
            p.grad = gr.clone()
            q.grad.copy_(gr.to(dev))                                                         # gradients are views of the flat buffer
        opt_ref.step()
        opt.step()
        for p, q in zip(ref, mine):
            assert torch.allclose(q.detach().cpu(), p.detach(), rtol=2e-6, atol=1e-9), (it, (q.detach().cpu() - p.detach()).abs().max())
    # the parameters are still views of one buffer, and the state round-trips
    assert all(q.data.untyped_storage().data_ptr() == opt.flat_p.untyped_storage().data_ptr() for q in mine)
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    assert opt.step_count == steps


def test_flat_adam_matches_torch_on_the_emulator():
    from emu.build_emu import build
    _run(_lib.load(build()), 'cpu')


def test_grad_scale_is_the_mean_over_ranks():
    from emu.build_emu import build
    lib = _lib.load(build())
    a = [torch.nn.Parameter(torch.ones(9))]
    b = [torch.nn.Parameter(torch.ones(9))]
    oa, ob = FlatAdam(a, lib=lib), FlatAdam(b, lib=lib)
    a[0].grad.fill_(4.0)
    b[0].grad.fill_(1.0)
    oa.step(grad_scale=0.25)                                                                 # a summed gradient of 4 ranks
    ob.step()
    assert torch.equal(a[0].detach(), b[0].detach())


def test_step_invalidates_the_engines_packed_weights():
    class M:
        n = 0

        def repack(self):
            self.n += 1
    from emu.build_emu import build
    m = M()
    opt = FlatAdam([torch.nn.Parameter(torch.ones(8))], lib=_lib.load(build()), model=m)
    opt.step()
    opt.step()
    assert m.n == 2


def test_cpu_parameters_without_the_emulator_fail_loudly():
    with pytest.raises((RuntimeError, ImportError, OSError)):
        FlatAdam([torch.nn.Parameter(torch.ones(4))]).step()


@pytest.mark.gpu
def test_flat_adam_matches_torch_on_the_mi355x():
    _run(None, 'cuda', steps=8)


def test_stray_gradients_and_torch_checkpoints():
    """(i) after `module.zero_grad()` (set_to_none) autograd allocates fresh .grad tensors that are not views of the flat
    buffer -- step() must still follow them; (ii) the checkpoint schema is torch.optim.Adam's, both ways."""
    from emu.build_emu import build
    lib = _lib.load(build())
    g = torch.Generator().manual_seed(3)
    shapes = [(6, 5), (7,), (3, 2, 2)]
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    opt_ref = torch.optim.Adam(ref, lr=1e-2)
    opt = FlatAdam(mine, lr=1e-2, lib=lib)
    for it in range(3):
        for p, q in zip(ref, mine):
            gr = torch.randn(*p.shape, generator=g)
            p.grad = gr.clone()
            if it == 1:
                q.grad = None                           # what zero_grad(set_to_none=True) leaves behind ...
            q.grad = gr.clone() if it else q.grad.copy_(gr)   # ... and what autograd then allocates: a stray tensor
        if it == 2:
            mine[0].data = mine[0].data.clone()         # a re-homed parameter (model.to(), load_state_dict(assign=True))
        opt_ref.step()
        opt.step()
        for p, q in zip(ref, mine):
            assert torch.allclose(q.detach(), p.detach(), rtol=2e-6, atol=1e-9)
            assert q.grad.untyped_storage().data_ptr() == opt.flat_g.untyped_storage().data_ptr()
    # torch -> flat
    opt2 = FlatAdam([torch.nn.Parameter(p.detach().clone()) for p in ref], lr=5.0, lib=lib)
    opt2.load_state_dict(opt_ref.state_dict())
    assert opt2.step_count == 3 and opt2.lr == 1e-2
    assert torch.allclose(opt2.exp_avg, opt.exp_avg, rtol=1e-5, atol=1e-7)
    # flat -> torch
    fresh = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    opt3 = torch.optim.Adam(fresh, lr=1.0)
    opt3.load_state_dict(opt.state_dict())
    for p, q, r in zip(ref, mine, fresh):
        gr = torch.randn(*p.shape, generator=g)
        p.grad, r.grad = gr.clone(), gr.clone()
        q.grad.copy_(gr)
    opt_ref.step()
    opt3.step()
    opt.step()
    for p, q, r in zip(ref, mine, fresh):
        assert torch.allclose(r.detach(), p.detach(), rtol=2e-6, atol=1e-7), (r.detach() - p.detach()).abs().max()
        assert torch.allclose(q.detach(), p.detach(), rtol=2e-6, atol=1e-7), (q.detach() - p.detach()).abs().max()
    with pytest.raises(ValueError):
        bad = opt_ref.state_dict()
        bad['state'][0]['exp_avg'] = torch.zeros(2)
        opt2.load_state_dict(bad)
