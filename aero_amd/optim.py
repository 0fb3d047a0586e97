"""Optimizer of the generator (reference train.py:83: `torch.optim.Adam(generator.parameters(), lr, betas=(0.9, beta2))`;
stepped by solver.py:602-605) as ONE fused pass over a flat parameter buffer (aero_adam_step, csrc/k_optim.h).

FlatAdam re-homes the parameters into a single contiguous fp32 buffer (each `p.data` becomes a view, as
DistributedDataParallel does with its buckets) and gives every parameter a gradient view into a second flat buffer, so

    * `zero_grad()` is one memset,
    * a gradient all-reduce over RCCL is one collective on the flat buffer (`distrib.sum_gradients`; the 1 / world-size factor
      is applied inside the fused step),
    * `step()` is one kernel launch: 28 bytes of HBM traffic per parameter.

The arithmetic and its order are torch's single-tensor Adam (no amsgrad, no weight decay): tests/test_optim.py compares
against torch.optim.Adam step by step.  There is no CPU path: without the HIP library `step()` raises.
"""
import ctypes as C

import torch

from . import _lib


class FlatAdam:
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, lib=None, model=None):
        """model: the `Aero` module these parameters belong to (optional).  The kernel writes the weights behind autograd's back, so
        version counters do not move: after every step `model.repack()` tells the device engine to re-pack them."""
        self.model = model
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('FlatAdam: no parameters')
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError('FlatAdam: parameters must be fp32 tensors on one device')
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.lib = lib
        sizes = [p.numel() for p in self.params]
        # every parameter starts on a 16-byte boundary of the flat buffer (4 floats): the kernel moves float4
        offs, n = [], 0
        for sz in sizes:
            offs.append(n)
            n += (sz + 3) // 4 * 4
        self.n = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._offs, self._sizes = offs, sizes
        with torch.no_grad():
            for p, o, sz in zip(self.params, offs, sizes):
                self.flat_p[o:o + sz].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + sz].view_as(p)
                p.grad = self.flat_g[o:o + sz].view_as(p)
        self.step_count = 0
        self.fresh = True                                        # flat_g is all zeros (nothing accumulated since zero_grad)
        if model is not None:                                    # the HIP backward writes its gradients straight into flat_g
            import weakref
            for m in (model, getattr(model, 'module', None)):    # (distrib.DataParallel: the wrapper and the generator inside)
                if isinstance(m, torch.nn.Module):
                    object.__setattr__(m, '_grad_sink', weakref.ref(self))

    def zero_grad(self, set_to_none=False):
        """(gradients stay views of the flat buffer: set_to_none is accepted for API compatibility and ignored)"""
        self._reattach(drop_stray_grads=True)                    # (first: a stray p.grad must not be copied into the zeroed buffer)
        self.flat_g.zero_()
        self.fresh = True

    def accepts(self, param_ptrs, offs, n, dev):
        """aero_amd.train.AeroFunction: are these (data pointers of) parameters exactly the ones of this optimizer, laid out in its
        flat buffers at these offsets, with their .grad still views of flat_g?  Then the backward may write into flat_g directly."""
        if n != self.n or len(param_ptrs) != len(self.params) or list(offs) != self._offs or self.flat_g.device != dev:
            return False
        esz, bp, bg = 4, self.flat_p.data_ptr(), self.flat_g.data_ptr()
        for p, ptr, o in zip(self.params, param_ptrs, self._offs):
            g = p.grad
            if ptr != bp + o * esz or g is None or g.data_ptr() != bg + o * esz:
                return False
        return True

    def _reattach(self, drop_stray_grads=False):
        """The kernel reads ONLY the flat buffers.  `module.zero_grad()` (set_to_none=True is torch's default), `model.to()` or
        `load_state_dict(assign=True)` replace `p.grad` / `p.data` by fresh tensors that are no longer views of them: copy such
        strays in and re-home the parameter, so that a step always follows the gradients autograd produced.  A gradient that is
        None counts as zero (torch.optim.Adam would skip that parameter; here its moments still decay)."""
        esz = self.flat_p.element_size()
        with torch.no_grad():
            for p, o, sz in zip(self.params, self._offs, self._sizes):
                if p.data_ptr() != self.flat_p.data_ptr() + o * esz:
                    if p.device != self.flat_p.device or p.dtype != torch.float32 or p.numel() != sz:
                        raise RuntimeError('FlatAdam: a parameter changed device, dtype or size after the optimizer was built')
                    self.flat_p[o:o + sz].copy_(p.detach().reshape(-1))
                    p.data = self.flat_p[o:o + sz].view_as(p)
                g = p.grad
                if g is None:
                    self.flat_g[o:o + sz].zero_()
                    p.grad = self.flat_g[o:o + sz].view_as(p)
                elif g.data_ptr() != self.flat_g.data_ptr() + o * esz:
                    if not drop_stray_grads:
                        self.flat_g[o:o + sz].copy_(g.detach().reshape(-1).to(self.flat_g.dtype))
                    p.grad = self.flat_g[o:o + sz].view_as(p)

    def step(self, grad_scale=1.0):
        lib = self.lib or _lib.load()
        if not self.flat_p.is_cuda and self.lib is None:
            raise RuntimeError('FlatAdam.step: parameters must live on the MI355X (no CPU path)')
        self._reattach()
        stream = torch.cuda.current_stream(self.flat_p.device).cuda_stream if self.flat_p.is_cuda else 0
        if self.flat_p.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture (aero_amd.train.CapturedStep): the step count must advance at every REPLAY, so the bias
            # corrections come from device memory, uploaded by before_replay()
            if self._bc is None:
                raise RuntimeError('FlatAdam: call prepare_capture() before capturing a step in a HIP graph')
            lib.call('aero_adam_step_dev', self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                     self.n, C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps), self._bc.data_ptr(),
                     C.c_float(grad_scale), stream)
            if self.model is not None and hasattr(self.model, 'repack'):
                self.model.repack()
            return
        self.step_count += 1
        lib.call('aero_adam_step', self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                 self.n, C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps), self.step_count,
                 C.c_float(grad_scale), stream)
        if self.model is not None and hasattr(self.model, 'repack'):
            self.model.repack()

    # ---- HIP-graph replay (aero_amd.train.CapturedStep)
    _bc = None

    _BC_SLOTS = 8

    def prepare_capture(self):
        """allocate the device-side bias-correction pair the captured step reads"""
        if self._bc is None:
            self._bc = torch.ones(2, dtype=torch.float32, device=self.flat_p.device)
            # a RING of pinned upload slots: replays are asynchronous, and a host that runs ahead would overwrite a single pinned pair
            # before the copy of the previous replay has executed.  A slot is reused only after the event behind its last copy is done.
            host = torch.ones(self._BC_SLOTS, 2, dtype=torch.float32)
            self._bc_host = host.pin_memory() if self.flat_p.is_cuda else host
            self._bc_events = [None] * self._BC_SLOTS
            self._bc_slot = 0

    def before_replay(self):
        """advance the step count and upload {1 - beta1^t, sqrt(1 - beta2^t)} for the replay that follows"""
        self.step_count += 1
        i = self._bc_slot
        self._bc_slot = (i + 1) % self._BC_SLOTS
        if self._bc_events[i] is not None:
            self._bc_events[i].synchronize()
        slot = self._bc_host[i]
        slot[0] = 1.0 - self.betas[0] ** self.step_count
        slot[1] = (1.0 - self.betas[1] ** self.step_count) ** 0.5
        self._bc.copy_(slot, non_blocking=True)
        if self.flat_p.is_cuda:
            ev = self._bc_events[i] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat_p.device))
            self._bc_events[i] = ev

    # ---- checkpoints: the schema of torch.optim.Adam.state_dict() (the reference stores it under the checkpoint's optimizer entry,
    # src/solver.py:111-118 / model_serializer.py), so checkpoints move both ways between the two optimizers
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o, sz) in enumerate(zip(self.params, self._offs, self._sizes)):
                state[i] = {'step': torch.tensor(float(self.step_count)),
                            'exp_avg': self.exp_avg[o:o + sz].view_as(p).clone(),
                            'exp_avg_sq': self.exp_avg_sq[o:o + sz].view_as(p).clone()}
        group = {'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False, 'maximize': False,
                 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'decoupled_weight_decay': False,
                 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        if 'param_groups' not in sd:                             # the flat layout this class wrote before round 3
            if sd['exp_avg'].numel() != self.n or sd['exp_avg_sq'].numel() != self.n:
                raise ValueError('FlatAdam.load_state_dict: flat moment buffers of the wrong size')
            self.step_count = int(sd['step'])
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])
            self.lr, self.betas, self.eps = float(sd['lr']), tuple(sd['betas']), float(sd['eps'])
            return
        groups = sd['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.params):
            raise ValueError('FlatAdam.load_state_dict: expected one parameter group with %d parameters' % len(self.params))
        g = groups[0]
        if g.get('amsgrad') or g.get('weight_decay'):
            raise ValueError('FlatAdam.load_state_dict: amsgrad / weight decay are not supported')
        self.lr, self.betas, self.eps = float(g['lr']), (float(g['betas'][0]), float(g['betas'][1])), float(g['eps'])
        state = sd['state']
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, (p, o, sz) in enumerate(zip(self.params, self._offs, self._sizes)):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            if tuple(st['exp_avg'].shape) != tuple(p.shape) or tuple(st['exp_avg_sq'].shape) != tuple(p.shape):
                raise ValueError(f'FlatAdam.load_state_dict: moment shape mismatch for parameter {i}')
            self.exp_avg[o:o + sz].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + sz].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('FlatAdam.load_state_dict: parameters with different step counts (one fused step count is kept)')
        self.step_count = steps.pop() if steps else 0
