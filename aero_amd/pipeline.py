"""Several batches in flight: the serving loop of the drop-in (the reference's predict.py:76-80 / enhance.py:11-15 run one forward at a
time and wait for it).

One forward of the model is a chain of ~90 launches of which a fifth (by time) are latency-bound -- the recurrent LSTM kernels keep a
quarter of the CUs busy for 200 dependent steps, the attention and the 1-D FTB convs not many more.  `HipEngine.forward` already cuts ONE
batch into two halves on two HIP streams, but the halves run the same kernel sequence from the same start, so mostly the same kernel meets
itself.  Batches that do not depend on each other can do better: `BatchPipeline` enqueues batch i on stream i mod depth without waiting
for batch i - 1, the streams drift apart, and the recurrent phase of one batch runs under the MFMA / bandwidth-bound phases of the
others.  Measured on the MI355X at the bench workload (64 clips per batch, tools/bench_pipelined.py): 10.30 ms per batch one at a time on
one stream, 10.00 ms with the two half-batch streams, 9.2 ms with three batches in flight; results are bit-identical to the single-stream
forward of each batch (every batch runs the single-stream kernel sequence, only on its own stream).

Stream semantics are PyTorch's: `submit` orders the batch behind everything the caller's current stream has issued so far (its input is
ready), `result` makes the caller's current stream wait for that batch.  Collect results a few submissions late (or call `drain`), not
right after each `submit` -- a `result` immediately behind its `submit` serialises the batches again.
"""
import os
import weakref

import torch


class _Ticket:
    __slots__ = ('out', 'event', 'stream', 'host', 'keep', '__weakref__')

    def __init__(self, out, event, stream, host=False, keep=None):
        self.out, self.event, self.stream, self.host, self.keep = out, event, stream, host, keep


def _tensors(out):
    if torch.is_tensor(out):
        yield out
    elif isinstance(out, (tuple, list)):
        for o in out:
            yield from _tensors(o)


def parse_schedule(text):
    """'stages=mmlldddd;l=prio:-1;d=mask:0:224' -> {'stages': 'mmlldddd', 'l': ('prio', -1), 'd': ('mask', 0, 224)}.
    `stages`: one letter per stage of the forward (encoder layers 1-4, decoder layers 5-8; the STFT / normalisation run with the first, the
    iSTFT with the last); 'm' is the batch's own stream of the ring.  Every other letter names a stream KIND of which each slot of the ring
    gets its own object: 'prio:p' a stream of dispatch priority p, 'mask:lo:hi' a stream restricted to the CUs [lo, hi), 'plain' a
    default stream.  A kind marked 'shared' (e.g. 'l=prio:-1:shared') is ONE stream for all slots -- TIMING EXPERIMENTS ONLY: the per-slot
    pairing that keeps the caching allocator's per-stream pools safe (see submit) does not hold for it.  'stagger=s' / 'lstm=<kind>': see
    BatchPipeline.  These schedules are the round-6 experiments of DESIGN.md 4.4 (priorities and CU masks measured NOT to pay); the serving
    loop's default is plain streams with `waits`."""
    out = {}
    for item in text.split(';'):
        if not item.strip():
            continue
        k, v = item.split('=')
        k, v = k.strip(), v.strip()
        if k in ('stages', 'lstm'):
            out[k] = v
            continue
        if k == 'stagger':
            out[k] = float(v)
            continue
        parts = v.split(':')
        shared = parts[-1] == 'shared'
        if shared:
            parts = parts[:-1]
        out[k] = (parts[0],) + tuple(int(x) for x in parts[1:]) + (('shared',) if shared else ())
    return out


class BatchPipeline:
    """pipe = BatchPipeline(model, depth=3);  t = pipe.submit(x);  ...;  y = pipe.result(t)   (inference, model.eval()).
    `waits` ('auto' = the default): event waits between consecutive batches that keep the streams out of phase.
    `schedule` (a dict or the text form of parse_schedule): which HIP stream each stage of a batch is issued on -- experiments, DESIGN.md 4.4a/b."""

    def __init__(self, model, depth=3, schedule=None, stagger=0, waits='auto'):
        self.model = model
        self.depth = max(1, int(depth))
        self.schedule = parse_schedule(schedule) if isinstance(schedule, str) else schedule
        # stagger = s > 0: batch i + 1 does not START before batch i has finished stage s of its forward (1-4: encoder layers, 5-8: decoder
        # layers).  Left alone, the streams do not drift into an even spacing: two of the three lock onto each other and run the same kernel
        # families side by side (completion gaps 13.6 / 13.3 / 0.3 ms instead of 8.7 / 8.7 / 8.7; tools/dbg/pipeline_fill_drain.py)
        self.stagger = float((self.schedule or {}).get('stagger', stagger) or 0)
        if self.schedule is not None and set(self.schedule.setdefault('stages', 'mmmmmmmm')) == {'m'} and 'lstm' not in self.schedule:
            self.schedule = None
        # waits: [(a, b), ...] -- having finished stage a (0: before it starts) a batch waits until the batch before it has finished stage b;
        # `stagger = s` is the pair (0, s)
        if waits == 'auto':
            # the default of the serving loop (round 6, profiles/r06_waits_sweep*.txt, same-box A/B: 8.72 -> 8.50 ms, 8.87 -> 8.73, 8.75 -> 8.53 per
            # batch at K = 20, first completion 21 -> 16 ms): a batch starts when the batch before it enters its LAST encoder layer, and enters
            # its decoder when the batch before it enters its last decoder layer -- the MFMA-bound decoders run one after the other (two of them
            # side by side only time-slice the CUs) with the next batches' latency-bound encoder layers underneath, and no two batches can lock
            # into running the same layer at the same time
            ne, nd = len(getattr(model, 'encoder', ())), len(getattr(model, 'decoder', ()))
            waits = [] if (self.stagger or os.environ.get('AERO_PIPELINE_WAITS') == '0' or ne < 2 or nd < 2) else [(0, ne - 1), (ne, ne + nd - 1)]
        self.waits = [(float(a), float(b)) for a, b in (waits or [])]
        if self.stagger:
            self.waits.append((0.0, self.stagger))
        self._stage_ev = {}
        self._streams = {}
        self._kinds = {}
        self._slot_done = {}
        self._n = 0
        self._seen = set()
        self._open = []

    def _ring(self, dev):
        if dev not in self._streams:
            from .engine import side_streams
            self._streams[dev] = side_streams(dev, self.depth)     # (shared with the engine's half-batch streams: see there)
        return self._streams[dev]

    def _kind_stream(self, dev, slot, letter):
        """the stream of kind `letter` that belongs to slot `slot` of the ring"""
        if letter == 'm':
            return self._ring(dev)[slot]
        spec = self.schedule[letter]
        shared = spec[-1] == 'shared'
        key = (dev, letter, 0 if shared else slot)
        st = self._kinds.get(key)
        if st is None:
            from .engine import special_stream
            tag = ('pipe', letter, key[2])
            if spec[0] == 'prio':
                st = special_stream(dev, priority=spec[1], tag=tag)
            elif spec[0] == 'mask':
                st = special_stream(dev, cu_range=(spec[1], spec[2]), tag=tag)
            else:
                st = special_stream(dev, priority=0, tag=tag)
            self._kinds[key] = st
        return st

    def _all_streams(self, dev):
        return list(self._ring(dev)) + [s for (d, _, _), s in self._kinds.items() if d == dev]

    def submit(self, mix, to_host=False, **kw):
        """enqueue model(mix, **kw) (Aero.forward's keywords) and return a ticket for `result`.
        A HOST tensor `mix` is uploaded on the batch's own stream (pinned, asynchronous) and with `to_host` the outputs are downloaded on
        it into pinned host tensors: the copies of one batch then run next to the kernels of the others without any further stream object
        (the hardware queues are few: engine.side_streams); `result` of such a ticket waits for the batch on the host."""
        if self.model.training:
            raise RuntimeError('BatchPipeline is the inference loop: call model.eval() first')
        mdev = next(self.model.parameters()).device
        if mdev.type != 'cuda':
            with torch.no_grad():
                return _Ticket(self.model(mix, **kw), None, None)
        if mix.is_cuda and self.depth == 1 and not to_host:
            with torch.no_grad():
                return _Ticket(self.model(mix, **kw), None, None)
        host_in = None
        if not mix.is_cuda:
            host_in = mix if mix.is_pinned() else mix.pin_memory()
        dev = mdev
        cur = torch.cuda.current_stream(dev)
        eng = self.model._get_engine()
        ring = self._ring(dev)
        if eng._weights_key(dev) != eng._key:
            # the weights changed since the last pack: the packed images are about to be replaced (and their memory recycled on the caller's
            # stream) while batches in flight may still read them -- let the caller's stream wait for those batches first
            for s in self._all_streams(dev):
                cur.wait_stream(s)
        eng._prepare(dev)                                   # weights are (re-)packed on the caller's stream, which the batch stream waits for
        slot = self._n % self.depth
        self._n += 1
        hook = None
        if self.schedule is None:
            st = ring[slot]
            if self.waits:
                prev_evs = self._stage_ev.get(dev, {})
                mine = self._stage_ev[dev] = {}
                for a, b in self.waits:
                    if a == 0.0 and b in prev_evs:
                        st.wait_event(prev_evs[b])

                def hook(i, st=st, prev_evs=prev_evs, mine=mine):
                    if isinstance(i, str):
                        return
                    for a, b in self.waits:
                        if abs(i - b) < 1e-6 and b not in mine:
                            ev = torch.cuda.Event()
                            ev.record(st)
                            mine[b] = ev
                    for a, b in self.waits:
                        if a != 0.0 and abs(i - a) < 1e-6 and b in prev_evs:
                            st.wait_event(prev_evs[b])
        else:
            # Stages of this batch on different streams of its slot.  One stage follows the other (fork = event + wait), so a batch is
            # still ONE chain; what changes is which queue -- priority, CU set -- its launches wait in next to the other batches'.
            # The caching allocator keeps one pool per stream and reuses a freed block without waiting for OTHER streams: a block of
            # stream s that a later stage read on stream s' returns to s's pool and is handed out again by the NEXT batch of this slot,
            # which therefore starts behind this batch's last stage (`_slot_done`).  (A 'shared' kind breaks that pairing: experiments only.)
            stages = self.schedule['stages']
            st = self._kind_stream(dev, slot, stages[0])
            prev = self._slot_done.get((dev, slot))
            if prev is not None:
                st.wait_event(prev)
            state = {'cur': st}

            def hook(i, stages=stages, state=state, dev=dev, slot=slot):
                if isinstance(i, str):                      # 'lstm+' / 'lstm-': the recurrent launches on the kind named by schedule['lstm']
                    if 'lstm' not in self.schedule:
                        return
                    if i == 'lstm+':
                        state['back'] = state['cur']
                        nxt = self._kind_stream(dev, slot, self.schedule['lstm'])
                    else:
                        nxt = state['back']
                elif not isinstance(i, int) or i >= len(stages):
                    return
                else:
                    nxt = self._kind_stream(dev, slot, stages[i])
                if nxt is not state['cur']:
                    ev = torch.cuda.Event()
                    ev.record(state['cur'])
                    nxt.wait_event(ev)
                    torch.cuda.set_stream(nxt)
                    state['cur'] = nxt
        st.wait_stream(cur)
        if host_in is None:
            mix.record_stream(st)
        # the whole batch on this stream as the plain eager kernel sequence: no half-batch split inside a pipelined batch, and NOT the
        # AERO_GRAPH replay path -- HipEngine._forward_graph keeps one (graph, static input, static output) per shape whatever the stream, so
        # two batches in flight would share the graph's static buffers
        saved = eng.streams, eng.use_graph, eng.stage_hook
        eng.streams, eng.use_graph, eng.stage_hook = 1, False, hook
        try:
            with torch.cuda.stream(st), torch.no_grad():
                if host_in is not None:
                    mix = host_in.to(dev, non_blocking=True)
                out = self.model(mix, **kw)
                if to_host:
                    def down(t):
                        if not torch.is_tensor(t):
                            return t
                        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                        h.copy_(t, non_blocking=True)
                        return h
                    out = tuple(down(t) for t in out) if isinstance(out, (tuple, list)) else down(out)
                if self.schedule is not None:
                    st = state['cur']                       # the stream of the batch's last stage: the one the ticket's event belongs to
        finally:
            eng.streams, eng.use_graph, eng.stage_hook = saved
        key = (str(dev), tuple(mix.shape), tuple(sorted(kw.items())))
        if key not in self._seen:
            # the first batch of a shape builds the engine's lazily created device tables (window, envelope, DFT table, constant
            # buffers) on ITS stream: the other streams must not run ahead of that
            self._seen.add(key)
            for s in self._all_streams(dev):
                if s is not st:
                    s.wait_stream(st)
            cur.wait_stream(st)                             # ... nor a direct model(x) on the caller's stream right behind this submit
        ev = torch.cuda.Event()
        ev.record(st)
        if self.schedule is not None:
            self._slot_done[(dev, slot)] = ev
        t = _Ticket(out, ev, st, host=to_host, keep=host_in)
        # tickets are held WEAKLY: a caller that drops a ticket (or dies between submit and result) frees its outputs, pinned buffers and
        # event with it instead of leaving them in this list for the life of the pipeline
        self._open = [r for r in self._open if r() is not None]
        self._open.append(weakref.ref(t))
        return t

    def result(self, ticket):
        """the output of a submitted batch, ordered into the caller's current stream (`to_host` tickets: complete on the host)"""
        if ticket.event is not None:
            if ticket.host:
                ticket.event.synchronize()
            else:
                cur = torch.cuda.current_stream(ticket.stream.device)
                cur.wait_event(ticket.event)
                for t in _tensors(ticket.out):
                    t.record_stream(cur)
            ticket.event = None
            ticket.keep = None
            self._open = [r for r in self._open if r() is not None and r() is not ticket]
        return ticket.out

    def drain(self):
        """make the caller's current stream wait for every batch submitted so far"""
        for r in list(self._open):
            t = r()
            if t is not None:
                self.result(t)
        self._open = []
        for dev in self._streams:                           # (batches whose tickets were dropped: their streams are waited for all the same)
            cur = torch.cuda.current_stream(dev)
            for s in self._all_streams(dev):
                cur.wait_stream(s)

    def run(self, batches, to_host=False, **kw):
        """all of `batches` (an iterable of inputs), results in order; at most `depth` outputs are held un-collected"""
        outs, tickets = [], []
        for x in batches:
            tickets.append(self.submit(x, to_host=to_host, **kw))
            if len(tickets) > self.depth:
                outs.append(self.result(tickets.pop(0)))
        outs.extend(self.result(t) for t in tickets)
        return outs
