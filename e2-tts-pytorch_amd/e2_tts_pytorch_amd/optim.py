"""Fused optimizer side of the training step over flat fp32 buffers (kernel group K19):
what /root/reference/e2_tts_pytorch/trainer.py:272-279 does with accelerate + adam_atan2_pytorch.Adopt + ema_pytorch.EMA
as ~10 passes over every parameter, done in one reduction pass and one update pass per contiguous run of parameters.

    opt = FusedAdopt(model, lr=1e-4, max_grad_norm=1.0)     # trainer.py:146,183
    loss.backward(); opt.step(); opt.zero_grad()            # trainer.py:270-277
    ema = FusedEMA(model); ema.update()                     # trainer.py:170,279

Parameters whose storage AND gradient storage are adjacent (every backbone parameter: they are views of the Transformer's
flat buffers) are merged into one run = one kernel launch.  No torch fallback: the update itself always runs in the HIP
kernels (`ops.adopt_step` / `ops.ema_update`).
"""
from __future__ import annotations

import copy

import torch

from . import ops
from .backbone import Transformer


def _runs(pairs, keys=None):
    """[(a, b)] fp32 tensor pairs -> [(a_flat, b_flat)]: tensors that are adjacent in memory in BOTH lists (up to the 7
    alignment-padding elements the flat layout puts between slots; nothing ever writes those, they stay zero) are merged
    into one flat view each = one kernel launch per run.  A run must start 16-byte aligned.  keys (optional, one per
    pair): only pairs with equal keys are merged (per-parameter step counts); returns [(a_flat, b_flat, key)] then."""
    order = sorted(range(len(pairs)), key=lambda i: pairs[i][0].data_ptr())
    groups, cur = [], []
    for i in order:
        a, b = pairs[i]
        if cur:
            pa, pb = pairs[cur[-1]]
            gap_a = a.data_ptr() - (pa.data_ptr() + pa.numel() * 4)
            gap_b = b.data_ptr() - (pb.data_ptr() + pb.numel() * 4)
            same = (a.untyped_storage().data_ptr() == pa.untyped_storage().data_ptr() and
                    b.untyped_storage().data_ptr() == pb.untyped_storage().data_ptr())      # one allocation each
            if not (same and gap_a == gap_b and 0 <= gap_a <= 28) or (keys is not None and keys[i] != keys[cur[-1]]):
                groups.append(cur)
                cur = []
        cur.append(i)
    if cur:
        groups.append(cur)
    runs = []
    for idx in groups:
        a0, b0 = pairs[idx[0]]
        al, bl = pairs[idx[-1]]
        n = (al.data_ptr() + al.numel() * 4 - a0.data_ptr()) // 4
        if (a0.data_ptr() | b0.data_ptr()) & 15 or len(idx) == 1:
            runs.extend((pairs[i][0].view(-1), pairs[i][1].view(-1)) + ((keys[i],) if keys is not None else ()) for i in idx)
        else:
            runs.append((torch.as_strided(a0, (n,), (1,)), torch.as_strided(b0, (n,), (1,))) + ((keys[idx[0]],) if keys is not None else ()))
    return runs


def _grad_base(slots, n):
    """the flat fp32 gradient buffer (n elements) that every slot's .grad is a view of at its layout offset, or None"""
    q0, off0 = slots[0]
    g0 = q0.grad
    if g0 is None or g0.dtype != torch.float32:
        return None
    st = g0.untyped_storage()
    start = g0.data_ptr() - off0 * 4 - st.data_ptr()            # byte offset of element 0 of the flat buffer
    if start < 0 or start % 4 or start + n * 4 > st.nbytes():
        return None
    base_ptr = st.data_ptr() + start
    for q, off in slots:
        g = q.grad
        if g is None or not g.is_contiguous() or g.data_ptr() != base_ptr + off * 4 or g.untyped_storage().data_ptr() != st.data_ptr():
            return None
    return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, start // 4, (n,), (1,))


class FusedAdopt:
    """ADOPT (adam_atan2_pytorch.adopt.Adopt defaults: betas (0.9, 0.99), eps 1e-6, decoupled weight decay) with the
    global-norm gradient clip of `accelerator.clip_grad_norm_` folded in; state (m, v) lives in flat buffers per run.

    `steps` is kept PER PARAMETER and a parameter without a gradient is skipped, as Adopt does (trainer.py:183,275): on a
    step whose classifier-free-guidance coin dropped the text (25 % of the reference's training steps, e2_tts.py:1261) the
    text embedding has `.grad is None`, and the backbone's text-stream parameters -- whose slots of the flat gradient
    buffer then hold zeros rather than None -- are treated the same way (`Transformer._text_grad_live`): parameter,
    moments and step count stay as they are.  The backbone remains ONE launch: its text-stream slots are a second
    parameter group inside the flat buffer (e2k_adopt_step_groups)."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.99), eps=1e-6, weight_decay=0., max_grad_norm=1.0, decoupled_wd=True,
                 process_group=None):
        self.model = model
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        # adam_atan2_pytorch.adopt.Adopt: `wd /= init_lr` when decoupled_wd (the default), and the decay `p.mul_(1 - lr * wd)` is applied
        # before anything else on every step a parameter has a gradient -- the first (v = g^2 only) included
        self.decoupled_wd, self._init_lr = bool(decoupled_wd), lr
        # the data-parallel group the text-live flag is made global over under a stock DDP without the shim (None: the default group).
        # A job whose replicas are a SUBGROUP of the default group (or in which not every rank of the default group runs this optimizer)
        # must pass its group here: the flag exchange is a collective
        self.process_group = process_group
        self.steps = [0] * len(self.params)            # per parameter, as Adopt's state['steps']
        self._state = {}           # data_ptr of a parameter STORAGE -> (m, v) flat fp32 buffers mirroring that storage
        self._loaded = {}          # parameter index -> (m, v) from load_state_dict, not yet copied into a run
        self._backbones = [m for m in model.modules() if isinstance(m, Transformer)]
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._ranges = {}          # id(backbone) -> (device int32 (nr, 2) text ranges, set of text parameter ids)
        self._ema = None           # a FusedEMA whose backbone averages are moved inside the ADOPT pass (attach_ema)

    def attach_ema(self, ema):
        """fold `ema.update()` for the backbone into this optimizer's pass (e2k_adopt_step_ema): on the steps on which the coming
        `ema.update()` moves the average, the ADOPT kernel moves it while the new parameter value is in registers (8 B per element
        instead of a 12-B pass of its own), and `ema.update()` then only handles the few parameters outside the backbone.  This relies on
        the trainer's order -- one `ema.update()` after every `step()` (trainer.py:275-279) -- and a second `step()` without it raises."""
        assert ema is None or ema.online is self.model, 'the EMA must follow the model this optimizer updates'
        self._ema, self._ema_seen = ema, None
        return self

    @property
    def step_count(self):
        """optimizer steps taken (= the largest per-parameter count)"""
        return max(self.steps) if self.steps else 0

    def _globalise_text_live(self, dev):
        """`Transformer._text_grad_live` (did the text stream's parameters receive a gradient this step?) decides whether those
        parameters are skipped, so it must be the same on every data-parallel rank: the classifier-free-guidance coin that drops the
        text is flipped per rank (e2_tts.py:1261-1262), and the reference's DistributedDataParallel all-reduces its used-parameter map
        (trainer.py:155-162).  ddp.DataParallel / enable_overlap_under_ddp make the flag global during the pass; under a stock DDP
        without the shim it is this rank's own, and one MAX all-reduce over the default group per optimizer step -- a point every
        rank of a data-parallel job reaches equally often -- settles it here."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        local = [tr for tr in self._backbones if getattr(tr, '_text_grad_live', None) is not None and not getattr(tr, '_text_live_is_global', False)]
        # (every rank must take the same decision about calling the collective: which backbones ran a backward pass of ours since the
        #  last step is the same everywhere in a data-parallel job; the per-rank part is only the VALUE of the flag)
        if not local:
            return
        grp = self.process_group
        if grp is not None and dist.get_world_size(grp) <= 1:
            return
        t = torch.full((len(local),), 0, dtype=torch.int32, device=dev if dist.get_backend(grp) != 'gloo' else 'cpu')
        for i, tr in enumerate(local):
            if tr._text_grad_live:
                t[i] = 1
        # (one blocking collective + host read-back per optimizer step, only on this un-shimmed path: ddp.DataParallel and
        #  enable_overlap_under_ddp settle the flag during the pass through a pinned buffer and never come here)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
        for tr, v in zip(local, t.tolist()):
            tr._text_grad_live = bool(v)
            tr._text_live_is_global = True

    def _text_group(self, tr, dev):
        ent = self._ranges.get(id(tr))
        if ent is None or ent[0].device != dev:
            rs = tr.text_param_ranges()
            ids = {id(q) for q, off in tr._layout.slots if any(a <= off < b for a, b in rs)}
            ent = self._ranges[id(tr)] = (torch.tensor(rs, dtype=torch.int32, device=dev).reshape(-1, 2), ids)
        return ent

    # checkpoint format (trainer.py:202-228 stores optimizer.state_dict()) ------------------------------------------
    # torch.optim layout with adam_atan2_pytorch.adopt.Adopt's per-parameter keys ('steps', 'm', 'v'; SURVEY.md Appendix
    # A.10), parameters numbered in model.parameters() order as `Adopt(model.parameters(), ...)` numbers them.

    def _mv(self, t, create=True):
        """(m, v) slices for the contiguous fp32 range `t` (a parameter or a run of adjacent parameters).  The moments
        mirror the parameter STORAGE element for element (the backbone's flat buffer = one storage), so the same state is
        found however the gradients happen to be grouped into runs on a given step"""
        st = t.untyped_storage()
        key = st.data_ptr()
        if key not in self._state:
            if not create:
                return None
            n = st.nbytes() // 4
            self._state[key] = (torch.zeros(n, dtype=torch.float32, device=t.device), torch.zeros(n, dtype=torch.float32, device=t.device))
        m, v = self._state[key]
        o = (t.data_ptr() - key) // 4
        return m[o:o + t.numel()], v[o:o + t.numel()]

    def _views(self):
        """{parameter index: (m, v) views shaped like the parameter} for every parameter that has state"""
        out = {}
        for i, p in enumerate(self.params):
            mv = self._mv(p, create=False)
            if mv is not None:
                out[i] = (mv[0].view_as(p), mv[1].view_as(p))
        return out

    def state_dict(self):
        state = {i: dict(steps=self.steps[i], m=m.detach().clone(), v=v.detach().clone()) for i, (m, v) in sorted(self._views().items())}
        for i, (m, v) in self._loaded.items():              # loaded but not yet stepped
            state.setdefault(i, dict(steps=self.steps[i], m=m.clone(), v=v.clone()))
        group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay, decoupled_wd=self.decoupled_wd,
                     init_lr=self._init_lr, params=list(range(len(self.params))))
        return dict(state=dict(sorted(state.items())), param_groups=[group])

    def load_state_dict(self, sd):
        g = sd['param_groups'][0]
        assert len(g['params']) == len(self.params), 'parameter count differs from the checkpoint'
        self.lr, self.betas, self.eps = g['lr'], tuple(g['betas']), g['eps']
        self.weight_decay = g.get('weight_decay', 0.)
        self.decoupled_wd = bool(g.get('decoupled_wd', True))
        self._init_lr = g.get('init_lr', self._init_lr)         # (Adopt keeps it on the optimizer object, not in the checkpoint: ours survives a reload)
        self.steps = [0] * len(self.params)
        self._loaded = {}
        for i, st in sd['state'].items():
            p = self.params[int(i)]
            self.steps[int(i)] = int(st['steps'])
            self._loaded[int(i)] = (st['m'].to(device=p.device, dtype=torch.float32).reshape(p.shape),
                                    st['v'].to(device=p.device, dtype=torch.float32).reshape(p.shape))
        self._install_loaded()

    @torch.no_grad()
    def _install_loaded(self):
        """copy loaded per-parameter moments into the flat buffers of the runs that exist (runs are only known once
        gradients have been seen, so this is called again from step() after new runs were created)"""
        if not self._loaded:
            return
        for i, (m, v) in self._views().items():
            if i in self._loaded:
                lm, lv = self._loaded.pop(i)
                m.copy_(lm)
                v.copy_(lv)

    def zero_grad(self, set_to_none=True):
        """optimizer.zero_grad (trainer.py:277).  Backbones in persistent-gradient mode are skipped: their next backward
        pass overwrites the flat gradient buffer anyway, and detaching ~600 views only to re-attach them costs host time"""
        keep = set()
        for tr in self._backbones:
            if getattr(tr, '_persist_grads', False):
                keep.update(id(q) for q, _ in tr._layout.slots)
        for p in self.params:
            if id(p) in keep or p.grad is None:
                continue
            if set_to_none:
                p.grad = None
            else:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        pairs = [(p, p.grad) for p in self.params if p.grad is not None]
        if not pairs:
            return
        for p, g in pairs:
            assert p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
        dev = pairs[0][0].device
        gs = torch.zeros(1, dtype=torch.float64, device=dev) if self.max_grad_norm > 0 else None
        # a backbone whose gradients are the views of ONE flat buffer our backward produced is a single run over its
        # whole flat parameter buffer (alignment pads and the zero "holes" of the layout included: nothing writes their
        # gradients, so they stay exactly zero through the update)
        runs, taken, stepped = [], set(), []
        self._globalise_text_live(dev)
        ema, ema_decay, folded = self._ema, None, set()
        if ema is not None:
            # the contract is checked on EVERY step, not only on those whose update is folded: a loop that steps more often
            # than it updates the average fails on its first iteration, not around step 110 when the first fold comes due
            seen, self._ema_seen = getattr(self, '_ema_seen', None), ema.step
            if ema._folded is not None or (seen is not None and ema.step != seen + 1):
                self._ema_seen = None
                raise RuntimeError('FusedAdopt.step() with an attached EMA: exactly one ema.update() has to follow every step() '
                                   '(trainer.py:275-279); loops that update the average less often must detach it: attach_ema(None)')
            ema_decay = ema.pending_decay()             # None: the coming ema.update() copies or does nothing
        for tr in self._backbones:
            slots = getattr(getattr(tr, '_layout', None), 'slots', None)
            flat = getattr(tr, '_flat', None)
            if not slots or flat is None or slots[0][0].grad is None:
                continue
            pg = getattr(tr, '_pg', None)
            if getattr(tr, '_persist_grads', False) and pg is not None and slots[0][0].grad is pg.views[0]:
                base = pg.buf                       # persistent mode: the buffer is at hand, no need to re-derive it
            else:
                base = _grad_base(slots, flat.numel())
            if base is not None and all(q.data_ptr() == flat.data_ptr() + off * 4 for q, off in slots):
                ranges, text_ids = self._text_group(tr, flat.device)
                live = getattr(tr, '_text_grad_live', None) is not False      # (None: no backward pass of ours since the last step -- trust the gradients that are there)
                main = [self._index[id(q)] for q, _ in slots if id(q) not in text_ids and id(q) in self._index]
                text = [self._index[id(q)] for q, _ in slots if id(q) in text_ids and id(q) in self._index]
                # (all parameters of a group have stepped together since construction / load, so one count per group)
                kw = dict(step=self.steps[main[0]] if main else 0, ranges=ranges, step_b=self.steps[text[0]] if text else 0, active_b=live)
                te = ema.twin_of(tr) if ema_decay is not None else None
                if te is not None:
                    kw.update(ema=te._flat.view(-1), ema_decay=ema_decay)
                    folded.add(id(te))
                runs.append((flat.view(-1), base, kw))
                stepped += main + (text if live else [])
                taken.update(id(q) for q, _ in slots)
        # backbones that did not go through the flat path (gradients that are not views of one buffer at the layout offsets:
        # stock DDP bucket views, a re-packed parameter buffer, ...): on a step whose text stream ran on no rank their
        # text-stream parameters hold exact zeros, not None (backbone._param_grads says why), and must be skipped like a
        # parameter without a gradient -- no weight decay, no moment decay, no step count
        dead = set()
        for tr in self._backbones:
            slots = getattr(getattr(tr, '_layout', None), 'slots', None)
            if slots and id(slots[0][0]) not in taken and getattr(tr, '_text_grad_live', None) is False:
                dead |= tr._text_param_ids()
            tr._text_grad_live = None           # consumed: the next backward pass sets it again
        rest = [(p, g) for p, g in pairs if id(p) not in taken and id(p) not in dead]
        for pf, gf, k in _runs(rest, [self.steps[self._index[id(p)]] for p, _ in rest]):
            runs.append((pf, gf, dict(step=k)))
        stepped += [self._index[id(p)] for p, _ in rest]
        if gs is not None:
            for pf, gf, _ in runs:
                ops.sumsq(gf, gs)
        b1, b2 = self.betas
        nstate = len(self._state)
        mvs = [self._mv(pf) for pf, _, _ in runs]
        if len(self._state) != nstate:
            self._install_loaded()
        wd = self.weight_decay / self._init_lr if self.decoupled_wd else self.weight_decay
        for (pf, gf, kw), (m, v) in zip(runs, mvs):
            ops.adopt_step(pf, gf, m, v, kw.pop('step'), lr=self.lr, beta1=b1, beta2=b2, eps=self.eps,
                           weight_decay=wd, max_grad_norm=self.max_grad_norm, gsumsq=gs, **kw)
        if folded:
            ema._folded = (ema.step, ema_decay, folded)
        for p, _ in pairs:                                     # the kernels wrote behind autograd's back
            torch.autograd.graph.increment_version(p)
        for i in stepped:
            self.steps[i] += 1


class FusedEMA:
    """ema_pytorch.EMA(model) defaults (beta 0.9999, update_after_step 100, update_every 10, inv_gamma 1, power 2/3):
    `.ema_model` is a deep copy whose parameters follow the online model."""

    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10, inv_gamma=1., power=2. / 3.):
        self.online, self.ema_model = model, copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
        self.inv_gamma, self.power = inv_gamma, power
        self.step, self.initted = 0, False
        self._folded = None        # (step, decay, {id(backbone of the copy)}): averages FusedAdopt.step() already moved for the coming update()
        for m in self.ema_model.modules():            # deepcopy clones every parameter separately: re-establish the
            if isinstance(m, Transformer):            # flat storage of the copy, so that it is one run like the original
                m._flat = None
                m.enable_persistent_grads(False)      # (the copy never sees a backward pass)
                m._sync(next(m.parameters()).device)

    # checkpoint format: ema_pytorch.EMA is an nn.Module holding `ema_model` plus the buffers `initted` and `step`
    # (include_online_model=False keeps the online model out of it, trainer.py:170-174)
    def state_dict(self):
        sd = {f'ema_model.{k}': v for k, v in self.ema_model.state_dict().items()}
        sd['initted'] = torch.tensor(self.initted)
        sd['step'] = torch.tensor(self.step)
        return sd

    def load_state_dict(self, sd):
        inner = {k[len('ema_model.'):]: v for k, v in sd.items() if k.startswith('ema_model.')}
        self.ema_model.load_state_dict(inner, strict=True)
        self.initted = bool(sd['initted'])
        self.step = int(sd['step'])
        self._folded = None                # (a folded update that was pending belongs to the state that was just replaced)

    def twin_of(self, tr):
        """the copy's backbone that follows the online backbone `tr`, if both are whole flat buffers of the same layout"""
        for te, to in zip((m for m in self.ema_model.modules() if isinstance(m, Transformer)),
                          (m for m in self.online.modules() if isinstance(m, Transformer))):
            if to is tr:
                return te if self._flat_pair(te, to) else None
        return None

    @staticmethod
    def _flat_pair(te, to):
        fe, fo = getattr(te, '_flat', None), getattr(to, '_flat', None)
        se, so = getattr(getattr(te, '_layout', None), 'slots', None), getattr(getattr(to, '_layout', None), 'slots', None)
        if fe is None or fo is None or not se or not so or fe.numel() != fo.numel():
            return False
        return (all(q.data_ptr() == fe.data_ptr() + off * 4 for q, off in se) and
                all(q.data_ptr() == fo.data_ptr() + off * 4 for q, off in so))

    def pending_decay(self):
        """the decay the NEXT update() will move the averages with, or None when it will copy or do nothing"""
        step = self.step
        if step % self.update_every != 0 or step <= self.update_after_step or not self.initted:
            return None
        return self.current_decay(step + 1)

    def current_decay(self, step=None):
        epoch = max((self.step if step is None else step) - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.
        return min(max(1. - (1. + epoch / self.inv_gamma) ** -self.power, 0.), self.beta)

    @torch.no_grad()
    def update(self):
        step = self.step
        self.step += 1
        folded, self._folded = self._folded, None
        assert folded is None or folded[0] == step, 'the folded EMA update belongs to another step'
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step or not self.initted:
            for e, p in zip(self.ema_model.parameters(), self.online.parameters()):
                e.copy_(p)
            self.initted = True
            return
        decay = self.current_decay()
        assert folded is None or folded[1] == decay
        pairs = [(e, p.detach()) for e, p in zip(self.ema_model.parameters(), self.online.parameters()) if e.numel()]
        runs, taken = [], set()
        for te, to in zip((m for m in self.ema_model.modules() if isinstance(m, Transformer)),
                          (m for m in self.online.modules() if isinstance(m, Transformer))):
            if self._flat_pair(te, to):
                if folded is None or id(te) not in folded[2]:     # (folded: FusedAdopt.step() moved this average in its own pass)
                    runs.append((te._flat.view(-1), to._flat.view(-1)))          # whole flat buffers (holes / pads are zero in both)
                taken.update(id(q) for q, _ in te._layout.slots)
        runs += _runs([(e, p) for e, p in pairs if id(e) not in taken])
        for ef, pf in runs:
            ops.ema_update(ef, pf, decay)
        for e, _ in pairs:
            torch.autograd.graph.increment_version(e)


# ------------------------------------------------------------------------------------------------ checkpoints

def save_checkpoint(path, model, optimizer, ema, scheduler=None, step=0):
    """the reference trainer's single-file checkpoint (trainer.py:202-213): same keys, same nesting"""
    torch.save(dict(model_state_dict=model.state_dict(), optimizer_state_dict=optimizer.state_dict(),
                    ema_model_state_dict=ema.state_dict(),
                    scheduler_state_dict=scheduler.state_dict() if scheduler is not None else None, step=step), path)


def load_checkpoint(path, model, optimizer, ema, scheduler=None, map_location=None):
    """trainer.py:215-228; returns the step to resume from"""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ck['model_state_dict'])
    optimizer.load_state_dict(ck['optimizer_state_dict'])
    ema.load_state_dict(ck['ema_model_state_dict'])
    if scheduler is not None and ck.get('scheduler_state_dict') is not None:
        scheduler.load_state_dict(ck['scheduler_state_dict'])
    return ck['step']
