"""Training step engine: forward + backward + gradient exchange + clip + Adam,
optionally replayed as one HIP graph (the step of the reference's
Executor.train, /root/reference/openeat/utils/executor.py:36-63)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch

from openeat_amd import ops
from openeat_amd import planes as _planes
from openeat_amd.arena import ParamArena
from openeat_amd.ddp import GradAllReduce
from openeat_amd.optim import FusedAdam
from openeat_amd.utils import common


class TrainEngine:
    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, grad_clip: float = 5.0, accum_grad: int = 1,
                 n_allreduce_chunks: int = 4, static_shapes: bool = False, async_wgrad: bool = True,
                 parallel_decoders: bool = False, segmented: Optional[bool] = None):
        self.model = model
        ops.ASYNC_WGRAD = bool(async_wgrad)
        ops.PARALLEL_DECODERS = bool(parallel_decoders)
        self.parallel = bool(parallel_decoders)
        self.arena = ParamArena(model).activate()
        self.optimizer = FusedAdam(self.arena, lr=lr, max_grad_norm=grad_clip)
        self.reducer = GradAllReduce(self.arena.grad, n_allreduce_chunks)
        self.reducer.broadcast_parameters(self.arena.flat)
        # capture() with several ranks records the step as a chain of graphs cut where the eager step's backward hooks sit,
        # so that the gradient all-reduces start between the replays (None: exactly when world > 1; True on one rank
        # exercises the same path with no-op collectives)
        import os
        self.segmented = (self.reducer.active and os.environ.get("OE_SEGMENTED", "1") != "0") if segmented is None else bool(segmented)
        self._segments = None
        if self.reducer.active or self.segmented:
            self._install_overlap_hooks()
        self.accum_grad = max(1, int(accum_grad))
        self._micro = 0                          # micro-steps accumulated since the last optimizer step
        self.static_shapes = static_shapes
        dev = self.arena.flat.device
        self.seed_counter = torch.zeros(1, dtype=torch.int64, device=dev)      # advanced once per step on device
        ops.set_seed_device_counter(self.seed_counter)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._static: Dict[str, torch.Tensor] = {}
        self._out = None
        self._ln_table = None
        # step_cached(): one captured graph per batch shape, least recently used dropped first, all in ONE memory pool
        self._cache: "OrderedDict[tuple, Optional[tuple]]" = OrderedDict()
        self._pool = None
        self.cache_hits = self.cache_misses = 0
        self.cache_uncapturable = 0              # shapes whose capture was refused: they run eagerly (step_cached warns once per shape)

    def _install_overlap_hooks(self, layer_cuts=None, heads: bool = True):
        """Start the gradient all-reduce of the arena tail whose gradients are final while backward is still running:
        after the heads (gradient of the encoder output ready) and at the quarter points of the encoder stack; what is
        left after backward is the first quarter of the encoder and the input layer.
        layer_cuts / heads: other cut sets (set_overlap_cuts) - encoder layer indices whose input gradient marks a cut, and
        whether the encoder output is one."""
        us = self.arena.unit_start
        if getattr(self, "_hook_model", None) is None:
            self._hook_model = self.model            # (callers may wrap engine.model later: the hooks stay on this one)
        enc = getattr(self._hook_model, "encoder", None)
        if enc is None or "heads" not in us:
            return

        def tail_from(unit):
            def cb():
                if self._capturing or self._hooks_off:   # a HIP-graph capture holds no collectives: all of it goes after the graph
                    return
                ops.join_side_stream()          # weight-gradient GEMMs of the finished units run on the side stream
                ops.ln_table_flush()            # ... and their LayerNorms' parameter gradients are still per-block partials (eager table)
                self.reducer.reduce_tail(us[unit])
            return cb

        self._hook_model.grad_ready_hooks = {"encoder_out": tail_from("heads")} if heads else {}
        n = len(enc.encoders)
        cuts = sorted({(n * q) // 4 for q in (1, 2, 3)} - {0}) if layer_cuts is None else sorted(set(int(c) for c in layer_cuts) - {0})
        enc.grad_ready_hooks = {i: tail_from(f"enc{i}") for i in cuts if f"enc{i}" in us}      # default: quarter points of the stack

    def set_overlap_cuts(self, layer_cuts=None, heads: bool = True):
        """Choose where the multi-rank step is cut (eager: where backward hands finished arena tails to the collective; captured in
        segments: where one graph ends and the next begins).  None / True = the default (encoder output + quarter points: five
        graphs); e.g. ([n // 4], False) = two graphs, one all-reduce beside the second.  Drops any captured graph."""
        self.drop_graph()
        self._install_overlap_hooks(layer_cuts, heads)

    # one micro-step: loss (already divided by accum_grad) and its backward
    def _fwd_bwd(self, batch):
        ops.predrop_clear()
        ops.stamp("step starts")
        # eager steps use the LayerNorm parameter-gradient table too (one reduce launch per step instead of one per norm)
        own_table = ops.LN_EAGER_TABLE and ops.LN_TABLE is None and not self._capturing and self.arena.flat.is_cuda
        if own_table:
            ops.ln_table_begin(self.arena.flat.device, eager=True)
        try:
            return self._fwd_bwd_body(batch)
        finally:
            if own_table:
                ops.LN_TABLE = None

    def _fwd_bwd_body(self, batch):
        loss, acc = self.model(**batch)
        loss = loss / self.accum_grad if self.accum_grad != 1 else loss
        ops.stamp("fwd: losses done")
        loss.backward()
        ops.resolve_pending_ln()               # (a parked LayerNorm backward whose consumer never ran: none in a healthy step)
        ops.stamp("bwd: main chain done")
        ops.join_side_stream()                 # weight-gradient GEMMs running beside the backward chain
        ops.ln_table_flush()                   # captured graph: every LayerNorm's parameter-gradient partials in one launch
        ops.stamp("bwd: side streams joined")
        return loss.detach(), None if acc is None else acc.detach()

    def _finish(self):
        self.reducer()
        # inside a capture the learning rate is whatever replay() puts into lr_dev; everywhere else (eager steps, also
        # those taken beside a captured graph) it is the optimizer's current param_groups value
        self.optimizer.step(lr_from_device=self._capturing or self._replaying)      # (marks the weights' bf16 planes stale)
        self.seed_counter.add_(1)
        ops.stamp("optimizer done")

    _capturing = False
    _replaying = False
    _accum_capture = False      # the captured graph is ONE micro-step of an accumulated step (accum_grad > 1)
    _opt_graph = None           # ... and this one its clip + Adam
    _seg_keep = None
    _hooks_off = False          # capture()'s warm-up steps: same launch structure as the capture (whole arena reduced after backward)
    _split = False

    def step(self, batch: Dict[str, torch.Tensor], lr: Optional[float] = None):
        """Eager (micro-)step, any shapes.  With accum_grad = k the gradient arena is zeroed before the first of k
        calls, every call adds the gradient of loss / k, and only the k-th call exchanges gradients (the backward-overlap
        hooks and the final all-reduce), clips, runs Adam - /root/reference/openeat/utils/executor.py:42-63 (no_sync on
        the micro-steps in between, one optimizer step per k batches).  Returns (loss / k, acc) of this call."""
        common.STATIC_SHAPES = self.static_shapes
        ops.POS_PROJ_AHEAD = self.parallel and (not self.reducer.active)      # not beside the backward hooks' collectives
        if lr is not None:
            self.optimizer.set_lr(lr)
        if self._micro == 0:
            self.arena.zero_grad()
        boundary = self._micro + 1 >= self.accum_grad
        hooks_off, self._hooks_off = self._hooks_off, self._hooks_off or not boundary     # no collective before the last micro-step
        try:
            out = self._fwd_bwd(batch)
        finally:
            self._hooks_off = hooks_off
        if boundary:
            self._finish()
            self._micro = 0
        else:
            self._micro += 1
            self.seed_counter.add_(1)          # fresh dropout masks for the next micro-step
        return out

    # ---- HIP-graph path: fixed shapes, no host sync inside the step -----------------------------
    def capture(self, example_batch: Dict[str, torch.Tensor], warmup: int = 2, pool=None, _warm: bool = False):
        """Capture the step (see _capture_impl).  With several ranks the outcome is agreed across them: a capture that fails on
        one rank only would leave the ranks in different launch modes (graphs with all-reduces between them on some, eager
        hooks on others) - every rank then drops its graph and raises, and the caller falls back to eager steps everywhere."""
        err = None
        try:
            self._capture_impl(example_batch, warmup, pool, _warm)
        except Exception as e:                     # noqa: BLE001
            err = e
        ok = 0 if err is not None else 1
        if self.reducer.active and self.reducer.agree_min(ok) == 0:
            self._graph = self._segments = self._seg_keep = None
            if err is None:
                err = RuntimeError("TrainEngine.capture: another rank could not capture this step; staying eager on every rank")
        if err is not None:
            raise err

    def _capture_impl(self, example_batch: Dict[str, torch.Tensor], warmup: int = 2, pool=None, _warm: bool = False):
        """Capture the step as a HIP graph.  With one rank the whole step (incl. clip + Adam) is one graph;
        with several ranks the graph holds zero-grad + forward + backward and the gradient all-reduce and
        the 3-kernel optimizer step run right after it on the same stream (RCCL stays outside the graph).
        The `warmup` steps that precede the capture are REAL optimizer steps on `example_batch` (parameters, Adam
        moments, step count and dropout counter advance; a scheduler should count them); the capture itself executes
        nothing.
        Gradient accumulation (accum_grad = k > 1, executor.py:42-63): the graph holds ONE micro-step - forward + backward
        of loss / k accumulating into the gradient arena, no zero-grad, no update - and a second small graph holds clip +
        Adam; replay() zeroes the arena before the first of k micro-steps, replays the micro-step graph k times (fresh
        inputs each time) and the update graph after the k-th (behind the gradient all-reduce when there are several
        ranks: one whole-arena exchange per optimizer step, none on the micro-steps in between - DDP's no_sync)."""
        self._split = self.reducer.active
        self._accum_capture = self.accum_grad != 1
        if self._accum_capture and self._micro != 0:
            raise RuntimeError("TrainEngine.capture: call between optimizer steps, not in the middle of an accumulation")
        unjoined = 0
        # at least one eager step first: streams, events and lazily initialised state must exist before the capture (a
        # cold capture ended "unjoined"); step_cached() has just made that step itself (_warm)
        warmup = 0 if _warm else max(1, int(warmup))
        ops.POS_PROJ_AHEAD = self.parallel            # inside a capture the collectives all come after the graph
        common.STATIC_SHAPES = True
        self.static_shapes = True
        self._static = {k: v.clone() for k, v in example_batch.items()}
        # The graph is captured as ONE chain: parallel branches (the side-stream weight gradients of the eager step)
        # are replayed on several hardware queues with cross-queue waits, and measured slower than the plain chain
        # (24.3 vs 22.5 ms/step); the eager step keeps its side stream.
        import os
        async_wgrad = ops.ASYNC_WGRAD
        ops.ASYNC_WGRAD = ops.ASYNC_WGRAD and os.environ.get("OE_GRAPH_FORK", "0") == "1"      # tuning: keep the fork in the graph
        # weight gradients in groups of 48 behind one fork each (ops.WGRAD_DEFER; config 2: 19.6 -> 18.4 ms/step; 36..64 measured)
        ops.WGRAD_DEFER = int(os.environ.get("OE_WGRAD_DEFER", "48")) if (async_wgrad and not ops.ASYNC_WGRAD) else 0
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._hooks_off = True
                try:
                    for _ in range(warmup):
                        self.arena.zero_grad()
                        for _m in range(self.accum_grad):
                            self._fwd_bwd(self._static)
                            if _m + 1 < self.accum_grad:
                                self.seed_counter.add_(1)
                        self._finish()
                finally:
                    self._hooks_off = False
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            self._capturing = True
            try:
                # with a process group alive, another thread (the collective backend's watchdog) may touch the device
                # while we capture: only this thread's calls belong to the capture
                mode = "thread_local" if self.reducer.active else "global"
                ops.ln_table_begin(self.arena.flat.device)
                try:
                    # no pre-split operand crosses the capture's boundary in either direction (planes.capture_scope)
                    with _planes.capture_scope():
                        if self._accum_capture:
                            g, unjoined = self._capture_accum(pool, mode)
                        elif self.segmented:
                            g, unjoined = self._capture_segments(pool, mode)
                        else:
                            with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                                try:
                                    self.arena.grad.zero_()
                                    self._out = self._fwd_bwd(self._static)
                                    if not self._split:
                                        self._finish()
                                finally:
                                    unjoined = self._lead_forks_back()      # also when the step raised: capture_end comes next
                finally:
                    self._ln_table = ops.ln_table_end()        # the graph's launch reads this tensor: keep it alive
            finally:
                self._capturing = False
        finally:
            ops.ASYNC_WGRAD = async_wgrad
            ops.WGRAD_DEFER = 0
            ops.drop_deferred()                # a capture that failed midway must not leave launches behind for an eager step
        if unjoined:
            self._segments = None
            raise RuntimeError(f"TrainEngine.capture: {unjoined} forked stream(s) had not rejoined the capturing stream at the end "
                               "of the step (a fork without its join); they were joined to close the capture, the graph is dropped")
        self._graph = g

    def _capture_accum(self, pool, mode):
        """accum_grad > 1: graph 1 = one micro-step (forward + backward into the arena + the dropout counter's tick), graph 2 =
        clip + Adam (+ the weights' bf16 planes).  Returns (micro-step graph, unjoined-stream count)."""
        pool = pool if pool is not None else torch.cuda.graph_pool_handle()
        unjoined = 0
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
            try:
                self._out = self._fwd_bwd(self._static)
                self.seed_counter.add_(1)                  # every micro-step draws fresh dropout masks
            finally:
                unjoined += self._lead_forks_back()
        if self._opt_graph is None:                        # shape-independent: one for all cached shapes
            go = torch.cuda.CUDAGraph()
            with torch.cuda.graph(go, pool=pool, capture_error_mode=mode):
                self.optimizer.step(lr_from_device=True)
            self._opt_graph = go
        return g, unjoined

    def _capture_segments(self, pool, mode):
        """The step as a chain of HIP graphs in one memory pool (DistributedDataParallel's backward/all-reduce overlap,
        /root/reference/openeat/bin/train_ddp.py:212-219, without giving the graphs up): graph 0 = zero-grad + forward +
        backward of the heads down to the encoder output; then one graph per stretch of encoder layers between the cut
        points (ops.cut: where the eager step's hooks sit), the last one ending in the input layer.  After graph k the
        gradients of everything downstream of its cut are final: replay() hands that tail of the arena to the collective
        and goes straight on to graph k + 1.  Returns (first graph, unjoined-stream count)."""
        pool = pool if pool is not None else torch.cuda.graph_pool_handle()
        self._split = True                        # clip + Adam follow the last all-reduce, outside the graphs
        us = self.arena.unit_start
        segs, unjoined = [], 0
        ops.CUTS = []
        try:
            g0 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g0, pool=pool, capture_error_mode=mode):
                try:
                    self.arena.grad.zero_()
                    self._out = self._fwd_bwd(self._static)
                finally:
                    unjoined += self._lead_forks_back()
            cuts = list(reversed(ops.CUTS))       # backward order: the encoder output first, then the layer cuts from the top
            ops.CUTS = None
            pending = g0
            for name, upstream, leaf in cuts:
                if leaf.grad is None:
                    raise RuntimeError(f"segmented capture: no gradient reached the cut '{name}'")
                segs.append((pending, us[name]))  # after `pending`: floats [us[name], ...) of the gradient arena are final
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                    try:
                        torch.autograd.backward([upstream], [leaf.grad])
                        ops.join_side_stream()
                        ops.ln_table_flush()
                    finally:
                        unjoined += self._lead_forks_back()
                pending = g
            segs.append((pending, None))          # the rest of the arena goes with _finish()
        finally:
            ops.CUTS = None
        self._segments = segs
        self._seg_keep = cuts                     # the tapes' tensors live in the graphs' pool: keep the python objects too
        return g0, unjoined

    @staticmethod
    def _lead_forks_back() -> int:
        """Last thing inside a capture: every stream the step forked onto must be an ancestor of the capturing stream's
        next launch (ending a capture with unjoined work is an error - and has ended in SIGSEGV inside capture_end on
        this stack).  Offenders are joined so that the capture can be closed; the caller raises."""
        from openeat_amd import hip
        origin = torch.cuda.current_stream()
        bad = 0
        while True:
            n, first = hip.capture_unjoined_streams(origin, ops.forked_streams())
            if n == 0:
                return bad
            bad += 1
            origin.wait_stream(first)

    def drop_graph(self):
        """Forget the captured graph(s) (and give their memory pool back): subsequent steps are eager again."""
        self._graph = None
        self._segments = None
        self._seg_keep = None
        self._out = None
        self._static = {}
        self._ln_table = None
        self._opt_graph = None
        self._cache.clear()
        self._pool = None
        torch.cuda.empty_cache()

    # ---- ragged training: one graph per batch shape ------------------------------------------------
    def step_cached(self, batch: Dict[str, torch.Tensor], lr: Optional[float] = None, max_graphs: int = 64, agree_shapes: bool = True):
        """A training step on a batch of any shape, replayed from a captured graph when this shape has been seen before
        (/root/reference/openeat/dataset/dataset.py:337-364 forms length buckets: a handful of (B, T) shapes recur all
        epoch).  First sight of a shape: the step runs eagerly - a real step - and is then captured for the next time
        (the capture executes nothing).  At most `max_graphs` graphs are kept, least recently used dropped first; all of
        them allocate from one memory pool, so the activation memory held is that of the largest shape, not the sum.
        Callers bound the number of distinct shapes by padding to multiples (frames: the bucket's length_multiple;
        targets: pad_targets below).  agree_shapes=False: the caller guarantees that every rank sees the same shape
        sequence (synthetic benchmarks) - no per-call handshake."""
        key = tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in batch.items()))
        rec = self._cache.get(key, False)
        # Several ranks: ragged data gives the ranks different shape sequences, so one rank may hold a graph for its batch
        # while another sees its shape for the first time.  The launch mode decides which collectives a rank issues (a
        # replayed accumulation sends the whole arena once, an eager boundary step sends hook tails and a remainder) and a
        # cross-rank handshake inside a capture on SOME ranks would pair with the others' next loop handshake:
        # the ranks agree hit / miss first - one host-side integer MIN per call - and replay only if every rank can;
        # otherwise all of them take the eager step, and those that missed capture LOCALLY afterwards, no collective.
        multi = self.reducer.active and agree_shapes
        if multi and self.reducer.agree_min(1 if isinstance(rec, tuple) else 0) == 0:
            self.static_shapes = True
            out = self.step(batch, lr)
            if rec is not False:
                self._cache.move_to_end(key)
                return out
            self.cache_misses += 1
            return self._capture_for_cache(key, batch, out, max_graphs, local=True)
        if rec is False:
            self.cache_misses += 1
            self.static_shapes = True
            out = self.step(batch, lr)
            return self._capture_for_cache(key, batch, out, max_graphs, local=False)
        self._cache.move_to_end(key)
        if rec is None:
            return self.step(batch, lr)
        self.cache_hits += 1
        self._graph, self._static, self._out, self._ln_table, self._segments, self._seg_keep = rec
        try:
            return self.replay(batch, lr)
        finally:
            self._graph, self._static, self._out, self._ln_table, self._segments, self._seg_keep = None, {}, None, None, None, None

    def _capture_for_cache(self, key, batch, out, max_graphs: int, local: bool):
        """step_cached: the shape's first (eager) step has just run - capture it for the next time.  local: no cross-rank
        agreement about the outcome (the ranks agree per call instead; a rank whose capture fails keeps reporting a miss)."""
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        micro = self._micro                       # an accumulation may be under way: the capture executes nothing and must not disturb it
        try:
            self._micro = 0
            (self._capture_impl if local else self.capture)(batch, pool=self._pool, _warm=True)
            rec = (self._graph, self._static, self._out, self._ln_table, self._segments, self._seg_keep)
        except Exception as e:               # noqa: BLE001 - capture() has cleaned up after itself
            # A shape that does not capture keeps running eagerly (~1.7x slower at config 2) - never silently: warn once
            # per shape and count it.  What is NOT a property of the shape is re-raised: a fork that never rejoined is a
            # bug in the step's stream handling, a HIP / launch error is a broken device state.
            msg = f"{type(e).__name__}: {e}"
            if ("had not rejoined" in msg or "HIP error" in msg or "hipError" in msg or "CUDA error" in msg or "launch failed" in msg
                    or isinstance(e, (torch.cuda.OutOfMemoryError, MemoryError))):
                self._graph, self._static, self._out, self._ln_table, self._segments, self._seg_keep = None, {}, None, None, None, None
                raise
            import warnings
            self.cache_uncapturable += 1
            warnings.warn(f"TrainEngine.step_cached: batch shape {[tuple(v.shape) for v in batch.values()]} does not capture "
                          f"({msg}); it will run eagerly every time", RuntimeWarning, stacklevel=2)
            rec = None
        finally:
            self._micro = micro
        self._graph, self._static, self._out, self._ln_table, self._segments, self._seg_keep = None, {}, None, None, None, None
        self._cache[key] = rec
        while len(self._cache) > max(1, max_graphs):
            self._cache.popitem(last=False)
        return out

    def replay(self, batch: Optional[Dict[str, torch.Tensor]] = None, lr: Optional[float] = None):
        assert self._graph is not None
        ops.pack_tables_sweep()                # (a replayed pack launch must not touch weights that have died since the capture)
        self.arena.mark_step()                 # the replayed optimizer moves the weights: eager readers of their planes split again
        if batch is not None:
            for k, v in batch.items():
                self._static[k].copy_(v)
        if lr is not None:
            self.optimizer.set_lr(lr)
        self.optimizer.lr_dev.fill_(float(self.optimizer.param_groups[0]["lr"]))     # the graph's Adam reads lr_dev
        if self._accum_capture:
            if self._micro == 0:
                self.arena.grad.zero_()
            self._graph.replay()                                  # one micro-step: gradients of loss / k added to the arena
            if self._micro + 1 >= self.accum_grad:
                self.reducer()                                    # several ranks: the one exchange of this optimizer step
                self._opt_graph.replay()
                self._micro = 0
            else:
                self._micro += 1
            return self._out
        if self._segments:
            for g, final_from in self._segments:
                g.replay()
                if final_from is not None:
                    self.reducer.reduce_tail(final_from)      # async: the next graph runs beside the collective
        else:
            self._graph.replay()
        if self._split:
            self._replaying = True
            try:
                self._finish()
            finally:
                self._replaying = False
        return self._out


def pad_targets(targets: torch.Tensor, multiple: int = 16, ignore_id: int = -1) -> torch.Tensor:
    """Pad the label matrix (B, L) with ignore_id to the next multiple of `multiple` columns: fewer distinct batch shapes
    for step_cached (padding labels are ignored exactly as the collate's own padding is, dataset.py:219)."""
    L = targets.shape[1]
    Lp = -(-L // multiple) * multiple
    if Lp == L:
        return targets
    return torch.nn.functional.pad(targets, (0, Lp - L), value=ignore_id)
