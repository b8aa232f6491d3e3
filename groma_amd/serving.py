"""Continuous-batch decode loop + KV-cache manager.

The reference serves one request at a time, one python-driven forward per token
(groma/serve/model_worker.py:287-338: prefill with `model(input_ids, images=..., use_cache=True)`, then
`model(input_ids=[[token]], past_key_values=...)` per token, arg-max / multinomial on the host, stop on EOS or a
stop id).  Decode is HBM-bound on the 13.2 GB weight stream, and that stream is shared by every row of
a step -- so the MI355X design batches the decode steps of *different requests*:

  * an arena of `max_rows` KV slots ([rows, H, max_len, hd] per layer, K and V^T) -- the KV manager hands a slot
    to a request at admission and takes it back when the request finishes; nothing is copied or compacted.  With
    `grow_to` the arena is re-allocated at a larger `max_len` (live rows copied, the step re-captured) when a request
    arrives that would not fit -- the reference's cache has no bound but the model's positions (model_worker.py:287-338);
  * the requests admitted in one tick are prefilled TOGETHER (one batched forward into a staging cache, then each row's KV is moved
    to its slot).  Every kernel of the path is row- / image-independent, so a row of that batch is bit-identical to a batch-1
    `GromaModel.forward` of the same request (tests/test_fullsize_properties_gpu.py) and independent of its company;
  * with `overlap_admission=True` (round 6) that prefill runs on a worker thread and a stream of its own while the decode ticks of
    the live rows go on: the prefill is compute-bound, the ticks stream weights, and the two share the chip (measured,
    profiles/r06_overlap_probe.txt: the same work finishes 1.12-1.16x sooner, and a live row waits <= 15 ms for its next token
    during an admission instead of the whole 50 ms prefill).  The rows join at the first tick after their prefill has finished;
    a request's tokens are the same either way (rows are independent);
  * every decode step advances ALL occupied rows with one captured hipGraph: per-row positions live on the device
    (`pos_dev`, stride 1 -- csrc/decode.hip), idle rows are masked, the host only reads back the `max_rows` new ids;
  * rows are computed independently by every kernel of the step (GEMV rows, per-(row, head) attention), so a
    request's tokens do not depend on its neighbours or on when it was admitted (tests/test_serving_gpu.py).

Ragged lengths are exact here: each row attends to its own [0, pos] -- the padded-prefix artefact of batched HF
generate cannot occur because rows are never padded against each other.
"""
import threading
from collections import deque
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import engine, ops

I32, I64, F32 = torch.int32, torch.int64, torch.float32


class SlotTable:
    """KV-arena bookkeeping (host side, no device work): which row belongs to which request."""

    def __init__(self, n):
        self.n = n
        self.owner = [None] * n
        self._free = deque(range(n))

    def acquire(self, rid):
        if not self._free:
            return None
        s = self._free.popleft()
        self.owner[s] = rid
        return s

    def release(self, slot):
        if self.owner[slot] is None:
            raise ValueError(f"slot {slot} is not in use")
        self.owner[slot] = None
        self._free.append(slot)

    @property
    def n_free(self):
        return len(self._free)

    def active(self):
        return [s for s in range(self.n) if self.owner[s] is not None]


@dataclass
class Request:
    rid: int
    input_ids: torch.Tensor                 # [P] int64 (un-expanded prompt: <image> / <region> placeholders)
    image: torch.Tensor                     # [3, S, S] float
    max_new_tokens: int = 256
    refer_boxes: Optional[torch.Tensor] = None
    ground_boxes: Optional[torch.Tensor] = None
    eos_token_id: Optional[int] = None
    stop_token_id: Optional[int] = None     # model_worker.py:277-283 `stop` when it tokenises to one id
    seed: Optional[int] = None              # torch.manual_seed before the prefill (region shuffle); also keys the sampler
    temperature: float = 0.0                # model_worker.py:269,307-311: < 1e-4 -> arg-max, else softmax(logits / T) sampling
    tokens: List[int] = field(default_factory=list)
    pred_boxes: Optional[torch.Tensor] = None
    slot: Optional[int] = None
    prompt_len: int = 0                     # expanded length L in the cache
    done: bool = False
    error: Optional[str] = None


class _RowView:
    """KVCache interface over the first `n` rows of a cache, so GromaModel.forward prefills straight into it."""

    def __init__(self, arena, n):
        self.k = [t[:n] for t in arena.k]
        self.vt = [t[:n] for t in arena.vt]
        self.bs, self.smax, self.seq_len, self.sp = n, arena.smax, 0, getattr(arena, "sp", 1)

    def __len__(self):
        return len(self.k)

    def __getitem__(self, l):
        S = self.seq_len
        if self.sp == 2:  # reference-precision cache (engine.KVCache): the values the operand pairs stand for
            return (ops.unsplit(self.k[l])[:, :, :S], ops.unsplit(self.vt[l])[:, :, :, :S].transpose(2, 3))
        return (self.k[l][:, :, :S], self.vt[l][:, :, :, :S].transpose(2, 3))

    def __bool__(self):
        return True

    def grow(self, smax):
        raise RuntimeError("a KV slot cannot grow: the request exceeds max_len")


class ContinuousBatcher:
    @engine.model_entry(lambda self, *a, **kw: (a[0] if a else kw["model"]).precision)
    def __init__(self, model, max_rows=8, max_len=1024, use_graph=True, grow_to=None, overlap_admission=None, admit_min=2, admit_hold=8):
        if max_rows < 1 or max_rows > 64:
            raise ValueError("max_rows must be in 1..64 (1..8: the fused 8-row weight streams; 9..64: the matrix-unit weight streams, "
                             "csrc/gemm_skinny.hip / gemm_skinny_fp8.hip -- single-type 16-bit or e4m3 models; others fall back to the general kernels)")
        if max_len % 64 or max_len > 8192:
            raise ValueError("max_len must be a multiple of 64 and <= 8192")
        if grow_to is not None and (grow_to % 64 or grow_to < max_len or grow_to > 8192):
            raise ValueError("grow_to must be a multiple of 64 in max_len..8192")
        self.grow_to = int(grow_to or max_len)   # the arena may be re-allocated up to this many positions per slot (default: fixed)
        self.model, self.llm = model, model.llm
        self.rows, self.max_len, self.use_graph = max_rows, max_len, use_graph
        dev = model.device
        self.arena = self.llm.new_cache(max_rows, max_len, dev)
        self.staging = self.llm.new_cache(max_rows, max_len, dev)  # batched prefills land here, then move to their slots
        self.slots = SlotTable(max_rows)
        self.queue = deque()
        self.live = {}
        self._next_rid = 0
        # device-resident loop state: one entry per row
        self.tok = torch.zeros((max_rows,), dtype=I64, device=dev)
        self.nxt = torch.zeros((max_rows,), dtype=I64, device=dev)
        self.occupied = torch.zeros((max_rows,), dtype=I64, device=dev)   # gr_greedy_advance `unfinished`
        self.pos = torch.zeros((max_rows,), dtype=I32, device=dev)  # idle rows rewrite position 0 of their own (free) slot
        self.step_ctr = torch.zeros((1,), dtype=I32, device=dev)
        self.n_live = torch.zeros((1,), dtype=I32, device=dev)
        self._seq = torch.zeros((max_rows, 1), dtype=I64, device=dev)
        self.h = torch.zeros((max_rows, self.llm.T), dtype=F32, device=dev)
        self.inv_temp = torch.zeros((max_rows,), dtype=F32, device=dev)   # per-row 1/temperature (0 = greedy)
        self.seed = torch.zeros((max_rows,), dtype=I64, device=dev)       # per-row sampler seed
        self.graph = None
        self.steps = 0
        # overlapped admission: at most ONE prefill in flight (there is one staging cache), on `_side`; `_placed` orders the next
        # prefill's writes into the staging cache behind the previous batch's move into the arena
        if overlap_admission and not use_graph:
            raise ValueError("overlap_admission needs the captured decode step (use_graph=True): the eager step shares workspaces with the prefill")
        # admit_min: while rows are live, hold an admission until this many requests can go in ONE prefill (or the queue holds fewer):
        # a 1-image prefill costs 23 ms, a 4-image one 12 ms per image -- throughput for first-token latency; 1 = admit at once.  A held
        # request goes in after `admit_hold` ticks whatever has freed up (ragged traffic at 32 rows: 38.6 -> 44.9 img/s with 2,
        # profiles/r06_serve_admit_min.txt)
        self.admit_min, self.admit_hold, self._held = max(1, int(admit_min)), max(0, int(admit_hold)), 0
        self.overlap = bool(use_graph if overlap_admission is None else overlap_admission)   # default: on whenever the step is a graph
        self._job = None
        self._side = torch.cuda.Stream(device=dev) if self.overlap else None
        self._placed = None

    # ------------------------------------------------------------------ admission-time warm-up
    @engine.model_entry(lambda self, *a, **kw: self.model.precision)
    def warm_admission(self, input_ids, image, rows=1):
        """One throw-away admission prefill of `rows` copies of an example request BEFORE live traffic: the hipGraphs of the ViT,
        the region pyramid and the LLaMA prefill for that batch shape are captured on this pass (engine.GraphPool.first_sight)
        instead of inside the first three live admissions of the shape, and the workspace arenas reach their working size.
        Nothing of the example survives: the staging cache is overwritten by the next admission, no slot is taken."""
        if self.slots.n_free != self.rows:
            raise RuntimeError("warm_admission() must run before the first admission")
        if self.use_graph and self.graph is None:
            self._capture()
        k = max(1, min(int(rows), self.rows))
        ids = input_ids.reshape(1, -1).to(I64).cpu().repeat(k, 1)
        imgs = torch.stack([image] * k)
        def once():
            self.model.forward(input_ids=ids.clone(), images=imgs, use_cache=True, return_dict=True,
                               _cache=_RowView(self.staging, k), _seeds=[0] * k)
        # first pass EAGER: the workspace arenas grow to their working size and every kernel has had its first launch (lazy
        # function attributes, module load) -- a graph captured on this pass would be keyed on pre-growth addresses, never be
        # replayed, and sit in the LRU pools evicting useful ones.  Second pass: capture, on the settled addresses.
        # (overlapped admissions launch their GEMMs one workgroup per tile, ops.gemm_yield: capture that form)
        with engine.GraphPool.eager(), ops.gemm_yield(self.overlap):
            once()
        with engine.GraphPool.first_sight(), ops.gemm_yield(self.overlap):
            once()

    # ------------------------------------------------------------------ request intake
    def submit(self, input_ids, image, max_new_tokens=256, refer_boxes=None, ground_boxes=None, eos_token_id="config",
               stop_token_id=None, seed=None, temperature=0.0):
        if eos_token_id == "config":
            eos_token_id = self.model.generation_config.eos_token_id
        r = Request(self._next_rid, input_ids.reshape(-1).to(I64).cpu(), image, int(max_new_tokens), refer_boxes, ground_boxes,
                    eos_token_id, stop_token_id, seed, float(temperature))
        self._next_rid += 1
        self.queue.append(r)
        self.live[r.rid] = r
        return r.rid

    # ------------------------------------------------------------------ one decode step of every occupied row
    def _decode(self):
        llm = self.llm
        ops.embed_gather(self.tok, llm.w["embed"], llm.w["new_embed"], out=self.h)
        llm.forward(self.h, self.rows, 1, self.arena, pos_dev=self.pos, pos_stride=1)
        ops.sample_rows(llm.decode_logits(self.rows), llm.V, self.inv_temp, self.seed,
                        pos=self.pos, pos_stride=1, pos_off=1, out=self.nxt)  # greedy rows: inv_temp 0 -> arg-max
        ops.greedy_advance(self.nxt, self.tok, self.occupied, self._seq, self.pos, self.step_ctr, self.n_live,
                           eos=None, pad=0, inc_pos=2)

    def _capture(self):
        """Capture the step once, before any request owns a slot (the warm-up steps write KV at the idle position)."""
        side = torch.cuda.Stream(device=self.tok.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self._decode()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._decode()
        self.graph = g

    # ------------------------------------------------------------------ KV arena growth
    def _bound(self, r):
        """upper bound of a request's context: the expanded prompt (groma.py:317-357: <image> -> the image tokens, <region> -> two ids
        per selected region, at most max_region_num + the caller's refer / ground boxes) + its new tokens"""
        m = self.model
        extra = sum(int(b.shape[0]) for b in (r.refer_boxes, r.ground_boxes) if b is not None)
        return int(r.input_ids.numel()) + (m.vit.G // 2) ** 2 + 2 * (int(m.config.max_region_num) + extra) + r.max_new_tokens

    def _grow(self, need):
        """Re-allocate the arena (and the staging cache) at max(2 x max_len, need) positions per slot, capped by grow_to.  Live rows
        keep their KV prefix (K rows / V^T columns are copied; (hi, lo) pair storage interleaves in blocks of 32, so a prefix of
        64-multiples is a prefix there too) and their loop state; the decode step is re-captured on the new addresses.
        Failure-atomic: both new caches are allocated and the prefixes copied BEFORE anything of the batcher changes; if the device
        cannot hold them (peak = old arena + old staging + both new caches) the old caches stay in service, False is returned and
        the over-long request is refused by the ordinary length checks of _admit() -- live rows are never stranded.
        Cost of a successful growth in the middle of live traffic: the copy, three eager steps + one capture, one device sync."""
        old = self.max_len
        new_len = min(self.grow_to, max(2 * old, -(-int(need) // 64) * 64))
        if new_len <= old or new_len >= getattr(self, "_grow_failed_at", 1 << 30):
            return False
        dev = self.tok.device
        try:
            staging = self.llm.new_cache(self.rows, new_len, dev)
            arena = self.llm.new_cache(self.rows, new_len, dev)
            sp = getattr(arena, "sp", 1)
            for l in range(len(arena.k)):
                arena.k[l][:, :, :old].copy_(self.arena.k[l])
                arena.vt[l][..., : old * sp].copy_(self.arena.vt[l])
        except RuntimeError as e:   # (torch.cuda.OutOfMemoryError is a RuntimeError)
            staging = arena = None
            self._grow_failed_at = new_len       # do not retry this size for every over-long request of the queue
            self.last_grow_error = str(e)
            torch.cuda.empty_cache()
            return False
        # commit: from here on nothing can fail half-way
        stale = set(engine.cache_addresses(self.staging)[1:])
        self.staging, self.arena, self.max_len = staging, arena, new_len
        # prefill graphs captured against the old staging cache can never be replayed again: free their slots in the LRU pool
        self.llm.graphs.drop(lambda key: any(a in stale for a in key))
        if self.graph is not None:
            # the three eager warm-up steps in front of a capture advance every row: keep the loop state aside and put it back (what they
            # wrote into live rows' KV lies at positions >= pos, which the next real step rewrites before it attends to them)
            keep = (self.tok, self.nxt, self.occupied, self.pos, self.step_ctr, self.n_live, self._seq, self.h)
            saved = [t.clone() for t in keep]
            self.graph = None
            self._capture()
            for t, v in zip(keep, saved):
                t.copy_(v)
        return True

    def _grow_for(self, reqs):
        if self.max_len < self.grow_to:
            # (_bound is a worst case -- the real spliced length is only known after the prefill -- so a request that would just
            #  have fitted can trigger a growth; the alternative, prefilling first and re-admitting, costs a second prefill)
            need = max(self._bound(r) for r in reqs)
            if need > self.max_len:
                self._grow(need)   # False: the old caches stay; the length checks of the prefill refuse what does not fit

    # ------------------------------------------------------------------ admission: one batched prefill per tick
    def _admit(self, reqs):
        """Prefill `reqs` (<= free slots) in ONE forward and move each row's KV into its slot.  Row results of the
        prefill are bit-identical to a batch-1 forward of the same request (every kernel of the path is row- /
        image-independent, tests/test_fullsize_properties_gpu.py), so admission order and company never change a
        request's tokens; each request's region shuffle draws from its own seed."""
        self._grow_for(reqs)
        self._place(reqs, self._prefill(reqs))

    def _prefill(self, reqs):
        """The device half of an admission, on the CURRENT stream: one batched forward into the staging cache + each row's first token.
        Returns [(request, staging row, first token, 1/temperature)] for the requests that go on; touches neither the slot table nor
        the arena nor the loop state (with overlap_admission it runs on the worker thread while the decode ticks use those)."""
        m, k = self.model, len(reqs)
        P = max(r.input_ids.numel() for r in reqs)
        ids = torch.full((k, P), int(m.pad_token_id), dtype=I64)
        for i, r in enumerate(reqs):
            ids[i, : r.input_ids.numel()] = r.input_ids
        empty = torch.zeros((0, 4))
        rb = [r.refer_boxes if r.refer_boxes is not None else empty for r in reqs] if any(r.refer_boxes is not None for r in reqs) else None
        gb = [r.ground_boxes if r.ground_boxes is not None else empty for r in reqs] if any(r.ground_boxes is not None for r in reqs) else None
        view = _RowView(self.staging, k)
        try:
            out = m.forward(input_ids=ids, images=torch.stack([r.image for r in reqs]), refer_boxes=rb, ground_boxes=gb,
                            use_cache=True, return_dict=True, output_hidden_states=True, _cache=view,
                            _seeds=[r.seed for r in reqs])
        except RuntimeError as e:
            if k == 1:
                reqs[0].done, reqs[0].error = True, str(e)
                return []
            # isolate the offender: one by one.  Each survivor's KV must be in ITS OWN staging row when the batch is placed, and a
            # batch-1 prefill lands in row 0 -- so the survivors are moved to their rows as they come
            placed = []
            for i in range(k - 1, -1, -1):   # (last to first: row 0 is where every batch-1 prefill lands, so its own request goes last)
                one = self._prefill([reqs[i]])
                if one:
                    if i:
                        for l in range(len(self.staging.k)):
                            self.staging.k[l][i].copy_(self.staging.k[l][0])
                            self.staging.vt[l][i].copy_(self.staging.vt[l][0])
                    placed.append((reqs[i], i, one[0][2], one[0][3]))
            placed.reverse()
            return placed
        lengths = m._last_aux["lengths"]
        placed = []
        dev = self.tok.device
        for i, r in enumerate(reqs):
            r.prompt_len = int(lengths[i])
            r.pred_boxes = out.hidden_states[1]["pred_boxes"][i]
            if r.prompt_len + r.max_new_tokens > self.max_len:
                r.done, r.error = True, "prompt + max_new_tokens exceeds the KV slot (max_len)"
                continue
            it = 0.0 if r.temperature < 1e-4 else 1.0 / r.temperature
            first = int(ops.sample_rows(out.logits[i, r.prompt_len - 1][None].contiguous(), out.logits.shape[-1],
                                        torch.tensor([it], dtype=F32, device=dev),
                                        torch.tensor([int(r.seed or 0)], dtype=I64, device=dev),
                                        pos=torch.tensor([r.prompt_len], dtype=I32, device=dev))[0])
            placed.append((r, i, first, it))
        return placed

    def _place(self, reqs, placed):
        """The host half of an admission, on the decode stream between two ticks: emit the first tokens, hand out slots, move the rows'
        KV from the staging cache into the arena, arm the loop state."""
        rows, slots = [], []
        for r, i, first, it in placed:
            self._emit(r, first)
            if r.done:
                continue
            r.slot = self.slots.acquire(r.rid)
            rows.append(i)
            slots.append(r.slot)
            self.tok[r.slot] = first
            self.pos[r.slot] = r.prompt_len
            self.occupied[r.slot] = 1
            self.inv_temp[r.slot] = it
            self.seed[r.slot] = int(r.seed or 0)
        if rows:
            dev = self.tok.device
            src = torch.tensor(rows, dtype=I64, device=dev)
            dst = torch.tensor(slots, dtype=I64, device=dev)
            # only the prefix the prompts fill (whole 64-position blocks: a prefix in the pair layout too): what lies behind a row's
            # position is rewritten by its own decode steps before it is ever attended to
            n = min(self.max_len, -(-max(r.prompt_len for r, _, _, _ in placed if r.slot is not None) // 64) * 64)
            sp = getattr(self.arena, "sp", 1)
            for l in range(len(self.arena.k)):
                self.arena.k[l][:, :, :n].index_copy_(0, dst, self.staging.k[l][:, :, :n].index_select(0, src))
                self.arena.vt[l][..., : n * sp].index_copy_(0, dst, self.staging.vt[l][..., : n * sp].index_select(0, src))

    # ------------------------------------------------------------------ overlapped admission (worker thread + side stream)
    def _launch_admission(self, reqs):
        """start the prefill of `reqs` on the worker thread; the caller has made sure no other one is in flight"""
        job = {"reqs": reqs, "placed": [], "done": threading.Event(), "event": None, "error": None}
        if self._placed is not None:
            self._side.wait_event(self._placed)   # the previous batch has left the staging cache
        else:
            self._side.wait_stream(torch.cuda.current_stream())

        job["thread"] = threading.Thread(target=self._admission_work, args=(job,), name="groma-admission", daemon=True)
        self._job = job
        job["thread"].start()

    @engine.model_entry(lambda self, *a, **kw: self.model.precision)
    def _admission_work(self, job):
        """worker thread: the operand type, no-grad mode, current stream and GEMM grid form are all per thread"""
        try:
            with torch.cuda.stream(self._side), ops.gemm_yield():   # (one workgroup per GEMM tile: the ticks' kernels get CUs as tiles retire)
                job["placed"] = self._prefill(job["reqs"])
                ev = torch.cuda.Event()
                ev.record(self._side)
                job["event"] = ev
        except BaseException as e:   # never leave the batcher waiting for a job that died
            job["error"] = e
        finally:
            job["done"].set()

    def _finish_admission(self, wait=False):
        """place the rows of a finished prefill (wait=True: block until it has finished).  Returns the admission events."""
        job = self._job
        if job is None or not (wait or job["done"].is_set()):
            return []
        job["done"].wait()
        job["thread"].join()
        self._job = None
        if job["error"] is not None:
            for r in job["reqs"]:
                if not r.done:
                    r.done, r.error = True, f"admission failed: {job['error']}"
            if not isinstance(job["error"], Exception):
                raise job["error"]
        else:
            torch.cuda.current_stream().wait_event(job["event"])
            self._place(job["reqs"], job["placed"])
            self._placed = torch.cuda.Event()
            self._placed.record(torch.cuda.current_stream())
        return [(r.rid, r.tokens[-1] if r.tokens else None, r.done) for r in job["reqs"]]

    def _emit(self, r, token):
        r.tokens.append(token)
        if (r.eos_token_id is not None and token == r.eos_token_id) or \
                (r.stop_token_id is not None and token == r.stop_token_id) or len(r.tokens) >= r.max_new_tokens:
            self._finish(r)

    def _finish(self, r):
        r.done = True
        if r.slot is not None:
            s = r.slot
            self.occupied[s] = 0
            self.pos[s] = 0
            self.slots.release(s)
            r.slot = None

    # ------------------------------------------------------------------ scheduler tick
    @engine.model_entry(lambda self, *a, **kw: self.model.precision)
    def step(self):
        """Admit what fits, then advance every occupied row by one token.  Returns [(rid, token, done), ...]."""
        if self.use_graph and self.graph is None:
            if self.slots.n_free != self.rows:
                raise RuntimeError("capture must precede the first admission")
            self._capture()
        events = []
        if self.overlap:
            events += self._finish_admission()
            if self._job is None and self._may_admit():
                batch = [self.queue.popleft() for _ in range(min(len(self.queue), self.slots.n_free))]
                self._grow_for(batch)
                self._launch_admission(batch)
            if self._job is not None and not self.slots.active():
                events += self._finish_admission(wait=True)   # nothing to decode meanwhile: wait for the rows
        elif self._may_admit():
            batch = [self.queue.popleft() for _ in range(min(len(self.queue), self.slots.n_free))]
            self._admit(batch)
            for r in batch:
                events.append((r.rid, r.tokens[-1] if r.tokens else None, r.done))
        act = self.slots.active()
        if not act:
            return events
        if self.use_graph:
            self.graph.replay()
        else:
            self._decode()
        self.steps += 1
        new = self.tok.tolist()  # the only per-step host read: max_rows ids
        for s in act:
            r = self.live[self.slots.owner[s]]
            self._emit(r, int(new[s]))
            events.append((r.rid, r.tokens[-1], r.done))
        return events

    def _may_admit(self):
        n = min(len(self.queue), self.slots.n_free)
        if n == 0:
            return False
        if n >= min(self.admit_min, len(self.queue)) or self.slots.n_free == self.rows or self._held >= self.admit_hold:
            self._held = 0
            return True
        self._held += 1
        return False

    def run_until_done(self, max_steps=100000):
        for _ in range(max_steps):
            if not self.queue and not self.slots.active() and self._job is None:
                break
            self.step()
        return {rid: r for rid, r in self.live.items()}

    def result(self, rid, pop=True):
        r = self.live.pop(rid) if pop else self.live[rid]
        return r

    # ------------------------------------------------------------------ model_worker.generate_stream analogue
    def generate_stream(self, input_ids, image, max_new_tokens=256, stream_interval=1, **kw):
        """Yield the generated ids of ONE request every `stream_interval` tokens while other requests keep decoding
        (groma/serve/model_worker.py:287-338 yields the decoded text; tokenisation stays with the caller)."""
        rid = self.submit(input_ids, image, max_new_tokens=max_new_tokens, **kw)
        r = self.live[rid]
        sent = 0
        while not r.done:
            self.step()
            if r.error:
                raise RuntimeError(r.error)
            if len(r.tokens) - sent >= stream_interval or r.done:
                sent = len(r.tokens)
                yield list(r.tokens)
        self.live.pop(rid, None)
