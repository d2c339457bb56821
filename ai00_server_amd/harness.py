"""Mirror of the ai00-core code that *calls* the hot path (crates/ai00-core/src/run.rs), for tests and benches.

    InferLoop.run_pending     the `infer` task        run.rs:1072-1162  (per-slot FIFO, one request per slot per
                                                       step, loop until the RnnInput is consumed)
    greedy_process            `process` decode loop   run.rs:788-1020   with the arg-max sampler (Nucleus top_k=1,
                                                       sampler/nucleus.rs:77-89) and token 0 = stop (run.rs:855)
    perplexity                `perplexity`            run.rs:699-755    (RnnOption::Full consumer)
    ReplicaRouter             SURVEY 8(e)             request-level sharding over N independent engines (no collective)

Works against anything that has `.max_batch`, `.infer(RnnInput)` and `.state` like `runtime.Runtime`.
"""
from __future__ import annotations

import math
from collections import deque
from dataclasses import dataclass, field

import numpy as np

from .runtime import RnnInput, RnnInputBatch, RnnOption


@dataclass
class InferRequest:            # InferBatch::Run {batch, tokens, option, sender}  run.rs:1091-1098
    batch: int
    tokens: list
    option: RnnOption = RnnOption.Last
    outputs: list = field(default_factory=list)     # what `sender` would receive: one array per emitting infer call


class InferLoop:
    """The single consumer of InferBatch messages (run.rs:1072-1162)."""

    def __init__(self, runtime):
        self.rt = runtime
        self.queues: dict[int, deque] = {}

    def submit(self, req: InferRequest) -> InferRequest:          # schedule(): Run -> per-slot VecDeque
        self.queues.setdefault(req.batch, deque()).append(req)
        return req

    def run_pending(self) -> int:
        """`while batches.values().map(len).sum() > 0 { ... }` (run.rs:1120-1157). Returns #infer calls."""
        calls = 0
        while sum(len(q) for q in self.queues.values()) > 0:
            inference = [RnnInputBatch() for _ in range(self.rt.max_batch)]
            senders: dict[int, InferRequest] = {}
            for b, q in self.queues.items():                      # pop <= 1 request per slot (run.rs:1124-1130)
                if not q:
                    continue
                req = q.popleft()
                inference[b] = RnnInputBatch(list(req.tokens), req.option)
                senders[b] = req
            inp = RnnInput(inference)
            while inp.num_token() > 0:                            # run.rs:1134-1156
                inp, out = self.rt.infer(inp)
                calls += 1
                for b, o in enumerate(out):
                    if len(o) and b in senders:
                        senders[b].outputs.append(o)
        return calls


def greedy_process(loop: InferLoop, batch: int, prompt, max_tokens: int) -> list[int]:
    """Decode loop of one slot with the arg-max sampler; empty prompt => [0] (run.rs:489-492); token 0 stops."""
    tokens = list(prompt) if len(prompt) else [0]
    out = []
    for _ in range(max_tokens):
        req = loop.submit(InferRequest(batch, tokens, RnnOption.Last))
        loop.run_pending()
        logits = req.outputs[-1][-1]
        tok = int(np.argmax(logits))
        if tok == 0:                                              # run.rs:855
            break
        out.append(tok)
        tokens = [tok]
    return out


def perplexity(loop: InferLoop, batch: int, tokens, head: float | None = None) -> float:
    """run.rs:699-755: Full outputs, softmax without max-subtraction, ppl = -sum(ln p) / len(tokens')."""
    p = []
    toks = list(tokens) if head is not None else [0] + list(tokens)
    n = len(tokens)
    if head is not None:
        p.append(head)
    req = loop.submit(InferRequest(batch, toks, RnnOption.Full))
    loop.run_pending()
    index = 1
    for chunk in req.outputs:                                     # one array per infer call, split(1) per token
        for row in chunk:
            if len(p) >= n:
                break
            if index < len(toks):
                e = np.exp(row.astype(np.float32))
                p.append(float(e[toks[index]] / e.sum(dtype=np.float32)))
            index += 1
    return float(-sum(math.log(x) for x in p) / len(toks))


class NucleusSampler:
    """Host-side STATE of `NucleusSampler` (sampler/nucleus.rs:13-122) for the on-device sampling front-end: the
    penalty map lives here, its effect reaches the device as sparse logit adjustments (`adjustments()`), and the
    token the device picked is fed back through `update()` (nucleus.rs:104-119)."""

    def __init__(self, top_p=0.5, top_k=128, temperature=1.0, presence_penalty=0.3, frequency_penalty=0.3,
                 penalty_decay=0.99654026, bias: dict | None = None):
        self.top_p, self.top_k, self.temperature = float(top_p), int(top_k), float(temperature)
        self.ap, self.af, self.ad = np.float32(presence_penalty), np.float32(frequency_penalty), np.float32(penalty_decay)
        self.penalties: dict[int, np.float32] = {}
        self.bias = dict(bias or {})

    def init(self, model_tokens):                                    # nucleus.rs:49-59
        for index, token in enumerate(reversed(list(model_tokens))):
            pen = self.penalties.pop(int(token), self.ap)
            self.penalties[int(token)] = np.float32(pen + self.af * self.ad ** np.float32(index))

    def adjustments(self) -> dict:                                    # transform (nucleus.rs:61-67) + bias (run.rs:681-683)
        adj = {t: np.float32(-p) for t, p in self.penalties.items()}
        for t, b in self.bias.items():
            adj[t] = np.float32(adj.get(t, np.float32(0)) + np.float32(b))
        return adj

    def update(self, token: int):                                     # nucleus.rs:104-119
        for t in self.penalties:
            self.penalties[t] = np.float32(self.penalties[t] * self.ad)
        self.penalties[token] = np.float32(self.penalties[token] + self.af) if token in self.penalties else self.ap


class TypicalSampler(NucleusSampler):
    """Host-side state of `TypicalSampler` (sampler/typical.rs:11-140): the same penalty state machine as the nucleus
    sampler (init typical.rs:47-58, transform :62-68, update :122-133); the device applies tau / top_k / temperature."""
    kind = 1

    def __init__(self, tau=0.5, top_k=128, temperature=1.0, presence_penalty=0.3, frequency_penalty=0.3,
                 penalty_decay=0.99654026, bias: dict | None = None):
        super().__init__(top_p=0.0, top_k=top_k, temperature=temperature, presence_penalty=presence_penalty,
                         frequency_penalty=frequency_penalty, penalty_decay=penalty_decay, bias=bias)
        self.tau = float(tau)


class MirostatSampler:
    """Host-side state of `MirostatSampler` (sampler/mirostat.rs:11-90): `max_surprise` lives here; the device sorts,
    truncates at max_surprise, samples and returns the token surprise; `update()` applies mirostat.rs:85-87."""
    kind = 2
    top_k, temperature, top_p = 1, 1.0, 0.0

    def __init__(self, tau=3.0, rate=0.1):
        self.target, self.rate = np.float32(tau), np.float32(rate)
        self.max_surprise = np.float32(2.0 * tau)

    @property
    def tau(self):                                                    # what the device calls tau for kind 2
        return float(self.max_surprise)

    def init(self, model_tokens):                                     # mirostat.rs:38
        pass

    def adjustments(self) -> dict:                                    # transform is a no-op (mirostat.rs:40)
        return {}

    def update(self, token_surprise: float):
        err = np.float32(np.float32(token_surprise) - self.target)
        self.max_surprise = np.float32(min(np.float32(self.max_surprise - self.rate * err), np.float32(4.0) * self.target))


class StateJob:
    """`GenerateKind::State` requests as a batch job with SLOT TURNOVER — the documented `/embeddings` route (docs/doc-api/openai.md:
    376-437; `GenerateKind::State` run.rs:980-989; `api/oai/state.rs:29-40`) fed a list of documents.  The scheduling is the reference's
    (`queue` run.rs:488-626 -> `infer` steps run.rs:1121-1157 -> `finish`), the C++ form is `rwkv::Scheduler::queue / step / state`
    (include/rwkv_scheduler.hpp): every idle slot takes the next waiting document (an `Empty` slot: nothing cached matches a fresh document, so
    the state it starts from is the initial one, written from a device-resident snapshot like `state.write(backed.clone(), batch)`
    run.rs:1104), all busy slots ride the same `infer` call (`RWKV_OPTION_NONE`: state-only, no head GEMM), and a slot whose document has
    been consumed hands over its embedding — one layer's WKV rows, `rwkv_state_back_layer_async` into pinned memory on the engine's copy
    stream — and is free for the next document at once: the read-back of a finished document overlaps the prefill of the following ones.

    `run(docs)` returns (embeddings [n_docs, N, C] in pinned memory, number of infer calls)."""

    def __init__(self, runtime, layer: int | None = None, arena=None):
        from .runtime import PinnedArena
        self.rt = runtime
        self.layer = runtime.info.num_layer - 1 if layer is None else int(layer)
        self._Arena = arena or PinnedArena           # `arena(shape)` -> object with `.array`, `.close()` (tests pass a numpy one)
        # the initial state, device-resident (state.init() -> load -> read once; `backed.clone()` afterwards)
        runtime.state.load(runtime.state.init(), 0)
        self.zero = runtime.state.read(0)
        self.out = None

    def run(self, docs: list):
        rt, B = self.rt, self.rt.max_batch
        N, Cn = rt.info.head_size, rt.info.num_emb
        if self.out is None or self.out.array.shape[0] < len(docs):
            if self.out is not None:
                self.out.close()
            self.out = self._Arena((len(docs), N, Cn))
        out = self.out.array
        # documents as uint32 arrays, once: a step then hands the engine views (no per-token Python work per call); empty prompt => [0] (run.rs:489-492)
        arr = [np.asarray(d if len(d) else [0], dtype=np.uint32) for d in docs]
        waiting = deque(range(len(docs)))
        owner = [-1] * B                        # document id a slot works on
        none = np.zeros(0, np.uint32)
        rest = [none for _ in range(B)]         # its tokens not yet consumed
        calls = 0
        while waiting or any(o >= 0 for o in owner):
            for b in range(B):                  # `queue`: the next document takes an idle slot
                if owner[b] < 0 and waiting:
                    d = waiting.popleft()
                    rt.state.write(self.zero, b)
                    owner[b], rest[b] = d, arr[d]
            inp = RnnInput([RnnInputBatch(rest[b] if owner[b] >= 0 else none, RnnOption.NoOutput) for b in range(B)])
            inp, _ = rt.infer(inp)              # one step over <= token_chunk_size tokens of all busy slots
            calls += 1
            for b in range(B):
                if owner[b] < 0:
                    continue
                rest[b] = inp.batches[b].tokens
                if not len(rest[b]):            # `finish`: the embedding leaves on the copy stream, the slot is idle again
                    rt.state.embed_async(self.layer, b, out[owner[b]])
                    owner[b] = -1
        rt.state.sync()
        return out[:len(docs)], calls

    def close(self):
        if self.out is not None:
            self.out.close()
            self.out = None


class ReplicaRouter:
    """Request-level sharding over independent engines (one per GPU): round-robin for batch jobs, least-busy for
    interactive requests.  No collective: each replica owns its weights, slots and stream.  (The full policy —
    longest cached prefix first, full replicas skipped, one driving thread per engine — is the C++ `rwkv::ReplicaRouter`,
    include/rwkv_router.hpp; this class is the single-threaded form the Python tests drive.)"""

    def __init__(self, runtimes: list):
        self.loops = [InferLoop(r) for r in runtimes]
        self.busy = [0] * len(runtimes)

    def pick(self) -> int:
        """Least-busy replica that still has a free slot (ties: lowest index); -1 when every replica is full."""
        free = [j for j in range(len(self.busy)) if self.busy[j] < self.loops[j].rt.max_batch]
        if not free:
            return -1
        i = min(free, key=lambda j: self.busy[j])
        self.busy[i] += 1
        return i

    def release(self, i: int):
        self.busy[i] -= 1

    def generate(self, prompts: list, n_new: int) -> list:
        """Interactive requests: each prompt goes to the least-busy replica (`pick`), takes the next free slot there, and is
        decoded greedily (`greedy_process`); a prompt that finds every replica full waits for the current wave to finish."""
        out = [None] * len(prompts)
        todo = list(range(len(prompts)))
        while todo:
            wave = []
            used = [0] * len(self.loops)
            while todo:
                r = self.pick()
                if r < 0:
                    break
                i = todo.pop(0)
                wave.append((i, r, used[r]))
                used[r] += 1
            for i, r, slot in wave:
                self.loops[r].rt.state.load(self.loops[r].rt.state.init(), slot)
                out[i] = greedy_process(self.loops[r], slot, list(prompts[i]), n_new)
                self.release(r)
        return out

    @staticmethod
    def shard(n_items: int, rank: int, world: int) -> range:
        """Indices of a batch job owned by `rank` (documents round-robin)."""
        return range(rank, n_items, world)

    def embed_documents(self, docs: list, layer: int) -> list:
        """`/embeddings`-style job: prefill each document, return one layer's WKV rows as its embedding."""
        out = [None] * len(docs)
        for r, loop in enumerate(self.loops):
            mine = list(self.shard(len(docs), r, len(self.loops)))
            B = loop.rt.max_batch
            for g in range(0, len(mine), B):
                group = mine[g:g + B]
                for slot, d in enumerate(group):
                    loop.rt.state.load(loop.rt.state.init(), slot)
                    loop.submit(InferRequest(slot, list(docs[d]), RnnOption.NoOutput))   # state-only: no head GEMM, no logits
                loop.run_pending()
                for slot, d in enumerate(group):
                    out[d] = loop.rt.state.embed(layer, slot)
        return out
