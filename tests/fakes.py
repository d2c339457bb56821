"""Test doubles: an oracle-backed runtime with the same surface as ai00_server_amd.runtime.Runtime, so the host-side
mirror (harness, router) can be exercised on a machine without a GPU.  TEST INFRASTRUCTURE ONLY."""
import numpy as np

from ai00_server_amd.runtime import RnnInput, RnnOption


class _State:
    def __init__(self, rt):
        self.rt = rt
        self.pending = []

    @property
    def shape(self):
        return self.rt.ref.state_shape()

    def init(self):
        return self.rt.ref.init_state()

    def load(self, tensor, batch):
        self.rt.states[batch] = np.array(tensor, dtype=np.float32).reshape(self.rt.ref.init_state().shape)

    def back(self, batch):
        return self.rt.states[batch].copy()

    def read(self, batch):
        return self.rt.states[batch].copy()

    def write(self, tensor, batch):
        self.rt.states[batch] = tensor.copy()

    def embed(self, layer, batch):
        n = self.rt.ref.info.head_size
        return self.rt.states[batch][layer, 1:1 + n].copy()

    def embed_async(self, layer, batch, dst):
        """The real one returns at once and fills `dst` on the copy stream; here the copy is deferred to `sync()` so that a driver
        which re-used the slot too early (before the engine ordered the pack) or read `dst` before `sync()` is caught."""
        self.pending.append((dst, self.embed(layer, batch)))
        dst[...] = np.nan

    def sync(self):
        for dst, val in self.pending:
            dst[...] = val
        self.pending = []


class OracleRuntime:
    """CPU stand-in with Runtime's surface; chunks like the engine: <= token_chunk_size tokens per infer call."""

    def __init__(self, ref, max_batch=4, token_chunk_size=8):
        self.ref, self.max_batch, self.token_chunk_size = ref, max_batch, token_chunk_size
        self.states = [ref.init_state() for _ in range(max_batch)]
        self.state = _State(self)
        self.info = ref.info
        self.calls = 0

    def infer(self, inp: RnnInput):
        self.calls += 1
        budget = self.token_chunk_size
        outs = []
        active = [b for b in inp.batches if len(b.tokens)]
        share = max(1, budget // max(1, len(active)))
        for b, ib in enumerate(inp.batches):
            n = min(len(ib.tokens), share, budget)
            if n == 0:
                outs.append(np.zeros((0, self.ref.info.num_vocab), np.float32))
                continue
            budget -= n
            toks, ib.tokens = ib.tokens[:n], ib.tokens[n:]
            full = ib.option == RnnOption.Full
            lg = self.ref.forward(toks, self.states[b], full=True)
            if full:
                outs.append(lg)
            elif ib.option == RnnOption.NoOutput:
                outs.append(np.zeros((0, self.ref.info.num_vocab), np.float32))
            elif len(ib.tokens) == 0:
                outs.append(lg[-1:])
            else:
                outs.append(np.zeros((0, self.ref.info.num_vocab), np.float32))
        return inp, outs
