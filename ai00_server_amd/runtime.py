"""Host-side mirror of the `web-rwkv` surface that ai00-core uses, bound to librwkv_hip.so over ctypes.

Names follow the reference imports (crates/ai00-core/src/lib.rs:24-35, run.rs:22-31):

    Loader.info            lib.rs:587            -> ModelInfo
    ModelBuilder(...).quant(..).lora(..).build() lib.rs:484-516  -> Runtime (+ .state)
    Runtime.infer(RnnInput) -> (RnnInput, RnnOutput)   run.rs:1143
    RnnInput / RnnInputBatch / RnnOption                 run.rs:1128-1132
    State.init/load/back/read/write                      run.rs:477, 1099-1106
    softmax(context, [TensorCpu])                        run.rs:1179
    Tokenizer.encode/decode/token_index_to_bytes         lib.rs:375, run.rs:157-168,856

There is NO fallback: if the shared library is missing or no gfx950 device is present, construction
raises.  Nothing under `oracle/` is ever imported here.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
import time
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RWKV_HIP_LIB") or os.path.join(_HERE, "librwkv_hip.so")   # override: A/B builds (scripts/build_variant.py)


class RwkvError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rwkv error {code}: {msg}")
        self.code = code


class _ModelInfoC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("version", "num_layer", "num_emb", "num_hidden", "num_vocab", "num_head",
                                         "head_size", "reserved")]


class _LoraC(C.Structure):
    _fields_ = [("st_bytes", C.c_void_p), ("st_len", C.c_size_t), ("alpha", C.c_float)]


class _LoadDescC(C.Structure):
    _fields_ = [("adapter", C.c_int32), ("quant_layers", C.c_int32), ("quant_type", C.c_int32),
                ("precision", C.c_int32), ("max_batch", C.c_int32), ("token_chunk_size", C.c_int32),
                ("st_bytes", C.c_void_p), ("st_len", C.c_size_t), ("lora", C.POINTER(_LoraC)), ("n_lora", C.c_size_t)]


class _SlotInC(C.Structure):
    _fields_ = [("tokens", C.POINTER(C.c_uint32)), ("n_tokens", C.c_size_t), ("option", C.c_int32),
                ("reserved", C.c_int32)]


class _SampleC(C.Structure):
    _fields_ = [("top_p", C.c_float), ("top_k", C.c_int32), ("temperature", C.c_float), ("uniform", C.c_float),
                ("adj_tokens", C.POINTER(C.c_uint32)), ("adj_values", C.POINTER(C.c_float)), ("n_adj", C.c_size_t),
                ("kind", C.c_int32), ("tau", C.c_float), ("allow", C.POINTER(C.c_uint8))]


class _SlotOutC(C.Structure):
    _fields_ = [("logits", C.POINTER(C.c_float)), ("logits_capacity_rows", C.c_size_t), ("n_rows", C.c_size_t),
                ("n_consumed", C.c_size_t)]


# every symbol include/rwkv_abi.h declares: (restype, argtypes)
ABI_SYMBOLS = {
    "rwkv_last_error": (C.c_char_p, []),
    "rwkv_abi_version": (C.c_int32, []),
    "rwkv_device_count": (C.c_int32, []),
    "rwkv_device_name": (C.c_int32, [C.c_int32, C.c_char_p, C.c_size_t]),
    "rwkv_model_info_from_st": (C.c_int32, [C.c_void_p, C.c_size_t, C.POINTER(_ModelInfoC)]),
    "rwkv_engine_create": (C.c_int32, [C.POINTER(_LoadDescC), C.POINTER(C.c_void_p)]),
    "rwkv_engine_destroy": (None, [C.c_void_p]),
    "rwkv_engine_save_prefab": (C.c_int32, [C.c_void_p, C.c_char_p]),
    "rwkv_engine_info": (C.c_int32, [C.c_void_p, C.POINTER(_ModelInfoC)]),
    "rwkv_engine_device": (C.c_int32, [C.c_void_p]),
    "rwkv_engine_max_batch": (C.c_int32, [C.c_void_p]),
    "rwkv_engine_token_chunk_size": (C.c_int32, [C.c_void_p]),
    "rwkv_engine_weight_bytes": (C.c_uint64, [C.c_void_p]),
    "rwkv_infer": (C.c_int32, [C.c_void_p, C.POINTER(_SlotInC), C.POINTER(_SlotOutC)]),
    "rwkv_host_alloc": (C.c_int32, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "rwkv_host_free": (None, [C.c_void_p]),
    "rwkv_infer_sample": (C.c_int32, [C.c_void_p, C.POINTER(_SlotInC), C.POINTER(_SampleC), C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)]),
    "rwkv_plan_chunk": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]),
    "rwkv_state_len": (C.c_size_t, [C.c_void_p]),
    "rwkv_state_shape": (None, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "rwkv_state_init": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "rwkv_state_load": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "rwkv_state_back": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "rwkv_state_read": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "rwkv_state_write": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p]),
    "rwkv_dstate_free": (None, [C.c_void_p]),
    "rwkv_state_back_layer": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "rwkv_state_back_layer_async": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "rwkv_state_sync": (C.c_int32, [C.c_void_p]),
    "rwkv_read_init_state": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rwkv_softmax": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t]),
    "rwkv_tokenizer_create": (C.c_int32, [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rwkv_tokenizer_destroy": (None, [C.c_void_p]),
    "rwkv_tokenizer_encode": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t]),
    "rwkv_tokenizer_decode": (C.c_int64, [C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t, C.c_char_p, C.c_size_t]),
    "rwkv_tokenizer_token_bytes": (C.c_int64, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]),
    "rwkv_tokenizer_vocab_size": (C.c_int64, [C.c_void_p]),
    "rwkv_profile_family_name": (C.c_char_p, [C.c_int32]),
    "rwkv_profile_infer": (C.c_int32, [C.c_void_p, C.POINTER(_SlotInC), C.POINTER(_SlotOutC), C.POINTER(C.c_float),
                                       C.POINTER(C.c_int32)]),
    "rwkv_decode_greedy": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_float)]),
    "rwkv_bench_gemm": (C.c_int32, [C.c_int32] * 8 + [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
}
PROFILE_FAMILIES = 8

_lib = None


def lib() -> C.CDLL:
    """Load librwkv_hip.so (built in-tree by `python -m ai00_server_amd.build`).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built: run `python -m ai00_server_amd.build` "
                                    "(there is no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in ABI_SYMBOLS.items():
            fn = getattr(l, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def _check(code: int):
    if code != 0:
        raise RwkvError(code, lib().rwkv_last_error().decode(errors="replace"))


def _buf(data) -> tuple[C.c_void_p, int, object]:
    """Zero-copy view of bytes / bytearray / numpy / mmap as (ptr, len, keepalive)."""
    if isinstance(data, np.ndarray):
        arr = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    else:
        arr = np.frombuffer(data, dtype=np.uint8)
    return C.c_void_p(arr.ctypes.data), arr.size, arr


# ------------------------------------------------------------------------------------------------
class ModelVersion(enum.IntEnum):
    V5 = 5
    V6 = 6
    V7 = 7


class Quant(enum.IntEnum):       # `Quant` lib.rs:689-704 (SF4 unsupported)
    NONE = 0
    Int8 = 1
    NF4 = 2


class Precision(enum.IntEnum):   # reload.rs:89-94 (+ the raw mode of ABI 7, include/rwkv_abi.h)
    Fp16 = 0                     # f16 operands, the error-carrying launches hi + lo: within 1e-3 at 32 layers (the default)
    Fp32 = 1                     # hi + lo operands everywhere (fp32-class)
    Fp16Raw = 2                  # f16 operands everywhere: fastest, not tolerance-holding at depth


class RnnOption(enum.IntEnum):   # run.rs:716, 819
    Last = 0
    Full = 1
    NoOutput = 2                 # extension (RWKV_OPTION_NONE): state-only jobs such as `/embeddings`; no logits row is produced


@dataclass
class ModelInfo:
    version: ModelVersion
    num_layer: int
    num_emb: int
    num_hidden: int
    num_vocab: int
    num_head: int
    head_size: int


def _info_from_c(c: _ModelInfoC) -> ModelInfo:
    return ModelInfo(ModelVersion(c.version), c.num_layer, c.num_emb, c.num_hidden, c.num_vocab, c.num_head,
                     c.head_size)


class Loader:
    @staticmethod
    def info(st_bytes) -> ModelInfo:
        """`Loader::info(&SafeTensors)` (lib.rs:587, api/file.rs:113-116). Host only, needs no GPU."""
        p, n, keep = _buf(st_bytes)
        out = _ModelInfoC()
        _check(lib().rwkv_model_info_from_st(p, n, C.byref(out)))
        return _info_from_c(out)


def plan_chunk(n_tokens: list[int], token_chunk_size: int) -> list[int]:
    """Tokens of each slot that one `infer` call consumes (host-only mirror of the engine's chunk policy)."""
    n = len(n_tokens)
    a = (C.c_size_t * n)(*n_tokens)
    out = (C.c_int32 * n)()
    _check(lib().rwkv_plan_chunk(n, token_chunk_size, a, out))
    return list(out)


def list_adapters() -> list[str]:
    """`list_adapters` (lib.rs:339-349)."""
    l = lib()
    names = []
    for i in range(l.rwkv_device_count()):
        b = C.create_string_buffer(256)
        _check(l.rwkv_device_name(i, b, 256))
        names.append(b.value.decode())
    return names


@dataclass
class RnnInputBatch:             # RnnInputBatch::new(tokens, option) run.rs:1128
    tokens: list = field(default_factory=list)
    option: RnnOption = RnnOption.Last


class RnnInput:
    """RnnInput::new(batches, token_chunk_size) run.rs:1132.  `infer` consumes tokens from it."""

    def __init__(self, batches: list[RnnInputBatch], token_chunk_size: int | None = None):
        self.batches = batches
        self.token_chunk_size = token_chunk_size

    def num_token(self) -> int:   # run.rs:1136
        return sum(len(b.tokens) for b in self.batches)


class State:
    """`dyn State` (bundle.state(), lib.rs:494)."""

    def __init__(self, rt: "Runtime"):
        self._rt = rt

    @property
    def shape(self) -> tuple[int, int, int, int]:
        s = (C.c_size_t * 4)()
        lib().rwkv_state_shape(self._rt._h, s)
        return tuple(int(v) for v in s)

    def _np_shape(self):
        c, r, l, _ = self.shape
        return (l, r, c)

    def init(self) -> np.ndarray:                                  # run.rs:477, 950
        a = np.empty(self._np_shape(), np.float32)
        _check(lib().rwkv_state_init(self._rt._h, a.ctypes.data))
        return a

    def load(self, tensor: np.ndarray, batch: int) -> None:        # run.rs:1099
        a = np.ascontiguousarray(tensor, np.float32)
        if a.size != int(np.prod(self._np_shape())):
            raise RwkvError(-1, "state tensor has the wrong size")
        _check(lib().rwkv_state_load(self._rt._h, batch, a.ctypes.data))

    def back(self, batch: int) -> np.ndarray:                      # run.rs:1101
        a = np.empty(self._np_shape(), np.float32)
        _check(lib().rwkv_state_back(self._rt._h, batch, a.ctypes.data))
        return a

    def read(self, batch: int) -> "TensorGpu":                     # run.rs:1106
        h = C.c_void_p()
        _check(lib().rwkv_state_read(self._rt._h, batch, C.byref(h)))
        return TensorGpu(h)

    def write(self, tensor: "TensorGpu", batch: int) -> None:      # run.rs:1104
        _check(lib().rwkv_state_write(self._rt._h, batch, tensor._h))

    def embed(self, layer: int, batch: int) -> np.ndarray:
        """One layer's WKV rows [N, C] of a slot (docs/doc-api/openai.md:376-437 `/embeddings`)."""
        _, r, _, _ = self.shape
        c = self.shape[0]
        a = np.empty((r - 2, c), np.float32)
        _check(lib().rwkv_state_back_layer(self._rt._h, batch, layer, a.ctypes.data))
        return a

    def embed_async(self, layer: int, batch: int, dst: np.ndarray):
        """The same read-back, not waited for (rwkv_state_back_layer_async): `dst` is a [N, C] float32 view into pinned memory
        (`PinnedArena`), valid after `sync()`.  The slot may be re-used at once."""
        _check(lib().rwkv_state_back_layer_async(self._rt._h, batch, layer, dst.ctypes.data))

    def sync(self):
        _check(lib().rwkv_state_sync(self._rt._h))


class PinnedArena:
    """A float32 array in pinned host memory (rwkv_host_alloc): the destination of asynchronous read-backs (`State.embed_async`)
    and of direct logits copies.  `close()` (or garbage collection) frees it; views must not outlive it."""

    def __init__(self, shape):
        self.shape = tuple(int(x) for x in shape)
        n = int(np.prod(self.shape))
        self._ptr = C.c_void_p()
        _check(lib().rwkv_host_alloc(max(1, n) * 4, C.byref(self._ptr)))
        self.array = np.ctypeslib.as_array(C.cast(self._ptr, C.POINTER(C.c_float)), shape=(max(1, n),))[:n].reshape(self.shape)

    def close(self):
        if self._ptr:
            self.array = None
            lib().rwkv_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TensorGpu:
    """Device-resident state snapshot (`state.read` result); reusable for many `state.write`s."""

    def __init__(self, h):
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().rwkv_dstate_free(self._h)
            except Exception:
                pass
            self._h = None


class ModelBuilder:
    """`ModelBuilder::new(ctx, st).quant(map).lora(l).build_vN()` + `vN::Bundle::new(model, max_batch)` +
    `TokioRuntime::new(bundle)` (lib.rs:484-516) in one step."""

    def __init__(self, st_bytes, adapter: int = -1):
        self._st = st_bytes
        self._adapter = adapter
        self._quant_layers = 0
        self._quant_type = Quant.NONE
        self._lora: list[tuple[object, float]] = []

    def quant(self, layers: int, quant_type: Quant) -> "ModelBuilder":   # lib.rs:465: (0..quant) -> quant_type
        self._quant_layers, self._quant_type = int(layers), Quant(quant_type)
        return self

    def lora(self, st_bytes, alpha: float) -> "ModelBuilder":           # lib.rs:466-482
        self._lora.append((st_bytes, float(alpha)))
        return self

    def build(self, max_batch: int = 8, token_chunk_size: int = 128,
              precision: Precision = Precision.Fp16) -> "Runtime":
        return Runtime(self, max_batch, token_chunk_size, precision)


class Runtime:
    """`dyn Runtime<Rnn>` (lib.rs:112, 398)."""

    def __init__(self, b: ModelBuilder, max_batch: int, token_chunk_size: int, precision: Precision):
        l = lib()
        p, n, keep = _buf(b._st)
        loras = (_LoraC * max(1, len(b._lora)))()
        keeps = [keep]
        for i, (data, alpha) in enumerate(b._lora):
            lp, ln, lk = _buf(data)
            loras[i] = _LoraC(lp, ln, alpha)
            keeps.append(lk)
        d = _LoadDescC(b._adapter, b._quant_layers, int(b._quant_type), int(precision), max_batch, token_chunk_size,
                       p, n, loras if b._lora else None, len(b._lora))
        h = C.c_void_p()
        _check(l.rwkv_engine_create(C.byref(d), C.byref(h)))
        self._h = h
        self.max_batch = int(l.rwkv_engine_max_batch(h))
        self.token_chunk_size = int(l.rwkv_engine_token_chunk_size(h))    # what the engine was really built with
        ic = _ModelInfoC()
        _check(l.rwkv_engine_info(h, C.byref(ic)))
        self.info = _info_from_c(ic)
        self.state = State(self)
        V = self.info.num_vocab
        self._ins = (_SlotInC * max_batch)()
        self._outs = (_SlotOutC * max_batch)()
        # logits land in ONE pinned block (rwkv_host_alloc): the slots of a call get consecutive pieces of it in slot order, which
        # the engine copies device-to-host in one go; calls that need more rows than the block holds fall back to numpy buffers
        self._arena_rows = token_chunk_size + max_batch
        ap = C.c_void_p()
        _check(l.rwkv_host_alloc(self._arena_rows * V * 4, C.byref(ap)))
        self._arena_ptr = ap
        self._arena = np.ctypeslib.as_array(C.cast(ap, C.POINTER(C.c_float)), shape=(self._arena_rows, V))
        self._views = [None] * max_batch

    def close(self):
        if self._h:
            lib().rwkv_engine_destroy(self._h)
            self._h = None
            self._arena = None
            self._views = []
            lib().rwkv_host_free(self._arena_ptr)
            self._arena_ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device(self) -> int:
        return lib().rwkv_engine_device(self._h)

    @property
    def weight_bytes(self) -> int:
        return int(lib().rwkv_engine_weight_bytes(self._h))

    def save_prefab(self, path: str):
        """`ModelSerialize::serialize(file)` (lib.rs:131-154): the loaded (tiled / quantised / blended) model as one image;
        hand its bytes to `ModelBuilder` like a safetensors file (sniffed by content, lib.rs:585-588)."""
        _check(lib().rwkv_engine_save_prefab(self._h, os.fsencode(path)))

    def _prepare(self, inp: RnnInput):
        if len(inp.batches) != self.max_batch:
            raise RwkvError(-1, f"RnnInput must have max_batch={self.max_batch} entries")
        V = self.info.num_vocab
        keep = []
        needs = [0 if not len(ib.tokens) or ib.option == RnnOption.NoOutput else
                 (1 if ib.option == RnnOption.Last else min(len(ib.tokens), self.token_chunk_size)) for ib in inp.batches]
        pinned = sum(needs) <= self._arena_rows
        row = 0
        for b, ib in enumerate(inp.batches):
            toks = np.asarray(ib.tokens, dtype=np.uint32)
            keep.append(toks)
            if pinned:
                view = self._arena[row:row + needs[b]]
                row += needs[b]
            else:
                view = np.empty((max(1, needs[b]), V), np.float32)
            self._views[b] = view
            self._ins[b] = _SlotInC(toks.ctypes.data_as(C.POINTER(C.c_uint32)) if toks.size else None, toks.size,
                                    int(ib.option), 0)
            self._outs[b] = _SlotOutC(view.ctypes.data_as(C.POINTER(C.c_float)) if needs[b] else None, needs[b], 0, 0)
        return keep

    def _collect(self, inp: RnnInput):
        outs = []
        for b, ib in enumerate(inp.batches):
            o = self._outs[b]
            outs.append(self._views[b][:o.n_rows].copy())         # RnnOutputBatch: [n_rows, V]; empty if n_rows == 0
            # tokens handed over as a numpy array stay one (a view of the rest: the batch jobs keep their documents as arrays, so a step costs
            # no per-token Python work); lists stay lists
            ib.tokens = ib.tokens[o.n_consumed:] if isinstance(ib.tokens, np.ndarray) else list(ib.tokens[o.n_consumed:])
        return inp, outs

    def infer(self, inp: RnnInput):
        """`runtime.infer(input) -> (input, output)` (run.rs:1143): one step over <= token_chunk_size tokens."""
        keep = self._prepare(inp)
        _check(lib().rwkv_infer(self._h, self._ins, self._outs))
        del keep
        return self._collect(inp)

    def infer_sample(self, inp: RnnInput, samplers: list, uniforms: list):
        """On-device sampling front-end (rwkv_infer_sample).  `samplers[b]` is None or an object with `.top_k`,
        `.temperature`, `.adjustments() -> {token: delta_logit}` (penalties and bias merged) and either `.top_p` (nucleus) or
        `.kind = 1` + `.tau` (typical);
        `uniforms[b]` is the draw `fastrand::f32()` would make.  Returns (inp, [(token, prob) or None per slot])."""
        if len(inp.batches) != self.max_batch:
            raise RwkvError(-1, f"RnnInput must have max_batch={self.max_batch} entries")
        B = self.max_batch
        ins, sps = (_SlotInC * B)(), (_SampleC * B)()
        keep = []
        for b, ib in enumerate(inp.batches):
            toks = np.asarray(ib.tokens, dtype=np.uint32)
            keep.append(toks)
            ins[b] = _SlotInC(toks.ctypes.data_as(C.POINTER(C.c_uint32)) if toks.size else None, toks.size, 0, 0)
            s = samplers[b]
            if s is None:
                sps[b] = _SampleC(0.0, 1, 1.0, 0.0, None, None, 0, 0, 0.0, None)
                continue
            adj = s.adjustments()
            at = np.fromiter(adj.keys(), dtype=np.uint32, count=len(adj))
            av = np.fromiter(adj.values(), dtype=np.float32, count=len(adj))
            keep += [at, av]
            kind = int(getattr(s, "kind", 0))                       # 0 nucleus (top_p), 1 typical (tau)
            allow = getattr(s, "allow", None)                       # formatter mask: uint8 [num_vocab], 0 = forbidden (bnf.rs:35-38)
            if allow is not None:
                allow = np.ascontiguousarray(allow, dtype=np.uint8)
                if allow.size != self.info.num_vocab:
                    raise RwkvError(-1, "formatter mask must have num_vocab entries")
                keep.append(allow)
            sps[b] = _SampleC(getattr(s, "top_p", 0.0), s.top_k, s.temperature, uniforms[b],
                              at.ctypes.data_as(C.POINTER(C.c_uint32)) if at.size else None,
                              av.ctypes.data_as(C.POINTER(C.c_float)) if av.size else None, at.size,
                              kind, float(getattr(s, "tau", 0.0)),
                              allow.ctypes.data_as(C.POINTER(C.c_uint8)) if allow is not None else None)
        toks_o, probs_o = (C.c_uint32 * B)(), (C.c_float * B)()
        emitted, consumed = (C.c_uint8 * B)(), (C.c_size_t * B)()
        _check(lib().rwkv_infer_sample(self._h, ins, sps, toks_o, probs_o, emitted, consumed))
        del keep
        out = []
        for b, ib in enumerate(inp.batches):
            ib.tokens = list(ib.tokens[consumed[b]:])
            out.append((int(toks_o[b]), float(probs_o[b])) if emitted[b] else None)
        return inp, out

    # ---- measurement loops (bench.py): the same ABI calls with every per-step Python object hoisted out, so that the rate is
    # the library's, not the interpreter's
    def serve_loop_logits(self, first_tokens, n_steps: int) -> float:
        """`n_steps` single-token steps of `len(first_tokens)` slots through rwkv_infer with the logits of every slot copied to
        the (pinned) host block each step, as run.rs:809-832 receives them; returns seconds.  The token fed is the slot's first
        token every step: the arg-max over 65 536 floats per slot on the host is the sampler's cost, not the transport's."""
        B = len(first_tokens)
        toks = np.asarray(first_tokens, dtype=np.uint32).copy()
        for b in range(self.max_batch):
            self._ins[b] = _SlotInC(C.cast(toks.ctypes.data + 4 * b, C.POINTER(C.c_uint32)) if b < B else None, 1 if b < B else 0, 0, 0)
            self._outs[b] = _SlotOutC(self._arena[b:b + 1].ctypes.data_as(C.POINTER(C.c_float)) if b < B else None, 1 if b < B else 0, 0, 0)
        l, h, ins, outs = lib(), self._h, self._ins, self._outs
        _check(l.rwkv_infer(h, ins, outs))
        t = time.perf_counter()
        for _ in range(n_steps):
            rc = l.rwkv_infer(h, ins, outs)
            if rc:
                _check(rc)
        return time.perf_counter() - t

    def serve_loop_sample(self, first_tokens, n_steps: int, top_p=0.5, top_k=128, temperature=1.0, seed=0) -> float:
        """The same loop through rwkv_infer_sample (nucleus defaults of the reference, sampler/nucleus.rs:25-35): 8 bytes per slot
        come back instead of 256 KiB, and the token the device picked is fed to the next step.  Returns seconds."""
        B = len(first_tokens)
        toks = np.asarray(first_tokens, dtype=np.uint32).copy()
        u = np.random.default_rng(seed).random((n_steps + 1, self.max_batch))
        ins, sps = (_SlotInC * self.max_batch)(), (_SampleC * self.max_batch)()
        for b in range(self.max_batch):
            ins[b] = _SlotInC(C.cast(toks.ctypes.data + 4 * b, C.POINTER(C.c_uint32)) if b < B else None, 1 if b < B else 0, 0, 0)
            sps[b] = _SampleC(top_p, top_k, temperature, 0.0, None, None, 0, 0, 0.0, None)
        toks_o, probs_o = (C.c_uint32 * self.max_batch)(), (C.c_float * self.max_batch)()
        emitted, consumed = (C.c_uint8 * self.max_batch)(), (C.c_size_t * self.max_batch)()
        out_view = np.ctypeslib.as_array(toks_o)
        l, h = lib(), self._h
        _check(l.rwkv_infer_sample(h, ins, sps, toks_o, probs_o, emitted, consumed))
        t = time.perf_counter()
        sp_view = np.frombuffer(sps, dtype=np.dtype({"names": ["uniform"], "formats": [np.float32],
                                                     "offsets": [_SampleC.uniform.offset], "itemsize": C.sizeof(_SampleC)}))
        u32 = u.astype(np.float32)
        for s_ in range(n_steps):
            sp_view["uniform"][:B] = u32[s_, :B]                 # one strided store instead of B attribute writes
            rc = l.rwkv_infer_sample(h, ins, sps, toks_o, probs_o, emitted, consumed)
            if rc:
                _check(rc)
            toks[:B] = out_view[:B]
        return time.perf_counter() - t

    def profile_infer(self, inp: RnnInput):
        keep = self._prepare(inp)
        ms = (C.c_float * PROFILE_FAMILIES)()
        n = (C.c_int32 * PROFILE_FAMILIES)()
        _check(lib().rwkv_profile_infer(self._h, self._ins, self._outs, ms, n))
        del keep
        inp, outs = self._collect(inp)
        fam = {lib().rwkv_profile_family_name(i).decode(): (float(ms[i]), int(n[i])) for i in range(PROFILE_FAMILIES)
               if lib().rwkv_profile_family_name(i)}
        return inp, outs, fam

    def decode_greedy(self, first_tokens, n_steps: int):
        """Device-resident greedy decode of `len(first_tokens)` slots; returns (tokens [n_steps, n_slots], ms)."""
        ft = np.asarray(first_tokens, dtype=np.uint32)
        out = np.empty((n_steps, ft.size), np.uint32)
        ms = C.c_float()
        _check(lib().rwkv_decode_greedy(self._h, ft.size, ft.ctypes.data_as(C.POINTER(C.c_uint32)), n_steps,
                                        out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(ms)))
        return out, float(ms.value)

    def read_state(self, st_bytes) -> np.ndarray:
        """`vN::read_state(context, info, model)` (lib.rs:378-389)."""
        p, n, keep = _buf(st_bytes)
        a = np.empty(self.state._np_shape(), np.float32)
        _check(lib().rwkv_read_init_state(self._h, p, n, a.ctypes.data))
        return a


def bench_gemm(rows: int, K: int, fmt: int, T: int, hilo: bool = False, spb: int = 0, nmat: int = 16,
               iters: int = 200) -> tuple[float, float]:
    """Kernel microbench hook: (microseconds per launch inside a captured graph, blocks per launch)."""
    us, lds = C.c_float(), C.c_float()
    _check(lib().rwkv_bench_gemm(rows, K, fmt, T, int(hilo), spb, nmat, iters, C.byref(us), C.byref(lds)))
    return float(us.value), float(lds.value)


def softmax(rt: Runtime, tensors: list[np.ndarray]) -> list[np.ndarray]:
    """`web_rwkv::runtime::softmax::softmax(&context, Vec<TensorCpu>)` (run.rs:1179)."""
    if not tensors:
        return []
    ins = [np.ascontiguousarray(t, np.float32).reshape(-1) for t in tensors]
    outs = [np.empty_like(t) for t in ins]
    pi = (C.c_void_p * len(ins))(*[t.ctypes.data for t in ins])
    po = (C.c_void_p * len(ins))(*[t.ctypes.data for t in outs])
    _check(lib().rwkv_softmax(rt._h, pi, po, len(ins)))
    return outs


class Tokenizer:
    """`Tokenizer::new(&contents)` (lib.rs:375)."""

    def __init__(self, vocab_json: str | bytes):
        data = vocab_json.encode() if isinstance(vocab_json, str) else bytes(vocab_json)
        h = C.c_void_p()
        _check(lib().rwkv_tokenizer_create(data, len(data), C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().rwkv_tokenizer_destroy(self._h)
            except Exception:
                pass

    def encode(self, text: bytes) -> list[int]:                     # run.rs:157
        n = lib().rwkv_tokenizer_encode(self._h, text, len(text), None, 0)
        if n < 0:
            raise RwkvError(int(n), "no matching token found")
        buf = (C.c_uint32 * max(1, n))()
        lib().rwkv_tokenizer_encode(self._h, text, len(text), buf, n)
        return list(buf[:n])

    def decode(self, tokens) -> bytes:                              # run.rs:856
        t = np.asarray(tokens, dtype=np.uint32)
        p = t.ctypes.data_as(C.POINTER(C.c_uint32))
        n = lib().rwkv_tokenizer_decode(self._h, p, t.size, None, 0)
        if n < 0:
            raise RwkvError(int(n), "token index out of range")
        buf = C.create_string_buffer(max(1, n))
        lib().rwkv_tokenizer_decode(self._h, p, t.size, buf, n)
        return buf.raw[:n]

    def token_index_to_bytes(self) -> list[bytes]:                  # sampler/bnf.rs:15
        out = []
        for i in range(lib().rwkv_tokenizer_vocab_size(self._h)):
            n = lib().rwkv_tokenizer_token_bytes(self._h, i, None, 0)
            if n < 0:
                out.append(b"")
                continue
            buf = C.create_string_buffer(max(1, n))
            lib().rwkv_tokenizer_token_bytes(self._h, i, buf, n)
            out.append(buf.raw[:n])
        return out
