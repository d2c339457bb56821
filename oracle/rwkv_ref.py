"""CPU oracle for the RWKV V5.2 / V6 / V7 forward pass  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it.  The product path
(`ai00_server_amd/`, `librwkv_hip.so`) never imports, links or executes anything
under `oracle/`.

PARITY UNPINNED.  The reference (`/root/reference`, ai00_server @ 2025-10-24) holds
no tests, no golden vectors and no implementation of this path: the arithmetic lives
in the un-vendored crate `web-rwkv 0.10.18` (Cargo.lock:5530-5533) which needs
rustc + a wgpu/Vulkan adapter, neither of which exists here.  This oracle therefore
restates the *published* RWKV-5.2 / RWKV-6 / RWKV-7 inference formulas (BlinkDL,
`rwkv` pip package `model.py`, `rwkv_v7_demo.py`) and anchors the data contract on
the reference's own call sites:

  * on-disk tensor names / transposes / fp16:  assets/scripts/convert_safetensors.py:22-101
    (crates/converter/src/main.rs:8-22)
  * version sniffing:                          convert_safetensors.py:36-47
  * state slab shape [C, N+2, L, 1] + usage:   crates/ai00-core/src/run.rs:984-989, 1099-1106
  * output selection Last / Full:              run.rs:716-747, 809-832
  * quant = first `quant` layers, one type:    crates/ai00-core/src/lib.rs:465
  * perplexity softmax (exp/sum, no max-sub):  run.rs:735-741
  * empty prompt => [0], token 0 = stop:       run.rs:489-492, 855

All arithmetic is fp32 on weights rounded through fp16 exactly as the `.st` file
stores them (convert_safetensors.py:64 `.half()`).
"""
from __future__ import annotations

import json
import re
import struct
from dataclasses import dataclass

import numpy as np

HEAD_SIZE = 64
LN_EPS = 1e-5
GN_EPS = 64e-5  # BlinkDL: head_size_divisor 8 -> eps = 1e-5 * 8**2

QUANT_NONE, QUANT_INT8, QUANT_NF4 = 0, 1, 2
INT8_BLOCK = 128
NF4_BLOCK = 64

# QLoRA NormalFloat-4 quantiles (Dettmers et al. 2023, Appendix E).
NF4_TABLE = np.array(
    [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
     -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
     0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
     0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0], dtype=np.float32)
# thresholds between neighbouring code points (fp32); idx = #(x_norm > mid)
NF4_MID = ((NF4_TABLE[1:] + NF4_TABLE[:-1]) * np.float32(0.5)).astype(np.float32)
NF4_TABLE_F16 = NF4_TABLE.astype(np.float16)


def big_empty(shape, dtype) -> np.ndarray:
    """np.empty backed by an anonymous mmap with MADV_HUGEPAGE for large arrays: first-touch page
    faults are the dominant cost of building multi-GB checkpoints inside sandboxed containers."""
    import mmap
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if n < (64 << 20):
        return np.empty(shape, dtype)
    m = mmap.mmap(-1, n, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    try:
        m.madvise(14)  # MADV_HUGEPAGE
    except (OSError, ValueError, AttributeError):
        pass
    return np.frombuffer(m, dtype=dtype).reshape(shape)


# --------------------------------------------------------------------------------------
# safetensors I/O (plain, so the oracle does not depend on the product's parser)
# --------------------------------------------------------------------------------------
def st_serialize(tensors: dict[str, np.ndarray], metadata: dict | None = None) -> bytes:
    """Serialise to the safetensors layout `convert_safetensors.py:80` writes."""
    header = {}
    if metadata:
        header["__metadata__"] = metadata
    off = 0
    blobs = []
    for name, arr in tensors.items():
        arr = np.ascontiguousarray(arr)
        dt = {np.dtype(np.float16): "F16", np.dtype(np.float32): "F32"}[arr.dtype]
        n = arr.nbytes
        header[name] = {"dtype": dt, "shape": list(arr.shape), "data_offsets": [off, off + n]}
        blobs.append(arr)
        off += n
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    out = bytearray(8 + len(hj) + off)
    out[:8] = struct.pack("<Q", len(hj))
    out[8:8 + len(hj)] = hj
    p = 8 + len(hj)
    for arr in blobs:
        out[p:p + arr.nbytes] = arr.tobytes() if arr.nbytes < (1 << 20) else memoryview(arr).cast("B")
        p += arr.nbytes
    return bytes(out)


def st_deserialize(data: bytes) -> dict[str, np.ndarray]:
    (hl,) = struct.unpack("<Q", data[:8])
    header = json.loads(bytes(data[8:8 + hl]))
    out = {}
    base = 8 + hl
    for name, d in header.items():
        if name == "__metadata__":
            continue
        dt = {"F16": np.float16, "F32": np.float32}[d["dtype"]]
        a, b = d["data_offsets"]
        out[name] = np.frombuffer(data, dtype=dt, count=(b - a) // np.dtype(dt).itemsize,
                                  offset=base + a).reshape(d["shape"])
    return out


# --------------------------------------------------------------------------------------
# model info (mirrors web-rwkv `Loader::info`, lib.rs:587; fields per SURVEY 8(b))
# --------------------------------------------------------------------------------------
@dataclass
class ModelInfo:
    version: int        # 5, 6, 7
    num_layer: int
    num_emb: int
    num_hidden: int
    num_vocab: int
    num_head: int

    @property
    def head_size(self) -> int:
        return self.num_emb // self.num_head


def model_info(t: dict[str, np.ndarray]) -> ModelInfo:
    """Version sniffing follows convert_safetensors.py:36-47 on the *converted* names."""
    if "blocks.0.att.x_r" in t:
        version = 7
    elif "blocks.0.att.time_mix_x" in t:
        version = 6
    elif "blocks.0.att.ln_x.weight" in t and "blocks.0.att.gate.weight" in t:
        td = t["blocks.0.att.time_decay"]
        if td.ndim < 2 or td.shape[-1] <= 1:
            raise ValueError("RWKV v5.0/v5.1 checkpoints are not supported (need v5.2)")
        version = 5
    else:
        raise ValueError("unsupported model version (v4 or unknown)")
    L = 0
    while f"blocks.{L}.ln1.weight" in t:
        L += 1
    V, C = t["emb.weight"].shape
    F = t["blocks.0.ffn.key.weight"].shape[0]
    H = t["blocks.0.att.r_k"].shape[0] if version == 7 else t["blocks.0.att.time_first"].shape[0]
    return ModelInfo(version, L, C, F, V, H)


# --------------------------------------------------------------------------------------
# quantisation reference (build-defined block formats)
# --------------------------------------------------------------------------------------
def quant_int8(w16: np.ndarray):
    """Per 128-element block along the input dim: q = rint((x-b)/a), a=fp16((max-min)/255), b=min.
    Returns (q u8 [out,in], a f16 [out,in/128], b f16 [out,in/128])."""
    out_dim, in_dim = w16.shape
    assert in_dim % INT8_BLOCK == 0
    x = w16.astype(np.float32).reshape(out_dim, in_dim // INT8_BLOCK, INT8_BLOCK)
    mn = x.min(axis=2)
    mx = x.max(axis=2)
    a = ((mx - mn) / np.float32(255.0)).astype(np.float16)
    b = mn.astype(np.float16)
    a32 = a.astype(np.float32)
    safe = np.where(a32 > 0, a32, np.float32(1.0))
    q = np.rint((x - b.astype(np.float32)[..., None]) / safe[..., None])
    q = np.clip(q, 0, 255).astype(np.uint8)
    return q.reshape(out_dim, in_dim), a, b


def dequant_int8(q: np.ndarray, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """x^ = fp16(a*q + b) with ONE rounding (what v_pk_fma_f16 does); exact in float64."""
    out_dim, in_dim = q.shape
    qq = q.reshape(out_dim, in_dim // INT8_BLOCK, INT8_BLOCK).astype(np.float64)
    v = a.astype(np.float64)[..., None] * qq + b.astype(np.float64)[..., None]
    return v.astype(np.float16).reshape(out_dim, in_dim)


def quant_nf4(w16: np.ndarray):
    """Per 64-element block: absmax fp16, idx = #(x/absmax > mid_i). Returns (idx u8, absmax f16)."""
    out_dim, in_dim = w16.shape
    assert in_dim % NF4_BLOCK == 0
    x = w16.astype(np.float32).reshape(out_dim, in_dim // NF4_BLOCK, NF4_BLOCK)
    am = np.abs(x).max(axis=2).astype(np.float16)
    am32 = am.astype(np.float32)
    safe = np.where(am32 > 0, am32, np.float32(1.0))
    xn = x / safe[..., None]
    idx = (xn[..., None] > NF4_MID).sum(axis=-1).astype(np.uint8)
    return idx.reshape(out_dim, in_dim), am


def dequant_nf4(idx: np.ndarray, am: np.ndarray) -> np.ndarray:
    """x^ = fp16(absmax * fp16(Q[idx])) with one rounding (v_pk_mul_f16)."""
    out_dim, in_dim = idx.shape
    t = NF4_TABLE_F16.astype(np.float64)[idx.reshape(out_dim, in_dim // NF4_BLOCK, NF4_BLOCK)]
    v = am.astype(np.float64)[..., None] * t
    return v.astype(np.float16).reshape(out_dim, in_dim)


def fake_quant(w16: np.ndarray, quant_type: int) -> np.ndarray:
    if quant_type == QUANT_INT8:
        return dequant_int8(*quant_int8(w16))
    if quant_type == QUANT_NF4:
        return dequant_nf4(*quant_nf4(w16))
    return w16


# Matrices that `ModelBuilder::quant` quantises in layers 0..quant (lib.rs:465):
# only the big projection matrices; embedding, head, LoRA mats and vectors stay fp16.
def quantised_matrix_names(version: int) -> list[str]:
    att = ["att.receptance.weight", "att.key.weight", "att.value.weight", "att.output.weight"]
    if version in (5, 6):
        att.append("att.gate.weight")
    ffn = ["ffn.key.weight", "ffn.value.weight"]
    if version in (5, 6):
        ffn.append("ffn.receptance.weight")
    return att + ffn


# --------------------------------------------------------------------------------------
# the forward pass
# --------------------------------------------------------------------------------------
def _ln(x, w, b, eps=LN_EPS):
    x = x.astype(np.float32)
    m = x.mean(dtype=np.float32)
    v = ((x - m) ** 2).mean(dtype=np.float32)
    return (x - m) / np.sqrt(v + np.float32(eps)) * w + b


def _gn(x, H, w, b, eps=GN_EPS):
    x = x.reshape(H, -1).astype(np.float32)
    m = x.mean(axis=1, keepdims=True, dtype=np.float32)
    v = ((x - m) ** 2).mean(axis=1, keepdims=True, dtype=np.float32)
    return ((x - m) / np.sqrt(v + np.float32(eps))).reshape(-1) * w + b


def _sigmoid(x):
    return np.float32(1.0) / (np.float32(1.0) + np.exp(-x))


class RwkvRef:
    """fp32 reference of one RWKV model; one recurrent state slab per call (slots are independent,
    run.rs:1121-1130, so a batch is just this applied per slot)."""

    def __init__(self, tensors: dict[str, np.ndarray], quant_layers: int = 0,
                 quant_type: int = QUANT_NONE, lora: list | None = None):
        """`lora`: [(lora_tensors, alpha), ...] blended at load like `ModelBuilder::lora(Lora { data, blend: LoraBlend::full(alpha) })`
        (lib.rs:466-482).  As published in web-rwkv's loader (runtime/loader.rs, 0.10.x; not vendored in the reference tree —
        restated, UNPINNED): `full(alpha)` is the one pattern `blocks\.([0-9]+)\.([0-9a-zA-Z\.\_]+)` — per-block tensors only; a matrix
        `X.weight` with `X.lora.0` [in, r] / `X.lora.1` [out, r] in the file gets W += alpha * B A^T (factor [alpha, 1]); any other
        tensor the file holds under the model's own name is blended whole, v = alpha * l + (1 - alpha) * v (factor [alpha, 1 - alpha]:
        alpha = 1 replaces a fine-tuned vector).  Matrices are blended on the fp16 values in fp32 and rounded back once (then
        quantised, if their layer is); vectors stay fp32."""
        self.info = model_info(tensors)
        self.quant_layers = quant_layers
        self.quant_type = quant_type
        qn = set()
        if quant_type != QUANT_NONE:
            for l in range(min(quant_layers, self.info.num_layer)):
                for n in quantised_matrix_names(self.info.version):
                    qn.add(f"blocks.{l}.{n}")
        self.w: dict[str, np.ndarray] = {}
        for k, v in tensors.items():
            v16 = np.asarray(v, dtype=np.float16)
            vec32 = None
            in_scope = re.fullmatch(r"blocks\.[0-9]+\..+", k) is not None
            for lt, alpha in ((lora or []) if in_scope else []):
                stem = k[:-len(".weight")] if k.endswith(".weight") else k
                if stem + ".lora.0" in lt and stem + ".lora.1" in lt and v16.ndim == 2:
                    A = np.asarray(lt[stem + ".lora.0"], np.float16).astype(np.float32)
                    B = np.asarray(lt[stem + ".lora.1"], np.float16).astype(np.float32)
                    v16 = (v16.astype(np.float32) + np.float32(alpha) * (B @ A.T)).astype(np.float16)
                elif k in lt:
                    base = v16.astype(np.float32) if vec32 is None else vec32
                    vec32 = np.float32(alpha) * np.asarray(lt[k], np.float16).astype(np.float32).reshape(base.shape) + (np.float32(1.0) - np.float32(alpha)) * base
            if k in qn:
                v16 = fake_quant(v16, quant_type)
            dst = big_empty(v16.shape, np.float32)
            dst[...] = v16 if vec32 is None else vec32
            self.w[k] = dst

    # ---- state slab: [L][N+2][C] fp32 == shape [C, N+2, L, 1] fastest-dim-first (run.rs:987)
    def state_shape(self):
        i = self.info
        return (i.num_emb, i.head_size + 2, i.num_layer, 1)

    def init_state(self) -> np.ndarray:
        i = self.info
        return np.zeros((i.num_layer, i.head_size + 2, i.num_emb), dtype=np.float32)

    def read_init_state(self, st: dict[str, np.ndarray]) -> np.ndarray:
        """`vN::read_state` (lib.rs:385-387): WKV rows from `blocks.i.att.time_state`
        (stored with the last two dims transposed, convert_safetensors.py:100-101)."""
        i = self.info
        s = self.init_state()
        for l in range(i.num_layer):
            ts = np.asarray(st[f"blocks.{l}.att.time_state"], dtype=np.float16).astype(np.float32)
            ts = ts.transpose(0, 2, 1)  # undo the converter -> [H, i, j]
            # slab[l][1+i][h*N+j] = S_h[i][j]
            s[l, 1:1 + i.head_size, :] = ts.transpose(1, 0, 2).reshape(i.head_size, i.num_emb)
        return s

    # ---- one token ------------------------------------------------------------------
    def _token(self, token: int, state: np.ndarray, want_logits: bool):
        i, w = self.info, self.w
        H, N, C = i.num_head, i.head_size, i.num_emb
        x = _ln(w["emb.weight"][token], w["blocks.0.ln0.weight"], w["blocks.0.ln0.bias"])
        v_first = None
        for l in range(i.num_layer):
            p = f"blocks.{l}."
            xx = _ln(x, w[p + "ln1.weight"], w[p + "ln1.bias"])
            sx = state[l, 0].copy()
            S = state[l, 1:1 + N].reshape(N, H, N).transpose(1, 0, 2).copy()  # [H, i, j]
            state[l, 0] = xx
            if i.version == 5:
                att, S = self._att5(p, xx, sx, S)
            elif i.version == 6:
                att, S = self._att6(p, xx, sx, S)
            else:
                att, S, v_first = self._att7(p, l, xx, sx, S, v_first)
            state[l, 1:1 + N] = S.transpose(1, 0, 2).reshape(N, C)
            x = x + att
            xx = _ln(x, w[p + "ln2.weight"], w[p + "ln2.bias"])
            sx = state[l, N + 1].copy()
            state[l, N + 1] = xx
            x = x + (self._ffn7(p, xx, sx) if i.version == 7 else self._ffn56(p, xx, sx))
        if not want_logits:
            return None
        xo = _ln(x, w["ln_out.weight"], w["ln_out.bias"])
        return w["head.weight"] @ xo

    def _wkv56(self, p, r, k, v, wdec, u, S):
        H, N = self.info.num_head, self.info.head_size
        r, k, v = r.reshape(H, N), k.reshape(H, N), v.reshape(H, N)
        wdec, u = wdec.reshape(H, N), u.reshape(H, N)
        a = k[:, :, None] * v[:, None, :]                       # a[h,i,j] = k_i v_j
        out = np.einsum("hi,hij->hj", r, u[:, :, None] * a + S)  # out_j = sum_i r_i (u_i a_ij + S_ij)
        S = a + wdec[:, :, None] * S
        return out.reshape(-1).astype(np.float32), S.astype(np.float32)

    def _att5(self, p, xx, sx, S):
        w = self.w
        mix = lambda n: xx * w[p + f"att.time_mix_{n}"].reshape(-1) + sx * (1 - w[p + f"att.time_mix_{n}"].reshape(-1))
        r = w[p + "att.receptance.weight"] @ mix("r")
        k = w[p + "att.key.weight"] @ mix("k")
        v = w[p + "att.value.weight"] @ mix("v")
        g = w[p + "att.gate.weight"] @ mix("g")
        g = g * _sigmoid(g)
        wdec = np.exp(-np.exp(w[p + "att.time_decay"].reshape(-1)))
        out, S = self._wkv56(p, r, k, v, wdec, w[p + "att.time_first"].reshape(-1), S)
        y = _gn(out, self.info.num_head, w[p + "att.ln_x.weight"], w[p + "att.ln_x.bias"]) * g
        return w[p + "att.output.weight"] @ y, S

    def _att6(self, p, xx, sx, S):
        w = self.w
        C = self.info.num_emb
        dx = sx - xx
        z = xx + dx * w[p + "att.time_mix_x"].reshape(-1)
        m = np.tanh(w[p + "att.time_mix_w1"] @ z)               # [5*Dm]  (stored [5*Dm, C])
        w2 = w[p + "att.time_mix_w2"]                            # [5, C, Dm]
        Dm = w2.shape[2]
        m = m.reshape(5, Dm)
        xs = {}
        for c, n in enumerate("wkvrg"):                          # BlinkDL order: mw, mk, mv, mr, mg
            mc = w2[c] @ m[c]
            xs[n] = xx + dx * (w[p + f"att.time_mix_{n}"].reshape(-1) + mc)
        r = w[p + "att.receptance.weight"] @ xs["r"]
        k = w[p + "att.key.weight"] @ xs["k"]
        v = w[p + "att.value.weight"] @ xs["v"]
        g = w[p + "att.gate.weight"] @ xs["g"]
        g = g * _sigmoid(g)
        td = np.tanh(w[p + "att.time_decay_w1"] @ xs["w"])       # [Dd]  (stored [Dd, C])
        d = w[p + "att.time_decay"].reshape(-1) + w[p + "att.time_decay_w2"] @ td  # w2 stored [C, Dd]
        wdec = np.exp(-np.exp(d.astype(np.float32)))
        out, S = self._wkv56(p, r, k, v, wdec, w[p + "att.time_first"].reshape(-1), S)
        y = _gn(out, self.info.num_head, w[p + "att.ln_x.weight"], w[p + "att.ln_x.bias"]) * g
        return w[p + "att.output.weight"] @ y, S

    def _ffn56(self, p, xx, sx):
        w = self.w
        if self.info.version == 5:
            mk, mr = w[p + "ffn.time_mix_k"].reshape(-1), w[p + "ffn.time_mix_r"].reshape(-1)
            xk = xx * mk + sx * (1 - mk)
            xr = xx * mr + sx * (1 - mr)
        else:
            dx = sx - xx
            xk = xx + dx * w[p + "ffn.time_mix_k"].reshape(-1)
            xr = xx + dx * w[p + "ffn.time_mix_r"].reshape(-1)
        r = _sigmoid(w[p + "ffn.receptance.weight"] @ xr)
        k = np.maximum(w[p + "ffn.key.weight"] @ xk, 0) ** 2
        return r * (w[p + "ffn.value.weight"] @ k)

    def _att7(self, p, l, xx, sx, S, v_first):
        w = self.w
        H, N = self.info.num_head, self.info.head_size
        dx = sx - xx
        xm = {n: xx + dx * w[p + f"att.x_{n}"].reshape(-1) for n in "rwkvag"}
        r = w[p + "att.receptance.weight"] @ xm["r"]
        k = w[p + "att.key.weight"] @ xm["k"]
        v = w[p + "att.value.weight"] @ xm["v"]
        wd = w[p + "att.w2"] @ np.tanh(w[p + "att.w1"] @ xm["w"])
        a = _sigmoid(w[p + "att.a0"].reshape(-1) + w[p + "att.a2"] @ (w[p + "att.a1"] @ xm["a"]))
        g = w[p + "att.g2"] @ _sigmoid(w[p + "att.g1"] @ xm["g"])
        kk = (k * w[p + "att.k_k"].reshape(-1)).reshape(H, N)
        kk = kk / np.maximum(np.sqrt((kk * kk).sum(axis=1, keepdims=True)), np.float32(1e-12))
        kk = kk.reshape(-1)
        k = k * (1 + (a - 1) * w[p + "att.k_a"].reshape(-1))
        if l == 0:
            v_first = v
        else:
            v = v + (v_first - v) * _sigmoid(w[p + "att.v0"].reshape(-1) + w[p + "att.v2"] @ (w[p + "att.v1"] @ xm["v"]))
        wdec = np.exp(np.float32(-0.606531) * _sigmoid((w[p + "att.w0"].reshape(-1) + wd).astype(np.float32)))
        # S[h, i(value), j(key)]
        rh, kh, vh, kkh, ah, wh = (t.reshape(H, N) for t in (r, k, v, kk, a, wdec))
        sa = np.einsum("hij,hj->hi", S, -kkh)
        S = S * wh[:, None, :] + sa[:, :, None] * (kkh * ah)[:, None, :] + vh[:, :, None] * kh[:, None, :]
        S = S.astype(np.float32)
        out = np.einsum("hij,hj->hi", S, rh).reshape(-1)
        y = _gn(out, H, w[p + "att.ln_x.weight"], w[p + "att.ln_x.bias"])
        bonus = (rh * kh * w[p + "att.r_k"].reshape(H, N)).sum(axis=1, keepdims=True) * vh
        y = y + bonus.reshape(-1)
        return w[p + "att.output.weight"] @ (y * g), S, v_first

    def _ffn7(self, p, xx, sx):
        w = self.w
        xk = xx + (sx - xx) * w[p + "ffn.x_k"].reshape(-1)
        k = np.maximum(w[p + "ffn.key.weight"] @ xk, 0) ** 2
        return w[p + "ffn.value.weight"] @ k

    # ---- public: mirrors `runtime.infer` for ONE slot ---------------------------------
    def forward(self, tokens, state: np.ndarray, full: bool = False):
        """Consume `tokens` sequentially, mutate `state` in place.
        Returns logits [n_out, V]: Last -> 1 row (last token), Full -> one row per token
        (RnnOption::{Last,Full}, run.rs:716-747)."""
        rows = []
        n = len(tokens)
        for t, tok in enumerate(tokens):
            want = full or t == n - 1
            lg = self._token(int(tok), state, want)
            if want:
                rows.append(lg.astype(np.float32))
        return np.stack(rows) if rows else np.zeros((0, self.info.num_vocab), np.float32)

    def greedy(self, prompt, n_new: int, state: np.ndarray | None = None):
        """Greedy decode (Nucleus top_k=1 picks arg-max: sampler/nucleus.rs:77-89)."""
        state = self.init_state() if state is None else state
        toks = list(prompt) if len(prompt) else [0]          # run.rs:489-492
        out = []
        lg = self.forward(toks, state)[-1]
        for _ in range(n_new):
            t = int(np.argmax(lg))
            out.append(t)
            lg = self.forward([t], state)[-1]
        return out, state


def _ln_rows(x, w, b, eps=LN_EPS):
    x = x.astype(np.float32)
    m = x.mean(axis=-1, keepdims=True, dtype=np.float32)
    v = ((x - m) ** 2).mean(axis=-1, keepdims=True, dtype=np.float32)
    return (x - m) / np.sqrt(v + np.float32(eps)) * w + b


class RwkvRefBatch(RwkvRef):
    """Lock-step form of RwkvRef for multi-slot tests: B independent slots advance ONE token each per `step`
    (the batch `runtime.infer` is handed at run.rs:1121-1132, one RnnInputBatch per slot).  Formula for formula the
    same restatement as `RwkvRef._token` / `_att5` / `_att6` / `_att7` / `_ffn*` with a leading slot axis: every
    `W @ x` becomes `X @ W.T`, so the weights are streamed once per step instead of once per slot.  BLAS may sum a
    matrix-matrix product in a different order than a matrix-vector product, so the two forms agree to fp32 round-off
    (tests/test_oracle.py pins that), not bit for bit.  `states` is [B, L, N+2, C], mutated in place."""

    def step(self, tokens, states: np.ndarray, want_logits: bool = True):
        i, w = self.info, self.w
        H, N, C, B = i.num_head, i.head_size, i.num_emb, len(tokens)
        x = _ln_rows(w["emb.weight"][np.asarray(tokens, dtype=np.int64)], w["blocks.0.ln0.weight"], w["blocks.0.ln0.bias"])
        v_first = None
        for l in range(i.num_layer):
            p = f"blocks.{l}."
            xx = _ln_rows(x, w[p + "ln1.weight"], w[p + "ln1.bias"])
            sx = states[:, l, 0].copy()
            S = states[:, l, 1:1 + N].reshape(B, N, H, N).transpose(0, 2, 1, 3).copy()      # [B, H, i, j]
            states[:, l, 0] = xx
            if i.version == 5:
                att, S = self._batt5(p, xx, sx, S)
            elif i.version == 6:
                att, S = self._batt6(p, xx, sx, S)
            else:
                att, S, v_first = self._batt7(p, l, xx, sx, S, v_first)
            states[:, l, 1:1 + N] = S.transpose(0, 2, 1, 3).reshape(B, N, C)
            x = x + att
            xx = _ln_rows(x, w[p + "ln2.weight"], w[p + "ln2.bias"])
            sx = states[:, l, N + 1].copy()
            states[:, l, N + 1] = xx
            x = x + (self._bffn7(p, xx, sx) if i.version == 7 else self._bffn56(p, xx, sx))
        if not want_logits:
            return None
        xo = _ln_rows(x, w["ln_out.weight"], w["ln_out.bias"])
        return (xo @ w["head.weight"].T).astype(np.float32)

    def _bgn(self, x, wt, b):                                    # GroupNorm over each head, x [B, C]
        B = x.shape[0]
        H = self.info.num_head
        x = x.reshape(B, H, -1).astype(np.float32)
        m = x.mean(axis=2, keepdims=True, dtype=np.float32)
        v = ((x - m) ** 2).mean(axis=2, keepdims=True, dtype=np.float32)
        return ((x - m) / np.sqrt(v + np.float32(GN_EPS))).reshape(B, -1) * wt + b

    def _bwkv56(self, r, k, v, wdec, u, S):
        B = r.shape[0]
        H, N = self.info.num_head, self.info.head_size
        r, k, v, wdec = (t.reshape(B, H, N) for t in (r, k, v, wdec))
        u = u.reshape(1, H, N)
        a = k[:, :, :, None] * v[:, :, None, :]
        out = np.einsum("bhi,bhij->bhj", r, u[:, :, :, None] * a + S)
        S = a + wdec[:, :, :, None] * S
        return out.reshape(B, -1).astype(np.float32), S.astype(np.float32)

    def _batt5(self, p, xx, sx, S):
        w = self.w
        mix = lambda n: xx * w[p + f"att.time_mix_{n}"].reshape(-1) + sx * (1 - w[p + f"att.time_mix_{n}"].reshape(-1))
        r = mix("r") @ w[p + "att.receptance.weight"].T
        k = mix("k") @ w[p + "att.key.weight"].T
        v = mix("v") @ w[p + "att.value.weight"].T
        g = mix("g") @ w[p + "att.gate.weight"].T
        g = g * _sigmoid(g)
        wdec = np.exp(-np.exp(w[p + "att.time_decay"].reshape(-1)))
        wdec = np.broadcast_to(wdec, r.shape)
        out, S = self._bwkv56(r, k, v, wdec, w[p + "att.time_first"].reshape(-1), S)
        y = self._bgn(out, w[p + "att.ln_x.weight"], w[p + "att.ln_x.bias"]) * g
        return y @ w[p + "att.output.weight"].T, S

    def _batt6(self, p, xx, sx, S):
        w = self.w
        B = xx.shape[0]
        dx = sx - xx
        z = xx + dx * w[p + "att.time_mix_x"].reshape(-1)
        m = np.tanh(z @ w[p + "att.time_mix_w1"].T)
        w2 = w[p + "att.time_mix_w2"]
        Dm = w2.shape[2]
        m = m.reshape(B, 5, Dm)
        xs = {}
        for c, n in enumerate("wkvrg"):
            mc = m[:, c] @ w2[c].T
            xs[n] = xx + dx * (w[p + f"att.time_mix_{n}"].reshape(-1) + mc)
        r = xs["r"] @ w[p + "att.receptance.weight"].T
        k = xs["k"] @ w[p + "att.key.weight"].T
        v = xs["v"] @ w[p + "att.value.weight"].T
        g = xs["g"] @ w[p + "att.gate.weight"].T
        g = g * _sigmoid(g)
        td = np.tanh(xs["w"] @ w[p + "att.time_decay_w1"].T)
        d = w[p + "att.time_decay"].reshape(-1) + td @ w[p + "att.time_decay_w2"].T
        wdec = np.exp(-np.exp(d.astype(np.float32)))
        out, S = self._bwkv56(r, k, v, wdec, w[p + "att.time_first"].reshape(-1), S)
        y = self._bgn(out, w[p + "att.ln_x.weight"], w[p + "att.ln_x.bias"]) * g
        return y @ w[p + "att.output.weight"].T, S

    def _bffn56(self, p, xx, sx):
        w = self.w
        if self.info.version == 5:
            mk, mr = w[p + "ffn.time_mix_k"].reshape(-1), w[p + "ffn.time_mix_r"].reshape(-1)
            xk = xx * mk + sx * (1 - mk)
            xr = xx * mr + sx * (1 - mr)
        else:
            dx = sx - xx
            xk = xx + dx * w[p + "ffn.time_mix_k"].reshape(-1)
            xr = xx + dx * w[p + "ffn.time_mix_r"].reshape(-1)
        r = _sigmoid(xr @ w[p + "ffn.receptance.weight"].T)
        k = np.maximum(xk @ w[p + "ffn.key.weight"].T, 0) ** 2
        return r * (k @ w[p + "ffn.value.weight"].T)

    def _batt7(self, p, l, xx, sx, S, v_first):
        w = self.w
        B = xx.shape[0]
        H, N = self.info.num_head, self.info.head_size
        dx = sx - xx
        xm = {n: xx + dx * w[p + f"att.x_{n}"].reshape(-1) for n in "rwkvag"}
        r = xm["r"] @ w[p + "att.receptance.weight"].T
        k = xm["k"] @ w[p + "att.key.weight"].T
        v = xm["v"] @ w[p + "att.value.weight"].T
        wd = np.tanh(xm["w"] @ w[p + "att.w1"].T) @ w[p + "att.w2"].T
        a = _sigmoid(w[p + "att.a0"].reshape(-1) + (xm["a"] @ w[p + "att.a1"].T) @ w[p + "att.a2"].T)
        g = _sigmoid(xm["g"] @ w[p + "att.g1"].T) @ w[p + "att.g2"].T
        kk = (k * w[p + "att.k_k"].reshape(-1)).reshape(B, H, N)
        kk = kk / np.maximum(np.sqrt((kk * kk).sum(axis=2, keepdims=True)), np.float32(1e-12))
        kk = kk.reshape(B, -1)
        k = k * (1 + (a - 1) * w[p + "att.k_a"].reshape(-1))
        if l == 0:
            v_first = v
        else:
            v = v + (v_first - v) * _sigmoid(w[p + "att.v0"].reshape(-1) + (xm["v"] @ w[p + "att.v1"].T) @ w[p + "att.v2"].T)
        wdec = np.exp(np.float32(-0.606531) * _sigmoid((w[p + "att.w0"].reshape(-1) + wd).astype(np.float32)))
        rh, kh, vh, kkh, ah, wh = (t.reshape(B, H, N) for t in (r, k, v, kk, a, wdec))
        sa = np.einsum("bhij,bhj->bhi", S, -kkh)
        S = S * wh[:, :, None, :] + sa[:, :, :, None] * (kkh * ah)[:, :, None, :] + vh[:, :, :, None] * kh[:, :, None, :]
        S = S.astype(np.float32)
        out = np.einsum("bhij,bhj->bhi", S, rh).reshape(B, -1)
        y = self._bgn(out, w[p + "att.ln_x.weight"], w[p + "att.ln_x.bias"])
        bonus = (rh * kh * w[p + "att.r_k"].reshape(1, H, N)).sum(axis=2, keepdims=True) * vh
        y = y + bonus.reshape(B, -1)
        return (y * g) @ w[p + "att.output.weight"].T, S, v_first

    def _bffn7(self, p, xx, sx):
        w = self.w
        xk = xx + (sx - xx) * w[p + "ffn.x_k"].reshape(-1)
        k = np.maximum(xk @ w[p + "ffn.key.weight"].T, 0) ** 2
        return k @ w[p + "ffn.value.weight"].T

    def init_states(self, B: int) -> np.ndarray:
        i = self.info
        return np.zeros((B, i.num_layer, i.head_size + 2, i.num_emb), dtype=np.float32)

    def prefill(self, prompts, states: np.ndarray):
        """Feed ragged prompts (one per slot) in lock step; returns the last-token logits of every slot [B, V]."""
        B = len(prompts)
        last = np.zeros((B, self.info.num_vocab), np.float32)
        for t in range(max(len(p) for p in prompts)):
            act = [b for b in range(B) if t < len(prompts[b])]
            sub = states[act].copy()
            lg = self.step([prompts[b][t] for b in act], sub)
            states[act] = sub
            for j, b in enumerate(act):
                if t == len(prompts[b]) - 1:
                    last[b] = lg[j]
        return last

    def greedy_batch(self, first_tokens, n_steps: int, states: np.ndarray):
        """`n_steps` decode steps for every slot, arg-max fed back (Nucleus top_k=1, nucleus.rs:77-89); returns ids [n_steps, B]
        (row s = the token chosen after step s) and the logits of the last step."""
        cur = [int(t) for t in first_tokens]
        out = np.zeros((n_steps, len(cur)), np.int64)
        lg = None
        for s in range(n_steps):
            lg = self.step(cur, states)
            cur = [int(t) for t in np.argmax(lg, axis=1)]
            out[s] = cur
        return out, lg


def softmax_ref(logits: np.ndarray) -> np.ndarray:
    """`softmax::softmax` (run.rs:1179): numerically-stable softmax over the vocab, fp32."""
    x = logits.astype(np.float32)
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def perplexity_ref(logit_rows: np.ndarray, tokens, head: float | None = None) -> float:
    """run.rs:699-755: p_t = exp(l)[tok]/sum(exp(l)) (no max-subtraction), ppl = -mean(ln p)."""
    p = []
    toks = list(tokens) if head is not None else [0] + list(tokens)
    if head is not None:
        p.append(head)
    idx = 1
    for row in logit_rows:
        if idx < len(toks):
            e = np.exp(row.astype(np.float32))
            p.append(float(e[toks[idx]] / e.sum(dtype=np.float32)))
        idx += 1
    return float(-np.sum(np.log(np.array(p, dtype=np.float32))) / len(toks))


# --------------------------------------------------------------------------------------
# synthetic checkpoints: seeded, exact names/shapes of the converted `.st`
# --------------------------------------------------------------------------------------
CONFIGS = {
    # name: (version, L, C, F, V)
    "v5-0.4b": (5, 24, 1024, 3584, 65536),
    "v6-1.6b": (6, 24, 2048, 7168, 65536),
    "v6-3b": (6, 32, 2560, 8960, 65536),
    "v6-7b": (6, 32, 4096, 14336, 65536),
    "v7-2.9b": (7, 32, 2560, 10240, 65536),
    # tiny shapes for tests (same structure, minutes -> milliseconds)
    "v5-tiny": (5, 2, 128, 448, 512),
    "v6-tiny": (6, 2, 128, 448, 512),
    "v7-tiny": (7, 2, 128, 512, 512),
    "v6-small": (6, 3, 256, 1024, 1024),
    "v5-small": (5, 2, 256, 768, 1024),
    "v7-small": (7, 3, 256, 1024, 1024),
}


def synth_checkpoint(version: int, L: int, C: int, F: int, V: int, seed: int = 20251024,
                     fast: bool = False, alloc=None, shapes_only: bool = False) -> dict[str, np.ndarray]:
    """Seeded synthetic fp16 tensors in the converted `.st` layout (App. A.1 of SURVEY.md;
    names/transposes follow convert_safetensors.py:96-101 literally).
    `fast=True` fills the big matrices by tiling a 16M-sample random block (bench-only; same statistics).
    `alloc(name, shape)` may supply the destination arrays (e.g. views into a file buffer);
    `shapes_only=True` returns {name: shape} without generating anything."""
    rng = np.random.Generator(np.random.SFC64(seed))
    H, N = C // HEAD_SIZE, HEAD_SIZE
    t: dict = {}
    pools: dict = {}
    base_pool = None

    def put(name, shape, gen):
        shape = tuple(int(x) for x in shape)
        if shapes_only:
            t[name] = shape
            return
        dst = alloc(name, shape) if alloc is not None else np.empty(shape, np.float16)
        gen(dst)
        t[name] = dst

    def mat(name, o, i, std=None):
        std = (0.5 / np.sqrt(i)) if std is None else std

        def gen(dst):
            nonlocal base_pool
            if fast and o * i > (1 << 20):
                key = float(std)
                if key not in pools:
                    if base_pool is None:
                        base_pool = rng.standard_normal(1 << 24, dtype=np.float32)
                    pools[key] = (base_pool * np.float32(std)).astype(np.float16)
                pool = pools[key]
                flat = dst.reshape(-1)
                off = int(rng.integers(0, 1 << 20))
                pos = 0
                while pos < flat.size:
                    n = min(pool.size - off, flat.size - pos)
                    flat[pos:pos + n] = pool[off:off + n]
                    pos += n
                    off = 0
            else:
                dst[...] = (rng.standard_normal((o, i), dtype=np.float32) * np.float32(std)).astype(np.float16)
        put(name, (o, i), gen)

    def vec(name, shape, mean=0.0, std=0.02):
        shape = (shape,) if isinstance(shape, int) else shape
        put(name, shape, lambda d: d.__setitem__(Ellipsis, (mean + rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float16)))

    def uni(name, shape, lo, hi):
        put(name, shape, lambda d: d.__setitem__(Ellipsis, rng.uniform(lo, hi, size=shape).astype(np.float16)))

    mat("emb.weight", V, C, std=0.5)
    for l in range(L):
        p = f"blocks.{l}."
        if l == 0:
            vec(p + "ln0.weight", C, 1.0)
            vec(p + "ln0.bias", C)
        for ln in ("ln1", "ln2"):
            vec(p + ln + ".weight", C, 1.0)
            vec(p + ln + ".bias", C)
        a = p + "att."
        if version == 5:
            for n in "kvrg":
                uni(a + f"time_mix_{n}", (1, 1, C), 0, 1)
            uni(a + "time_decay", (H, N), -6, -0.5)
            vec(a + "time_first", (H, N), 0.0, 0.3)
        elif version == 6:
            Dm, Dd = (64, 128) if C >= 4096 else (32, 64)
            for n in "xwkvrg":
                uni(a + f"time_mix_{n}", (1, 1, C), 0, 1)
            vec(a + "time_mix_w1", (5 * Dm, C))          # orig [C,5Dm] transposed
            vec(a + "time_mix_w2", (5, C, Dm))           # orig [5,Dm,C] transposed
            uni(a + "time_decay", (1, 1, C), -6, -0.5)
            vec(a + "time_decay_w1", (Dd, C))
            vec(a + "time_decay_w2", (C, Dd))
            vec(a + "time_first", (H, N), 0.0, 0.3)
        else:
            Dw, Da, Dv, Dg = (96, 96, 64, 320) if C >= 2560 else (64, 64, 32, 128)
            if C < 1024:
                Dw, Da, Dv, Dg = 32, 32, 32, 64
            for n in "rwkvag":
                uni(a + f"x_{n}", (1, 1, C), 0, 1)
            uni(a + "w0", (1, 1, C), -7, -1)
            vec(a + "w1", (Dw, C)); vec(a + "w2", (C, Dw))
            vec(a + "a0", (1, 1, C), 0.0, 0.3)
            vec(a + "a1", (Da, C)); vec(a + "a2", (C, Da))
            vec(a + "v0", (1, 1, C), 0.0, 0.3)
            vec(a + "v1", (Dv, C)); vec(a + "v2", (C, Dv))
            vec(a + "g1", (Dg, C), 0.0, 0.05); vec(a + "g2", (C, Dg), 0.0, 0.05)
            vec(a + "k_k", (1, 1, C), 0.85, 0.05)
            vec(a + "k_a", (1, 1, C), 1.0, 0.02)
            vec(a + "r_k", (H, N), 0.0, 0.1)
        for n in ("receptance", "key", "value", "output") + (("gate",) if version != 7 else ()):
            mat(a + n + ".weight", C, C)
        vec(a + "ln_x.weight", C, 1.0)
        vec(a + "ln_x.bias", C)
        f = p + "ffn."
        if version == 7:
            uni(f + "x_k", (1, 1, C), 0, 1)
        else:
            uni(f + "time_mix_k", (1, 1, C), 0, 1)
            uni(f + "time_mix_r", (1, 1, C), 0, 1)
            mat(f + "receptance.weight", C, C)
        mat(f + "key.weight", F, C)
        mat(f + "value.weight", C, F)
    vec("ln_out.weight", C, 1.0)
    vec("ln_out.bias", C)
    mat("head.weight", V, C)
    return t


def synth_st(name: str, seed: int = 20251024, fast: bool = True):
    """Generate a synthetic checkpoint straight into one `.st` file image.
    Returns (file: np.uint8 array, tensors: {name: fp16 view into file})."""
    cfg = CONFIGS[name]
    shapes = synth_checkpoint(*cfg, seed=seed, shapes_only=True)
    header, off = {}, 0
    for k, shp in shapes.items():
        n = int(np.prod(shp)) * 2
        header[k] = {"dtype": "F16", "shape": list(shp), "data_offsets": [off, off + n]}
        off += n
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    buf = big_empty((8 + len(hj) + off,), np.uint8)
    buf[:8] = np.frombuffer(struct.pack("<Q", len(hj)), np.uint8)
    buf[8:8 + len(hj)] = np.frombuffer(hj, np.uint8)
    base = 8 + len(hj)

    def alloc(k, shp):
        a, b = header[k]["data_offsets"]
        return buf[base + a:base + b].view(np.float16).reshape(shp)

    tensors = synth_checkpoint(*cfg, seed=seed, fast=fast, alloc=alloc)
    return buf, tensors


def synth_named(name: str, seed: int = 20251024, fast: bool = False):
    return synth_checkpoint(*CONFIGS[name], seed=seed, fast=fast)


def synth_init_state(info: ModelInfo, seed: int = 7) -> dict[str, np.ndarray]:
    """A state-tuned `.state` file: `blocks.i.att.time_state` [H,N,N] (transposed, fp16)."""
    rng = np.random.Generator(np.random.SFC64(seed))
    H, N = info.num_head, info.head_size
    return {f"blocks.{l}.att.time_state":
            (rng.standard_normal((H, N, N), dtype=np.float32) * np.float32(0.1)).astype(np.float16)
            for l in range(info.num_layer)}


def synth_prompt(slot: int, n: int) -> list[int]:
    """token ids uniform in [1, 65529]; clipped to the vocab by the caller if smaller."""
    rng = np.random.default_rng(1234 + slot)
    return [int(v) for v in rng.integers(1, 65530, size=n)]


def algorithmic_bytes(info: ModelInfo, tensors_shapes: dict[str, tuple], quant_layers: int,
                      quant_type: int, batch: int) -> dict:
    """bytes(B) = W_q + B*(2*S + V*4); embedding table excluded."""
    width = {QUANT_NONE: 2.0, QUANT_INT8: 1.0 + 4.0 / INT8_BLOCK, QUANT_NF4: 0.5 + 2.0 / NF4_BLOCK}
    qn = set()
    if quant_type != QUANT_NONE:
        for l in range(min(quant_layers, info.num_layer)):
            for n in quantised_matrix_names(info.version):
                qn.add(f"blocks.{l}.{n}")
    wq = 0.0
    for k, shp in tensors_shapes.items():
        if k == "emb.weight":
            continue
        n = int(np.prod(shp))
        wq += n * (width[quant_type] if k in qn else 2.0)
    S = info.num_layer * (info.head_size + 2) * info.num_emb * 4
    return {"W_q": wq, "S": S, "per_step": wq + batch * (2 * S + info.num_vocab * 4)}


# --------------------------------------------------------------------------------------
# sampler reference (crates/ai00-core/src/sampler/nucleus.rs) — for the on-device front-end
# --------------------------------------------------------------------------------------
def nucleus_ref(probs: np.ndarray, top_p: float, top_k: int, temperature: float, u: float):
    """`NucleusSampler::sample` (nucleus.rs:69-101) with the random draw `u` made explicit.
    Sort descending, take top_k, keep while the cumulative sum BEFORE the element is <= top_p (the first is always kept),
    p^(1/T), renormalise, first element with u <= cumulative, else the FIRST element (`find_or_first`); `top_k = 0` keeps
    nothing and the `unwrap_or_default` at nucleus.rs:101 yields token 0.  Ties: the reference sorts with `voracious_sort`
    (nucleus.rs:76), an UNSTABLE radix sort, so the order of equal probabilities is not defined by the reference; this
    restatement (and the device kernel) fix it as lower id first.
    Returns (token, margin) where margin is the distance of the decisive comparison."""
    p = probs.astype(np.float32)
    if top_k < 1:
        return 0, 1.0
    order = np.lexsort((np.arange(p.size), -p))[:top_k]
    kept, cum = [], np.float32(0.0)
    for i in order:
        if cum > np.float32(top_p):
            break
        cum = np.float32(cum + p[i])
        kept.append(i)
    q = np.array([np.float32(p[i]) ** np.float32(1.0 / temperature) for i in kept], dtype=np.float32)
    s = np.float32(0.0)
    for x in q:
        s = np.float32(s + x)
    c, margin = np.float32(0.0), 1.0
    for i, x in zip(kept, q):
        c = np.float32(c + np.float32(x / s))
        margin = min(margin, abs(float(c) - u))
        if np.float32(u) <= c:
            return int(i), margin
    return int(kept[0]), margin


def typical_ref(probs: np.ndarray, tau: float, top_k: int, temperature: float, u: float, h_shift: float = 0.0):
    """`TypicalSampler::sample` (typical.rs:70-120) with the random draw `u` made explicit: over p > 0, surprise
    y = -ln p, entropy H = sum p y, sort by |y - H| ascending (ties: lower id first), take top_k, keep while the
    cumulative probability BEFORE the element is <= tau, p^(1/T), renormalise, first element with u <= cumulative, else
    the first.  Returns (token, margin of the decisive CDF comparison).  `h_shift` perturbs H: it is a many-term fp32 sum
    whose last bits depend on the summation order, and two keys on opposite sides of H swap when H moves by half their
    gap — a checker compares against the answers for a few shifts."""
    p = probs.astype(np.float32)
    ids = np.nonzero(p > 0)[0]
    y = -np.log(p[ids]).astype(np.float32)
    h = np.float32(0.0)
    for a, b in zip(p[ids], y):
        h = np.float32(h + np.float32(a * b))
    h = np.float32(h + np.float32(h_shift))
    key = np.abs(y - h).astype(np.float32)
    if top_k < 1:
        return 0, 1.0                                                   # `.take(0)` -> `unwrap_or_default` (typical.rs:87, 117)
    order = np.lexsort((ids, key))[:top_k]
    kept, cum = [], np.float32(0.0)
    for o in order:
        if cum > np.float32(tau):
            break
        cum = np.float32(cum + p[ids[o]])
        kept.append(o)
    q = np.array([np.float32(p[ids[o]]) ** np.float32(1.0 / temperature) for o in kept], dtype=np.float32)
    s_ = np.float32(0.0)
    for x in q:
        s_ = np.float32(s_ + x)
    c, margin = np.float32(0.0), 1.0
    for o, x in zip(kept, q):
        c = np.float32(c + np.float32(x / s_))
        margin = min(margin, abs(float(c) - u))
        if np.float32(u) <= c:
            return int(ids[o]), margin
    return int(ids[kept[0]]), margin


def mirostat_ref(probs: np.ndarray, max_surprise: float, u: float):
    """`MirostatSampler::sample` (mirostat.rs:44-84) with the draw explicit: sort descending (ties: lower id first),
    running sum, k = 1 + position of the first token whose surprise -log2 p exceeds max_surprise (all if none), draw
    u * sum against the running sum.  Returns (token, token_surprise = log2(sum) - log2(p), margin)."""
    p = probs.astype(np.float32)
    order = np.lexsort((np.arange(p.size), -p))
    with np.errstate(divide="ignore"):
        over = np.nonzero(-np.log2(p[order]) > np.float32(max_surprise))[0]
    k = int(over[0]) + 1 if over.size else p.size
    cum, c = [], np.float32(0.0)
    for i in order[:k]:
        c = np.float32(c + p[i])
        cum.append(c)
    total = cum[-1]
    r = np.float32(np.float32(u) * total)
    margin = 1.0
    for i, cc in zip(order[:k], cum):
        margin = min(margin, abs(float(cc) - float(r)))
        if r <= cc:
            return int(i), float(np.log2(total) - np.log2(p[i])), margin
    return int(order[0]), float(np.log2(total) - np.log2(p[order[0]])), margin


class NucleusRef:
    """State machine of `NucleusSampler` (nucleus.rs:13-122): penalties map, init/transform/update."""

    def __init__(self, top_p=0.5, top_k=128, temperature=1.0, presence_penalty=0.3, frequency_penalty=0.3,
                 penalty_decay=0.99654026):
        self.top_p, self.top_k, self.temperature = top_p, top_k, temperature
        self.ap, self.af, self.ad = presence_penalty, frequency_penalty, penalty_decay
        self.penalties: dict[int, float] = {}

    def init(self, model_tokens):                                    # nucleus.rs:49-59
        for index, token in enumerate(reversed(list(model_tokens))):
            pen = self.penalties.pop(int(token), np.float32(self.ap))
            self.penalties[int(token)] = np.float32(pen + np.float32(self.af) * np.float32(self.ad) ** np.float32(index))

    def transform(self, logits: np.ndarray) -> np.ndarray:            # nucleus.rs:61-67
        out = logits.astype(np.float32).copy()
        for t, pen in self.penalties.items():
            out[t] -= np.float32(pen)
        return out

    def update(self, token: int):                                     # nucleus.rs:104-119
        for t in self.penalties:
            self.penalties[t] = np.float32(self.penalties[t] * np.float32(self.ad))
        if token in self.penalties:
            self.penalties[token] = np.float32(self.penalties[token] + np.float32(self.af))
        else:
            self.penalties[token] = np.float32(self.ap)

    def sample(self, logits: np.ndarray, bias: dict | None, u: float) -> int:   # run.rs:664-697 + nucleus.rs:69-122
        x = self.transform(logits)
        for t, b in (bias or {}).items():
            x[t] += np.float32(b)
        tok, _ = nucleus_ref(softmax_ref(x[None])[0], self.top_p, self.top_k, self.temperature, u)
        self.update(tok)
        return tok
