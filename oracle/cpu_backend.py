"""oracle/cpu_backend.py — TEST INFRASTRUCTURE: ctypes front of oracle/cpu_backend.c, the compiled (C + OpenMP) second restatement of
the lock-step decode step for RWKV V5.2 / V6 / V7.  Same surface as `rwkv_ref.RwkvRefBatch` (`step`, `init_states`, `greedy_batch`), same
weights (checkpoint tensors rounded through fp16; quantised layers fake-quantised to the fp16 value the GPU dequantises to — Int8 and NF4 in C,
bit for bit `rwkv_ref.fake_quant`).  Used by tests/test_oracle.py (cross-check of the two restatements)
and by bench.py's `cpu_baseline` leg.  The product never imports this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import rwkv_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpu_backend.c")
LIB = os.path.join(HERE, "_build", "libcpu_backend.so")


def build(force: bool = False) -> str:
    """gcc -O3 -mavx2 -mfma -mf16c -fopenmp (a baseline every x86 server of the last decade has; not -march=native: the library built
    in one container also runs on the GPU box's host)."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-mavx2", "-mfma", "-mf16c", "-fopenmp", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror",
                               SRC, "-o", LIB, "-lm"])
    return LIB


_u16p, _f32p = C.POINTER(C.c_uint16), C.POINTER(C.c_float)


class _Layer(C.Structure):
    _fields_ = [(n, _f32p) for n in ("ln1w", "ln1b", "ln2w", "ln2b", "mix_x", "mix_w", "mix_k", "mix_v", "mix_r", "mix_g")] + \
               [("mix_w1", _u16p), ("mix_w2", _u16p), ("decay", _f32p), ("first", _f32p), ("decay_w1", _u16p), ("decay_w2", _u16p)] + \
               [(n, _u16p) for n in ("Wr", "Wk", "Wv", "Wg", "Wo")] + [("lnxw", _f32p), ("lnxb", _f32p), ("fmix_k", _f32p), ("fmix_r", _f32p)] + \
               [(n, _u16p) for n in ("Fk", "Fv", "Fr")] + \
               [(n, _f32p) for n in ("mix_a", "w0", "a0", "v0", "k_k", "k_a", "r_k")] + \
               [(n, _u16p) for n in ("w1", "w2", "a1", "a2", "v1", "v2", "g1", "g2")] + [(n, C.c_int32) for n in ("Dw", "Da", "Dv", "Dg")]


class _Model(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("version", "L", "C", "F", "V", "H", "Dm", "Dd")] + \
               [("emb", _u16p), ("head", _u16p), ("ln0w", _f32p), ("ln0b", _f32p), ("lnow", _f32p), ("lnob", _f32p), ("layers", C.POINTER(_Layer))]


def usable_cores() -> list[int]:
    """One logical CPU per physical core this process may run on, socket by socket (hyper-thread siblings dropped: a memory-bound
    GEMV gains nothing from them), capped by the cgroup CPU quota when the container has one."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, picked = set(), []
    try:
        cur = {}
        topo = {}
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k in ("processor", "physical id", "core id"):
                cur[k] = int(v)
            if not line.strip() and "processor" in cur:
                topo[cur["processor"]] = (cur.get("physical id", 0), cur.get("core id", cur["processor"]))
                cur = {}
        if "processor" in cur:
            topo[cur["processor"]] = (cur.get("physical id", 0), cur.get("core id", cur["processor"]))
        for cpu in sorted(allowed, key=lambda c: (topo.get(c, (0, c)), c)):
            key = topo.get(cpu, (0, cpu))
            if key not in seen:
                seen.add(key)
                picked.append(cpu)
    except OSError:
        picked = allowed
    try:                                                           # cgroup v2 quota: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            picked = picked[:max(1, int(int(q) / int(per)))]
    except (OSError, ValueError):
        pass
    return picked or allowed


class CpuBackend:
    def __init__(self, tensors: dict, quant_layers: int = 0, quant_type: int = R.QUANT_NONE, threads: int | None = None):
        self.lib = C.CDLL(build())
        self.lib.rwkv_cpu_pin.argtypes = [C.POINTER(C.c_int32), C.c_int]
        self.lib.rwkv_cpu_pin.restype = C.c_int
        self.lib.rwkv_cpu_place.argtypes = [_u16p, C.c_long, C.c_long]
        self.lib.rwkv_cpu_place.restype = C.c_void_p
        self.lib.rwkv_cpu_free.argtypes = [C.c_void_p]
        # team = one thread per physical core, pinned (RWKV_CPU_THREADS overrides the count: scaling experiments)
        cores = usable_cores()
        n = threads or int(os.environ.get("RWKV_CPU_THREADS", "0")) or len(cores)
        n = max(1, min(n, len(cores)))
        arr = (C.c_int32 * n)(*cores[:n])
        self.pin_failures = int(self.lib.rwkv_cpu_pin(arr, n))
        self._placed = []
        self.weight_bytes = 0
        self.lib.rwkv_cpu_step.restype = C.c_int
        self.lib.rwkv_cpu_step.argtypes = [C.POINTER(_Model), C.POINTER(C.c_int32), C.c_int, _f32p, _f32p]
        self.lib.rwkv_cpu_fake_quant_int8.argtypes = [_u16p, C.c_long, C.c_long]
        self.lib.rwkv_cpu_threads.restype = C.c_int
        self.lib.rwkv_cpu_fake_quant_nf4.argtypes = [_u16p, C.c_long, C.c_long, _f32p, _u16p]
        self._nf4_mid = np.ascontiguousarray(R.NF4_MID, dtype=np.float32)
        self._nf4_tab = np.ascontiguousarray(R.NF4_TABLE_F16, dtype=np.float16)
        self.info = i = R.model_info(tensors)
        if i.version not in (5, 6, 7):
            raise NotImplementedError("the compiled restatement covers V5.2, V6 and V7")
        qn = set()
        if quant_type != R.QUANT_NONE:
            for l in range(min(quant_layers, i.num_layer)):
                qn.update(f"blocks.{l}.{n}" for n in R.quantised_matrix_names(i.version))
        self._keep = []

        def mat(name):                                   # fp16 matrix, row-major, (fake-)quantised if the layer is
            a = np.ascontiguousarray(np.asarray(tensors[name], dtype=np.float16))
            if name in qn:
                a = a.copy()
                if quant_type == R.QUANT_INT8:
                    a2 = a.reshape(-1, a.shape[-1])
                    self.lib.rwkv_cpu_fake_quant_int8(a2.view(np.uint16).ctypes.data_as(_u16p), a2.shape[0], a2.shape[1])
                else:
                    a2 = a.reshape(-1, a.shape[-1])
                    self.lib.rwkv_cpu_fake_quant_nf4(a2.view(np.uint16).ctypes.data_as(_u16p), a2.shape[0], a2.shape[1],
                                                     self._nf4_mid.ctypes.data_as(_f32p), self._nf4_tab.view(np.uint16).ctypes.data_as(_u16p))
            # NUMA placement: the matrix the step reads is a copy first-touched by the threads that will stream it (rwkv_cpu_place)
            a2 = a.reshape(-1, a.shape[-1])
            ptr = self.lib.rwkv_cpu_place(a2.view(np.uint16).ctypes.data_as(_u16p), a2.shape[0], a2.shape[1])
            if not ptr:
                raise MemoryError(name)
            self._placed.append(ptr)
            if name != "emb.weight":
                self.weight_bytes += a.nbytes
            return C.cast(ptr, _u16p)

        def vec(name):                                   # fp32 of the fp16-rounded values, flat
            a = np.ascontiguousarray(np.asarray(tensors[name], dtype=np.float16).astype(np.float32).reshape(-1))
            self._keep.append(a)
            return a.ctypes.data_as(_f32p)

        none16, none32 = C.cast(None, _u16p), C.cast(None, _f32p)
        Dm = Dd = 0
        if i.version == 6:
            Dm = int(np.asarray(tensors["blocks.0.att.time_mix_w2"]).shape[2])
            Dd = int(np.asarray(tensors["blocks.0.att.time_decay_w1"]).shape[0])
        self._layers = (_Layer * i.num_layer)()
        for l in range(i.num_layer):
            p = f"blocks.{l}."
            y = self._layers[l]
            y.ln1w, y.ln1b, y.ln2w, y.ln2b = vec(p + "ln1.weight"), vec(p + "ln1.bias"), vec(p + "ln2.weight"), vec(p + "ln2.bias")
            if i.version == 7:
                for n in "rwkvag":
                    setattr(y, "mix_" + n, vec(p + "att.x_" + n))
                y.mix_x, y.mix_w1, y.mix_w2, y.decay_w1, y.decay_w2, y.decay, y.first = none32, none16, none16, none16, none16, none32, none32
                y.w0, y.a0, y.v0 = vec(p + "att.w0"), vec(p + "att.a0"), vec(p + "att.v0")
                y.k_k, y.k_a, y.r_k = vec(p + "att.k_k"), vec(p + "att.k_a"), vec(p + "att.r_k")
                for n in ("w1", "w2", "a1", "a2", "v1", "v2", "g1", "g2"):
                    setattr(y, n, mat(p + "att." + n))
                y.Dw, y.Da, y.Dv, y.Dg = (int(np.asarray(tensors[p + "att." + n]).shape[0]) for n in ("w1", "a1", "v1", "g1"))
                y.Wr, y.Wk, y.Wv = mat(p + "att.receptance.weight"), mat(p + "att.key.weight"), mat(p + "att.value.weight")
                y.Wg, y.Wo = none16, mat(p + "att.output.weight")
                y.lnxw, y.lnxb = vec(p + "att.ln_x.weight"), vec(p + "att.ln_x.bias")
                y.fmix_k, y.fmix_r = vec(p + "ffn.x_k"), none32
                y.Fk, y.Fv, y.Fr = mat(p + "ffn.key.weight"), mat(p + "ffn.value.weight"), none16
                continue
            for n in "kvrg":
                setattr(y, "mix_" + n, vec(p + "att.time_mix_" + n))
            if i.version == 6:
                y.mix_x, y.mix_w = vec(p + "att.time_mix_x"), vec(p + "att.time_mix_w")
                y.mix_w1, y.mix_w2 = mat(p + "att.time_mix_w1"), mat(p + "att.time_mix_w2")
                y.decay_w1, y.decay_w2 = mat(p + "att.time_decay_w1"), mat(p + "att.time_decay_w2")
            else:
                y.mix_x, y.mix_w, y.mix_w1, y.mix_w2, y.decay_w1, y.decay_w2 = none32, none32, none16, none16, none16, none16
            y.decay, y.first = vec(p + "att.time_decay"), vec(p + "att.time_first")
            y.Wr, y.Wk, y.Wv = mat(p + "att.receptance.weight"), mat(p + "att.key.weight"), mat(p + "att.value.weight")
            y.Wg, y.Wo = mat(p + "att.gate.weight"), mat(p + "att.output.weight")
            y.lnxw, y.lnxb = vec(p + "att.ln_x.weight"), vec(p + "att.ln_x.bias")
            y.fmix_k, y.fmix_r = vec(p + "ffn.time_mix_k"), vec(p + "ffn.time_mix_r")
            y.Fk, y.Fv, y.Fr = mat(p + "ffn.key.weight"), mat(p + "ffn.value.weight"), mat(p + "ffn.receptance.weight")
        self._model = _Model(i.version, i.num_layer, i.num_emb, i.num_hidden, i.num_vocab, i.num_head, Dm, Dd,
                             mat("emb.weight"), mat("head.weight"), vec("blocks.0.ln0.weight"), vec("blocks.0.ln0.bias"),
                             vec("ln_out.weight"), vec("ln_out.bias"), self._layers)

    def __del__(self):
        for p in getattr(self, "_placed", []):
            try:
                self.lib.rwkv_cpu_free(p)
            except Exception:
                pass
        self._placed = []

    @property
    def threads(self) -> int:
        return int(self.lib.rwkv_cpu_threads())

    def describe(self) -> str:
        return (f"C/OpenMP restatement (oracle/cpu_backend.c): fp16 weights as the GPU dequantises them, fp32 accumulate, one persistent team of "
                f"{self.threads} threads pinned one per physical core, weights first-touched by the threads that stream them")

    def stream_gbps(self, steps_per_s: float) -> float:
        """Weight bytes one step streams (every matrix once, the embedding row excluded) x steps/s: what the host's DRAM delivered."""
        return self.weight_bytes * steps_per_s / 1e9

    OPERAND_CLASSES = ("att", "lora1", "lora2", "wo", "ffn1", "fv", "mix1", "mix2", "decay2", "head")

    def set_operand_rounding(self, mask: int) -> None:
        """Error attribution only (scripts/fp16_error_attribution.py): bit i set -> the GEMM operands of OPERAND_CLASSES[i] are rounded to
        fp16 on the way in, as `Precision::Fp16` does on the GPU.  0 (the default) is the oracle every parity test uses."""
        self.lib.rwkv_cpu_set_operand_rounding.argtypes = [C.c_int]
        self.lib.rwkv_cpu_set_operand_rounding(int(mask))

    def init_states(self, B: int) -> np.ndarray:
        i = self.info
        return np.zeros((B, i.num_layer, i.head_size + 2, i.num_emb), dtype=np.float32)

    def step(self, tokens, states: np.ndarray, want_logits: bool = True):
        """One token per slot; `states` [B, L, N+2, C] float32 C-contiguous, updated in place; returns logits [B, V] or None."""
        B = len(tokens)
        assert states.dtype == np.float32 and states.flags.c_contiguous and states.shape[0] == B
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        logits = np.empty((B, self.info.num_vocab), np.float32) if want_logits else None
        rc = self.lib.rwkv_cpu_step(C.byref(self._model), tok.ctypes.data_as(C.POINTER(C.c_int32)), B, states.ctypes.data_as(_f32p),
                                    logits.ctypes.data_as(_f32p) if want_logits else C.cast(None, _f32p))
        if rc != 0:
            raise RuntimeError("rwkv_cpu_step failed")
        return logits

    def greedy_batch(self, first_tokens, n_steps: int, states: np.ndarray):
        cur = [int(t) for t in first_tokens]
        ids = np.zeros((n_steps, len(cur)), np.int64)
        lg = None
        for s in range(n_steps):
            lg = self.step(cur, states)
            cur = [int(x) for x in np.argmax(lg, axis=1)]
            ids[s] = cur
        return ids, lg
