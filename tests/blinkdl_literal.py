"""Second, independent restatement of the PUBLISHED BlinkDL single-token inference functions (rwkv pip package
`model.py`: att_one_v5_2 / att_one_v6_0 / ffn_one*, and `rwkv_v7_demo.py`: RWKV_x070_TMix_one / CMix_one), written
with torch in BlinkDL's ORIGINAL tensor layout.  Test-only: it pins the oracle's handling of the `.st` transposes
(convert_safetensors.py:96-101) and of the state conventions against a form that does the algebra differently
(batched matmuls on [H,N,N] instead of einsums)."""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float16).astype(np.float32))


class Literal:
    def __init__(self, st: dict, layout: str = "converted"):
        """`layout="converted"`: `st` is an ai00 `.st` (the converter's renames and transposes are undone here);
        `layout="blinkdl"`: `st` is a checkpoint as BlinkDL's trainer saves it (`time_maa_*`, `time_faaaa`, nothing
        transposed) — only the two renames are applied, so the arithmetic below runs on the ORIGINAL tensors."""
        if layout == "blinkdl":
            st = {k.replace("time_faaaa", "time_first").replace("time_maa", "time_mix"): a for k, a in st.items()}
        self.v = 7 if "blocks.0.att.x_r" in st else 6 if "blocks.0.att.time_mix_x" in st else 5
        w = {}
        for k, a in st.items():
            t = _t(a)
            # undo the converter: names containing these substrings were transposed on the last two dims
            if layout == "converted" and any(s in k for s in ["time_mix_w1", "time_mix_w2", "time_decay_w1", "time_decay_w2", "w1", "w2", "a1",
                                                               "a2", "g1", "g2", "v1", "v2", "time_state", "lora.0"]):
                t = t.transpose(-1, -2).contiguous()
            if k.endswith(".weight") and t.dim() == 2 and "emb" not in k and "ln" not in k:
                t = t.t().contiguous()          # BlinkDL inference stores linear weights as [in, out]: x @ w
            w[k] = t
        self.w = w
        self.L = sum(1 for k in st if k.endswith(".ln1.weight"))
        self.C = st["emb.weight"].shape[1]
        self.H = (st["blocks.0.att.r_k"] if self.v == 7 else st["blocks.0.att.time_first"]).shape[0]
        self.N = self.C // self.H

    def new_state(self):
        return [[torch.zeros(self.C), torch.zeros(self.H, self.N, self.N), torch.zeros(self.C)] for _ in range(self.L)]

    def forward(self, token, state):
        w, H, N = self.w, self.H, self.N
        x = F.layer_norm(w["emb.weight"][token], (self.C,), w["blocks.0.ln0.weight"], w["blocks.0.ln0.bias"])
        v_first = None
        for i in range(self.L):
            p = f"blocks.{i}."
            a = p + "att."
            xx = F.layer_norm(x, (self.C,), w[p + "ln1.weight"], w[p + "ln1.bias"])
            sx, s = state[i][0], state[i][1]
            if self.v == 5:
                kx = xx * w[a + "time_mix_k"].flatten() + sx * (1 - w[a + "time_mix_k"].flatten())
                vx = xx * w[a + "time_mix_v"].flatten() + sx * (1 - w[a + "time_mix_v"].flatten())
                rx = xx * w[a + "time_mix_r"].flatten() + sx * (1 - w[a + "time_mix_r"].flatten())
                gx = xx * w[a + "time_mix_g"].flatten() + sx * (1 - w[a + "time_mix_g"].flatten())
                r = (rx @ w[a + "receptance.weight"]).view(H, 1, N)
                k = (kx @ w[a + "key.weight"]).view(H, N, 1)
                v = (vx @ w[a + "value.weight"]).view(H, 1, N)
                g = F.silu(gx @ w[a + "gate.weight"])
                t_decay = torch.exp(-torch.exp(w[a + "time_decay"])).view(H, N, 1)
                t_first = w[a + "time_first"].view(H, N, 1)
                at = k @ v
                out = r @ (t_first * at + s)
                s = at + t_decay * s
                out = out.flatten()
                out = F.group_norm(out.unsqueeze(0), num_groups=H, weight=w[a + "ln_x.weight"], bias=w[a + "ln_x.bias"], eps=64e-5).squeeze(0)
                att = (out * g) @ w[a + "output.weight"]
            elif self.v == 6:
                dx = sx - xx
                xxx = xx + dx * w[a + "time_mix_x"].flatten()
                xxx = torch.tanh(xxx @ w[a + "time_mix_w1"]).view(5, 1, -1)       # w1 [C, 5*Dm]
                xxx = torch.bmm(xxx, w[a + "time_mix_w2"]).view(5, -1)             # w2 [5, Dm, C]
                mw, mk, mv, mr, mg = xxx.unbind(dim=0)
                wx = xx + dx * (w[a + "time_mix_w"].flatten() + mw)
                kx = xx + dx * (w[a + "time_mix_k"].flatten() + mk)
                vx = xx + dx * (w[a + "time_mix_v"].flatten() + mv)
                rx = xx + dx * (w[a + "time_mix_r"].flatten() + mr)
                gx = xx + dx * (w[a + "time_mix_g"].flatten() + mg)
                r = (rx @ w[a + "receptance.weight"]).view(H, 1, N)
                k = (kx @ w[a + "key.weight"]).view(H, N, 1)
                v = (vx @ w[a + "value.weight"]).view(H, 1, N)
                g = F.silu(gx @ w[a + "gate.weight"])
                ww = w[a + "time_decay"].flatten() + (torch.tanh(wx @ w[a + "time_decay_w1"]) @ w[a + "time_decay_w2"])
                ww = torch.exp(-torch.exp(ww.float())).view(H, N, 1)
                t_first = w[a + "time_first"].view(H, N, 1)
                at = k @ v
                out = r @ (t_first * at + s)
                s = at + ww * s
                out = out.flatten()
                out = F.group_norm(out.unsqueeze(0), num_groups=H, weight=w[a + "ln_x.weight"], bias=w[a + "ln_x.bias"], eps=64e-5).squeeze(0)
                att = (out * g) @ w[a + "output.weight"]
            else:
                xd = sx - xx
                f = lambda n: xx + xd * w[a + "x_" + n].flatten()
                xr, xw, xk, xv, xa, xg = f("r"), f("w"), f("k"), f("v"), f("a"), f("g")
                r = xr @ w[a + "receptance.weight"]
                ww = torch.tanh(xw @ w[a + "w1"]) @ w[a + "w2"]
                k = xk @ w[a + "key.weight"]
                v = xv @ w[a + "value.weight"]
                aa = torch.sigmoid(w[a + "a0"].flatten() + (xa @ w[a + "a1"]) @ w[a + "a2"])
                g = torch.sigmoid(xg @ w[a + "g1"]) @ w[a + "g2"]
                kk = F.normalize((k * w[a + "k_k"].flatten()).view(H, N), dim=-1, p=2.0).view(H * N)
                k = k * (1 + (aa - 1) * w[a + "k_a"].flatten())
                if i == 0:
                    v_first = v
                else:
                    v = v + (v_first - v) * torch.sigmoid(w[a + "v0"].flatten() + (xv @ w[a + "v1"]) @ w[a + "v2"])
                ww = torch.exp(-0.606531 * torch.sigmoid((w[a + "w0"].flatten() + ww).float()))
                vk = v.view(H, N, 1) @ k.view(H, 1, N)
                ab = (-kk).view(H, N, 1) @ (kk * aa).view(H, 1, N)
                s = s * ww.view(H, 1, N) + s @ ab.float() + vk.float()
                out = (s @ r.view(H, N, 1)).view(1, H * N)
                out = F.group_norm(out, num_groups=H, weight=w[a + "ln_x.weight"], bias=w[a + "ln_x.bias"], eps=64e-5).view(H * N)
                out = out + ((r * k * w[a + "r_k"].flatten()).view(H, N).sum(dim=-1, keepdim=True) * v.view(H, N)).view(H * N)
                att = (out * g) @ w[a + "output.weight"]
            state[i][0], state[i][1] = xx, s
            x = x + att
            f_ = p + "ffn."
            xx = F.layer_norm(x, (self.C,), w[p + "ln2.weight"], w[p + "ln2.bias"])
            sx = state[i][2]
            if self.v == 5:
                kx = xx * w[f_ + "time_mix_k"].flatten() + sx * (1 - w[f_ + "time_mix_k"].flatten())
                rx = xx * w[f_ + "time_mix_r"].flatten() + sx * (1 - w[f_ + "time_mix_r"].flatten())
            elif self.v == 6:
                kx = xx + (sx - xx) * w[f_ + "time_mix_k"].flatten()
                rx = xx + (sx - xx) * w[f_ + "time_mix_r"].flatten()
            else:
                kx = xx + (sx - xx) * w[f_ + "x_k"].flatten()
            vx = torch.relu(kx @ w[f_ + "key.weight"]) ** 2
            out = vx @ w[f_ + "value.weight"]
            if self.v != 7:
                out = torch.sigmoid(rx @ w[f_ + "receptance.weight"]) * out
            state[i][2] = xx
            x = x + out
        x = F.layer_norm(x, (self.C,), w["ln_out.weight"], w["ln_out.bias"])
        return (x @ w["head.weight"]).numpy()
