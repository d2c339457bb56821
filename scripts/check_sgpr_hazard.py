"""Build-time lint (CPU, hipcc only): inline-asm VMEM instructions whose scalar base was written by a VALU instruction (v_readlane / v_readfirstlane:
an SGPR the register allocator spilled to a VGPR lane, or a value it thought divergent) fewer than five instructions earlier.  gfx9 needs five wait
states between a VALU write of an SGPR and a VMEM read of it; hipcc's hazard recogniser inserts them for its own instructions but does not look inside
an asm statement (round 6: the stream-K hand-off stores faulted on exactly this).

    python scripts/check_sgpr_hazard.py [part ...]        # compiles rwkv_kernels.hip -DRWKV_PART=k to assembly (device only) and scans it
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ai00_server_amd", "csrc", "rwkv_kernels.hip")
VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)")
SREG = re.compile(r"s\[(\d+):(\d+)\]|\bs(\d+)\b")
VALU_SGPR_WRITE = re.compile(r"^\s*(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)")

def scan(path):
    bad = []
    window = []                     # (index of real instruction, sgpr written by VALU)
    n = 0
    in_asm = False
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"): in_asm = True; continue
        if t.startswith(";;#ASMEND"): in_asm = False; continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        m = VALU_SGPR_WRITE.match(line)
        if m: window.append((n, int(m.group(2))))
        if in_asm and VMEM.match(line):
            used = set()
            for a, b, c in SREG.findall(line):
                if a: used.update(range(int(a), int(b) + 1))
                elif c: used.add(int(c))
            for idx, reg in window:
                if reg in used and n - idx <= 5 and "s_nop 4" not in prev_asm_text:
                    bad.append((ln, t, reg, n - idx))
        if in_asm:
            prev_asm_text = (prev_asm_text + " " + t) if asm_open else t
            asm_open = True
        else:
            asm_open = False; prev_asm_text = ""
        n += 1
        window = [(i, r) for i, r in window if n - i <= 6]
    return bad

asm_open = False; prev_asm_text = ""
parts = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 3, 5]
rc = 0
for k in parts:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, f"p{k}.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DRWKV_PART={k}", "-S", "--cuda-device-only", SRC, "-o", out])
        bad = scan(out)
        print(f"part {k}: {len(bad)} inline-asm VMEM instruction(s) read an SGPR a VALU wrote <= 5 instructions earlier")
        for b in bad[:10]: print("   line %d: %s   (s%d written %d instructions before)" % b)
        rc |= 1 if bad else 0
sys.exit(rc)
