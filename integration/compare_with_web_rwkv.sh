#!/bin/bash
# The one missing pin of this repository, as ONE command for whoever has what this build image lacks: Rust, a Vulkan adapter for
# web-rwkv, an MI355X for librwkv_hip, and a real RWKV checkpoint (`.st`).  It builds the stock ai00_server (web-rwkv 0.10.18 over
# wgpu / Vulkan) and the patched one (integration/ai00-core.patch: the same ai00-core over this repository's C ABI), serves the same
# model from both with greedy sampling, and diffs
#   (1) the token stream of /api/oai/completions for a set of prompts         -> north_star: "bit-exact for token ids",
#   (2) the state slab of /api/oai/states for the same prompts                -> north_star: "within 1e-3 on logits / embeddings"
#       (the slab's WKV rows are what the /embeddings route returns, docs/doc-api/openai.md:376-437).
# Both servers run with `precision = "Fp32"` unless PRECISION=Fp16 is given (reload.rs:89-94): Fp32 is the mode in which both sides
# are exact to fp32 round-off; in Fp16 expect the operand-rounding noise documented in DESIGN.md 1.
#
#   AI00=/path/to/ai00_server MODEL=/path/to/model.st [QUANT=0] [QUANT_TYPE=Int8] [PRECISION=Fp32] [N_TOKENS=256] integration/compare_with_web_rwkv.sh
#
# Exit status 0: every token stream identical and every slab within TOL (default 1e-3 absolute).  Nothing here runs in this repository's CI
# (no rustc, no Vulkan ICD, no weights in the build image: SURVEY.md 8c); it is the procedure of INTEGRATION.md 3 made executable.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
: "${AI00:?path to an ai00_server checkout (Ai00-X/ai00_server)}"
: "${MODEL:?path to a .st checkpoint}"
QUANT=${QUANT:-0}; QUANT_TYPE=${QUANT_TYPE:-Int8}; PRECISION=${PRECISION:-Fp32}; N_TOKENS=${N_TOKENS:-256}; TOL=${TOL:-1e-3}
WORK=$(mktemp -d)
trap 'kill $(jobs -p) 2>/dev/null || true; rm -rf "$WORK"' EXIT
command -v cargo >/dev/null || { echo "cargo not found: this script needs a Rust toolchain" >&2; exit 2; }

python3 -m ai00_server_amd.build >/dev/null
export RWKV_HIP_LIB_DIR="$HERE/../ai00_server_amd" LD_LIBRARY_PATH="$HERE/../ai00_server_amd:${LD_LIBRARY_PATH:-}"

build_tree () {   # $1 = destination, $2 = "stock" | "hip"
    git -C "$AI00" worktree add --detach "$1" HEAD >/dev/null
    if [ "$2" = hip ]; then
        (cd "$1" && git apply "$HERE/ai00-core.patch" &&
            sed -i "s#path = \"../../../integration/rwkv-hip\"#path = \"$HERE/rwkv-hip\"#" crates/ai00-core/Cargo.toml crates/ai00-server/Cargo.toml)
    fi
    (cd "$1" && cargo build --release -p ai00-server)
}
config () {       # $1 = tree, $2 = port
    mkdir -p "$1/assets/models"
    ln -sf "$MODEL" "$1/assets/models/$(basename "$MODEL")"
    sed -e "s#^name = .*#name = \"$(basename "$MODEL")\"#" -e "s#^precision = .*#precision = \"$PRECISION\"#" \
        -e "s#^quant = .*#quant = $QUANT#" -e "s#^quant_type = .*#quant_type = \"$QUANT_TYPE\"#" \
        -e "s#^max_batch = .*#max_batch = 8#" -e "s#^port = .*#port = $2#" "$1/assets/configs/Config.toml" > "$1/compare.toml"
}
serve () {        # $1 = tree, $2 = port
    (cd "$1" && ./target/release/ai00-server --config compare.toml > "$WORK/server_$2.log" 2>&1) &
    for _ in $(seq 1 600); do curl -sf "http://127.0.0.1:$2/api/oai/models" >/dev/null && return 0; sleep 1; done
    echo "server on port $2 did not come up: $WORK/server_$2.log" >&2; tail -20 "$WORK/server_$2.log" >&2; exit 3
}

build_tree "$WORK/stock" stock
build_tree "$WORK/hip" hip
config "$WORK/stock" 65531; config "$WORK/hip" 65532
serve "$WORK/stock" 65531; serve "$WORK/hip" 65532

python3 - "$N_TOKENS" "$TOL" <<'PY'
import json, sys, urllib.request
import numpy as np
n_tokens, tol = int(sys.argv[1]), float(sys.argv[2])
PROMPTS = ["The Eiffel tower is located in the city of", "def fibonacci(n):\n", "User: What is the capital of Norway?\n\nAssistant:",
           "In a shocking finding, scientists discovered a herd of dragons living in a remote valley.", "1, 1, 2, 3, 5, 8,",
           "春眠不觉晓,", "<s>" * 3, "A" * 700]   # the last one crosses token_chunk_size = 256: prefill in several steps

def post(port, route, body):
    req = urllib.request.Request(f"http://127.0.0.1:{port}/api/oai/{route}", json.dumps(body).encode(), {"Content-Type": "application/json"})
    return json.loads(urllib.request.urlopen(req, timeout=600).read())

bad = 0
for i, p in enumerate(PROMPTS):
    # greedy: NucleusSampler with top_k = 1 keeps the arg-max only (nucleus.rs:77-89); penalties off so that sampling is a pure function of the logits
    body = {"prompt": [p], "max_tokens": n_tokens, "stop": [], "sampler": {"type": "Nucleus", "top_k": 1, "top_p": 0.0, "temperature": 1.0,
            "presence_penalty": 0.0, "frequency_penalty": 0.0, "penalty_decay": 1.0}}
    a, b = post(65531, "completions", body), post(65532, "completions", body)
    ta, tb = a["choices"][0]["text"], b["choices"][0]["text"]
    same = ta == tb
    first = next((k for k, (x, y) in enumerate(zip(ta, tb)) if x != y), min(len(ta), len(tb)))
    print(f"prompt {i}: completions {'IDENTICAL' if same else f'DIFFER at character {first}'} ({len(ta)} vs {len(tb)} characters)")
    bad += 0 if same else 1
    sa, sb = post(65531, "states", {"input": [p]}), post(65532, "states", {"input": [p]})
    xa, xb = np.asarray(sa["data"][0]["data"], np.float64), np.asarray(sb["data"][0]["data"], np.float64)   # StateData { data, shape: [C, N + 2, L, 1] } (api/oai/state.rs:43-49)
    if xa.shape != xb.shape or sa["data"][0]["shape"] != sb["data"][0]["shape"]:
        print(f"prompt {i}: state shapes differ {sa['data'][0]['shape']} vs {sb['data'][0]['shape']}")
        bad += 1
        continue
    err = float(np.abs(xa - xb).max())
    print(f"prompt {i}: state slab max-abs difference {err:.3e} (|ref|inf {np.abs(xa).max():.3f}, tolerance {tol:g})")
    bad += 0 if err <= tol else 1
print("RESULT:", "web-rwkv and librwkv_hip agree" if bad == 0 else f"{bad} mismatches")
sys.exit(0 if bad == 0 else 1)
PY
