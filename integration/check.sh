#!/bin/bash
# One-command check for a machine that has Rust (this repository's build image has none):
#   1. builds librwkv_hip.so, 2. type-checks rwkv-hip-sys and rwkv-hip against it,
#   3. (optional, AI00=<path to an ai00_server checkout>) applies ai00-core.patch and runs `cargo check -p ai00-core -p ai00-server`.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
python3 -m ai00_server_amd.build >/dev/null
export RWKV_HIP_LIB_DIR="$HERE/../ai00_server_amd"
(cd "$HERE/rwkv-hip-sys" && cargo check)
(cd "$HERE/rwkv-hip" && cargo check)
if [ -n "${AI00:-}" ]; then
    (cd "$AI00" && git apply --check "$HERE/ai00-core.patch" && git apply "$HERE/ai00-core.patch" &&
        sed -i "s#path = \"../../../integration/rwkv-hip\"#path = \"$HERE/rwkv-hip\"#" crates/ai00-core/Cargo.toml crates/ai00-server/Cargo.toml &&
        cargo check -p ai00-core -p ai00-server)
fi
echo "integration check ok"
