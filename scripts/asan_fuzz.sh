#!/bin/bash
# Dev tool (CPU): host code of the library (rwkv_engine.cpp, tokenizer.cpp) rebuilt with AddressSanitizer + UBSan and linked with the
# existing kernel objects into /tmp/asan/librwkv_hip.so, then tests/cpp/fuzz_cpu_entry_points.cpp (mutated safetensors headers,
# vocabularies, byte strings, token ids, chunk plans) and the C++ host tests run against it.  No GPU needed.
#   bash scripts/asan_fuzz.sh [iterations]      (100000 iterations take about two minutes)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-100000}
O=/tmp/asan
mkdir -p $O
python - <<PY
import sys
sys.path.insert(0, "$R")
from oracle import rwkv_ref as R
for name in ("v5-tiny", "v6-tiny", "v7-tiny"):
    open("$O/%s.st" % name, "wb").write(R.st_serialize(R.synth_named(name)))
PY
python -c "import sys; sys.path.insert(0, '$R'); from ai00_server_amd import build; build.build(verbose=False)"
for f in rwkv_engine tokenizer; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -x hip -fsanitize=address,undefined -fno-omit-frame-pointer -c $R/ai00_server_amd/csrc/$f.cpp -o $O/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $O/librwkv_hip.so $R/ai00_server_amd/csrc/rwkv_kernels.p*.o $O/rwkv_engine.o $O/tokenizer.o
CXX=/opt/rocm/lib/llvm/bin/clang++
RT=$(dirname $($CXX -print-file-name=libclang_rt.asan-x86_64.so))
for t in fuzz_cpu_entry_points scheduler_test router_test sampler_test; do
  $CXX -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -shared-libsan -pthread $R/tests/cpp/$t.cpp -o $O/$t -L$O -lrwkv_hip -Wl,-rpath,$O -Wl,-rpath,$RT
done
$O/fuzz_cpu_entry_points $N $R/tests/golden/vocab_sample.json $O/v5-tiny.st $O/v6-tiny.st $O/v7-tiny.st
$O/scheduler_test; $O/router_test; $O/sampler_test > /dev/null && echo "sampler_test: ok"
# the router once more under ThreadSanitizer (host threads only; against the product build of the library)
g++ -O1 -g -std=c++17 -fsanitize=thread -pthread $R/tests/cpp/router_test.cpp -o $O/router_tsan -L$R/ai00_server_amd -lrwkv_hip -Wl,-rpath,$R/ai00_server_amd
for i in 1 2 3; do $O/router_tsan; done
