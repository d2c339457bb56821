"""CPU: the C++ scheduling core (include/rwkv_scheduler.hpp — slot choice, prefix cache, continuous batching; mirror of
run.rs:289-331, 441-662, 1113-1157) against a fake engine.  Compiled with g++ here, no GPU and no HIP involved."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_scheduler_core(built_lib, tmp_path):
    exe = str(tmp_path / "scheduler_test")
    pkg = os.path.join(ROOT, "ai00_server_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror",
                           os.path.join(ROOT, "tests", "cpp", "scheduler_test.cpp"), "-o", exe,
                           "-L" + pkg, "-lrwkv_hip", "-Wl,-rpath," + pkg])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "scheduler_test: ok" in out.stdout, out.stdout + out.stderr


def test_cpp_replica_router(built_lib, tmp_path):
    """include/rwkv_router.hpp over 8 fake engines: prefix affinity, least-busy placement, full replicas skipped, one driving thread
    per replica; answers equal the single-engine answers (replicas only, no collective)."""
    exe = str(tmp_path / "router_test")
    pkg = os.path.join(ROOT, "ai00_server_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread",
                           os.path.join(ROOT, "tests", "cpp", "router_test.cpp"), "-o", exe,
                           "-L" + pkg, "-lrwkv_hip", "-Wl,-rpath," + pkg])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "router_test: ok" in out.stdout, out.stdout + out.stderr
