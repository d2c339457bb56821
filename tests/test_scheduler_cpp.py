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


def test_cpp_sampler_state_machines_match_the_python_mirrors(tmp_path):
    """include/rwkv_sampler.hpp (host-side state of Nucleus / Typical / Mirostat, sampler/*.rs) against ai00_server_amd/harness.py on
    one scripted scenario: penalty maps after init and after every update, the merged adjustment lists, the parameters handed to
    rwkv_infer_sample, Mirostat's max_surprise trajectory including its 4 tau cap."""
    import numpy as np
    from ai00_server_amd import harness as H
    exe = str(tmp_path / "sampler_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "cpp", "sampler_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    prompt = [5, 9, 5, 3, 9, 9, 120, 5]
    picks = [9, 44, 5, 44, 44, 7, 120, 9]

    def parse(line):
        parts = line.split()[1:]
        return {int(p.split(":")[0]): float(p.split(":")[1]) for p in parts}

    def check_penalty_sampler(s, tag, got_lines):
        s.init(prompt)
        want = [s.adjustments()]
        for t in picks:
            s.update(t)
            want.append(s.adjustments())
        assert len(got_lines) == len(want)
        for ln, w in zip(got_lines, want):
            assert ln.startswith(tag)
            g = parse(ln)
            assert sorted(g) == sorted(int(k) for k in w)
            for k, v in w.items():
                assert abs(g[int(k)] - float(v)) <= 2e-6 * max(1.0, abs(float(v))), (tag, k, g[int(k)], float(v))

    nuc = H.NucleusSampler(presence_penalty=0.4, frequency_penalty=0.25, penalty_decay=0.99, bias={44: 1.5, 3: -2.0})
    check_penalty_sampler(nuc, "nucleus", lines[0:9])
    p = lines[9].split()
    assert p[0] == "params" and [float(x) for x in p[1:5]] == [0.5, 128, 1.0, 0.25] and int(p[5]) == len(nuc.adjustments()) and int(p[6]) == 0
    typ = H.TypicalSampler(tau=0.7)
    check_penalty_sampler(typ, "typical", lines[10:19])
    p = lines[19].split()
    assert [float(x) for x in p[1:5]] == [0.0, 128, 1.0, 0.5] and int(p[6]) == 1 and abs(float(p[7]) - 0.7) < 1e-6
    m = H.MirostatSampler(tau=3.0, rate=0.1)
    traj = [float(m.max_surprise)]
    for x in (2.5, 7.25, 0.125, 3.0, 12.0, 1.0, 0.5, 0.25, 0.0, 0.0, 0.0, 0.0):
        m.update(x)
        traj.append(float(m.max_surprise))
    got = [float(x) for x in lines[20].split()[1:]]
    assert np.allclose(got, traj, rtol=1e-6, atol=0) and max(got) <= 12.0 + 1e-6
    p = lines[21].split()
    assert int(p[6]) == 2 and abs(float(p[7]) - traj[-1]) <= 1e-6 * traj[-1] and int(p[5]) == 0
