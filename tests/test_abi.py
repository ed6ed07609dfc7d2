"""The C-ABI library loads on a machine without a GPU and exports every symbol include/er_hip.h declares
(no compute calls here)."""
import os
import re

import pytest

from elasticreconstruction_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "er_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(er_[a-z0-9_]+)\s*\(", txt)))


def test_header_matches_binding_list():
    assert header_symbols() == sorted(_ffi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = _ffi.lib()
    missing = [s for s in header_symbols() if not hasattr(L, s)]
    assert not missing, "liber_hip.so lacks %s" % missing
    assert L.er_abi_version() >= 1


def test_no_cpu_fallback():
    """Without a HIP device every constructor must fail loudly (never compute on the host)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from elasticreconstruction_amd.tsdf import TSDFVolume
    with pytest.raises(_ffi.ErError):
        TSDFVolume()


def test_host_programs_and_every_constructor_refuse_to_run_without_a_gpu(tmp_path):
    """The three drop-in programs, the clouds and the FragmentOptimizer handle: an error message and a non-zero exit /
    an exception, never a computation on the host."""
    import subprocess
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from elasticreconstruction_amd import formats
    from elasticreconstruction_amd.icp import Cloud
    from elasticreconstruction_amd.fopt import FragmentOptimizer
    x = (np.random.default_rng(0).random((100, 3)) * 2 + 0.2).astype(np.float32)
    n = np.tile(np.array([0, 0, 1], np.float32), (100, 1))
    with pytest.raises(_ffi.ErError):
        Cloud(x, n, 0.05)
    with pytest.raises(_ffi.ErError):
        FragmentOptimizer(2, 8, 3.0)
    d = str(tmp_path)
    eye = np.eye(4)
    formats.save_log(os.path.join(d, "traj.log"), [formats.FramedTransformation(0, 0, 1, eye), formats.FramedTransformation(1, 1, 2, eye)])
    formats.save_log(os.path.join(d, "reg.log"), [formats.FramedTransformation(0, 1, 2, eye)])
    np.zeros((2, 480 * 640), np.uint16).tofile(os.path.join(d, "frames.raw"))
    for i in range(2):
        formats.save_pcd_xyzn(os.path.join(d, "cloud_bin_%d.pcd" % i), x, n)
    bin_ = os.path.join(ROOT, "elasticreconstruction_amd", "bin")
    for cmd in (["Integrate", "--ref_traj", "traj.log", "-oni", "frames.raw"],
                ["BuildCorrespondence", "--reg_traj", "reg.log", "--registration"],
                ["FragmentOptimizer", "--rigid", "--num", "2", "--registration", "reg.log", "--rgbdslam", "traj.log", "--interval", "1", "--dir", "./"]):
        r = subprocess.run([os.path.join(bin_, cmd[0])] + cmd[1:], cwd=d, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "no HIP device" in r.stderr, (cmd[0], r.returncode, r.stderr)
        assert not os.path.exists(os.path.join(d, "world.pcd")) and not os.path.exists(os.path.join(d, "reg_output.log")) \
            and not os.path.exists(os.path.join(d, "pose.log"))


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/er_hip.h compiles as strict C99 and a C program linked against liber_hip.so runs: the boundary is a C ABI,
    not a C++ or Python one (examples/abi_probe.c)."""
    import subprocess
    lib_dir = os.path.join(ROOT, "elasticreconstruction_amd")
    exe = str(tmp_path / "abi_probe")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "abi_probe.c"), "-L", lib_dir, "-ler_hip", "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi " in r.stdout and ("no HIP device" in r.stdout or "empty volume: 0 units" in r.stdout), r.stdout


def test_product_never_imports_oracle():
    """The product path must not reference oracle/ (only tests/, smoke() and bench.py's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "elasticreconstruction_amd")
    for dp, _, files in os.walk(pkg):
        if "_build" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                src = open(os.path.join(dp, f), errors="replace").read()
                code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith(("//", "#", "*", '"""')))
                assert "pyoracle" not in code and "libtsdf_oracle" not in code and "libicp_oracle" not in code \
                    and "libref_tsdf" not in code, "%s references the oracle" % f
