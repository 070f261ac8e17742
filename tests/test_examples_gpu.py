"""Callers written in C against the reference API only (examples/*.c: the flows of samples/smm/specialized.cpp,
samples/cp2k/cp2k.cpp, samples/utilities/wrap/{dgemm,autobatch}.c, samples/xgemm/kernel.c, samples/spmdm/spmdm.c, samples/pyfr/pyfr_driver_asp_reg.c and samples/blocked_gemm/blocked_gemm.c) are compiled with gcc, linked against libxsmm.so and run on the GPU
box: the drop-in claim, end to end. They check themselves against plain loops and return 0."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["smm_caller.c", "spmdm_caller.c", "pyfr_caller.c", "blocked_caller.c", "smm_functor.cpp", "cp2k_caller.cpp", "blas_wrap_caller.c", "autobatch_caller.c", "lowp_caller.c"])
def test_c_caller_runs_on_the_gpu(xs, torch_gpu, tmp_path, name):
    libdir = os.path.dirname(xs.LIB_PATH)
    exe = tmp_path / name.split(".")[0]
    cc = ["gcc", "-std=c99"] if name.endswith(".c") else ["g++", "-std=c++11"]
    wrap = ["-Wl,--wrap=dgemm_,--wrap=sgemm_"] if name.startswith("blas_wrap") else []  # the reference's call wrapper link line
    subprocess.run(cc + ["-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name),
                         "-o", str(exe)] + wrap + [ "-L", libdir, "-lxsmm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    assert name.split(".")[0] in res.stdout
