import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Tests name the kernel they expect right after the first call of a shape: compile in the calling thread (the product's default
# is the helper thread, with the pre-compiled kernel serving meanwhile -- tests/test_jit.py::test_jit_never_blocks_a_batch_call)
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")
# The product specialises a descriptor from 16 items on (the compiler runs on a helper thread; the pre-compiled generic kernel
# serves meanwhile, with the same bits). The tests compile in the calling thread (above), and hundreds of small batches of random
# shapes would each wait for hiprtc: here batches below 1024 items stay on the pre-compiled kernels -- which also keeps those
# kernels covered -- unless a test asks for the specialised ones itself (LIBXSMM_AMD_JIT_MINBATCH=1 inside the test).
os.environ.setdefault("LIBXSMM_AMD_JIT_MINBATCH", "1024")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def xs():
    """the product's ctypes binding (libxsmm-1_amd/__init__.py); builds lib/libxsmm.so if it is missing"""
    mod = importlib.import_module("libxsmm-1_amd")
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle binding (test infrastructure)"""
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def torch_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return torch
