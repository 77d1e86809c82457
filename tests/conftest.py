"""pytest configuration: registers the ``gpu`` marker and puts the product package
(``approximategps.jl_amd/`` holds the importable ``approxgp`` package) and the oracle on sys.path."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for multi-process GPU tests (read at HSA initialisation)

# torch bundles its own copy of the HIP runtime; in a process that uses both torch.cuda and libsvgp_mi355x.so it has to be
# loaded first (the library then binds to the already loaded runtime; the other order leaves torch without GPUs)
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "approximategps.jl_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
