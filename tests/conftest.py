import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture
def golden():
    return load_golden


def hashed_uniform(shape, salt, lo=-1.0, hi=1.0):
    """Same exact-integer-hash generator as tests/golden/gen_golden.py (inputs too big to store)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64) + np.uint64(salt) * np.uint64(0x9E3779B1)
    i = (i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    i ^= i >> np.uint64(15)
    i = (i * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    i ^= i >> np.uint64(13)
    u = (i >> np.uint64(8)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


# GPU tests that run the layers (GAT heads / SpGAT / SpKBGAT / GraphConvolution) are run once per GEMM family: the layer
# picks its family by problem size ("auto": exact-fp32 MFMA GEMMs below ~6 GFLOP per product, split-precision above), so on
# the small golden / oracle shapes each family has to be forced to get odd-shape and tile-tail coverage of its own.
GEMM_FAMILIES = ("2", "1", "0")     # values of recon_amd.gat_layers._GEMM_BX3 (shared by gcn_layers): f16 x 2, bf16 x 3, exact fp32
_NO_FAMILY = ("test_graph_build", "test_hub_tables", "test_small_mm", "test_thin_weight", "test_index_range", "test_gcn_beyond", "test_propagation_beyond", "test_batches_above", "test_fuzz", "test_sgemm", "test_spmm", "test_full_size_cfg2_split_precision_vs_fp32_gemm",
              "test_block_adjacency", "test_propagation", "test_start_entity", "test_gpgnn", "test_phased_backward",
              "test_sampler_golden", "test_sampler_vs", "test_sampler_rejects", "test_gcn_bf16", "test_cfg3b", "test_cfg3a_bf16",
              "test_propagate_blocks", "test_gcn_stack", "test_wide_backward", "test_gather_rows_pair")       # propagation / bf16 stack kernels: no GEMM family involved


def pytest_generate_tests(metafunc):
    if "gemm_family" not in metafunc.fixturenames:
        return
    is_gpu = metafunc.definition.get_closest_marker("gpu") is not None
    name = metafunc.definition.originalname
    if is_gpu and not name.startswith(_NO_FAMILY):
        metafunc.parametrize("gemm_family", GEMM_FAMILIES, indirect=True, ids=["gemm_hx2", "gemm_bx3", "gemm_f32"])


@pytest.fixture(autouse=True)
def gemm_family(request, monkeypatch):
    fam = getattr(request, "param", None)
    if fam is None and request.node.get_closest_marker("gpu") is not None:
        fam = "2"                    # un-parametrized GPU tests: the benchmark's family
    if fam is not None:
        from recon_amd import gat_layers
        monkeypatch.setattr(gat_layers, "_GEMM_BX3", fam)
    yield fam


@pytest.fixture
def recon_config():
    """`recon_config("RECON_PROP_FWD", "w")`: one run-time switch of the library (csrc/config.hip) for the rest of the test, restored afterwards."""
    from recon_amd import _lib
    prev = {}

    def set_(name, value):
        before = _lib.config_set(name, value)
        prev.setdefault(name, before)
    yield set_
    for name, value in prev.items():
        _lib.config_set(name, value)
