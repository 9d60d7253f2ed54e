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
    # The library is never compiled implicitly by the product (chromegcn_amd/_build.py).  The test session builds
    # it here -- explicitly, once, before anything has touched the GPU -- when the tree's sources are newer than
    # the in-tree .so (content hash, not mtime: the prebuilt library that travels to the GPU box stays valid).
    if os.environ.get("PYTEST_XDIST_WORKER") is None:
        from chromegcn_amd import _build
        if _build.is_stale() and _build.hipcc_path() is not None:
            _build.build_library()
        # the test-only variants (tests/test_gpu_ring_stress.py: slowed ring teams; never the product library): only a session
        # that selects gpu tests needs them (the CPU tier compiled two whole libraries it never opened)
        expr = getattr(config.option, "markexpr", "") or ""
        wants_gpu = "gpu" in expr and "not gpu" not in expr
        if wants_gpu and _build.hipcc_path() is not None and any(_build.variant_is_stale(v) for v in _build.TEST_VARIANTS):
            try:
                _build.build_test_variants()
            except Exception as e:   # the ring-stress tests skip without their variants; everything else is unaffected
                print("chromegcn_amd: test variants not built: %s" % e)


# Order of the -m gpu tier (the driver runs it with -x, so whatever comes first gates everything behind it): the tests
# that compare the HIP path with the oracle / the reference-recorded vectors first, the remaining single-process GPU tests
# next, and every test that starts other processes (launchers, rendezvous, sockets: infrastructure, not arithmetic) LAST,
# so that a hiccup there can never again cost the parity evidence (GPUTEST_r04: 1 launcher test failed, 201 parity tests
# never ran).
_GPU_ORDER = ["test_gpu_parity", "test_gpu_fullsize_oracle", "test_gpu_epoch_oracle", "test_gpu_modules", "test_gpu_loop", "test_gpu_head", "test_gpu_graph",
              "test_gpu_stat_acc", "test_gpu_runner", "test_gpu_sgd_fuse", "test_gpu_dropout_sgd", "test_gpu_sliced_routes", "test_gpu_band",
              "test_gpu_ring_stress", "test_gpu_torch_ops", "test_gpu_metrics", "test_gpu_handoff", "test_gpu_saliency",
              "test_gpu_cabi_errors", "test_gpu_fullsize"]
_SPAWNING_LAST = ["test_gpu_e2e", "test_gpu_rccl_one_rank", "test_gpu_two_rank", "test_bench_launcher"]


def _tier(item):
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if mod in _SPAWNING_LAST:
        return (2, _SPAWNING_LAST.index(mod))
    if mod in _GPU_ORDER:
        return (0, _GPU_ORDER.index(mod))
    return (1, 0)


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) when no device is visible, so a plain
    `pytest tests` on a CPU box stays green; `-m gpu` on the GPU box runs them for real -- parity first, process-spawning
    tests last (stable sort: the order inside a file is kept)."""
    items.sort(key=_tier)
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have = False
    if have:
        # The host oracle's fp32 sums depend on torch's thread count (a hub row of 10^4 neighbours, a cancelling column sum):
        # pin it for the GPU tier, so that "HIP against the oracle" is the same comparison on every box whatever its core
        # count (round 6: one gate-bias gradient sat at 2e-4 of its own accidentally tiny value on one lease and not on others)
        torch.set_num_threads(min(8, os.cpu_count() or 1))
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load
