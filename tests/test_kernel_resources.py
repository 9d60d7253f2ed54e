"""No kernel of the shipped library may spill a register or use scratch memory (k_layer_dense<S,256> shipped with
20 spilled VGPRs and 84 B of scratch per lane, and nothing in the tests noticed).  chromegcn_amd/_build.py compiles with
-Rpass-analysis=kernel-resource-usage and writes what the compiler reports for every kernel next to the library
(libchromegcn_hip.so.resources.json, keyed by the hash of the sources it was built from); this test reads it -- building the
library first when it is stale (tests/conftest.py does at session start) -- and fails on any offender."""
import pytest

from chromegcn_amd import _build


def _resources():
    res = _build.kernel_resources()
    if res is None and _build.hipcc_path() is not None:
        _build.build_library(force=True)
        res = _build.kernel_resources()
    return res


def test_no_kernel_spills_or_uses_scratch():
    res = _resources()
    if res is None:
        pytest.skip("no hipcc and no resource report for the current sources")
    assert len(res) >= 300, "expected the library's ~370 kernel instantiations, found %d" % len(res)
    # (SGPR "spills" of the kernels with two dozen arguments live in lanes of a VGPR -- v_writelane / v_readlane, no memory:
    # scratch == 0 is what says so -- and are not failures; VGPR spills and scratch are)
    bad = [k for k in res if k.get("spill", 0) or k.get("scratch", 0)]
    assert not bad, "kernels with spills / scratch:\n" + "\n".join(
        "  %s: %s VGPRs, %s spilled, %s SGPRs spilled, %s B scratch per lane" % (k["name"], k.get("vgprs"), k.get("spill"), k.get("sgpr_spill"), k.get("scratch"))
        for k in bad)


def test_the_report_covers_the_kernels_of_the_hot_path():
    res = _resources()
    if res is None:
        pytest.skip("no hipcc and no resource report for the current sources")
    names = " ".join(k["name"] for k in res)
    for frag in ("k_layer_dense256", "k_layer_dense", "k_layer_fwd", "k_aggregate_sliced", "k_bwd_sliced", "k_bwd_rowlocal_ring",
                 "k_bwd_rowlocal256s", "k_head_fused_rs", "k_head_fused", "k_band_aggregate", "k_sddmm", "k_rs_pass"):
        assert frag in names, frag
    occ = {k["name"]: k.get("occupancy") for k in res}
    assert all(v is not None and int(v) >= 1 for v in occ.values())
