"""Device metrics (cgcn_multilabel_metrics) against the per-label values recorded from the reference's
utils/metrics.py helpers (G5) and against the oracle (scikit-learn, the reference's dependency) on larger
random inputs.  Curve arithmetic is fp64 on the device; results are returned as fp32."""
import numpy as np
import pytest
import torch

from chromegcn_amd import metrics as M
from oracle import chromegcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
PAIRS = [("ref_auroc", "auroc"), ("ref_aupr", "aupr"), ("ref_fdr", "recall_at_fdr"), ("ref_ap", "average_precision")]


def test_metrics_match_reference_golden(golden):
    z = golden("g5_metrics.npz")
    m = M.multilabel_metrics(torch.from_numpy(z["preds"]).to(DEV), torch.from_numpy(z["targets"]).to(DEV))
    for k_ref, k in PAIRS:
        np.testing.assert_allclose(m[k].cpu().numpy(), z[k_ref], rtol=2e-6, atol=2e-7, equal_nan=True, err_msg=k)


@pytest.mark.parametrize("n,C", [(1, 3), (65, 2), (5000, 103), (40000, 17)])
def test_metrics_match_oracle(n, C):
    rng = np.random.RandomState(n + C)
    tg = (rng.rand(n, C) < rng.rand(C) * 0.5).astype(np.float32)
    pr = (rng.rand(n, C) * 0.7 + 0.3 * tg * rng.rand(n, C)).astype(np.float32)
    pr[:, 0] = np.round(pr[:, 0], 2)   # heavy ties: runs of equal scores cross the 4096-element chunk boundaries
    if C > 1:
        pr[:, 1] = 0.5                 # one run spanning every chunk
    if C > 2:
        pr[:, 2] = np.round(pr[:, 2], 1)
    want = O.multilabel_metrics_np(tg.astype(np.float64), pr)
    got = M.multilabel_metrics(torch.from_numpy(pr).to(DEV), torch.from_numpy(tg).to(DEV))
    for k in ("auroc", "aupr", "recall_at_fdr", "average_precision"):
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k], rtol=3e-6, atol=3e-7, equal_nan=True, err_msg=k)


def test_metrics_order_signed_scores_and_signed_zero():
    """Scores need not be probabilities: logits (both signs), -0.0 / +0.0 (one threshold for sklearn) and subnormals
    must rank exactly as numbers do (the sort key is an order-preserving integer image of the float)."""
    rng = np.random.RandomState(5)
    n, C = 3000, 5
    tg = (rng.rand(n, C) < 0.3).astype(np.float32)
    pr = (rng.randn(n, C) * 3 + tg).astype(np.float32)
    pr[::7, 0] = 0.0
    pr[3::7, 0] = -0.0
    pr[::5, 1] = np.float32(1e-42) * rng.randint(-3, 4, size=pr[::5, 1].shape).astype(np.float32)  # subnormals of both signs
    pr[:, 2] = -np.abs(pr[:, 2])       # all negative
    want = O.multilabel_metrics_np(tg.astype(np.float64), pr)
    got = M.multilabel_metrics(torch.from_numpy(pr).to(DEV), torch.from_numpy(tg).to(DEV))
    for k in ("auroc", "aupr", "recall_at_fdr", "average_precision"):
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k], rtol=3e-6, atol=3e-7, equal_nan=True, err_msg=k)


def test_compute_metrics_keys_and_aggregation(golden):
    z = golden("g5_metrics.npz")
    out = M.compute_metrics(torch.from_numpy(z["preds"]), torch.from_numpy(z["targets"]), 1.25, None, 0.5)
    for k in ["mAP", "meanAUC", "medianAUC", "allAUC", "allFDR", "meanAUPR", "medianAUPR", "allAUPR", "meanFDR",
              "medianFDR", "loss", "time"]:  # utils/evals.py:107-118
        assert k in out
    assert abs(out["meanAUC"] - np.nanmean(z["ref_auroc"])) < 1e-6
    assert abs(out["meanAUPR"] - np.mean(z["ref_aupr"])) < 1e-6
    assert abs(out["meanFDR"] - np.mean(z["ref_fdr"])) < 1e-6
    assert abs(out["mAP"] - np.mean(z["ref_ap"])) < 1e-6
    assert out["loss"] == 1.25 and out["time"] == 0.5
    with pytest.raises(RuntimeError):
        M.multilabel_metrics(torch.zeros(4, 2), torch.zeros(4, 2))


@pytest.mark.parametrize("n,C", [(1, 1), (2, 3), (63, 2), (4096, 3), (4097, 5), (70000, 9), (12289, 103), (37, 2600)])
def test_probability_path_equals_general_path_bit_for_bit(n, C):
    """cgcn_multilabel_metrics_nonneg (32-bit keys, the library's own segmented radix sort: partial last tiles, one-tile and
    many-tile labels, the flat pack and -- thousands of labels -- the tile pack) against cgcn_multilabel_metrics (64-bit keys,
    one device-wide sort) on probabilities: same curve kernels behind both, so the results must be the same bits."""
    g = torch.Generator().manual_seed(n * 131 + C)
    tg = (torch.rand(n, C, generator=g) < 0.2).float()
    pr = torch.sigmoid(torch.randn(n, C, generator=g) * 2 + tg)
    pr[:, 0] = (pr[:, 0] * 50).round() / 50          # heavy ties
    if C > 1:
        pr[:, 1] = 0.5 + pr[:, 1] * 1e-4             # every key in a handful of top digits (skew)
    if C > 2:
        pr[::3, 2] = 0.0                             # exact zeros, and -0.0 folded onto them
        pr[1::3, 2] = -0.0
    pr, tg = pr.to(DEV), tg.to(DEV)
    fast = M._metrics_raw(pr, tg, 0.5, nonneg=True).cpu()
    slow = M._metrics_raw(pr, tg, 0.5, nonneg=False).cpu()
    assert int(fast[4 * C:].view(torch.int32).item()) == 0
    a, b = fast[:4 * C].numpy(), slow[:4 * C].numpy()
    assert np.array_equal(a, b, equal_nan=True), np.abs(np.nan_to_num(a) - np.nan_to_num(b)).max()


def test_probability_path_on_views_that_are_not_16_byte_aligned():
    g = torch.Generator().manual_seed(3)
    n, C = 5000, 7
    buf_p = torch.rand(n * C + 3, generator=g).to(DEV)
    buf_t = (torch.rand(n * C + 3, generator=g) < 0.3).float().to(DEV)
    pr, tg = buf_p[1:1 + n * C].view(n, C), buf_t[3:3 + n * C].view(n, C)   # storage offsets 4 and 12 bytes
    assert pr.data_ptr() % 16 != 0 and pr.is_contiguous()
    fast = M._metrics_raw(pr, tg, 0.5, nonneg=True).cpu()[:4 * C].numpy()
    want = O.multilabel_metrics_np(tg.cpu().numpy().astype(np.float64), pr.cpu().numpy())
    for j, k in enumerate(("auroc", "aupr", "recall_at_fdr", "average_precision")):
        np.testing.assert_allclose(fast[j * C:(j + 1) * C], want[k], rtol=3e-6, atol=3e-7, equal_nan=True, err_msg=k)


def test_probability_path_reports_scores_that_are_not_probabilities():
    pr = torch.rand(300, 4)
    pr[17, 2] = -0.25
    tg = (torch.rand(300, 4) < 0.5).float()
    flat = M._metrics_raw(pr.to(DEV), tg.to(DEV), 0.5, nonneg=True)
    assert int(flat[16:].view(torch.int32).item()) != 0
    pr[17, 2] = float("nan")
    flat = M._metrics_raw(pr.to(DEV), tg.to(DEV), 0.5, nonneg=True)
    assert int(flat[16:].view(torch.int32).item()) != 0


def test_metrics_of_an_empty_split_are_nan_on_both_paths():
    pr, tg = torch.zeros(0, 5, device=DEV), torch.zeros(0, 5, device=DEV)
    for nonneg in (True, False):
        flat = M._metrics_raw(pr, tg, 0.5, nonneg=nonneg).cpu()
        assert torch.isnan(flat[:20]).all()
        if nonneg:
            assert int(flat[20:].view(torch.int32).item()) == 0


def test_recall_at_fdr_when_precision_is_exactly_the_cutoff():
    """Positives and negatives alternating down the ranking: precision is EXACTLY 1/2 at every even depth (tp = fp), so
    1 - precision <= 0.5 holds there with equality in the reference's float64 quotient (utils/metrics.py:153-154) -- at
    depths such as 14, 22, 26, 28 a product with a reciprocal lands one ulp below 1/2 and would miss the deepest such
    point.  Every label stops alternating at another depth; scores are distinct."""
    C, n = 40, 400
    tg = np.zeros((n, C), dtype=np.float32)
    for c in range(C):
        k = 1 + c                                   # c-th label alternates P N for 2 (c + 1) elements, then only negatives
        tg[0:2 * k:2, c] = 1.0
    pr = np.repeat(np.linspace(0.99, 0.01, n, dtype=np.float32)[:, None], C, axis=1)
    want = O.multilabel_metrics_np(tg.astype(np.float64), pr)
    got = M.multilabel_metrics(torch.from_numpy(pr).to(DEV), torch.from_numpy(tg).to(DEV))
    np.testing.assert_allclose(got["recall_at_fdr"].cpu().numpy(), want["recall_at_fdr"], rtol=1e-6, atol=0, err_msg="recall_at_fdr")
    assert (want["recall_at_fdr"] == 1.0).all()     # the deepest point with tp = fp holds every positive
    for k in ("auroc", "aupr", "average_precision"):
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k], rtol=3e-6, atol=3e-7, equal_nan=True, err_msg=k)
