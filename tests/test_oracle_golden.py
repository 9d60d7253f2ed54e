"""Pins oracle/ to the golden vectors recorded from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from oracle import chromegcn_oracle as O
from helpers import coo_to_csr, csr_from, state_from

torch.set_num_threads(1)


# ------------------------------------------------------------------ G1: process_graph
def test_g1_process_graph_matches_reference(golden):
    z = golden("g1_process_graph.npz")
    for name in z["cases"]:
        a_in = csr_from(z, "%s_in" % name)
        n = a_in.shape[0]
        for adj_type in ["hic", "constant", "both", "none"]:
            key = "%s_%s" % (name, adj_type)
            if key + "_row" not in z.files:
                continue
            ref_row, ref_col, ref_val = z[key + "_row"], z[key + "_col"], z[key + "_val"]
            t = O.process_graph(adj_type, {"c": a_in}, n, "c")
            assert t.dtype == torch.float32 and t._indices().dtype == torch.int64
            got = sp.coo_matrix((t._values().numpy(), (t._indices()[0].numpy(), t._indices()[1].numpy())), shape=(n, n)).tocsr()
            ref = coo_to_csr(z, key, n)
            # no duplicate coordinates on the reference side
            assert len(set(zip(ref_row.tolist(), ref_col.tolist()))) == len(ref_row)
            got.sort_indices(); ref.sort_indices()
            # structure identical up to explicit zeros (the 'hic' branch can keep a stored 0
            # only for negative inputs, which the contract excludes) ...
            assert (got != ref).nnz == 0
            # ... and values bit-identical where both store an entry
            d = (got - ref)
            assert d.nnz == 0 or np.all(d.data == 0)
            np.testing.assert_array_equal(got.toarray(), ref.toarray())


def test_g1_small_n_band_is_clipped():
    # the reference raises for n < 7 (np.ones(negative), utils/util_methods.py:141);
    # the restatement clips the band instead -- documented deviation
    a = O.normalized_adjacency("constant", None, 3).toarray()
    np.testing.assert_allclose(a, np.full((3, 3), 1 / 3), rtol=1e-7)


# ------------------------------------------------------------------ G2: one gated layer
def test_g2_layer_forward_backward(golden):
    z = golden("g2_gated_layer.npz")
    for name in z["cases"]:
        adj_type = name.split("_")[-1]
        a_in = csr_from(z, name + "_in")
        n = a_in.shape[0]
        a = O.normalized_adjacency(adj_type, a_in, n)
        ref_adj = coo_to_csr(z, name + "_adj", n)
        np.testing.assert_array_equal(a.toarray(), ref_adj.toarray())
        X, W, b, wg, cg = (z[name + "_" + k] for k in ["X", "W", "b", "wg", "cg"])
        f = O.layer_forward_np(a, X, W, b, wg, float(cg[0]))
        for k, tol in [("U", 2e-5), ("Z", 2e-5), ("Xn", 2e-5)]:
            np.testing.assert_allclose(f[k], z[name + "_" + k], atol=tol, rtol=tol, err_msg=name + k)
        np.testing.assert_allclose(f["g"], z[name + "_g"].ravel(), atol=2e-5, rtol=2e-5)
        bw = O.layer_backward_np(a, X, W, wg, f["Z"], f["g"], z[name + "_Gup"])
        for k in ["dX", "dW", "db"]:
            np.testing.assert_allclose(bw[k], z[name + "_" + k], atol=3e-5, rtol=3e-5, err_msg=name + k)
        np.testing.assert_allclose(bw["dwg"], z[name + "_dwg"].ravel(), atol=3e-5, rtol=3e-5)
        np.testing.assert_allclose(bw["dcg"], z[name + "_dcg"].ravel()[0], atol=3e-5, rtol=3e-5)


# ------------------------------------------------------------------ G3: model
def _mk(name, z):
    _, d, L = name.split("_")
    d = int(d[1:]); L = int(L[1:])
    init = state_from(z, name + "_init")
    c = init["out.weight"].shape[0]
    m = O.GatedGCNOracle(d, c, 0.0, L)
    m.load_state_dict(init)
    return m, d, L, c


def test_g3_state_dict_keys_match_reference(golden):
    z = golden("g3_model.npz")
    for name in z["cases"]:
        m, d, L, c = _mk(name, z)
        want = set(state_from(z, name + "_init").keys())
        assert set(m.state_dict().keys()) == want


def test_g3_eval_forward(golden):
    z = golden("g3_model.npz")
    for name in z["cases"]:
        m, d, L, c = _mk(name, z)
        a_in = csr_from(z, name + "_in"); n = a_in.shape[0]
        adj = O.process_graph("hic", {"c": a_in}, n, "c")
        m.eval()
        with torch.no_grad():
            _, lf, gates, _ = m(torch.from_numpy(z[name + "_xf"]), adj)
            _, lr, _, _ = m(torch.from_numpy(z[name + "_xr"]), adj)
        np.testing.assert_allclose(lf.numpy(), z[name + "_eval_logits_f"], atol=2e-5, rtol=2e-5)
        np.testing.assert_allclose(lr.numpy(), z[name + "_eval_logits_r"], atol=2e-5, rtol=2e-5)
        np.testing.assert_allclose(gates[0].numpy(), z[name + "_eval_g1"], atol=1e-5, rtol=1e-5)
        if L == 2:
            np.testing.assert_allclose(gates[1].numpy(), z[name + "_eval_g2"], atol=1e-5, rtol=1e-5)


def test_g3_train_two_sgd_steps(golden):
    z = golden("g3_model.npz")
    for name in z["cases"]:
        m, d, L, c = _mk(name, z)
        a_in = csr_from(z, name + "_in"); n = a_in.shape[0]
        adj = O.process_graph("hic", {"c": a_in}, n, "c")
        tgt = torch.from_numpy(z[name + "_tgt"])
        opt = O.make_sgd(m, 0.25)
        m.train()
        xf = torch.from_numpy(z[name + "_xf"]).requires_grad_(True)
        xr = torch.from_numpy(z[name + "_xr"]).requires_grad_(True)
        opt.zero_grad()
        _, pf, _, _ = m(xf, adj); _, pr, _, _ = m(xr, adj)
        loss = F.binary_cross_entropy_with_logits((pf + pr) / 2, tgt)
        loss.backward()
        assert abs(loss.item() - float(z[name + "_train_loss"])) < 1e-5
        np.testing.assert_allclose(xf.grad.numpy(), z[name + "_train_dxf"], atol=1e-6, rtol=1e-4)
        np.testing.assert_allclose(xr.grad.numpy(), z[name + "_train_dxr"], atol=1e-6, rtol=1e-4)
        for k, p in m.named_parameters():
            np.testing.assert_allclose(p.grad.numpy(), z["%s_grad_%s" % (name, k)], atol=2e-6, rtol=1e-4, err_msg=k)
        opt.step()
        post = state_from(z, name + "_post")
        for k, v in m.state_dict().items():
            np.testing.assert_allclose(v.numpy(), post[k].numpy(), atol=1e-5, rtol=1e-5, err_msg=k)
        opt.zero_grad()
        _, pf, _, _ = m(xf, adj); _, pr, _, _ = m(xr, adj)
        loss2 = F.binary_cross_entropy_with_logits((pf + pr) / 2, tgt)
        loss2.backward(); opt.step()
        assert abs(loss2.item() - float(z[name + "_train_loss2"])) < 2e-5
        post2 = state_from(z, name + "_post2")
        for k, v in m.state_dict().items():
            np.testing.assert_allclose(v.numpy(), post2[k].numpy(), atol=2e-5, rtol=2e-5, err_msg=k)


def test_reference_layer_rule_quirk():
    # models/ChromeModels.py:25 -- a second layer only when layers == 2
    assert O.GatedGCNOracle(128, 5, 0.0, 4, reference_layer_rule=True).n_layers == 1
    assert O.GatedGCNOracle(128, 5, 0.0, 2, reference_layer_rule=True).n_layers == 2
    assert O.GatedGCNOracle(128, 5, 0.0, 4).n_layers == 4


# ------------------------------------------------------------------ G4: stage loop
def test_g4_finetune_loop(golden):
    z = golden("g4_finetune_loop.npz")
    chroms = [str(c) for c in z["chroms"]]
    feats, graphs = {}, {}
    for c in chroms:
        graphs[c] = csr_from(z, c + "_in")
        feats[c] = {"forward": torch.from_numpy(z[c + "_xf"]), "backward": torch.from_numpy(z[c + "_xr"]),
                    "target": torch.from_numpy(z[c + "_tgt"])}
    init = state_from(z, "init")
    m = O.GatedGCNOracle(128, init["out.weight"].shape[0], 0.0, 2)
    m.load_state_dict(init)
    opt = O.make_sgd(m, 0.25)
    losses = []
    for e in range(2):
        preds, tg, tot = O.finetune_epoch(m, feats, graphs, opt, "train", "hic")
        np.testing.assert_allclose(preds.numpy(), z["train_preds_e%d" % e], atol=5e-5, rtol=5e-5)
        losses.append(tot)
    ref_tr = z["train_losses"]
    assert abs(losses[0] - ref_tr[:3].sum()) < 1e-4 and abs(losses[1] - ref_tr[3:].sum()) < 1e-4
    preds, tg, tot = O.finetune_epoch(m, feats, graphs, opt, "valid", "hic")
    np.testing.assert_allclose(preds.numpy(), z["eval_preds"], atol=5e-5, rtol=5e-5)
    assert abs(tot - z["eval_losses"].sum()) < 1e-4
    final = state_from(z, "final")
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.numpy(), final[k].numpy(), atol=5e-5, rtol=5e-5, err_msg=k)
    assert tg.shape[0] == sum(feats[c]["target"].shape[0] for c in chroms)


# ------------------------------------------------------------------ G5: metrics
def test_g5_metrics_oracle_matches_reference(golden):
    z = golden("g5_metrics.npz")
    m = O.multilabel_metrics_np(z["targets"].astype(np.float64), z["preds"])
    for k_ref, k in [("ref_auroc", "auroc"), ("ref_aupr", "aupr"), ("ref_fdr", "recall_at_fdr"), ("ref_ap", "average_precision")]:
        np.testing.assert_allclose(m[k], z[k_ref], rtol=1e-12, atol=1e-12, equal_nan=True, err_msg=k)
