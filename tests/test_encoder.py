"""The window encoder's contract (config 5), CPU side: golden G7 recorded from the reference's Expecto + strand
wrapper (models/WindowModels.py:9-87, models/NonStrandSpecific.py:81-94) built under a torch seed; the build's own
encoder, built under the same seed, must have the same state_dict layout and reproduce its outputs."""
import json

import numpy as np
import torch

from chromegcn_amd import encoder as E


def test_encoder_reproduces_reference_golden(golden):
    z = golden("g7_encoder.npz")
    meta = json.loads(str(z["meta"]))
    torch.manual_seed(meta["seed"])
    enc = E.WindowEncoder(meta["nclass"], meta["seq_length"])
    sd = enc.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]]              # a reference Expecto checkpoint loads unchanged
    assert [v.numel() for v in sd.values()] == z["param_numel"].tolist()
    g = torch.Generator().manual_seed(meta["bn_seed"])
    with torch.no_grad():
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    # same seed, same construction order => the same parameters the reference encoder drew (checksums per tensor)
    np.testing.assert_allclose([float(v.double().sum()) for v in enc.state_dict().values()], z["param_sums"], rtol=1e-12, atol=1e-9)
    enc.eval()
    pair = E.StrandPair(enc, {"a": 0, "c": 1, "g": 2, "t": 3, "n": 4})
    tokens = torch.from_numpy(z["tokens"])
    torch.set_num_threads(1)
    with torch.no_grad():
        x_f, x_r, y, a, b = pair(tokens)
    assert a is None and b is None
    np.testing.assert_allclose(x_f.numpy(), z["x_f"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(x_r.numpy(), z["x_r"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(y.numpy(), z["logits"], atol=1e-5, rtol=1e-5)


def test_reverse_complement_and_shapes():
    pair = E.StrandPair(E.WindowEncoder(3, 320))
    t = torch.tensor([[0, 1, 2, 3, 4, 0, 0]])
    assert pair.reverse_complement(t).tolist() == [[3, 3, 4, 0, 1, 2, 3]]       # reverse, then a<->t, c<->g, n stays
    assert torch.equal(pair.reverse_complement(pair.reverse_complement(t)), t)
    tab = E.complement_table({"a": 3, "c": 2, "g": 1, "t": 0, "n": 4})          # a vocabulary in another order
    assert tab.tolist() == [3, 2, 1, 0, 4]
    assert E.positions_after_convs(2000) == 106 and E.WindowEncoder(5, 2000).flat_width == 960 * 106
    enc = E.WindowEncoder(5, 320).eval()
    x, y, none = enc(torch.randint(0, 5, (2, 320)))
    import pytest
    with pytest.raises(ValueError):
        E.WindowEncoder(5, 200)          # too short for the convolution stack
    assert tuple(x.shape) == (2, 128) and tuple(y.shape) == (2, 5) and none is None
