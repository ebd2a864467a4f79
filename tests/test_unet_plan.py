"""Host logic of the U-Net execution plan (uforecon_amd/unet3d.py, ops.py) that needs no GPU: the cache of the weights'
planes follows the weight TENSOR OBJECT and its version (never an address), and the table of activation bounds hands a
plane layer's output bound to the next layer (CostRegNet / CostRegNetWeight, code1/encoder_utils/fmt/module.py:469-543)."""
import gc

import torch

from uforecon_amd import ops, unet3d


def test_planes_cache_is_keyed_on_the_tensor_object_and_its_version():
    ops._PLANES.clear()
    w = torch.zeros(16, 16, 3, 3, 3)
    ws1, ready1 = ops._planes_lookup(w, None, False, 1024)
    ws2, ready2 = ops._planes_lookup(w, None, False, 1024)
    assert not ready1 and ready2 and ws2 is ws1                       # second call: the planes are there
    _, ready_flip = ops._planes_lookup(w, None, True, 1024)            # mirrored taps: other planes
    assert not ready_flip
    _, again = ops._planes_lookup(w, None, False, 1024)
    assert not again                                                   # ... which replaced the entry
    w.add_(1.0)                                                        # in-place update: version bump
    _, after_update = ops._planes_lookup(w, None, False, 1024)
    assert not after_update
    # a NEW tensor in the old one's place (same id is possible after collection, same address likely): never "ready"
    key = id(w)
    del w
    gc.collect()
    for _ in range(50):
        w2 = torch.zeros(16, 16, 3, 3, 3)
        _, r = ops._planes_lookup(w2, None, False, 1024)
        assert not r, "a stale entry answered for a new tensor"
        if id(w2) == key:
            break
        del w2
    # a second head is part of the signature
    w3, h1, h2 = torch.zeros(8, 8, 3, 3, 3), torch.zeros(1, 8, 3, 3, 3), torch.zeros(1, 8, 3, 3, 3)
    assert not ops._planes_lookup(w3, h1, False, 64)[1] and ops._planes_lookup(w3, h1, False, 64)[1]
    assert not ops._planes_lookup(w3, h2, False, 64)[1]


def test_bounds_table_hands_a_layers_bound_to_the_next(monkeypatch):
    calls = []
    monkeypatch.setattr(ops, "absmax", lambda t: calls.append(id(t)) or torch.tensor([float(t.abs().max())]))
    b = unet3d._Bounds()
    x, y = torch.randn(4, 4), torch.randn(4, 4)
    assert float(b.of(x)) == float(x.abs().max()) and calls == [id(x)]          # unknown tensor: measured
    b.put(y, torch.tensor([7.0]))
    assert float(b.of(y)) == 7.0 and calls == [id(x)]                            # known tensor: no pass
    b.put(x, None)                                                               # a layer that reports no bound changes nothing
    b.of(x)
    assert len(calls) == 2
    # the table holds the tensor it describes: an id reused by another tensor cannot inherit a bound
    z = torch.randn(4, 4)
    b._b[id(z)] = (y, torch.tensor([1.0]))
    assert float(b.of(z)) == float(z.abs().max())
