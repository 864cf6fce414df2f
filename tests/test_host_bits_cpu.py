"""CPU: small host-side pieces of round 5 that need no GPU -- the loss-term scaling (one multiply by a cached constant vector +
unbind instead of slice / divide / multiply per term), the pinned packed-weight cache, the cached batch constants."""
import torch

from sassd import autograd as AG
from sassd import detector as D
from sassd import train as T


def test_scaled_terms_equal_the_per_term_formulation():
    sums = torch.tensor([3.5, -1.25, 0.75], requires_grad=True)
    b = 2
    terms = D._scaled_terms(sums, (2.0 / b, 1.0 / b, 0.2 / b))
    assert all(t.shape == (1,) for t in terms)
    ref_in = sums.detach().clone().requires_grad_(True)
    ref = (ref_in[0:1] / b * 2, ref_in[1:2] / b, ref_in[2:3] / b * .2)
    for t, r in zip(terms, ref):
        assert abs(float(t) - float(r)) <= 1e-7 * max(1.0, abs(float(r)))
    w = torch.tensor([1.0, -2.0, 0.5])
    sum(t.sum() * wi for t, wi in zip(terms, w)).backward()
    sum(r.sum() * wi for r, wi in zip(ref, w)).backward()
    assert torch.allclose(sums.grad, ref_in.grad, rtol=1e-6, atol=0)
    # two terms (the auxiliary head)
    s2 = torch.tensor([4.0, 6.0], requires_grad=True)
    a, c = D._scaled_terms(s2, (0.5, 0.5))
    (a + c).sum().backward()
    assert float(a) == 2.0 and float(c) == 3.0 and torch.equal(s2.grad, torch.tensor([0.5, 0.5]))
    # the constant vector is cached per (device, values)
    assert D._const_f32(sums.device, (1.0, 0.5, 0.1)) is D._const_f32(sums.device, (1.0, 0.5, 0.1))


def test_pack_cache_pins_its_source_and_stays_bounded():
    c = AG._PackCache()
    keep = []
    for i in range(c.MAX + 10):
        src = torch.zeros(4)
        c.put(("k", i), ("gen", i), torch.full((2,), float(i)), src)
        keep.append(src.data_ptr())
    assert len(c) <= c.MAX
    gen, pack, src = c[("k", c.MAX + 9)]
    assert gen == ("gen", c.MAX + 9) and float(pack[0]) == c.MAX + 9 and src.data_ptr() == keep[-1]
    assert ("k", 0) not in c                               # the oldest entries were dropped (a dropped entry = a re-pack)
    # while an entry lives its source tensor lives: no other tensor can be handed that address
    live = {v[2].data_ptr() for v in c.values()}
    fresh = [torch.zeros(4) for _ in range(64)]
    assert not (live & {t.data_ptr() for t in fresh})
    # overwriting a key does not grow the cache
    n = len(c)
    c.put(("k", c.MAX + 9), ("gen", "new"), torch.zeros(1), torch.zeros(1))
    assert len(c) == n and c[("k", c.MAX + 9)][0] == ("gen", "new")
    # least recently USED goes first: a long-lived entry that keeps being hit survives any number of transient keys
    c2 = AG._PackCache()
    c2.put("param", 0, torch.zeros(1), torch.zeros(1))
    for i in range(3 * c2.MAX):
        assert c2.get("param") is not None
        c2.put(("transient", i), 0, torch.zeros(1), torch.zeros(1))
    assert "param" in c2 and len(c2) <= c2.MAX and ("transient", 0) not in c2
    assert c2.get("missing") is None


def test_cached_batch_constants():
    dev = torch.device("cpu")
    a = D._arange_i32(dev, 7)
    assert a.dtype == torch.int32 and a.tolist() == list(range(7)) and D._arange_i32(dev, 7) is a
    an = torch.arange(21, dtype=torch.float32).view(3, 7)
    b2 = T._batched_anchors(an, 2)
    assert b2.shape == (2, 3, 7) and b2.is_contiguous() and torch.equal(b2[0], an) and torch.equal(b2[1], an)
    assert T._batched_anchors(an, 2) is b2 and T._batched_anchors(an, 3).shape == (3, 3, 7)
    an.add_(1)                                              # an in-place edit of the anchors invalidates the cached batch
    assert torch.equal(T._batched_anchors(an, 2)[1], an)
