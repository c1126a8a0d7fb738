"""CPU: size-independent properties of the host-side logic (hypothesis; no device, no oracle needed).
  * the fp16+8 storage (marconet_amd/mxfmt.py, the packers the device kernels are tested against): the decoded error
    (also of a second trip) is bounded by the block's largest value, the scale byte is the documented function of the block maximum, zero blocks stay zero;
  * shard arithmetic (marconet_amd/pipeline.py): shard_range / balance_shards are exact partitions with the load bound of the greedy."""
import math

import pytest
import torch
from hypothesis import given, settings, strategies as st

from marconet_amd import mxfmt
from marconet_amd.pipeline import balance_shards, shard_range

SETTINGS = dict(max_examples=60, deadline=None, derandomize=True)          # fixed example set: the tier must not flake


@settings(**SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), log_scale=st.integers(-20, 14), blocks=st.integers(1, 3), sparsity=st.sampled_from([0.0, 0.5, 0.97]))
def test_fp16p8_activation_round_trip_properties(seed, log_scale, blocks, sparsity):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((5, 32 * blocks), generator=g) * (2.0 ** log_scale)
    x = x * (torch.rand(x.shape, generator=g) >= sparsity)
    x[0, :32] = 0.0                                                        # an all-zero block
    b = mxfmt.pack_act(x)
    y = mxfmt.unpack_act(b, x.shape[-1])
    # padding bytes and zero blocks
    bb = b.reshape(5, blocks, 128)
    assert int(bb[..., 97:].max()) == 0
    assert torch.equal(y[0, :32], torch.zeros(32))
    # scale byte = max(floor(log2 max|hi|), -15) - 7 + 127 for a non-zero block (the floor: round 6, fp16-subnormal blocks), 0 for an all-zero one; from the HI halves
    hi = x.reshape(5, blocks, 32).to(torch.float16).float()
    m = hi.abs().amax(-1)
    exp = torch.where(m > 0, (torch.floor(torch.log2(m.double())).float() - 7 + 127).clamp(105, 254), torch.zeros_like(m))
    assert torch.equal(bb[..., 96].float(), exp)
    # error bound: hi carries 11 significant bits of each value; the lo byte adds 4 more bits of the BLOCK's scale:
    # |v - decode| <= 2^-16 * 2^floor(log2 max|hi|) (half an e4m3 ulp at the top binade of lo) wherever fp16 itself is normal
    bound = torch.pow(2.0, torch.floor(torch.log2(m.clamp_min(1e-30).double())).float() - 16).unsqueeze(-1)
    err = (x.reshape(5, blocks, 32) - y.reshape(5, blocks, 32)).abs()
    normal = (m > 2.0 ** -7).unsqueeze(-1)                                  # block scale >= 2^-14: lo * 2^11 / s stays inside e4m3's normal range
    assert bool(((err <= bound) | ~normal).all()), float((err / bound)[normal.expand_as(err)].max())
    # a second trip through the format moves a decoded value by no more than the same bound (NOT byte-idempotent: hi + lo8 may round to the
    # neighbouring half when lo sits at half an ulp of hi — the device converters are therefore compared by value, not by bytes)
    y2 = mxfmt.unpack_act(mxfmt.pack_act(y), x.shape[-1])
    assert bool((((y2 - y).abs().reshape(5, blocks, 32) <= bound) | ~normal).all())
    # never worse than plain fp16
    assert bool((err <= (x.reshape(5, blocks, 32) - hi).abs() + 1e-45).all())


@settings(**SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), o=st.integers(1, 9), log_scale=st.integers(-12, 6))
def test_fp16p8_weight_packer_properties(seed, o, log_scale):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn((o, 3, 3, 32), generator=g) * (2.0 ** log_scale)
    buf = mxfmt.pack_weight(w)
    rows = buf[: o * 9 * 128].reshape(o, 3, 3, 1, 128)
    tail = buf[o * 9 * 128:]
    assert tail.numel() == (o + 15) // 16 * 16 and int(tail[o:].max() if tail.numel() > o else 0) == 0
    hi = rows[..., 0:64].contiguous().view(torch.float16).float()
    assert torch.equal(hi, (w * mxfmt.WSCALE).reshape(o, 3, 3, 1, 32).to(torch.float16).float())
    # per-output-channel scale byte: E8M0 of s * 2^-11 with s = 2^(floor(log2 max|hi|) - 7), clamped so that the byte is >= 0
    m = hi.abs().reshape(o, -1).amax(-1)
    e8 = (torch.floor(torch.log2(m.clamp_min(1e-30).double())).float() - 7 + 127).clamp(11, 254)
    assert torch.equal(tail[:o].float(), e8 - 11)
    # hi8 * s reproduces hi to 4 significant bits (e4m3), never above the fp8 maximum
    s = torch.pow(2.0, e8 - 127).reshape(o, 1, 1, 1, 1)
    inv = [mxfmt.PERM.index(i) for i in range(32)]
    hi8 = torch.cat([rows[..., 80:96], rows[..., 112:128]], -1).contiguous().view(torch.float8_e4m3fn).float()[..., inv]
    assert bool(((hi8 * s - hi).abs() <= hi.abs() * 2.0 ** -4 + s * 2.0 ** -9).all())


@settings(**SETTINGS)
@given(total=st.integers(0, 300), world=st.integers(1, 9))
def test_shard_range_is_an_exact_balanced_partition(total, world):
    spans = [shard_range(total, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


@settings(**SETTINGS)
@given(widths=st.lists(st.integers(16, 512), min_size=1, max_size=80), world=st.integers(1, 8), seed=st.integers(0, 999))
def test_balance_shards_partitions_and_bounds_the_load(widths, world, seed):
    g = torch.Generator().manual_seed(seed)
    counts = [int(v) for v in torch.randint(0, 17, (len(widths),), generator=g)]
    parts = balance_shards(widths, counts, world)
    assert len(parts) == world
    assert sorted(b for p in parts for b in p) == list(range(len(widths)))          # every image exactly once
    cost = [108.0 + 3.69 + 484.1 * min(512, (w + 63) // 64 * 64) / 512.0 + 89.03 * n for w, n in zip(widths, counts)]
    load = [sum(cost[b] for b in p) for p in parts]
    # longest-processing-time greedy: no rank exceeds the mean by more than one job
    assert max(load) <= sum(cost) / world + max(cost) + 1e-6
    # inside a rank the images are ordered by bucket width (few distinct widths in a row)
    for p in parts:
        wb = [min(512, (widths[b] + 63) // 64 * 64) for b in p]
        assert wb == sorted(wb)
