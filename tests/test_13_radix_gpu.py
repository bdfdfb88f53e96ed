"""GPU: the stable LSD radix sort of the binning phase (keys bit-exact, ties by index) on its own,
through the C ABI test hook -- both implementations (three launches per pass; one launch per pass with decoupled look-back)
against torch's stable sort, at ragged sizes around the tile boundaries, every digit split and adversarial key sets."""
import ctypes as C

import pytest
import torch

from opengaussian_amd import _lib

pytestmark = pytest.mark.gpu


def _sort(keys, key_bits, variant, dev):
    lib = _lib.lib()
    n = keys.numel()
    tmp = torch.empty(int(lib.ogs_selftest_radix_tmp_bytes(n)), dtype=torch.uint8, device=dev)
    k = [keys.clone(), torch.full_like(keys, -1)]
    v = [torch.arange(n, dtype=torch.int32, device=dev), torch.full((n,), -1, dtype=torch.int32, device=dev)]
    res = C.c_int32(-1)
    _lib.check(lib.ogs_selftest_radix_sort(k[0].data_ptr(), v[0].data_ptr(), k[1].data_ptr(), v[1].data_ptr(), n, key_bits, variant,
                                           tmp.data_ptr(), C.byref(res), torch.cuda.current_stream().cuda_stream), "radix")
    torch.cuda.synchronize()
    return k[res.value], v[res.value]


def _check(keys, key_bits, dev, what):
    mask = (1 << key_bits) - 1
    want_k, want_v = torch.sort(keys.to(torch.int64) & 0xFFFFFFFF & mask, stable=True)
    for variant in (0, 1):
        k, v = _sort(keys, key_bits, variant, dev)
        got_digits = k.to(torch.int64) & 0xFFFFFFFF & mask
        assert torch.equal(got_digits, want_k), f"{what}: keys, variant {variant}"
        assert torch.equal(v.to(torch.int64), want_v), f"{what}: stable order (ties by index), variant {variant}"
        assert torch.equal(k, keys[v.to(torch.int64)]), f"{what}: keys travel with their values, variant {variant}"


@pytest.mark.parametrize("n", [1, 2, 63, 64, 255, 1023, 1024, 1025, 4095, 4096, 4097, 100_000, 262_144, 262_145, 1_000_003])
def test_radix_sort_matches_stable_sort_full_keys(gpu_device, n):
    g = torch.Generator().manual_seed(n)
    depth = torch.rand(n, generator=g) * 99.8 + 0.2                      # positive view depths -> their bit patterns
    _check(depth.view(torch.int32).to(gpu_device), 32, gpu_device, f"depth bits n={n}")
    ties = torch.randint(0, 7, (n,), generator=g, dtype=torch.int32)     # heavy ties: stability carries the order
    _check(ties.to(gpu_device), 32, gpu_device, f"ties n={n}")


@pytest.mark.parametrize("key_bits", [1, 4, 7, 8, 9, 11, 13, 16, 17, 24, 25, 31])
def test_radix_sort_every_digit_split(gpu_device, key_bits):
    n = 70_001
    g = torch.Generator().manual_seed(key_bits)
    keys = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int32) & ((1 << key_bits) - 1)
    _check(keys.to(gpu_device), key_bits, gpu_device, f"{key_bits}-bit tile ids")


def test_radix_sort_adversarial_keys(gpu_device):
    n = 300_007
    dev = gpu_device
    _check(torch.zeros(n, dtype=torch.int32, device=dev), 32, dev, "all equal")                 # one digit run spans every tile
    _check(torch.arange(n, dtype=torch.int32, device=dev).flip(0).contiguous(), 32, dev, "descending")
    _check(torch.arange(n, dtype=torch.int32, device=dev), 32, dev, "already sorted")
    alt = (torch.arange(n, device=dev) % 2).to(torch.int32) * 0x7FFFFFFF
    _check(alt.contiguous(), 32, dev, "two extreme values alternating")


def _sort_drop(keys, key_bits, variant, dev):
    lib = _lib.lib()
    n = keys.numel()
    tmp = torch.empty(int(lib.ogs_selftest_radix_tmp_bytes(n)), dtype=torch.uint8, device=dev)
    k = [keys.clone(), torch.full_like(keys, -1)]
    v = [torch.arange(n, dtype=torch.int32, device=dev), torch.full((n,), -1, dtype=torch.int32, device=dev)]
    res = C.c_int32(-1)
    _lib.check(lib.ogs_selftest_radix_sort(k[0].data_ptr(), v[0].data_ptr(), k[1].data_ptr(), v[1].data_ptr(), n, key_bits, variant | 2,
                                           tmp.data_ptr(), C.byref(res), torch.cuda.current_stream().cuda_stream), "radix")
    torch.cuda.synchronize()
    kept = res.value >> 1
    return k[res.value & 1][:kept], v[res.value & 1][:kept], kept


@pytest.mark.parametrize("n,key_bits,frac", [(1, 13, 1.0), (64, 6, 0.5), (1025, 8, 0.9), (4097, 13, 0.52), (100_000, 12, 0.52),
                                            (262_145, 13, 0.3), (1_000_003, 13, 0.52), (300_007, 17, 0.0), (70_001, 9, 1.0)])
def test_radix_sort_drop_mode_is_the_stable_sort_of_the_kept_keys(gpu_device, n, key_bits, frac):
    """Default binning mode: the first pass of the tile sort leaves out the keys 0xFFFFFFFF (the (Gaussian, tile) pairs that cannot
    reach their tile) and the later passes run on the survivors.  Both implementations, one and more passes, none / some / all
    dropped: the result is torch's stable sort of the kept keys, values (= original positions) in step, count reported."""
    g = torch.Generator().manual_seed(n * 31 + key_bits)
    keys = torch.randint(0, 1 << key_bits, (n,), generator=g, dtype=torch.int64)
    dropped = torch.rand(n, generator=g) < frac
    keys_in = torch.where(dropped, torch.full_like(keys, 0xFFFFFFFF), keys).to(torch.int32).to(gpu_device)   # wraps to -1
    keep_idx = torch.nonzero(~dropped).flatten()
    want_k, order = torch.sort(keys[keep_idx], stable=True)
    want_v = keep_idx[order]
    for variant in (0, 1):
        k, v, kept = _sort_drop(keys_in, key_bits, variant, gpu_device)
        assert kept == keep_idx.numel(), (variant, kept, keep_idx.numel())
        assert torch.equal(k.to(torch.int64).cpu() & 0xFFFFFFFF, want_k), f"keys, variant {variant}"
        assert torch.equal(v.to(torch.int64).cpu(), want_v), f"stable order, variant {variant}"
