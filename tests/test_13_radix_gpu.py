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
