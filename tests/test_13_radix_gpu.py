"""GPU: the stable LSD radix sort of the binning phase (keys bit-exact, ties by index) on its own,
through the C ABI test hook -- both implementations (three launches per pass; one launch per pass with decoupled look-back)
against torch's stable sort, at ragged sizes around the tile boundaries, every digit split and adversarial key sets."""
import ctypes as C

import pytest
import torch

from opengaussian_amd import _lib

pytestmark = pytest.mark.gpu


def _sort(keys, key_bits, variant, dev, items=0):
    lib = _lib.lib()
    n = keys.numel()
    tmp = torch.empty(int(lib.ogs_selftest_radix_tmp_bytes(n)), dtype=torch.uint8, device=dev)
    k = [keys.clone(), torch.full_like(keys, -1)]
    v = [torch.arange(n, dtype=torch.int32, device=dev), torch.full((n,), -1, dtype=torch.int32, device=dev)]
    res = C.c_int32(-1)
    _lib.check(lib.ogs_selftest_radix_sort(k[0].data_ptr(), v[0].data_ptr(), k[1].data_ptr(), v[1].data_ptr(), n, key_bits, variant,
                                           items, None, tmp.data_ptr(), C.byref(res), torch.cuda.current_stream().cuda_stream),
               "radix")
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


def _sort_drop(keys, key_bits, variant, dev, n_dev=None):
    lib = _lib.lib()
    n = keys.numel()
    tmp = torch.empty(int(lib.ogs_selftest_radix_tmp_bytes(n)), dtype=torch.uint8, device=dev)
    k = [keys.clone(), torch.full_like(keys, -1)]
    v = [torch.arange(n, dtype=torch.int32, device=dev), torch.full((n,), -1, dtype=torch.int32, device=dev)]
    res = C.c_int32(-1)
    _lib.check(lib.ogs_selftest_radix_sort(k[0].data_ptr(), v[0].data_ptr(), k[1].data_ptr(), v[1].data_ptr(), n, key_bits, variant | 2,
                                           0, None if n_dev is None else n_dev.data_ptr(), tmp.data_ptr(), C.byref(res),
                                           torch.cuda.current_stream().cuda_stream), "radix")
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


@pytest.mark.parametrize("items", [4, 16])
def test_one_launch_sort_with_a_forced_tile_size_at_4m_keys(gpu_device, items):
    """The configuration of the round-3 memory fault (gpurun_out/ab_S1M-1080p_sweep4.err: one-launch passes with 1024-key tiles
    over the 7.4 M (Gaussian, tile) pairs of the headline scene; DESIGN.md section 3 "the round-3 fault"): 4 Mi + 3 keys through
    radix_onesweep_kernel<4> need 4 x 4097 x 256 status words, which ogs_selftest_radix_tmp_bytes / sort_tmp_bytes must cover.
    The tile size is an explicit argument of the hook now (no environment switch); 16 for the other instantiation."""
    n = (4 << 20) + 3
    g = torch.Generator().manual_seed(items)
    keys = torch.randint(0, 1 << 13, (n,), generator=g, dtype=torch.int32).to(gpu_device)        # 13-bit tile ids, two passes
    k, v = _sort(keys, 13, 1, gpu_device, items=items)
    want_k, want_v = torch.sort(keys.to(torch.int64), stable=True)
    assert torch.equal(k.to(torch.int64), want_k)
    assert torch.equal(v.to(torch.int64), want_v)
    lib = _lib.lib()
    tiles4 = (n + 1023) // 1024
    assert int(lib.ogs_selftest_radix_tmp_bytes(n)) >= 4 * 256 * 4 * tiles4, "scratch must hold four 1024-key-tile status tables"


@pytest.mark.parametrize("variant", [0, 1])
def test_drop_mode_with_a_device_count_of_zero_reports_zero_kept(gpu_device, variant):
    """a deferred-capacity render phase whose view sees nothing runs the tile sort with *n_dev == 0.  Tile 0 of
    the one-launch pass used to return before storing the kept count, which the later passes, the tile ranges and pack read from
    (uninitialised) device memory.  The hook poisons the word first: both implementations must report 0."""
    n = 5000
    keys = torch.randint(0, 1 << 13, (n,), dtype=torch.int32).to(gpu_device)
    n_dev = torch.zeros(1, dtype=torch.int32, device=gpu_device)
    _, _, kept = _sort_drop(keys, 13, variant, gpu_device, n_dev=n_dev)
    assert kept == 0
    n_dev.fill_(1234)                                   # and a true count below the capacity
    k, v, kept = _sort_drop(keys, 13, variant, gpu_device, n_dev=n_dev)
    want_k, order = torch.sort(keys[:1234].to(torch.int64), stable=True)
    assert kept == 1234
    assert torch.equal(k.to(torch.int64), want_k) and torch.equal(v.to(torch.int64), order)


def test_look_back_bound_of_zero_polls_raises(gpu_device):
    """a one-launch pass whose look-back wait runs out used to set a device word nobody read -- wrong order,
    rc 0.  Fault injection through the hook (variant bit 2: bound = 0 polls; with 1000 tiles some workgroup always finds a
    predecessor unpublished): the call must fail with OGS_ERR_DEVICE, the sticky word must be clear afterwards, and the next sort
    must be correct again."""
    lib = _lib.lib()
    n = 1_000_003
    keys = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32).to(gpu_device)
    with pytest.raises(_lib.OgsError, match="look-back"):
        _sort(keys, 32, 1 | 4, gpu_device, items=4)
    assert lib.ogs_check_async_status() == 0
    _check(keys, 32, gpu_device, "after the injected fault")
