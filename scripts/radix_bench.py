"""Measurement: the binning phase's radix sort on its own, three-launch passes (variant 0) against one-launch passes with
decoupled look-back (variant 1), over the sizes of the depth sort (32-bit keys, 4 passes) and of the tile sort (13-bit
keys, 2 passes).  Checks every result against torch's stable sort.  usage: python scripts/radix_bench.py > profiles/r03_radix_variants.json"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opengaussian_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
out = {"unit": "us per sort (HIP events, median of 20)", "rows": []}
for key_bits in (32, 13):
    for n in (50_000, 100_000, 250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000):
        g = torch.Generator().manual_seed(n + key_bits)
        if key_bits == 32:   # depth bits of z ~ U(0.2, 10): what the depth sort sees
            keys = (torch.rand(n, generator=g) * 9.8 + 0.2).view(torch.int32).to(dev)
        else:
            keys = torch.randint(0, 8160, (n,), generator=g, dtype=torch.int32).to(dev)
        vals = torch.arange(n, dtype=torch.int32, device=dev)
        want_k, order = torch.sort(keys.to(torch.int64), stable=True)
        row = {"n": n, "key_bits": key_bits}
        for variant in (0, 1):
            tmp = torch.empty(int(lib.ogs_selftest_radix_tmp_bytes(n)), dtype=torch.uint8, device=dev)
            k = [keys.clone(), torch.empty_like(keys)]
            v = [vals.clone(), torch.empty_like(vals)]
            res = C.c_int32(0)
            stream = torch.cuda.current_stream().cuda_stream
            ts = []
            for rep in range(23):
                k[0].copy_(keys); v[0].copy_(vals)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.ogs_selftest_radix_sort(k[0].data_ptr(), v[0].data_ptr(), k[1].data_ptr(), v[1].data_ptr(), n, key_bits,
                                                       variant, 0, None, tmp.data_ptr(), C.byref(res), stream), "radix")
                e1.record(); e1.synchronize()
                if rep >= 3:
                    ts.append(e0.elapsed_time(e1) * 1e3)
            ok = bool(torch.equal(k[res.value].to(torch.int64), want_k) and torch.equal(v[res.value].to(torch.int64), order))
            ts.sort()
            row["legacy_3_launch_us" if variant == 0 else "onesweep_1_launch_us"] = round(ts[len(ts) // 2], 1)
            row["ok_" + str(variant)] = ok
        out["rows"].append(row)
        sys.stderr.write(json.dumps(row) + "\n")
print(json.dumps(out, indent=1))
