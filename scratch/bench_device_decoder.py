"""Faulty-party case at config-3 shape on the device-resident IncrementalDecoder: n=64, t=21, C chunks, `liars` senders
send garbage in every chunk and arrive first.  usage: python scratch/bench_device_decoder.py [C] [liars]"""
import random
import sys
import time

import torch

sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context  # noqa: E402
from honeybadgermpc_amd.device import BatchOpen, DeviceIncrementalDecoder  # noqa: E402
from honeybadgermpc_amd.offline import random_elements  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    c = int(sys.argv[1]) if len(sys.argv) > 1 else 47663
    liars = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n, t = 64, 21
    d = t + 1
    rnd = random.Random(3)
    ctx = Context.get(P)
    coeffs = random_elements(P, c * d)
    enc = BatchOpen(P, n, t, max_shares=c * d).r1_encode(coeffs).view(n, c, 4).clone()      # row i = sender i's column
    bad = rnd.sample(range(n), liars)
    for i in bad:
        enc[i] = random_elements(P, c)
    order = bad + [i for i in rnd.sample(range(n), n) if i not in bad]
    for rep in range(2):
        dec = DeviceIncrementalDecoder(P, n, t, batch_size=c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        used = 0
        for idx in order:
            dec.add(idx, enc[idx])
            used += 1
            if dec.done():
                break
        res, errs = dec.get_results()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert sorted(errs) == sorted(bad), (errs, bad)
        assert torch.equal(res.reshape(-1, 4), coeffs)
        print(f"C={c} liars={liars}: done after {used} columns, {dec.launches} robust launches, {dt * 1e3:.1f} ms -> {c * d / dt / 1e6:.1f} M shares/s (rep {rep})")


main()
