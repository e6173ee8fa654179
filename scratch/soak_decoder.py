"""Randomised soak of device.DeviceIncrementalDecoder against the host mirror of the reference's IncrementalDecoder:
random (n, t, batch), liars with random corruption patterns, random arrival orders; state compared after every add.
usage: python scratch/soak_decoder.py [seconds] [seed]"""
import random
import sys
import time

sys.path.insert(0, ".")
from honeybadgermpc_amd._capi import Context  # noqa: E402
from honeybadgermpc_amd.device import DeviceIncrementalDecoder  # noqa: E402
from honeybadgermpc_amd.field import GF  # noqa: E402
from honeybadgermpc_amd.polynomial import EvalPoint  # noqa: E402
from honeybadgermpc_amd.reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = Context.get(P)
    t_end = time.time() + budget
    trials = robust = 0
    while time.time() < t_end:
        n = rnd.choice([4, 5, 7, 8, 10, 13, 16, 22, 31])
        t = rnd.randrange(1, (n - 1) // 3 + 1)
        c = rnd.choice([1, 2, 3, 17, 64, 130])
        omega = rnd.random() < 0.3
        point = EvalPoint(GF(P), n, use_omega_powers=omega)
        xs = [point(i).value for i in range(n)]
        polys = [[rnd.randrange(P) for _ in range(t + 1)] for _ in range(c)]
        cols = [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]
        liars = rnd.sample(range(n), rnd.randrange(0, t + 1))
        for i in liars:
            hit = rnd.choice([range(c), [c - 1], [0], rnd.sample(range(c), max(1, c // 2))])
            for j in hit:
                cols[i][j] = (cols[i][j] + 1 + rnd.randrange(P - 1)) % P
        order = list(range(n))
        rnd.shuffle(order)
        algo = Algorithm.FFT if omega else Algorithm.VANDERMONDE
        host = IncrementalDecoder(EncoderFactory.get(point, algo), DecoderFactory.get(point, algo),
                                  RobustDecoderFactory.get(t, point, algorithm=Algorithm.GAO), degree=t, batch_size=c, max_errors=t)
        dev = DeviceIncrementalDecoder(P, n, t, batch_size=c, use_omega_powers=omega)
        for step, idx in enumerate(order):
            host.add(idx, cols[idx])
            dev.add(idx, ctx.upload_ints(cols[idx]))
            key = (n, t, c, omega, liars, order, step)
            assert dev.done() == host.done(), key
            assert dev._confirmed_errors == host._confirmed_errors and dev._z == host._z and dev._num_decoded == host._num_decoded, key
            if host.done():
                hres, herr = host.get_results()
                dres, derr = dev.get_results()
                assert derr == herr and ctx.download_ints(dres.reshape(-1, 4)) == [v for row in hres for v in row], key
                assert [list(r) for r in hres] == polys, key
                break
        assert host.done()
        trials += 1
        robust += 1 if dev.launches else 0
    print(f"soak: {trials} random decodes agreed step by step ({robust} went through the robust path)")


main()
