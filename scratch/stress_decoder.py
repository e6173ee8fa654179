"""Randomised differential run of the device decoder's plan-free robust path against the host mirror of the reference's IncrementalDecoder:
random shapes, liar counts, corruption patterns (everywhere / one chunk / a few chunks / last chunk), arrival orders; after EVERY column the
two must agree on done / confirmed errors / arrival list / polynomials decoded, raise the same exception at the same column (Welch-Berlekamp),
and return the shared polynomials.   usage: python scratch/stress_decoder.py [seconds] [seed]"""
import random, sys, time
sys.path.insert(0, '.')
from honeybadgermpc_amd._capi import Context
from honeybadgermpc_amd.device import DeviceIncrementalDecoder
from honeybadgermpc_amd.field import GF
from honeybadgermpc_amd.polynomial import EvalPoint
from honeybadgermpc_amd.reed_solomon import Algorithm, DecoderFactory, EncoderFactory, IncrementalDecoder, RobustDecoderFactory
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
ctx = Context.get(P)
t_end = time.time() + budget
runs = fails = raised = robust_runs = 0
while time.time() < t_end:
    n = rnd.choice([7, 10, 13, 16, 22, 31])
    if rnd.random() < 0.08:
        n = rnd.choice([40, 64, 64, 100, 130, 130, 200, 256])   # three K-blocks on the small-entry kernel; n >= 100: full-size entries; n >= 130: the builder's helper workgroups, the probe over six workgroups
    t = rnd.randrange(3, (n - 1) // 3 + 1) if (n - 1) // 3 >= 3 else (n - 1) // 3
    if t < 3:
        continue
    c = rnd.choice([1, 2, 5, 17, 40]) if n <= 31 else rnd.choice([1, 3, 9])
    use_omega = rnd.random() < 0.3
    robust = rnd.choice(["gao", "gao", "wb"])
    point = EvalPoint(GF(P), n, use_omega_powers=use_omega)
    xs = [point(i).value for i in range(n)]
    polys = [[rnd.randrange(P) for _ in range(t + 1)] for _ in range(c)]
    if rnd.random() < 0.4:
        # polynomials with leading zeros: what chunk_data's zero padding makes of the last chunk of every open (utils/misc.py:33-48), and
        # where Gao decodes past floor((n' - k) / 2) errors while Welch-Berlekamp does not (tests/golden/welch_berlekamp_low_degree.json)
        for j in ([c - 1] if rnd.random() < 0.5 else range(c)):
            if rnd.random() < 0.7:
                keep = rnd.randrange(0, t + 1)
                polys[j] = polys[j][:keep] + [0] * (t + 1 - keep)
    cols = [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]
    liars = rnd.sample(range(n), min(n, rnd.randrange(0, t + 2) + (rnd.randrange(0, t) if rnd.random() < 0.1 else 0)))          # sometimes one liar too many, now and then many
    # half of the runs: the liars are coordinated -- on the chunks they hit they all send the values of ONE other polynomial (which may
    # share up to t points with the true one: then honest senders "agree" with the fake too), the word Gao may decode to instead
    fake = None
    if rnd.random() < 0.5:
        fake = []
        for poly in polys:
            shared = rnd.sample(range(n), rnd.randrange(0, t + 1))
            # true + m(X) prod (X - x_s): equal to the true polynomial at the shared points, degree <= t
            q = [rnd.randrange(1, P)] + [0] * t
            deg = 0
            for s_ in shared:
                nq = [0] * (t + 1)
                for e in range(deg + 1):
                    nq[e + 1] = (nq[e + 1] + q[e]) % P
                    nq[e] = (nq[e] - q[e] * xs[s_]) % P
                q, deg = nq, deg + 1
            fake.append([(a + b) % P for a, b in zip(poly, q)])
    for i in liars:
        kind = rnd.randrange(4)
        hit = {0: range(c), 1: [c - 1], 2: [rnd.randrange(c)], 3: rnd.sample(range(c), max(1, c // 3))}[kind]
        for j in hit:
            if fake is not None:
                cols[i][j] = sum(co * pow(xs[i], e, P) for e, co in enumerate(fake[j])) % P
            else:
                cols[i][j] = (cols[i][j] + 1 + rnd.randrange(P - 1)) % P
    order = list(range(n))
    rnd.shuffle(order)
    if rnd.random() < 0.5:
        order = liars + [i for i in order if i not in liars]
    codec = Algorithm.FFT if use_omega else Algorithm.VANDERMONDE
    host = IncrementalDecoder(EncoderFactory.get(point, codec), DecoderFactory.get(point, codec),
                              RobustDecoderFactory.get(t, point, algorithm=Algorithm.GAO if robust == "gao" else Algorithm.WELCH_BERLEKAMP),
                              degree=t, batch_size=c, max_errors=t)
    # the two ways a transport feeds the decoder (copied columns / received in place) and the two things a round asks of it (every
    # coefficient / the constant terms only: what R1 forwards) -- all four combinations
    in_place = rnd.random() < 0.5
    want = "constant" if rnd.random() < 0.35 else "all"
    buf = ctx.empty(n * c).view(n, c, 4) if in_place else None
    # (round 6) half the decoders defer their verdicts: up to three more columns are announced while the quorum's launch is out, and the states
    # are compared once the verdict is in and the late columns have been replayed
    defer, busy = rnd.random() < 0.5, rnd.random() < 0.5
    dev = DeviceIncrementalDecoder(P, n, t, batch_size=c, robust=robust, use_omega_powers=use_omega, columns=buf, want=want, defer_verdict=defer, stream_busy=busy)
    bad = None
    held = 0
    for step, idx in enumerate(list(order) + [None]):
        if idx is None:
            # every column is in: a verdict still out is waited for, then the final comparison
            if not (defer and dev.pending()):
                break
            if dev.done() != host.done() or dev._confirmed_errors != host._confirmed_errors or (not host.done() and (dev._z != host._z or dev._num_decoded != host._num_decoded)):
                bad = ("state at the end", host.done(), dev.done())
            elif host.done():
                hres, _ = host.get_results()
                dres, _ = dev.get_results()
                width = dres.shape[1]
                if ctx.download_ints(dres.reshape(-1, 4)) != [v for row in hres for v in (list(row) + [0] * (t + 1 - len(row)))[:width]]:
                    bad = ("result at the end",)
            break
        hexc = dexc = None
        try:
            host.add(idx, cols[idx])
        except BaseException as e:  # noqa: BLE001
            hexc = e
        try:
            if in_place:
                if dev.accepts(idx):
                    dev.slot(idx).copy_(ctx.upload_ints(cols[idx]))
                dev.add(idx)
            else:
                dev.add(idx, ctx.upload_ints(cols[idx]))
        except BaseException as e:  # noqa: BLE001
            dexc = e
        if hexc is not None and dexc is None and defer and dev.pending():
            try:                                  # the verdict is still out: what the reference raised at this column comes when it is in
                dev.done()
            except BaseException as e:  # noqa: BLE001
                dexc = e
        if (hexc is None) != (dexc is None) or (hexc is not None and type(hexc) is not type(dexc)):
            bad = ("exception", step, repr(hexc), repr(dexc)); break
        if hexc is not None:
            raised += 1; break
        if defer and dev.pending() and held < 3 and not host.done() and rnd.random() < 0.7:
            held += 1
            continue
        if dev.done() != host.done() or dev._confirmed_errors != host._confirmed_errors or dev._z != host._z or dev._num_decoded != host._num_decoded:
            bad = ("state", step, host.done(), dev.done(), sorted(host._confirmed_errors), sorted(dev._confirmed_errors), host._num_decoded, dev._num_decoded); break
        if host.done():
            hres, _ = host.get_results()
            dres, _ = dev.get_results()
            # (the reference's Welch-Berlekamp rows come with their trailing zeros stripped, reed_solomon_wb.py:151 over polynomial.py:14-20;
            # the device decoder's result is a (C, degree + 1, limbs) tensor: compared padded)
            width = dres.shape[1]                 # 1: a constant-term decoder that finished on its optimistic step
            if ctx.download_ints(dres.reshape(-1, 4)) != [v for row in hres for v in (list(row) + [0] * (t + 1 - len(row)))[:width]]:
                bad = ("result", step)
            break
    robust_runs += dev.probes + dev.radius_verdicts + dev.launches > 0
    if bad:
        fails += 1
        print("FAIL", n, t, c, use_omega, robust, "liars", liars, "order", order, bad, flush=True)
    runs += 1
print(f"stress_decoder: {runs} decodes ({robust_runs} reached the robust phase, {raised} ended in the reference's own exception), {fails} failures (seed {seed}, {budget:.0f} s)")
