"""Randomised end-to-end run of batch_reconstruct_device (device_reconstruction.py): n parties in one process over in-memory queues, up to t
of them Byzantine in both rounds (random columns, columns of ONE other polynomial per chunk -- coordinated --, a shifted copy of the honest
column, truncated payloads, silence), batch sizes that leave a padded last chunk, both point policies.  Every honest party must open exactly
the secrets (batch_reconstruction.py:88-227's guarantee).   usage: python scratch/stress_reconstruct.py [seconds] [seed]"""
import asyncio
import random
import sys
import time

sys.path.insert(0, '.')
from honeybadgermpc_amd import wire  # noqa: E402
from honeybadgermpc_amd._capi import Context  # noqa: E402
from honeybadgermpc_amd.device_reconstruction import batch_reconstruct_device  # noqa: E402
from honeybadgermpc_amd.field import GF  # noqa: E402
from honeybadgermpc_amd.polynomial import EvalPoint  # noqa: E402

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
ctx = Context.get(P)


class Net:
    def __init__(self, n):
        self.q = [asyncio.Queue() for _ in range(n)]

    def send(self, i, tamper):
        def _send(dest, msg):
            out = tamper(dest, msg)
            if out is not None:
                self.q[dest].put_nowait((i, out))
        return _send

    def recv(self, i):
        return self.q[i].get


t_end = time.time() + budget
runs = fails = 0
while time.time() < t_end:
    n = rnd.choice([4, 7, 10, 13, 16])
    t = rnd.randrange(1, (n - 1) // 3 + 1)
    b = rnd.choice([1, 2, t, t + 1, t + 2, 3 * (t + 1) - 1, rnd.randrange(1, 80)])
    use_omega = rnd.random() < 0.3
    point = EvalPoint(GF(P), n, use_omega_powers=use_omega)
    xs = [point(i).value for i in range(n)]
    secrets = [rnd.choice([0, 1, P - 1, rnd.randrange(P)]) if rnd.random() < 0.2 else rnd.randrange(P) for _ in range(b)]
    polys = [[s] + [rnd.randrange(P) for _ in range(t)] for s in secrets]
    shares = [[sum(co * pow(xs[i], e, P) for e, co in enumerate(poly)) % P for poly in polys] for i in range(n)]
    bad = set(rnd.sample(range(n), rnd.randrange(0, t + 1)))
    modes = {i: rnd.choice(["random", "shift", "truncate", "silent", "late-chunk"]) for i in bad}
    shift = rnd.randrange(1, P)

    def tamper_for(i):
        def honest(dest, msg):
            return msg

        def tamper(dest, msg):
            tag, blob = msg
            mode = modes[i]
            vals = wire.unpack_ints(blob)
            if mode == "silent":
                return None if rnd.random() < 0.7 else msg
            if mode == "truncate":
                return (tag, blob[: max(0, len(blob) - rnd.randrange(1, 40))])
            if mode == "shift":
                return (tag, wire.pack_ints([(v + shift) % P for v in vals], P))
            if mode == "late-chunk":
                vals = list(vals)
                vals[-1] = (vals[-1] + 1 + rnd.randrange(P - 1)) % P
                return (tag, wire.pack_ints(vals, P))
            return (tag, wire.pack_ints([rnd.randrange(P) for _ in vals], P))

        return tamper if i in bad else honest

    async def main():
        net = Net(n)
        tasks = [asyncio.ensure_future(batch_reconstruct_device(ctx.upload_ints(shares[i]), P, t, n, i, net.send(i, tamper_for(i)), net.recv(i),
                                                               use_omega_powers=use_omega)) for i in range(n)]
        honest = [tasks[i] for i in range(n) if i not in bad]
        done, pending = await asyncio.wait(honest, timeout=60)
        for tk in tasks:
            if not tk.done():
                tk.cancel()
        await asyncio.gather(*tasks, return_exceptions=True)
        return [tasks[i].result() if (tasks[i] in done and not tasks[i].cancelled() and tasks[i].exception() is None) else tasks[i] for i in range(n)], len(pending)

    results, stuck = asyncio.run(main())
    ok = stuck == 0
    for i in range(n):
        if i in bad:
            continue
        r = results[i]
        if isinstance(r, asyncio.Future) or r is None or ctx.download_ints(r) != secrets:
            ok = False
    if not ok:
        fails += 1
        print("FAIL", n, t, b, use_omega, "bad", sorted(bad), modes, "stuck", stuck, flush=True)
    runs += 1
print(f"stress_reconstruct: {runs} opens of {{4..16}} parties with up to t Byzantine senders, {fails} failures (seed {seed}, {budget:.0f} s)")
