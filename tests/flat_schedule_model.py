"""Index model of k_mm8w_flat's schedule (honeybadgermpc_amd/csrc/hb_mfma_wide.hip): the pass list cut into one range a workgroup,
four passes a round, one short round whose passes are cut along the K-blocks, chunk tiles in a ring of LDS slots, the followers' sums
in ring space no tile of the round lives in.  tests/test_flat_schedule.py checks the invariants the kernel relies on."""
PART_Q = 1088        # MM8W_PART_Q: a wave's partial sums in uint4


def part_off(k, busy_a, busy_b, bufsz):
    off, i = 0, 0
    while True:
        for _ in range(2):
            if busy_a >= 0 and off < (busy_a + 1) * bufsz and off + PART_Q > busy_a * bufsz:
                off = (busy_a + 1) * bufsz
            if busy_b >= 0 and off < (busy_b + 1) * bufsz and off + PART_Q > busy_b * bufsz:
                off = (busy_b + 1) * bufsz
        if i == k:
            return off
        off += PART_Q
        i += 1


def lds_bytes(n_rt, nkb, nb):
    return (n_rt * 64 + nb * nkb * 4 * 64 + 128 + 272) * 16 + (16 * nkb + 32 * n_rt) * 4


def ring_slots(n_rt, nkb, limit=156 * 1024):
    """mm8w_flat_slots without the cost model: the smallest ring that holds a round's tiles, the next round's and the partial sums"""
    if n_rt < 4 or nkb < 8:
        return 0
    bufsz = nkb * 256
    for cand in range(3, 9):
        if lds_bytes(n_rt, nkb, cand) > limit:
            return 0
        ok = all(part_off(2, a, -1, bufsz) + PART_Q <= cand * bufsz + 128 and
                 part_off(1, a, (a + 1) % cand, bufsz) + PART_Q <= cand * bufsz + 128 for a in range(cand))
        if ok:
            return cand
    return 0


def workgroup(b, grid, n_pass, n_rt, nkb, nb):
    """-> list of rounds; a round = (list of (wave, pass, kb0, length, part, tile slot), tiles requested behind it, pieces)"""
    rho = (b & 7) * (grid >> 3) + (b >> 3) if grid % 8 == 0 else b
    q_lo, q_rem = divmod(n_pass, grid)
    p_first = rho * q_lo + min(rho, q_rem)
    q = q_lo + (1 if rho < q_rem else 0)
    t0 = p_first // n_rt
    full, left = q >> 2, q & 3
    n_rounds = full + (1 if left else 0)

    def last_tile(r):
        return (p_first + 4 * r + (4 if r < full else left) - 1) // n_rt

    issued_hi = t0 - 1
    rounds, loads0 = [], []
    if n_rounds:
        while issued_hi < last_tile(0):
            issued_hi += 1
            loads0.append((issued_hi, (issued_hi - t0) % nb))
    for r in range(n_rounds):
        pf = p_first + 4 * r
        cnt_r = 4 if r < full else left
        pieces = 1 if (r < full or cnt_r == 3) else (4 if cnt_r == 1 else 2)
        work = []
        for wave in range(4):
            po, kb0, ln, part = wave, 0, nkb, 0
            if pieces == 4:
                bl, ex = nkb >> 2, nkb & 3
                po, part, ln, kb0 = 0, wave, bl + (1 if wave < ex else 0), wave * bl + min(wave, ex)
            elif pieces == 2:
                l0 = (nkb + 1) >> 1
                po, part = wave >> 1, wave & 1
                ln, kb0 = (nkb - l0, l0) if part else (l0, 0)
            if pieces > 1 or wave < cnt_r:
                p = pf + po
                tile = p // n_rt
                work.append((wave, p, kb0, ln, part, (tile - t0) % nb))
        loads = []
        if r + 1 < n_rounds:
            while issued_hi < last_tile(r + 1):
                issued_hi += 1
                loads.append((issued_hi, (issued_hi - t0) % nb))
        rounds.append((work, loads, pieces))
    return p_first, q, t0, loads0, rounds
