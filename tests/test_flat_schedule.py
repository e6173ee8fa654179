"""The balanced launch of the full-size matrix-core kernel (k_mm8w_flat): invariants of its schedule on the index model
(tests/flat_schedule_model.py mirrors the kernel's arithmetic; the kernel itself is held to the oracle by the -m gpu tests)."""
import pytest

from flat_schedule_model import PART_Q, part_off, ring_slots, workgroup


@pytest.mark.parametrize("n_rt, nkb, n_tiles, grid", [
    (11, 11, 382, 256),      # config 5's shard, R2: 171 rows x 86 terms
    (6, 11, 382, 256),       # R1 / the 86 x 86 decode
    (4, 8, 2979, 256), (7, 9, 1000, 256), (5, 16, 300, 256), (13, 10, 999, 250), (4, 12, 257, 64), (9, 13, 129, 8),
])
def test_every_pass_once_and_tiles_resident(n_rt, nkb, n_tiles, grid):
    nb = ring_slots(n_rt, nkb)
    if nb == 0:
        pytest.skip("shape does not qualify")
    bufsz = nkb * 256
    n_pass = n_tiles * n_rt
    seen = {}
    for b in range(grid):
        p_first, q, t0, loads0, rounds = workgroup(b, grid, n_pass, n_rt, nkb, nb)
        resident = {}                       # slot -> tile
        for tile, slot in loads0:
            resident[slot] = tile
        covered = {}
        for ri, (work, loads, pieces) in enumerate(rounds):
            in_use = set()
            for wave, p, kb0, ln, part, slot in work:
                assert p_first <= p < p_first + q
                assert resident.get(slot) == p // n_rt, "a pass reads a tile that is not in its slot"
                in_use.add(slot)
                assert ln >= 2 and 0 <= kb0 and kb0 + ln <= nkb
                covered.setdefault(p, []).append((kb0, ln, part))
            # the next round's tiles are requested while this round's are being read: never into a slot in use
            for tile, slot in loads:
                assert slot not in in_use, "DMA into a slot a wave of this round still reads"
                resident[slot] = tile
            if pieces > 1:
                assert ri == len(rounds) - 1 and not loads
                busy = sorted(in_use)
                sa, sb = busy[0], busy[-1]
                regions = [part_off(k, sa, sb, bufsz) for k in range(pieces - 1 if pieces == 4 else 2)]
                for k, off in enumerate(regions):
                    assert off + PART_Q <= nb * bufsz + 128
                    for s in busy:
                        assert off >= (s + 1) * bufsz or off + PART_Q <= s * bufsz, "partial sums over a tile of the round"
                    for off2 in regions[:k]:
                        assert abs(off - off2) >= PART_Q
        for p, parts in covered.items():
            assert p not in seen
            seen[p] = b
            parts.sort()
            assert parts[0][0] == 0 and sum(ln for _, ln, _ in parts) == nkb
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(len(parts) - 1))
            assert [pt for _, _, pt in parts] == list(range(len(parts))), "piece 0 (the leader) takes the first K-blocks"
    assert len(seen) == n_pass


def test_config5_rounds():
    """R2 of config 5's shard: four full rounds and at most a quarter pass a wave; R1: two and a quarter"""
    for n_rt, want_full in ((11, 4), (6, 2)):
        nb = ring_slots(n_rt, 11)
        assert nb == 3
        worst = 0
        for b in range(256):
            _, q, _, _, rounds = workgroup(b, 256, 382 * n_rt, n_rt, 11, nb)
            assert q // 4 == want_full
            worst = max(worst, len(rounds))
            if q % 4 == 1:
                assert rounds[-1][2] == 4 and sorted(w[3] for w in rounds[-1][0]) == [2, 3, 3, 3]
        assert worst == want_full + 1
