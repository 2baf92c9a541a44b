"""CPU model of the stream-K schedule of k_gemm9_tn<EPI, 2> (veritasfi_amd/csrc/vf_transformer.hip): the same integer arithmetic, checked
for the properties the kernel relies on -- every K-tile of every tile is computed exactly once, every segment is at least two K-tiles
long, a workgroup has at most one dump (walked first) and one finish (walked last), a dump counts on exactly the workgroup that finishes
its tile, a finisher waits for exactly the workgroups below it that hold the rest of its tile, and the tile order is a permutation."""
import itertools

import pytest


def bound(w, nx, nk, GL):
    b = w * (nx * nk) // GL
    r = b % nk
    if r == 1:
        b -= 1
    elif r == nk - 1:
        b += 1
    return b


def tile_of(tl, nx, GL, base):
    R = (nx + GL - 1) // GL
    lastc = nx - (R - 1) * GL
    Rm1 = R - 1 if R > 1 else 1
    if tl < lastc * R:
        col, row = tl // R, tl % R
    else:
        v2 = tl - lastc * R
        col = lastc + v2 // Rm1
        row = v2 - (col - lastc) * Rm1
    return base + row * GL + col


def schedule(Mt, Nt, nk, G):
    ntiles, GL = Mt * Nt, G // 8
    q8, r8 = ntiles >> 3, ntiles & 7
    cover = {}
    counts_on, waits_for, dumps = {}, {}, {}
    for wg in range(G):
        xcd, wl = wg & 7, wg >> 3
        nx = q8 + (1 if xcd < r8 else 0)
        base = xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8
        lo, hi = bound(wl, nx, nk, GL), bound(wl + 1, nx, nk, GL)
        assert lo <= hi
        if hi == lo:
            continue
        first_t, last_t = lo // nk, (hi - 1) // nk
        n_items = last_t - first_t + 1
        dump0 = hi % nk != 0
        fin = lo % nk != 0 and not (n_items == 1 and dump0)
        for idx in range(n_items):
            tl = last_t - idx
            s, e = max(lo, tl * nk), min(hi, (tl + 1) * nk)
            assert e - s >= 2, ("short segment", wg, tl, s, e)
            pt = tile_of(tl, nx, GL, base)
            assert base <= pt < base + nx
            is_dump = e != (tl + 1) * nk
            is_fin = (not is_dump) and s != tl * nk
            assert is_dump == (dump0 and idx == 0)
            assert is_fin == (fin and idx == n_items - 1)
            for k in range(s - tl * nk, e - tl * nk):
                assert (pt, k) not in cover
                cover[(pt, k)] = wg
            if is_dump:
                tile_end, f = (last_t + 1) * nk, wl + 1
                while bound(f + 1, nx, nk, GL) < tile_end:
                    f += 1
                counts_on[wg] = f * 8 + xcd
                dumps[wg] = pt
            if is_fin:
                tile_start, P = first_t * nk, 1
                while bound(wl - P, nx, nk, GL) > tile_start:
                    P += 1
                waits_for[wg] = (P, pt)
    return ntiles, cover, counts_on, waits_for, dumps


@pytest.mark.parametrize("Mt,Nt,nk", [(26, 12, 12), (50, 12, 12), (50, 3, 48), (100, 3, 48), (200, 3, 48), (26, 9, 12), (27, 3, 48),
                                        (200, 12, 12), (13, 16, 16), (26, 4, 64), (31, 5, 24), (8, 8, 24), (33, 7, 12)])
def test_stream_k_schedule_covers_every_k_tile_once(Mt, Nt, nk):
    G = 256
    if (Mt * Nt // 8) * nk // (G // 8) < 6:
        pytest.skip("below the host's gate")
    ntiles, cover, counts_on, waits_for, dumps = schedule(Mt, Nt, nk, G)
    assert len(cover) == ntiles * nk
    assert set(cover) == set(itertools.product(range(ntiles), range(nk)))
    # the finisher of tile pt hears from exactly the dumpers of pt, all of them below it on its XCD
    for f, (P, pt) in waits_for.items():
        ds = sorted(d for d, t in dumps.items() if t == pt)
        assert len(ds) == P, (f, P, ds)
        assert all(counts_on[d] == f for d in ds)
        assert ds == [f - 8 * j for j in range(P, 0, -1)]
    assert set(counts_on.values()) <= set(waits_for)
    assert sum(P for P, _ in waits_for.values()) == len(dumps)
