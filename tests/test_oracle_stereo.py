"""Pins the oracle's Frame::ComputeStereoMatches restatement (oracle/orb_stereo.cpp) with a literal numpy
re-statement on a small pair, plus geometric sanity against the known synthetic disparity field."""
import numpy as np
from viorb_amd.synth import make_stereo_pair, KITTI_K


def literal_stereo(kl, dl, kr, dr, pl, pr, sf, isf, bf, fx):
    f = np.float32
    N = len(kl); u = np.full(N, -1, np.float32); dep = np.full(N, -1, np.float32)
    nrows = pl[0].shape[0]
    rows = [[] for _ in range(nrows)]
    for iR, k in enumerate(kr):
        r = f(f(2.0) * sf[k["octave"]])
        for yi in range(int(np.floor(f(k["y"] - r))), int(np.ceil(f(k["y"] + r))) + 1):
            if 0 <= yi < nrows: rows[yi].append(iR)
    mb = f(f(bf) / f(fx)); maxD = f(f(bf) / mb)
    pairs = []
    for iL, k in enumerate(kl):
        cand = rows[int(k["y"])]
        if not cand: continue
        minU, maxU = f(k["x"] - maxD), f(k["x"])
        if maxU < 0: continue
        best, bi = 100, 0
        for iR in cand:
            q = kr[iR]
            if q["octave"] < k["octave"] - 1 or q["octave"] > k["octave"] + 1: continue
            if minU <= q["x"] <= maxU:
                dist = int(np.unpackbits(dl[iL] ^ dr[iR]).sum())
                if dist < best: best, bi = dist, iR
        if best >= 75: continue
        sc = isf[k["octave"]]
        rnd = lambda v: float(np.floor(f(v) + f(0.5)))          # round() of a non-negative float
        cu, cv, cr = int(rnd(f(k["x"] * sc))), int(rnd(f(k["y"] * sc))), int(rnd(f(kr[bi]["x"] * sc)))
        IL, IR = pl[k["octave"]].astype(np.int32), pr[k["octave"]].astype(np.int32)
        if cr < 0 or cr + 11 >= IR.shape[1]: continue
        PL = IL[cv - 5:cv + 6, cu - 5:cu + 6] - IL[cv, cu]
        dists = []
        for inc in range(-5, 6):
            PR = IR[cv - 5:cv + 6, cr + inc - 5:cr + inc + 6] - IR[cv, cr + inc]
            dists.append(int(np.abs(PL - PR).sum()))
        binc = int(np.argmin(dists)) - 5
        if binc in (-5, 5): continue
        d1, d2, d3 = f(dists[binc + 4]), f(dists[binc + 5]), f(dists[binc + 6])
        with np.errstate(divide="ignore", invalid="ignore"):
            delta = f(f(d1 - d3) / f(f(2.0) * f(f(d1 + d3) - f(f(2.0) * d2))))
        if not (delta >= -1 and delta <= 1): continue
        bu = f(sf[k["octave"]] * f(f(f(cr) + f(binc)) + delta))
        disp = f(k["x"] - bu)
        if disp >= 0 and disp < maxD:
            if disp <= 0: disp = f(0.01); bu = f(k["x"] - f(0.01))
            dep[iL] = f(f(bf) / disp); u[iL] = bu; pairs.append((min(dists), iL))
    if pairs:
        pairs.sort()
        th = f(f(1.5) * f(1.4)) * f(pairs[len(pairs) // 2][0])
        for sd, iL in pairs:
            if not (f(sd) < th): u[iL] = -1; dep[iL] = -1
    return u, dep


def test_stereo_oracle_equals_literal_restatement(oracle):
    left, right, disp = make_stereo_pair(21, 480, 320)
    el, er = oracle.Extractor(600), oracle.Extractor(600)
    kl, dl = el(left); kr, dr = er(right)
    u, d, sad = oracle.stereo_match(el, er, kl, dl, kr, dr, KITTI_K["bf"], KITTI_K["fx"])
    t = el.tables()
    pl = [el.level(l) for l in range(8)]; pr = [er.level(l) for l in range(8)]
    wu, wd = literal_stereo(kl, dl, kr, dr, pl, pr, t["scale"], t["inv_scale"], KITTI_K["bf"], KITTI_K["fx"])
    np.testing.assert_array_equal(u, wu)
    np.testing.assert_array_equal(d, wd)
    m = u >= 0
    assert m.sum() > 0.3 * len(kl)
    # geometry: the recovered disparity follows the synthetic field (right(x) = left(x + d))
    est = kl["x"][m] - u[m]
    true = disp[np.clip(kl["y"][m].astype(int), 0, 319), np.clip(u[m].astype(int), 0, 479)]
    assert np.median(np.abs(est - true)) < 1.0
    assert (d[m] > 0).all() and (u[~m] == -1).all() and (d[~m] == -1).all()
