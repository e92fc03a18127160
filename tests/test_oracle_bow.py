"""Pins the oracle's DBoW2 transform and SearchByBoW restatements (oracle/bow.cpp) with literal numpy/pure-Python
re-derivations (the reference has no vectors for them: parity unpinned, see oracle/bow.h)."""
import numpy as np
import pytest
from viorb_amd.synth import make_vocabulary, descriptors_near_words

POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def ham(a, b):
    return int(POP[np.bitwise_xor(a, b)].sum())


def descend(voc, d, levelsup):
    node, cur, lvl, nid = 0, 0, 0, 0
    nid_level = voc["L"] - levelsup
    while True:
        lvl += 1
        ch = voc["child_ids"][voc["child_start"][cur]:voc["child_start"][cur + 1]]
        dist = [ham(d, voc["desc"][c]) for c in ch]
        cur = int(ch[int(np.argmin(dist))])             # argmin = first minimum, as the strict '<' of the reference
        if lvl == nid_level:
            nid = cur
        if voc["child_start"][cur + 1] == voc["child_start"][cur]:
            break
    return voc["word_id"][cur], voc["weight"][cur], nid


@pytest.mark.parametrize("k,L,levelsup", [(10, 4, 2), (8, 5, 4), (3, 6, 4), (10, 3, 4)])
def test_transform_matches_literal_descent(oracle, k, L, levelsup):
    voc = make_vocabulary(1, k, L)
    assert len(voc["word_id"]) == sum(k ** l for l in range(L + 1)) and (voc["word_id"] >= 0).sum() == k ** L
    desc = np.concatenate([descriptors_near_words(2, voc, 150), np.random.default_rng(3).integers(0, 256, (50, 32), dtype=np.uint8)])
    r = oracle.bow_transform(voc, desc, levelsup)
    bow = {}
    for i, d in enumerate(desc):
        w, wt, nid = descend(voc, d, levelsup)
        assert (r["word"][i], r["weight"][i], r["node"][i]) == (w, wt, nid)
        if wt > 0:
            bow[int(w)] = bow.get(int(w), 0.0) + wt
    keys = sorted(bow)
    norm = 0.0
    for kk in keys:
        norm += abs(bow[kk])
    assert list(r["bow_ids"]) == keys
    assert np.array_equal(r["bow_vals"], np.array([bow[kk] / norm for kk in keys]))
    if L - levelsup <= 0:
        assert (r["node"] == 0).all()


def literal_search(kd, ka, kn, kh, fd, fa, fn, ratio, ori):
    nF = len(fd); match = [-1] * nF; hist = [[] for _ in range(30)]; nm = 0
    nodes = sorted(set(int(x) for x in kn if x >= 0) & set(int(x) for x in fn if x >= 0))
    for nd in nodes:
        for i in np.nonzero(kn == nd)[0]:
            if not kh[i]:
                continue
            b1, b2, bi = 256, 256, -1
            for j in np.nonzero(fn == nd)[0]:
                if match[j] >= 0:
                    continue
                d = ham(kd[i], fd[j])
                if d < b1:
                    b2, b1, bi = b1, d, j
                elif d < b2:
                    b2 = d
            if b1 <= 50 and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                match[bi] = int(i); nm += 1
                if ori:
                    rot = np.float32(ka[i]) - np.float32(fa[bi])
                    if rot < 0:
                        rot = np.float32(rot + np.float32(360.0))
                    v = float(np.float32(rot * np.float32(1.0 / 30)))
                    b = int(np.floor(v + 0.5))
                    hist[0 if b == 30 else b].append(bi)
    if ori:
        cnt = [len(h) for h in hist]
        order = sorted(range(30), key=lambda i: (-cnt[i], i))
        m1, m2, m3 = cnt[order[0]], cnt[order[1]], cnt[order[2]]
        keep = [order[0]] if m1 > 0 else []
        if m1 > 0 and m2 > 0 and not (m2 < 0.1 * m1):
            keep.append(order[1])
            if m3 > 0 and not (m3 < 0.1 * m1):
                keep.append(order[2])
        for b in range(30):
            if b not in keep:
                for j in hist[b]:
                    match[j] = -1; nm -= 1
    return nm, np.array(match, np.int32)


@pytest.mark.parametrize("seed,ori", [(0, True), (1, False), (2, True)])
def test_search_by_bow_matches_literal(oracle, seed, ori):
    rng = np.random.default_rng(seed)
    voc = make_vocabulary(5, 6, 5)
    nK, nF = 300, 340
    kd = descriptors_near_words(seed, voc, nK, 4)
    # frame descriptors: noisy copies of a subset of the key frame's + fresh ones, shuffled
    src = rng.permutation(nK)[:220]
    fd = kd[src].copy()
    for _ in range(10):
        b = rng.integers(0, 256, len(fd)); fd[np.arange(len(fd)), b >> 3] ^= (1 << (b & 7)).astype(np.uint8)
    fd = np.concatenate([fd, descriptors_near_words(seed + 50, voc, nF - len(fd), 4)])
    ka = rng.uniform(0, 360, nK).astype(np.float32)
    fa = np.concatenate([ka[src] + rng.choice([5.0, 5.0, 5.0, 130.0], len(src)) + rng.normal(0, 3, len(src)), rng.uniform(0, 360, nF - len(src))]).astype(np.float32) % np.float32(360)
    perm = rng.permutation(nF); fd, fa = fd[perm], fa[perm]
    kn = oracle.bow_transform(voc, kd)["node"].copy(); fn = oracle.bow_transform(voc, fd)["node"].copy()
    kn[rng.random(nK) < 0.02] = -1                       # stopped words: not in the FeatureVector
    kh = (rng.random(nK) < 0.8).astype(np.uint8)
    n, m = oracle.search_by_bow(kd, ka, kn, kh, fd, fa, fn, 0.7, ori)
    n2, m2 = literal_search(kd, ka, kn, kh, fd, fa, fn, 0.7, ori)
    assert n == n2 and np.array_equal(m, m2)
    assert n > 60                                        # the planted correspondences are found
