"""GPU parity (bit-exact) of the DBoW2 tree descent and ORBmatcher::SearchByBoW (csrc/bow_matcher.hip) against the oracle."""
import ctypes as C
import numpy as np
import pytest
from viorb_amd.synth import make_vocabulary, descriptors_near_words

pytestmark = pytest.mark.gpu


def _kps(angles):
    from viorb_amd import KP_DTYPE
    k = np.zeros(len(angles), KP_DTYPE); k["angle"] = angles; k["class_id"] = -1
    return k


@pytest.mark.parametrize("k,L,levelsup,n", [(10, 4, 2, 700), (8, 5, 4, 1000), (3, 6, 4, 333), (10, 3, 4, 64), (17, 3, 1, 500), (10, 6, 4, 2000)])
def test_transform_bit_exact(oracle, k, L, levelsup, n):
    from viorb_amd import ORBVocabulary
    voc = make_vocabulary(11, k, L)
    desc = np.concatenate([descriptors_near_words(12, voc, n - n // 4), np.random.default_rng(13).integers(0, 256, (n // 4, 32), dtype=np.uint8)])
    ref = oracle.bow_transform(voc, desc, levelsup)
    V = ORBVocabulary(voc)
    word, weight, node = V.transform_features(desc, levelsup)
    assert np.array_equal(word, ref["word"]) and np.array_equal(weight, ref["weight"]) and np.array_equal(node, ref["node"])
    ids, vals, fnode = V.transform(desc, levelsup)
    assert np.array_equal(ids, ref["bow_ids"]) and np.array_equal(vals, ref["bow_vals"])
    assert np.array_equal(fnode, np.where(ref["weight"] > 0, ref["node"], -1))
    V.close()


def test_vocabulary_argument_checks():
    from viorb_amd import ORBVocabulary, ViorbError
    voc = make_vocabulary(1, 4, 3)
    bad = dict(voc); bad["child_ids"] = voc["child_ids"].copy(); bad["child_ids"][5] = len(voc["word_id"]) + 3
    with pytest.raises(ViorbError):
        ORBVocabulary(bad)
    bad = dict(voc); bad["child_start"] = voc["child_start"].copy(); bad["child_start"][3] = bad["child_start"][2] - 1
    with pytest.raises(ViorbError):
        ORBVocabulary(bad)


def _pair(oracle, voc, seed, nK, nF, n_common, noise):
    rng = np.random.default_rng(seed)
    kd = descriptors_near_words(seed, voc, nK, 4)
    src = rng.permutation(nK)[:n_common]
    fd = kd[src].copy()
    for _ in range(noise):
        b = rng.integers(0, 256, len(fd)); fd[np.arange(len(fd)), b >> 3] ^= (1 << (b & 7)).astype(np.uint8)
    fd = np.concatenate([fd, descriptors_near_words(seed + 50, voc, nF - len(fd), 4)])
    ka = rng.uniform(0, 360, nK).astype(np.float32)
    fa = np.concatenate([ka[src] + rng.choice([5.0, 5.0, 5.0, 130.0], len(src)) + rng.normal(0, 3, len(src)), rng.uniform(0, 360, nF - len(src))]).astype(np.float32) % np.float32(360)
    perm = rng.permutation(nF); fd, fa = fd[perm], fa[perm]
    kn = oracle.bow_transform(voc, kd)["node"].copy(); fn = oracle.bow_transform(voc, fd)["node"].copy()
    kn[rng.random(nK) < 0.02] = -1; fn[rng.random(nF) < 0.02] = -1
    kh = (rng.random(nK) < 0.8).astype(np.uint8)
    return kd, ka, kn, kh, fd, fa, fn


@pytest.mark.parametrize("seed,k,L,nK,nF,ori", [(0, 6, 5, 300, 340, True), (1, 6, 5, 1000, 1000, False), (2, 10, 6, 1000, 970, True),
                                                 (3, 10, 4, 500, 600, True), (4, 10, 6, 2000, 2000, True), (5, 4, 5, 64, 1, True),
                                                 (6, 8, 5, 8000, 7600, True)])      # more keypoints than the work arrays fit in LDS (k_search_by_bow<true>)
def test_search_by_bow_bit_exact(oracle, seed, k, L, nK, nF, ori):
    """(k=10, L=4: levelsup 4 puts every feature under the root, the search degenerates to greedy brute force.)"""
    from viorb_amd import SearchByBoW
    voc = make_vocabulary(20 + seed, k, L)
    kd, ka, kn, kh, fd, fa, fn = _pair(oracle, voc, seed, nK, nF, min(nK, nF) * 2 // 3, 10)
    n_ref, m_ref = oracle.search_by_bow(kd, ka, kn, kh, fd, fa, fn, 0.7, ori)
    n, m = SearchByBoW(_kps(ka), kd, kn, kh, _kps(fa), fd, fn, 0.7, ori)
    assert n == n_ref and np.array_equal(m, m_ref)
    if min(nK, nF) >= 300:
        assert n > 0.1 * min(nK, nF)


def test_search_by_bow_batched_device(oracle):
    """Device-resident batched form: 6 pairs of different sizes in one launch."""
    import torch
    from viorb_amd import lib, KP_DTYPE
    from viorb_amd.capi import check
    voc = make_vocabulary(31, 8, 5)
    cap, B = 1024, 6
    sizes = [(1000, 1000), (640, 1024), (1, 5), (0, 10), (10, 0), (1024, 333)]
    kk = np.zeros((B, cap), KP_DTYPE); fk = np.zeros((B, cap), KP_DTYPE)
    kd = np.zeros((B, cap, 32), np.uint8); fd = np.zeros((B, cap, 32), np.uint8)
    kn = np.full((B, cap), -1, np.int32); fn = np.full((B, cap), -1, np.int32); kh = np.zeros((B, cap), np.uint8)
    refs = []
    for b, (nK, nF) in enumerate(sizes):
        if nK == 0 or nF == 0:
            refs.append((0, np.full(nF, -1, np.int32))); continue
        a = _pair(oracle, voc, 40 + b, nK, nF, min(nK, nF) // 2, 8)
        kd[b, :nK], kk[b, :nK]["angle"], kn[b, :nK], kh[b, :nK] = a[0], a[1], a[2], a[3]
        fd[b, :nF], fk[b, :nF]["angle"], fn[b, :nF] = a[4], a[5], a[6]
        refs.append(oracle.search_by_bow(*a, 0.7, True))
    dev = torch.device("cuda", 0)
    up = lambda x: torch.from_numpy(x.view(np.uint8).reshape(-1).copy()).to(dev)
    t = [up(x) for x in (kk, kd, kn, kh)] + [torch.tensor([s[0] for s in sizes], dtype=torch.int32, device=dev)] + \
        [up(x) for x in (fk, fd, fn)] + [torch.tensor([s[1] for s in sizes], dtype=torch.int32, device=dev)]
    match = torch.full((B, cap), -7, dtype=torch.int32, device=dev); nm = torch.zeros(B, dtype=torch.int32, device=dev)
    check(lib().viorb_search_by_bow_device(*[C.c_void_p(x.data_ptr()) for x in t], cap, B, 0.7, 1, C.c_void_p(match.data_ptr()), C.c_void_p(nm.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    match, nm = match.cpu().numpy(), nm.cpu().numpy()
    for b, (nK, nF) in enumerate(sizes):
        assert nm[b] == refs[b][0] and np.array_equal(match[b, :nF], refs[b][1]) and (match[b, nF:] == -1).all()
