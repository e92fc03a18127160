"""Brute-force Hamming matcher (north_star "Hamming brute-force", SURVEY.md §8b viorb_match_bruteforce): the oracle's scan against
a numpy bit-count definition (CPU), and the HIP kernel against the oracle through the C ABI (GPU)."""
import ctypes as C
import numpy as np
import pytest


def _numpy_bruteforce(q, c):
    """Definition written from scratch: distance matrix by bit counts, first minimum, second smallest value of the scan
    'if d < b1: b2 = b1; b1 = d elif d < b2: b2 = d' (the second-smallest distance counted with multiplicity)."""
    if len(c) == 0:
        return np.full(len(q), 256, np.int32), np.full(len(q), 256, np.int32), np.full(len(q), -1, np.int32)
    x = q[:, None, :] ^ c[None, :, :]
    d = np.unpackbits(x, axis=2).sum(2).astype(np.int32)
    idx = d.argmin(1).astype(np.int32)                  # first minimum
    best = d.min(1)
    ds = np.sort(np.concatenate([d, np.full((len(q), 1), 256, np.int32)], 1), 1)
    second = np.minimum(ds[:, 1], 256).astype(np.int32)
    return best.astype(np.int32), second, idx


def _cases():
    rng = np.random.default_rng(42)
    out = []
    for nq, nc in ((1, 1), (5, 0), (7, 3), (64, 64), (257, 255), (300, 513), (1000, 1000)):
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8); c = rng.integers(0, 256, (nc, 32), dtype=np.uint8)
        if nc > 10:                                        # exact duplicates and near duplicates: ties on the minimum
            c[5] = c[2]; q[0] = c[2]; c[nc - 1] = c[2]
            if nq > 3:
                q[3] = c[7]; q[3, 0] ^= 1
        out.append((q, c))
    return out


def test_oracle_bruteforce_is_the_definition(oracle):
    for q, c in _cases():
        b, s, i = oracle.match_bruteforce(q, c)
        wb, ws, wi = _numpy_bruteforce(q, c)
        assert np.array_equal(b, wb) and np.array_equal(s, ws) and np.array_equal(i, wi)


@pytest.mark.gpu
def test_gpu_bruteforce_equals_oracle(oracle):
    import viorb_amd
    for q, c in _cases():
        b, s, i = viorb_amd.match_bruteforce(q, c)
        wb, ws, wi = oracle.match_bruteforce(q, c)
        assert np.array_equal(b, wb) and np.array_equal(s, ws) and np.array_equal(i, wi), (len(q), len(c))


@pytest.mark.gpu
def test_gpu_bruteforce_batched_on_extracted_descriptors(oracle):
    """Device form on real ORB descriptors of consecutive synthetic frames (ragged counts, capacity rows, non-default stream)."""
    import torch
    import viorb_amd
    from viorb_amd.capi import lib, check
    from viorb_amd.synth import make_vi_stream
    s = make_vi_stream(3, 3)
    ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=3)
    dev = torch.device("cuda", 0)
    ex.extract_batch_device(torch.from_numpy(np.ascontiguousarray(s["frames"][:3])).to(dev))
    torch.cuda.synchronize()
    kps, desc, count, status, cap = ex.results_device()
    n = np.array([ex.download(b)[0].shape[0] for b in range(3)], np.int32)
    d_host = [ex.download(b)[1] for b in range(3)]
    # queries: frames 0,1,2; candidates: frames 1,2,0 (rolled copy on the device)
    dd = torch.empty((3, cap, 32), dtype=torch.uint8, device=dev)
    check(lib().viorb_memcpy_dtod_async(C.c_void_p(dd.data_ptr()), C.c_void_p(desc), 3 * cap * 32, None))
    cand = torch.roll(dd, -1, 0).contiguous()
    ncand = torch.from_numpy(np.roll(n, -1)).to(dev)
    best, second, idx = (torch.full((3, cap), -7, dtype=torch.int32, device=dev) for _ in range(3))
    st = torch.cuda.Stream(device=dev)
    st.wait_stream(torch.cuda.current_stream(dev))
    check(lib().viorb_match_bruteforce_device(C.c_void_p(desc), C.c_void_p(count), cap, C.c_void_p(cand.data_ptr()), C.c_void_p(ncand.data_ptr()),
                                              cap, 3, C.c_void_p(best.data_ptr()), C.c_void_p(second.data_ptr()), C.c_void_p(idx.data_ptr()),
                                              C.c_void_p(st.cuda_stream)))
    st.synchronize()
    for b in range(3):
        wb, ws, wi = oracle.match_bruteforce(d_host[b], d_host[(b + 1) % 3])
        assert np.array_equal(best[b, :n[b]].cpu().numpy(), wb) and np.array_equal(second[b, :n[b]].cpu().numpy(), ws)
        assert np.array_equal(idx[b, :n[b]].cpu().numpy(), wi)
        assert (best[b, n[b]:].cpu().numpy() == -7).all()                  # rows beyond the count are not written
        assert (wb < 60).mean() > 0.3                                       # consecutive frames: many true matches


def test_bruteforce_no_cpu_fallback():
    import viorb_amd
    if viorb_amd.lib().viorb_device_count() > 0:
        pytest.skip("a HIP device is present")
    q = np.zeros((2, 32), np.uint8)
    with pytest.raises(viorb_amd.ViorbError) as e:
        viorb_amd.match_bruteforce(q, q)
    assert e.value.code == -2
