"""-m gpu parity of Frame::ComputeStereoMatches (a14, KITTI-shaped config) against the oracle: mvuRight / mvDepth
bit-exact (integer Hamming + integer SAD + the same float expressions)."""
import numpy as np
import pytest
import viorb_amd
from viorb_amd.extractor import ComputeStereoMatches
from viorb_amd.synth import make_stereo_pair, KITTI_K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,w,h,nf", [(100, 1241, 376, 2000), (101, 1241, 376, 2000), (7, 752, 480, 1000),
                                         (102, 1241, 376, 5000)])      # 5000: more features than the sort / SAD arrays fit in LDS (k_stereo_match<true>)
def test_stereo_matches_equal_oracle(oracle, seed, w, h, nf):
    if viorb_amd.lib().viorb_device_count() < 1:
        pytest.fail("no HIP device visible")
    left, right, _ = make_stereo_pair(seed, w, h)
    gl, gr = viorb_amd.ORBextractor(nf, 1.2, 8, 20, 7), viorb_amd.ORBextractor(nf, 1.2, 8, 20, 7)
    kl, dl = gl(left); kr, dr = gr(right)
    ol, orr = oracle.Extractor(nf), oracle.Extractor(nf)
    okl, odl = ol(left); okr, odr = orr(right)
    np.testing.assert_array_equal(kl, okl); np.testing.assert_array_equal(kr, okr)
    u, d, n = ComputeStereoMatches(gl, gr, KITTI_K["bf"], KITTI_K["fx"])
    ou, od, osad = oracle.stereo_match(ol, orr, okl, odl, okr, odr, KITTI_K["bf"], KITTI_K["fx"])
    np.testing.assert_array_equal(u[:len(ou)], ou)
    np.testing.assert_array_equal(d[:len(od)], od)
    assert n == int((ou >= 0).sum()) and n > 0.4 * len(okl)
    assert (u[len(ou):] == -1).all()


def test_stereo_batched_one_handle(oracle):
    """Left and right images in one batched handle (images 0..1 = left, 2..3 = right)."""
    import torch, ctypes as C
    from viorb_amd.capi import lib, check, ptr
    pairs = [make_stereo_pair(s, 752, 480)[:2] for s in (11, 12)]
    imgs = torch.from_numpy(np.stack([p[0] for p in pairs] + [p[1] for p in pairs])).cuda()
    ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=4)
    ex.extract_batch_device(imgs)
    u = torch.zeros((2, ex.cap), dtype=torch.float32, device="cuda"); d = torch.zeros_like(u); n = torch.zeros(2, dtype=torch.int32, device="cuda")
    check(lib().viorb_stereo_match_device(ex.h, 0, ex.h, 2, 2, KITTI_K["bf"], KITTI_K["fx"], ptr(u), ptr(d), ptr(n), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    for p in range(2):
        ol, orr = oracle.Extractor(1000), oracle.Extractor(1000)
        okl, odl = ol(pairs[p][0]); okr, odr = orr(pairs[p][1])
        ou, od, _ = oracle.stereo_match(ol, orr, okl, odl, okr, odr, KITTI_K["bf"], KITTI_K["fx"])
        np.testing.assert_array_equal(u[p, :len(ou)].cpu().numpy(), ou)
        np.testing.assert_array_equal(d[p, :len(od)].cpu().numpy(), od)
        assert n[p].item() == int((ou >= 0).sum())
