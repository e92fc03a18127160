"""-m gpu parity tests of the HIP extractor against the CPU oracle, stage by stage and end to end,
all through the C ABI (include/viorb.h). Bit-exact: this is integer/byte/index work plus float ops
that are individually rounded the same way on both sides."""
import os
import numpy as np
import pytest
import viorb_amd
from viorb_amd.synth import make_image, warp_image

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

CASES = {                      # name: (seed, w, h, nfeatures)
    "euroc0": (0, 752, 480, 1000),
    "euroc1": (1, 752, 480, 1000),
    "kitti": (100, 1241, 376, 2000),
    "synth720p": (1000, 1280, 720, 1500),
    "small": (5, 160, 120, 300),
    "odd": (9, 333, 257, 400),
}


@pytest.fixture(scope="module")
def gpu():
    if viorb_amd.lib().viorb_device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (and never fall back)")
    return True


def run_pair(oracle, seed, w, h, nf, **kw):
    img = make_image(seed, w, h)
    ex = viorb_amd.ORBextractor(nf, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    ox = oracle.Extractor(nf, 1.2, 8, 20, 7)
    okps, odesc = ox(img)
    return img, ex, kps, desc, ox, okps, odesc


@pytest.mark.parametrize("name", list(CASES))
def test_stage_parity(gpu, oracle, name):
    seed, w, h, nf = CASES[name]
    img, ex, kps, desc, ox, okps, odesc = run_pair(oracle, seed, w, h, nf)
    # a2 pyramid
    for l in range(8):
        np.testing.assert_array_equal(ex.level(l), ox.level(l), err_msg="pyramid level %d" % l)
    # a3 FAST candidates, order included
    for l in range(8):
        oc = ox.level_keypoints(l, candidates=True)
        gc = ex.debug_level_points(l, kept=False)
        want = np.stack([oc["x"], oc["y"], oc["response"]], 1).astype(np.int32).reshape(-1, 3)
        np.testing.assert_array_equal(gc, want, err_msg="FAST candidates level %d" % l)
    # a4 quadtree
    for l in range(8):
        ok = ox.level_keypoints(l)
        gk = ex.debug_level_points(l, kept=True)
        want = np.stack([ok["x"], ok["y"], ok["response"]], 1).astype(np.int32).reshape(-1, 3)
        np.testing.assert_array_equal(gk, want, err_msg="quadtree level %d" % l)
    # a6 blur (the oracle only blurs levels that have keypoints, like the reference)
    for l in range(8):
        ob = ox.level(l, blurred=True)
        if ob is not None:
            np.testing.assert_array_equal(ex.level(l, blurred=True), ob, err_msg="blur level %d" % l)
    # a5 + a7 + a8 final records
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        np.testing.assert_array_equal(kps[f], okps[f], err_msg=f)
    np.testing.assert_array_equal(kps["angle"], okps["angle"], err_msg="orientation (fastAtan2 of integer moments)")
    np.testing.assert_array_equal(desc, odesc)


@pytest.mark.parametrize("w,h", [(200, 97), (201, 130), (202, 257), (203, 96), (129, 100), (257, 131)])
def test_blur_borders_every_width_class(gpu, oracle, w, h):
    """The streaming blur patches the right image border by byte permutes that depend on w mod 4, and the strips / bands end at
    different lanes and rows for every size: all four classes, strip ends next to the border (129 = one pixel into a second
    32-dword strip, 257 = one pixel into a third), all levels."""
    img = make_image(77 + w, w, h)
    ex = viorb_amd.ORBextractor(300, 1.2, 8, 20, 7)
    ex(img)
    ox = oracle.Extractor(300, 1.2, 8, 20, 7)
    ox(img)
    checked = 0
    for l in range(8):
        ob = ox.level(l, blurred=True)
        if ob is not None:
            np.testing.assert_array_equal(ex.level(l, blurred=True), ob, err_msg="blur level %d of %dx%d" % (l, w, h))
            checked += 1
    assert checked >= 3


def test_golden_fixtures(gpu):
    for name in sorted(os.listdir(GOLD)):
        if not name.startswith("extract_"):
            continue
        g = np.load(os.path.join(GOLD, name))
        ex = viorb_amd.ORBextractor(int(g["nfeat"]), 1.2, 8, 20, 7)
        k, d = ex(make_image(int(g["seed"]), int(g["w"]), int(g["h"])))
        np.testing.assert_array_equal(k, g["kps"], err_msg=name)
        np.testing.assert_array_equal(d, g["desc"], err_msg=name)


def test_batched_device_api_matches_single(gpu, oracle):
    import torch
    B = 5
    imgs = np.stack([make_image(20 + b) for b in range(B)])
    ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=8)
    t = torch.from_numpy(imgs).cuda()
    ex.extract_batch_device(t)
    torch.cuda.synchronize()
    ox = oracle.Extractor(1000, 1.2, 8, 20, 7)
    for b in range(B):
        k, d = ex.download(b)
        ok, od = ox(imgs[b])
        np.testing.assert_array_equal(k, ok)
        np.testing.assert_array_equal(d, od)
    # second call on the same handle, fewer images, non-default stream, strided rows
    pad = torch.zeros((3, 480, 800), dtype=torch.uint8, device="cuda")
    pad[:, :, :752] = t[:3]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    ex.extract_batch_device(pad[:, :, :752], stream=s)
    s.synchronize()
    for b in range(3):
        k, d = ex.download(b)
        ok, od = ox(imgs[b])
        np.testing.assert_array_equal(k, ok)
        np.testing.assert_array_equal(d, od)


def test_batch_of_21_takes_the_sub_launch_and_side_stream_paths(gpu, oracle):
    """A batch of 16 or more goes through the XCD-aware placement with image offsets, FAST in several sub-launches (a batch that is not a
    multiple of 8 leaves a short last one) and the blur on the handle's second stream: every image must come out as the oracle's."""
    import torch
    B = 21
    imgs = np.stack([make_image(300 + b, 200, 150) for b in range(B)])
    ex = viorb_amd.ORBextractor(300, 1.2, 8, 20, 7, max_batch=B)
    ox = oracle.Extractor(300, 1.2, 8, 20, 7)
    t = torch.from_numpy(imgs).cuda()
    for _ in range(2):                                  # the second call re-uses the side stream and its events
        ex.extract_batch_device(t)
        torch.cuda.synchronize()
        for b in range(B):
            k, d = ex.download(b)
            ok, od = ox(imgs[b])
            np.testing.assert_array_equal(k, ok, err_msg="image %d" % b)
            np.testing.assert_array_equal(d, od, err_msg="image %d" % b)


def test_batch_of_72_images(gpu, oracle):
    """A batch with more (image, level) quadtrees than one round of workgroup slots and more FAST cells than one sub-launch: every image
    must come out as the oracle's, whatever ran beside it."""
    import torch
    B, D = 72, 6
    base = [make_image(500 + b, 320, 240) for b in range(D)]
    imgs = np.stack([base[b % D] for b in range(B)])
    ex = viorb_amd.ORBextractor(500, 1.2, 8, 20, 7, max_batch=B)
    ox = oracle.Extractor(500, 1.2, 8, 20, 7)
    ex.extract_batch_device(torch.from_numpy(imgs).cuda())
    torch.cuda.synchronize()
    ref = [ox(base[b]) for b in range(D)]
    for b in range(B):
        k, d = ex.download(b)
        np.testing.assert_array_equal(k, ref[b % D][0], err_msg="image %d" % b)
        np.testing.assert_array_equal(d, ref[b % D][1], err_msg="image %d" % b)


def test_level_with_more_quadtree_roots_than_a_quarter_of_its_quota(gpu, oracle):
    """592x158 with 172 features over 8 levels of 1.1: level 7 is 304x81 with 6 quadtree roots and a quota of 15; the unchecked first
    round keeps 24, the image returns 191 keypoints where sum(quota + 2) = 188 — the size-aware bound (viorb_extractor_max_keypoints_for)
    sizes the buffers, the result is the oracle's; a caller buffer of the generic size is an error, never a truncation."""
    import ctypes as C
    from viorb_amd.capi import ViorbError, KP_DTYPE, ptr, check
    img = make_image(7023, 592, 158)
    ok, od = oracle.Extractor(172, 1.1, 8, 12, 7)(img)
    ex = viorb_amd.ORBextractor(172, 1.1, 8, 12, 7)
    generic = ex.cap
    assert len(ok) > generic
    k, d = ex(img)
    np.testing.assert_array_equal(k, ok); np.testing.assert_array_equal(d, od)
    assert ex.cap >= len(ok) > generic
    kb, db, n = np.zeros(generic, KP_DTYPE), np.zeros((generic, 32), np.uint8), C.c_int()
    with pytest.raises(ViorbError):
        check(ex.L.viorb_extract(ex.h, ptr(img), 592, 158, img.strides[0], ptr(kb), ptr(db), generic, C.byref(n)))
    assert ex.capacity_for(752, 480) == generic                 # an ordinary size: the two bounds agree


def test_edge_inputs(gpu, oracle):
    ex = viorb_amd.ORBextractor(300, 1.2, 8, 20, 7)
    k, d = ex(np.full((120, 160), 77, np.uint8))               # textureless: zero keypoints everywhere
    assert len(k) == 0 and d.shape == (0, 32)
    rng = np.random.default_rng(1)
    noise = rng.integers(0, 256, (240, 320), dtype=np.uint8)     # maximal corner density
    ex2 = viorb_amd.ORBextractor(500, 1.2, 8, 20, 7)
    k, d = ex2(noise)
    ok, od = oracle.Extractor(500, 1.2, 8, 20, 7)(noise)
    np.testing.assert_array_equal(k, ok)
    np.testing.assert_array_equal(d, od)
    # image size change on the same handle re-configures
    k, d = ex2(make_image(3, 400, 300))
    ok, od = oracle.Extractor(500, 1.2, 8, 20, 7)(make_image(3, 400, 300))
    np.testing.assert_array_equal(k, ok)
    np.testing.assert_array_equal(d, od)


def test_two_threshold_cells(gpu, oracle):
    """Reference :797-807: a cell is detected at iniThFAST and, only when that leaves it empty, again at minThFAST. A low-contrast image
    (nearly every corner weaker than 20) takes the second pass in almost every cell; a half-and-half image mixes both kinds of cells in one frame."""
    base = make_image(11, 480, 360)
    weak = (base.astype(np.int32) // 8 + 100).astype(np.uint8)
    mixed = base.copy(); mixed[:, 240:] = weak[:, 240:]
    for img, nf in ((weak, 500), (mixed, 700)):
        k, d = viorb_amd.ORBextractor(nf, 1.2, 8, 20, 7)(img)
        ok, od = oracle.Extractor(nf, 1.2, 8, 20, 7)(img)
        assert len(ok) > 30
        np.testing.assert_array_equal(k, ok)
        np.testing.assert_array_equal(d, od)
    kw, _ = viorb_amd.ORBextractor(500, 1.2, 8, 20, 7)(weak)
    assert (kw["response"] < 20).mean() > 0.8, "most cells of the low-contrast image must come from the minThFAST pass"


def test_properties_full_size(gpu):
    """Size-independent checks at the bench configuration: determinism, bounds, and that a warped view
    of the same scene re-detects most features with small Hamming distance."""
    img = make_image(77)
    ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7)
    k1, d1 = ex(img)
    k2, d2 = ex(img)
    np.testing.assert_array_equal(k1, k2)
    np.testing.assert_array_equal(d1, d2)
    assert 900 <= len(k1) <= 1016 and (np.diff(k1["octave"]) >= 0).all()
    assert (k1["x"] >= 19).all() and (k1["x"] <= 752 - 19).all() and (k1["y"] >= 19).all() and (k1["y"] <= 480 - 19).all()
    kw, dw = ex(warp_image(img, 3.0, -2.0, 1.0, seed=3))
    bits = np.unpackbits(d1[:, None, :] ^ dw[None, :, :], axis=2).sum(axis=2)
    assert (bits.min(axis=1) <= 50).mean() > 0.5


@pytest.mark.parametrize("w,h,nf,sf,nl,ini,mn", [(640, 480, 800, 1.2, 1, 20, 7), (640, 480, 1200, 1.1, 12, 20, 7), (500, 375, 600, 1.5, 4, 30, 10),
                                                 (96, 80, 100, 1.2, 3, 20, 7), (131, 97, 150, 1.2, 8, 12, 5), (1280, 720, 1500, 1.2, 8, 20, 7),
                                                 # per-level quotas beyond what LDS holds: 3000 features over 8 levels (the global-scratch launch needs > 64 KB of LDS),
                                                 # 1728 on ONE level and 2393 on three (node list and sort keys in global scratch, every level through that launch)
                                                 (752, 480, 3000, 1.2, 8, 20, 7), (741, 663, 1728, 1.5, 1, 20, 7), (680, 567, 2393, 1.5, 3, 30, 5)])
def test_constructor_parameter_variations(gpu, oracle, w, h, nf, sf, nl, ini, mn):
    """Other pyramid depths / scale factors / thresholds / tiny images (one or two large FAST cells per level, vanishing upper
    levels) and the 1280x720 / 1500-feature configuration of BASELINE.json: keypoints and descriptors bit-exact."""
    img = make_image(900 + w + nl, w, h)
    k, d = viorb_amd.ORBextractor(nf, sf, nl, ini, mn)(img)
    ok, od = oracle.Extractor(nf, sf, nl, ini, mn)(img)
    np.testing.assert_array_equal(k, ok)
    np.testing.assert_array_equal(d, od)


@pytest.mark.parametrize("kind", ["sin5", "noise", "salt"])
def test_dense_corner_images(gpu, oracle, kind):
    """Images on which FAST fires almost everywhere: a 5-px two-dimensional sinusoid (48 % of the pixels are corners: more than the FAST
    kernel's corner list holds per cell, so its dense NMS path runs), uniform noise (26 % corners, every pre-test polarity combination)
    and sparse salt noise (isolated corners, cells that come back empty at iniThFAST and are redone at minThFAST). Candidate lists in
    cv::FAST order, the quadtree's selection and the final keypoints / descriptors must equal the oracle's."""
    w, h = (336, 256) if kind == "noise" else (400, 300)
    y, x = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(17)
    if kind == "sin5":
        img = np.clip(128 + 63 * np.sin(2 * np.pi * x / 5) + 63 * np.sin(2 * np.pi * y / 5), 0, 255).astype(np.uint8)
    elif kind == "noise":
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        img = np.full((h, w), 90, np.uint8)
        img[rng.integers(0, h, 600), rng.integers(0, w, 600)] = rng.integers(100, 125, 600)       # weak dots: only minThFAST sees most of them
        img[rng.integers(0, h, 60), rng.integers(0, w, 60)] = 255
    ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    ox = oracle.Extractor(1000, 1.2, 8, 20, 7)
    okps, odesc = ox(img)
    for l in range(8):
        oc = ox.level_keypoints(l, candidates=True)
        want = np.stack([oc["x"], oc["y"], oc["response"]], 1).astype(np.int32).reshape(-1, 3)
        np.testing.assert_array_equal(ex.debug_level_points(l, kept=False), want, err_msg="%s: FAST candidates level %d" % (kind, l))
    assert len(kps) == len(okps) and (kps == okps).all() and (desc == odesc).all()


@pytest.mark.parametrize("w,h,nf", [(752, 480, 1000), (1280, 720, 1500)])
def test_uniform_noise_images_have_no_candidate_cap(gpu, oracle, w, h, nf):
    """Uniform noise at the two bench sizes: tens of thousands of FAST candidates per level (the reference has no limit: its
    vToDistributeKeys.reserve(nfeatures * 10) is only a reserve, src/ORBextractor.cc:779) — far beyond the 8192 the LDS quadtree
    holds, so the over-size levels take the global-scratch instantiation of the same kernel. Everything must still equal the oracle."""
    img = np.random.default_rng(w).integers(0, 256, (h, w), dtype=np.uint8)
    ex = viorb_amd.ORBextractor(nf, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    ox = oracle.Extractor(nf, 1.2, 8, 20, 7)
    okps, odesc = ox(img)
    ncand = [len(ox.level_keypoints(l, candidates=True)) for l in range(8)]
    assert max(ncand) > 20000 and sum(n > 8192 for n in ncand) >= 2, ncand
    for l in range(8):
        oc = ox.level_keypoints(l, candidates=True)
        want = np.stack([oc["x"], oc["y"], oc["response"]], 1).astype(np.int32).reshape(-1, 3)
        np.testing.assert_array_equal(ex.debug_level_points(l, kept=False), want, err_msg="FAST candidates level %d" % l)
        ok = ox.level_keypoints(l)
        wantk = np.stack([ok["x"], ok["y"], ok["response"]], 1).astype(np.int32).reshape(-1, 3)
        np.testing.assert_array_equal(ex.debug_level_points(l, kept=True), wantk, err_msg="quadtree level %d" % l)
    assert len(kps) == len(okps) and (kps == okps).all() and (desc == odesc).all()


@pytest.mark.parametrize("tiers,node_x10", [("0", 50), ("1:512,2:256", 25), ("3:2048,5:1280", 25), ("0:3072,3:2048", 10), ("3:2048", 12)])
def test_quadtree_level_tiers_fall_through_to_the_full_capacity_launch(gpu, oracle, monkeypatch, tiers, node_x10):
    """The higher pyramid levels take their own first quadtree launches with smaller LDS plans (VIORB_OCT_TIERS, default "3:2048"); a level with
    more candidates than its plan holds must fall through to the full-capacity launch with the same result. "1:512,2:256" puts every level of a
    752x480 frame above its plan, "0" is the single first launch. The first launches' node lists are sized VIORB_OCT_NODE_X10 / 10 x quota + 64
    (default 2.5 x; 5 x can never run out): a level that runs out of node slots is left to the full-capacity launch before anything of it
    is written — 1.0 x and 1.2 x make most levels do that."""
    monkeypatch.setenv("VIORB_OCT_TIERS", tiers)
    monkeypatch.setenv("VIORB_OCT_NODE_X10", str(node_x10))
    img, ex, kps, desc, ox, okps, odesc = run_pair(oracle, 3, 752, 480, 1000)
    assert len(kps) == len(okps) and len(kps) > 900
    for f in ("x", "y", "angle", "response", "octave"):
        np.testing.assert_array_equal(kps[f], okps[f], err_msg=f)
    np.testing.assert_array_equal(desc, odesc)
