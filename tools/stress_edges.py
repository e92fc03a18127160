"""Degenerate inputs through the host-buffer drop-ins against the oracle: empty and one-element keypoint / point / observation sets,
a frame that sees nothing of the last one. `python tools/stress_edges.py` on a GPU box; a development aid."""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as oracle
oracle.lib()
import viorb_amd
from viorb_amd import frontend as fe
from viorb_amd.capi import KP_DTYPE
from viorb_amd.synth import make_periodic_stream, make_two_view_problem, plane_points_f32, cam_pose_from_navstate, local_points_f32, make_vocabulary
fails = 0


def check(name, f):
    global fails
    try:
        f(); print("ok   ", name)
    except viorb_amd.capi.ViorbError as e:
        print("refused", name, str(e)[:120])
    except AssertionError as e:
        fails += 1; print("FAIL ", name, str(e)[:200])
    except Exception as e:
        fails += 1; print("ERROR", name, repr(e)[:200]); traceback.print_exc(limit=2)


s = make_periodic_stream(5, 3)
ex = viorb_amd.ORBextractor(1000, 1.2, 8, 20, 7); tab = ex.tables()
k0, d0 = ex(s["frames"][0]); k1, d1 = ex(s["frames"][1])
Pw0 = plane_points_f32(np.stack([k0["x"], k0["y"]], 1), s["pose_true"][0], s["cam"])
flags0 = np.full(len(k0), 5, np.uint8)
Rcw, tcw = cam_pose_from_navstate(s["ns_true"][1], s["cam"]); pose12 = np.concatenate([Rcw.ravel(), tcw]).astype(np.float32)
bounds = (0.0, 752.0, 0.0, 480.0)
M = fe.ORBmatcher(0.9, True)


def sbp(kc, dc, kl, dl, fl, Pw, th=15.0):
    n, m = M.SearchByProjection(kc, dc, bounds, pose12, s["cam"][:4], tab["scale"], kl, fl, Pw, dl, th)
    nr, mr = oracle.search_by_projection_frame(kc, dc, bounds, pose12, s["cam"][:4], tab["scale"], fl, Pw, dl, kl["octave"], kl["angle"], th)
    assert n == nr and np.array_equal(m, mr), (n, nr)


e_k, e_d = np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
check("search_by_projection: full", lambda: sbp(k1, d1, k0, d0, flags0, Pw0))
check("search_by_projection: no current keypoints", lambda: sbp(e_k, e_d, k0, d0, flags0, Pw0))
check("search_by_projection: one current keypoint", lambda: sbp(k1[:1], d1[:1], k0, d0, flags0, Pw0))
check("search_by_projection: no last keypoints", lambda: sbp(k1, d1, e_k, e_d, np.zeros(0, np.uint8), np.zeros((0, 3), np.float32)))
check("search_by_projection: one last keypoint", lambda: sbp(k1, d1, k0[:1], d0[:1], flags0[:1], Pw0[:1]))
check("search_by_projection: no map points on the last frame", lambda: sbp(k1, d1, k0, d0, np.zeros(len(k0), np.uint8), Pw0))
check("search_by_projection: all points behind the camera", lambda: sbp(k1, d1, k0, d0, flags0, (Pw0 * np.float32([1, 1, -1])).astype(np.float32)))


def local(pts_f, pfl, pdesc, kc=k1, dc=d1):
    log_sf = np.float32(np.log(np.float64(tab["scale"][1])))
    owner = np.zeros(len(kc), np.uint8)
    n, m = fe.SearchLocalPoints(kc, dc, bounds, pose12, s["cam"][:4], tab["scale"], pts_f, pfl, pdesc, 1.0, 0.8, owner)
    nr, mr, _ = oracle.search_local_points(kc, dc, bounds, pose12, s["cam"][:4], tab["scale"], log_sf, pts_f, pfl, pdesc, 1.0, 0.8, owner)
    assert n == nr and np.array_equal(m, mr), (n, nr)


pf0 = local_points_f32(k0["octave"], s["pose_true"][0], Pw0, tab["scale"])
check("search_local_points: full", lambda: local(pf0, flags0, d0))
check("search_local_points: one point", lambda: local(pf0[:1], flags0[:1], d0[:1]))
check("search_local_points: no current keypoints", lambda: local(pf0, flags0, d0, e_k, e_d))
check("undistort: one point", lambda: np.testing.assert_array_equal(fe.UndistortKeyPoints(np.float32([[10, 20]]), np.float32(s["cam"][:4]), np.float32([-0.28, 0.07, 0.0002, 1e-5, 0])),
                                                                     oracle.undistort_points(np.float32([[10, 20]]), np.float32(s["cam"][:4]), np.float32([-0.28, 0.07, 0.0002, 1e-5, 0]))))
voc = make_vocabulary(3, 6, 4)
V = fe.ORBVocabulary(voc)


def bow(desc):
    ref = oracle.bow_transform(voc, desc, 2); w, wt, nd = V.transform_features(desc, 2)
    assert np.array_equal(w, ref["word"]) and np.array_equal(nd, ref["node"])


check("bow transform: one descriptor", lambda: bow(d0[:1]))
check("bow transform: 3 descriptors", lambda: bow(d0[:3]))
p = make_two_view_problem(3, 50, 60, 20)


def tri(sel1, sel2):
    a = (p["k1"][sel1], p["d1"][sel1], p["hp1"][sel1], p["ur1"][sel1], p["node1"][sel1], p["k2"][sel2], p["d2"][sel2], p["hp2"][sel2], p["ur2"][sel2], p["node2"][sel2],
         p["F12"], p["Cw1"], p["pose2"], p["intr4"], p["sf"], p["level_sigma2"], False, True)
    nr, mr = oracle.search_for_triangulation(*a); n, m = fe.SearchForTriangulation(*a)
    assert n == nr and np.array_equal(m, mr)


check("triangulation: one keypoint each", lambda: tri(slice(0, 1), slice(0, 1)))
check("triangulation: one against many", lambda: tri(slice(0, 1), slice(None)))
print("failures", fails)
