"""viorb_amd — MI355X-native per-frame front-end for VIORB (ORB extract / match / IMU / pose solve).

The product is libviorb_hip.so (HIP kernels behind the C ABI of include/viorb.h). This package is the
thin Python mirror of the reference's C++ interface used by tests and bench.py; it has no CPU
fallback: if the library is missing, `viorb_amd.lib()` raises.
"""
from .capi import lib, ViorbError, KP_DTYPE  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .frontend import Frontend, ORBmatcher, PoseOptimization, PoseOptimizationSE3, LocalBundleAdjustmentNavState, LocalBundleAdjustmentNavStateBatch, LocalBundleAdjustment, LocalBundleAdjustmentBatch, ORBVocabulary, SearchByBoW, SearchForTriangulation, Fuse, preintegrate, descriptor_distance, match_bruteforce, SearchLocalPoints, UndistortKeyPoints, ComputeImageBounds  # noqa: F401
