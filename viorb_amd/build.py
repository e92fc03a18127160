"""Builds viorb_amd/libviorb_hip.so (all HIP kernels + the C ABI of include/viorb.h) for gfx950.
hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "libviorb_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         # every float a*b+c in the kernels is two roundings unless fma() is written out: bit-exact
         # parity with the CPU oracle depends on it
         "-ffp-contract=off",
         # Machine LICM hoists the f64 literal pairs of every inlined sin/cos/atan/sqrt expansion out of the solvers' loops and keeps them
         # live across the whole kernel (k_pose_opt_vi: 375 registers per lane with it, 194 without, and at a 256 cap the "constants"
         # were spilled to scratch and reloaded inside the polynomial chains). Nothing else in the library gains from it.
         "-mllvm", "-disable-machine-licm"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.cpp")))


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(HERE, "csrc", "*.inc")) \
        + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + os.environ.get("VIORB_HIPCC_FLAGS", "").split() + ["-I", os.path.join(ROOT, "include"), "-o", SO] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force=True, verbose=True)
