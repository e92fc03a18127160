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


def _headers():
    return glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(HERE, "csrc", "*.inc")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def build(force=False, verbose=False, out=None, extra_flags=None, objdir=None, scratch=False):
    """One object per source (compiled in parallel, re-used while neither the source, a header nor the flags changed), then one link.
    `scratch`: recompile every object. `out` / `extra_flags` / `objdir`: an experiment build beside the product library (load it with VIORB_LIBRARY=...)."""
    out = out or SO
    if not force and out == SO and not needs_build():
        return SO
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = (extra_flags if extra_flags is not None else os.environ.get("VIORB_HIPCC_FLAGS", "").split())
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra) + ["-I", os.path.join(ROOT, "include")]
    objdir = objdir or os.path.join(HERE, "build", "obj" if not extra else "obj_" + "".join(c if c.isalnum() else "_" for c in " ".join(extra)))
    os.makedirs(objdir, exist_ok=True)
    stamp = " ".join(cflags)
    hdr_t = max(os.path.getmtime(h) for h in _headers())

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        fl = obj + ".flags"
        fresh = os.path.exists(obj) and os.path.exists(fl) and open(fl).read() == stamp and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t)
        if fresh and not scratch:
            return obj
        cmd = [hipcc] + cflags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        open(fl, "w").write(stamp)
        return obj

    with ThreadPoolExecutor(max_workers=int(os.environ.get("VIORB_BUILD_JOBS", "6"))) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:          # python -m viorb_amd.build OUT.so [-DFLAG ...]: an experiment build
        build(force=True, verbose=True, out=os.path.abspath(sys.argv[1]), extra_flags=sys.argv[2:])
    else:
        build(force=True, verbose=True)
