"""Build recipe for libdpd_hip.so (hipcc, gfx950 only).  Run here on the CPU: hipcc cross-compiles."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdpd_hip.so")
ASAN_LIB = os.environ.get("DPD_ASAN_OUT", "/tmp/libdpd_hip_asan.so")   # the sanitizer build never replaces the product library
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
EXTRA = os.environ.get("DPD_EXTRA_FLAGS", "").split()   # experiments only, e.g. -DBWD_WAVES_PER_SIMD=1
# DPD_ASAN=1: AddressSanitizer on the HOST side of the library only (argument validation, workspace arithmetic,
# launch set-up), for the CPU-side tests; the device code is built as always.  Run the tests of such a build with
# LD_PRELOAD=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so) and DPD_HIP_LIBRARY=<the library>.
ASAN = os.environ.get("DPD_ASAN", "") not in ("", "0")
# -ffp-contract=off: the reference computes I*I + Q*Q and friends with separate roundings; fused
# multiply-adds are written explicitly where they are wanted.
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs; the kernels consume them on the VALU right
# away, and the AGPR form costs one v_accvgpr_read/write per value.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-Wno-unused-result", "-I", os.path.join(os.path.dirname(HERE), "include")]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "dpd_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, jobs=8):
    if ASAN:
        return _build(ASAN_LIB, os.path.join(os.path.dirname(ASAN_LIB), "dpd_hip_asan_obj"), verbose, jobs)
    if not force and not needs_build():
        return LIB
    return _build(LIB, os.path.join(HERE, "build"), verbose, jobs)


def _build(lib_path, objdir, verbose, jobs):
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + EXTRA + (["-Xarch_host", "-fsanitize=address", "-g"] if ASAN else []) + ["-c", src, "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x"); cmd.insert(2, "hip")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        if len(procs) >= jobs:
            s, p = procs.pop(0)
            if p.wait() != 0:
                raise RuntimeError(f"hipcc failed on {s}")
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + (["-fsanitize=address", "-fno-gpu-sanitize"] if ASAN else []) + objs + ["-o", lib_path]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv)
