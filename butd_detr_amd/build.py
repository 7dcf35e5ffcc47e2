"""hipcc build recipe for the gfx950 C-ABI library (no cmake, no torch headers).

    python -m butd_detr_amd.build [--force]

Compiles every ``csrc/*.hip`` translation unit for ``--offload-arch=gfx950`` and links them into
``butd_detr_amd/lib/libbutd_detr_hip.so`` -- in-tree, so the built library travels with the repo
snapshot to the GPU box.  hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB_PATH = os.path.join(LIBDIR, "libbutd_detr_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE,
          "-Wall", "-Wno-unused-function"]
# Per-TU flags.  The index kernels must round exactly like the oracle: no FMA contraction.
# Wave priority of the MAIN queue's kernels (s_setprio at kernel entry; every unit but the index ops the prefetch branch
# runs): the captured step's prefetch branches -- the next batch's furthest-point sampling on 8 CUs for 3.4 ms, its
# language model -- share CUs with the step's own kernels; with the step's waves ahead in the issue arbitration the
# level-1 sampling costs the step 0.41 instead of 0.61 ms: 20.51 -> 20.28 ms per step (profiles/r06_side_branches.txt).
MAIN_PRIO = ["-DBUTD_MAIN_PRIO=3"]
MAIN_UNITS = ("gemm_ops.hip", "attention_ops.hip", "mlp_ops.hip", "sa_ops.hip", "sa_last_bwd.hip", "sa_fused.hip",
              "sa_first_linear.hip", "criterion_ops.hip", "lsap_ops.hip", "optim_ops.hip")
TU_FLAGS = {
    "pointnet2_ops.hip": ["-ffp-contract=off"],
    "fps_pruned.hip": ["-ffp-contract=off"],
    "ball_query_grid.hip": ["-ffp-contract=off"],
    "criterion_ops.hip": ["-ffp-contract=off"],
    "augment_ops.hip": ["-ffp-contract=off"],
    "rowwise_ops.hip": ["-ffp-contract=off"],
    # MFMA accumulators in VGPRs (the "VGPR form" of the matrix instructions): the softmax / epilogue code reads them in
    # place instead of through v_accvgpr_read / _write (60 of the ~430 instructions of a forward attention key tile, 12 876
    # static ones in gemm_ops); measured round 5: attention backward 442 -> 426 us, step 23.33 -> 23.13 ms (profiles/r05_vgpr_form.txt)
    "attention_ops.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "gemm_ops.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "sa_last_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "sa_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
}


_PROBED = {}


def _supported(flags):
    """Hidden LLVM options (-mllvm ...) are probed once on an empty translation unit: a hipcc that lacks or renamed one
    builds without it (with a warning) instead of failing the whole library -- the flag is worth ~1 % of a step."""
    key = tuple(flags)
    if key not in _PROBED:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void butd_probe() {}\n")
            rc = subprocess.call([HIPCC, "--offload-arch=gfx950", "-c", src, "-o", os.path.join(d, "probe.o")] + list(flags),
                                 stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _PROBED[key] = rc == 0
        if rc != 0:
            print(f"butd_detr_amd.build: {' '.join(flags)} is not accepted by {HIPCC}: building without it", file=sys.stderr)
    return _PROBED[key]


def _tu_flags(name):
    flags = TU_FLAGS.get(name, [])
    if "-mllvm" in flags and not _supported(flags):
        flags = []
    return flags + (MAIN_PRIO if name in MAIN_UNITS else [])


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(src, flags):
    h = hashlib.sha1()
    h.update(" ".join(flags).encode())
    with open(src, "rb") as f:
        h.update(f.read())
    for hdr in sorted(os.listdir(INCLUDE)):
        with open(os.path.join(INCLUDE, hdr), "rb") as f:
            h.update(f.read())
    for hdr in sorted(os.listdir(CSRC)):
        if hdr.endswith((".h", ".hpp", ".cuh")):
            with open(os.path.join(CSRC, hdr), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile (only what changed) and link; returns the library path."""
    os.makedirs(OBJDIR, exist_ok=True)
    objs, relink = [], force or not os.path.exists(LIB_PATH)
    for name in _sources():
        src = os.path.join(CSRC, name)
        flags = COMMON + _tu_flags(name)
        obj = os.path.join(OBJDIR, name.replace(".hip", ".o"))
        stamp_file = obj + ".stamp"
        stamp = _stamp(src, flags)
        old = open(stamp_file).read() if os.path.exists(stamp_file) else ""
        if force or old != stamp or not os.path.exists(obj):
            cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp_file, "w") as f:
                f.write(stamp)
            relink = True
        objs.append(obj)
    if relink:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
