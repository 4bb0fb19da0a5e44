"""Build libaudiogpt_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m audiogpt_amd.build [--force]

Objects are cached under audiogpt_amd/csrc/_build keyed by source mtime; the shared object is written
next to this file so that it travels with the repo snapshot to the GPU box.

Diagnostic variants (tracing / sanitizer rows) are separate libraries beside the product one -- they never
replace it; load one with AUDIOGPT_AMD_LIB=<path>:
    MAA_BUILD_ROCTX=1      libaudiogpt_mi355x_roctx.so      per-DDIM-step roctx ranges (csrc/ddim.cpp) for rocprofv3 --marker-trace
    MAA_BUILD_ASAN=1       libaudiogpt_mi355x_asan.so       -fsanitize=address on host AND device code (gfx950:xnack+, -O1 -g): run with
                                                            HSA_XNACK=1 and LD_PRELOAD=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so)
    MAA_BUILD_NO_TUNING=1  libaudiogpt_mi355x_notuning.so   deployment build: the MAA_* test / A-B switches compiled out (runtime.cpp)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libaudiogpt_mi355x.so")
SOURCES = ["igemm_f32.hip", "igemm_bf16.hip", "igemm_dma.hip", "igemm_dma2.hip", "igemm_pp.hip", "calib.hip", "nsf.hip", "diffsinger.hip", "halo_conv1d.hip", "encoders.hip", "spectral.hip", "flash_attn.hip", "norm.hip", "misc.hip", "runtime.cpp", "blocks.cpp", "unet.cpp", "vae.cpp",
           "vocoder.cpp", "diffnet.cpp", "encoders.cpp", "clap_audio.cpp", "ddim.cpp", "api.cpp"]
ARCH = "gfx950"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form"]
LINK = []
VARIANT = ""
if os.environ.get("MAA_BUILD_ROCTX") == "1":      # per-DDIM-step roctx ranges (csrc/ddim.cpp) for rocprofv3 --marker-trace
    FLAGS = FLAGS + ["-DMAA_ROCTX"]
    LINK = ["-L/opt/rocm/lib", "-lroctx64"]
    VARIANT += "_roctx"
if os.environ.get("MAA_BUILD_ASAN") == "1":       # address sanitizer, host and device (the device side needs xnack+ code objects)
    ARCH = "gfx950:xnack+"
    FLAGS = ["--offload-arch=" + ARCH, "-O1", "-g"] + FLAGS[2:] + ["-fsanitize=address", "-shared-libsan"]
    LINK = LINK + ["-fsanitize=address", "-shared-libsan"]
    VARIANT += "_asan"
if os.environ.get("MAA_BUILD_NO_TUNING") == "1":  # deployment build: the MAA_* test / A-B switches are compiled out (runtime.cpp)
    FLAGS = FLAGS + ["-DMAA_NO_TUNING"]
    VARIANT += "_notuning"
if VARIANT:
    OUT = os.path.join(HERE, "libaudiogpt_mi355x%s.so" % VARIANT)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp")))
    for f in files + [os.path.join(os.path.dirname(HERE), "include", "maa.h")]:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    # content-hash stamp: a snapshot copied to the GPU box (fresh mtimes) must not trigger a rebuild there
    stamp = OUT + ".hash"
    digest = _source_hash()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return OUT
    hipcc = _hipcc()
    bdir = os.path.join(CSRC, "_build" + VARIANT)
    os.makedirs(bdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "maa.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(bdir, s + ".o")
        if force or _newer(src, obj) or os.path.getmtime(obj) < hdr_time:
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-8000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    if jobs:
        if verbose:
            print("[audiogpt_amd.build] compiling %d file(s) for %s%s" % (len(jobs), ARCH, " (%s)" % VARIANT[1:] if VARIANT else ""), flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(bdir, s + ".o") for s in SOURCES]
    if jobs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs + LINK
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-8000:])
        if verbose:
            print("[audiogpt_amd.build] wrote", OUT, flush=True)
    with open(stamp, "w") as f:
        f.write(digest)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
