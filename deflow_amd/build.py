"""Build libdeflow_amd.so (the C-ABI library with every HIP kernel) in-tree for gfx950.

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under deflow_amd/_build/.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libdeflow_amd.so")
SOURCES = ["conv.hip", "conv_x3p.hip", "conv_wgrad.hip", "conv_bf16.hip", "elementwise.hip", "pillarize.hip", "pillar_bands.hip", "decoder.hip", "decoder3.hip", "decoder4.hip", "decoder_bf16.hip", "decoder3_bwd.hip", "decoder_wgrad.hip", "decoder_bwd.hip", "misc.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-result"]
# Per-file extras.  pillarize.hip is built WITHOUT the SLP vectoriser, i.e. without packed-fp32 instructions (v_pk_fma_f32 ...):
# pillarize.hip is built WITHOUT the SLP vectoriser.  Round 3 found that the pillar feature net's backward kernels (df_pfn_bwd_stats /
# _weights) returned wrong sums -- up to 1e-1 relative, ~4 % of the repetitions -- in the two-ranks-on-one-GPU tests, and stopped doing
# so once their v_pk_*_f32 instructions were gone.  Round 4 bisected it (tools/archive/pfn_race_probe*.sh, record: profiles/r04_pfn_race_probe.txt):
#   * a second PROCESS is not needed: a second host thread of the same process driving another stream reproduces it (a side stream fed
#     by the same thread does not -- at the small test size the two streams then alternate instead of overlapping);
#   * the neighbour that triggers it is ONE kernel: the GRU decoder's forward (df_gru_decoder_fwd).  The convolutions (fp32, fp16x2, bf16),
#     weight gradients, BatchNorm / upsample passes, sparse kernels, gather, hipBLASLt bf16 GEMMs and register-resident streams of each
#     MFMA instruction the library uses (16x16x32_bf16, 32x32x16_bf16, 16x16x4_f32, 16x16x32_f16) do not;
#   * the wrong values are lane-local (<= 8 of a workgroup's 64 statistics: one lane's accumulators), the inputs are constants;
#   * it is not an uninitialised read (a neighbour that leaves NaN in 240 VGPRs and all of LDS on every CU changes nothing), not a stray
#     write (LDS / register canaries beside the GRU kernel stay intact), not an aliasing of the two processes' code (identical
#     libraries fail alike), the waits in the SLP build's ISA cover every load and LDS read (tools: a linear + loop-carried scan), and
#     v_pk_fma_f32 checked against 2 x v_fma_f32 in the same lane never disagrees beside any MFMA stream (tools/archive/pk_mfma_hazard.hip);
#   * the REST of the library keeps the SLP vectoriser (130 kernels with packed-fp32 instructions, profiles/r04_packed_fp32_audit.txt)
#     and the whole training step, fp32 and bf16, is bit-reproducible over 3000 repetitions beside that same neighbour.
# So: these two kernels' SLP-built code + the GRU forward kernel running at the same time, cause below the ISA level not identified.
# Inside the engine the pairing cannot occur (the GRU forward runs in the forward pass, the pillar backward at the end of the backward,
# on one stream); two processes or threads sharing a GPU can produce it, and this flag is what keeps them exact.  These kernels are
# launch-latency-bound: the flag costs nothing measurable.
EXTRA_FLAGS = {"pillarize.hip": ["-fno-slp-vectorize"]}


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "deflow_amd.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), _deps_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", srcp, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed on {src}")
    return obj


def build_variant(name: str, extra_flags, verbose: bool = False) -> str:
    """an alternate library deflow_amd/_build/<name>/lib<name>.so with extra hipcc flags on every file (A/B experiments:
    DF_LIB=<path> python ...)"""
    out = os.path.join(BUILD, name)
    os.makedirs(out, exist_ok=True)

    def one(src):
        obj = os.path.join(out, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if not (os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), _deps_mtime())):
            r = subprocess.run([HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), *extra_flags, "-c", srcp, "-o", obj], capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"hipcc failed on {src}")
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, SOURCES))
    lib = os.path.join(out, f"lib{name}.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link failed")
    return lib


BUILD_INFO = os.path.join(HERE, "_build_info.json")


def write_build_info() -> None:
    """deflow_amd/_build_info.json: the git state the library was built from -- HEAD, whether the tree was dirty, and for every file
    under profiles/ the commit that last touched it.  The GPU box's snapshot has no .git, so bench.py's staleness fields
    (`model_error_budget.source_commit` / `head_commit`) read this file there.  Travels with the snapshot (it is
    git-ignored like the .so: a build product)."""
    import json
    import time
    root = os.path.dirname(HERE)

    def git(*a):
        try:
            r = subprocess.run(["git", "-C", root, *a], capture_output=True, text=True, timeout=20)
            return r.stdout.strip() if r.returncode == 0 else None
        except Exception:   # noqa: BLE001
            return None

    head = git("rev-parse", "--short", "HEAD")
    if head is None:
        return                  # not a git checkout (the GPU box): keep whatever file travelled here
    info = {"head_commit": head, "dirty": bool(git("status", "--porcelain", "--untracked-files=no")), "built_at": int(time.time()), "profiles": {}}
    pdir = os.path.join(root, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_parity_report.jsonl") or f == "pmc_traffic.json":
                info["profiles"]["profiles/" + f] = git("log", "-1", "--format=%h", "--", "profiles/" + f)
    with open(BUILD_INFO, "w") as fh:
        json.dump(info, fh, indent=1)


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    write_build_info()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force:
        for s in srcs:
            o = os.path.join(BUILD, s.replace(".hip", ".o"))
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return LIB


def build_probe(verbose: bool = False) -> str:
    """tools/mfma_peak_probe.hip -> deflow_amd/_build/mfma_peak_probe: the stand-alone measurement of what the 16-bit matrix pipe
    sustains on this chip (zero / random operands), which bench.py runs beside the training step for `roofline.sustained_*`.
    A measuring tool, not part of the library."""
    src = os.path.join(os.path.dirname(HERE), "tools", "mfma_peak_probe.hip")
    out = os.path.join(BUILD, "mfma_peak_probe")
    os.makedirs(BUILD, exist_ok=True)
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-w", "-o", out, src]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed on mfma_peak_probe.hip")
    return out


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
    try:
        print(build_probe(verbose=True))
    except RuntimeError as e:   # the probe is a measuring tool: a failure there must not fail the library build
        print("warning:", e)
