"""Which kernels of libdeflow_amd.so contain packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)?

Round 3 found (tools/pfn_bwd_stress.py) that the pillar feature net's backward kernels returned wrong sums in ~4 % of their launches
while ANOTHER PROCESS ran bf16-MFMA kernels on the same GPU, and stopped doing so once their packed-fp32 instructions were gone
(deflow_amd/build.py: pillarize.hip is built with -fno-slp-vectorize).  One process per GPU -- the deployment -- never showed it.
This script lists every kernel the hazard could apply to: it compiles each source to device assembly with the library's own
flags (hipcc --cuda-device-only -S; CPU only, no GPU needed) and counts the instructions per kernel.

    python tools/pk_audit.py > profiles/r04_packed_fp32_audit.txt
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deflow_amd import build as B   # noqa: E402


def audit(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [B.HIPCC, *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), "--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            return src, None, r.stderr[-500:]
        rows, cur, n = [], None, {}
        for line in open(out):
            m = re.match(r"^(_Z\w+|df_\w+):", line)
            if m:
                cur = m.group(1)
                n[cur] = [0, 0]
                continue
            if cur is None:
                continue
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                cur = None
                continue
            t = line.strip()
            if re.match(r"v_pk_(fma|mul|add)_f32", t):
                n[cur][0] += 1
            elif re.match(r"v_(mfma|smfmac)", t):
                n[cur][1] += 1
        for k, (pk, mf) in n.items():
            rows.append((k, pk, mf))
        return src, rows, ""


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True)
        return r.stdout.splitlines()
    except OSError:
        return names


def main():
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(audit, B.SOURCES))
    print("# kernels of libdeflow_amd.so with packed-fp32 VALU instructions (v_pk_fma_f32 | v_pk_mul_f32 | v_pk_add_f32), by source file")
    print("# flags:", " ".join(B.FLAGS), "| per-file extras:", B.EXTRA_FLAGS)
    tot_k = tot_pk = 0
    for src, rows, err in res:
        if rows is None:
            print(f"{src}: COMPILE FAILED {err}")
            continue
        hit = [(k, pk, mf) for k, pk, mf in rows if pk > 0]
        names = demangle([k for k, _, _ in hit])
        print(f"\n## {src}: {len(hit)} of {len(rows)} kernels / device functions contain packed-fp32 instructions")
        for (k, pk, mf), nm in sorted(zip(hit, names), key=lambda t: -t[0][1]):
            nm = re.sub(r"\(anonymous namespace\)::", "", nm)
            print(f"  {pk:5d} v_pk_*_f32  {mf:5d} mfma   {nm[:150]}")
        tot_k += len(hit)
        tot_pk += sum(pk for _, pk, _ in hit)
    print(f"\n# total: {tot_k} kernels, {tot_pk} packed-fp32 instructions")


if __name__ == "__main__":
    main()
