"""CPU: the C-ABI library builds/loads and exports every symbol include/deflow_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from deflow_amd import build
    return build.build()


def _declared():
    src = open(os.path.join(ROOT, "include", "deflow_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(df_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.df_version.restype = ctypes.c_int
    assert lib.df_version() >= 100


def test_binding_table_matches_header(lib_path):
    from deflow_amd import _lib
    assert sorted(_lib._SIGS) == _declared()


def test_struct_layouts():
    from deflow_amd import _lib
    assert ctypes.sizeof(_lib.DfImg) == 56 and _lib.DfImg.img_stride.offset == 32 and _lib.DfImg.elt.offset == 48   # + elt (round 3)
    assert ctypes.sizeof(_lib.DfGeom) == 48
    assert ctypes.sizeof(_lib.DfGruWeights) == 80 and ctypes.sizeof(_lib.DfGruWeightsT) == 24


def test_missing_library_fails_loudly(monkeypatch):
    from deflow_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdeflow_amd.so")
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        _lib.load()


def test_state_dict_keys_match_reference_layout():
    """head.* keys are pinned by the reference classes [REF decoder.py:143-153]; embedder/backbone follow upstream names."""
    import deflow_amd
    m = deflow_amd.DeFlow()
    sd = m.state_dict()
    for k, shape in {"head.offset_encoder.weight": (64, 3), "head.gru.convz.weight": (128, 192, 1),
                     "head.gru.convr.bias": (128,), "head.gru.convq.weight": (128, 192, 1),
                     "head.decoder.0.weight": (32, 192), "head.decoder.2.weight": (3, 32),
                     "embedder.feature_net.pfn_layers.0.0.weight": (32, 9),
                     "backbone.encoder_step_1.0.conv.weight": (64, 32, 3, 3),
                     "backbone.decoder_step1.u1_u2.0.weight": (256, 512, 1, 1),
                     "backbone.decoder_step4.weight": (64, 64, 3, 3)}.items():
        assert tuple(sd[k].shape) == shape, k
    assert sum(p.numel() for p in m.parameters()) == 6891939


def test_entry_points_reject_bad_arguments_without_launching(lib_path):
    """argument validation happens before any kernel launch, so it can be exercised without a GPU: NULL buffers and
    inconsistent shapes must come back as negative DF_E_* codes, never as a crash or a silent success"""
    import ctypes as C
    from deflow_amd._lib import DfImg, load
    lib = load()
    null_img = DfImg(0, 0, 0, 0, 0, 0, 1, 0, 0)
    bad = DfImg(0x1000, 2, 8, 8, 48, 48, 2, 8 * 8 * 48, 0)          # 48 channels: not a multiple of 64
    ok64 = DfImg(0x1000, 2, 8, 8, 64, 64, 2, 8 * 8 * 64, 0)
    P = C.c_void_p
    assert lib.df_sparse_conv3x3(P(0), P(0), 2, null_img, P(0), P(0), null_img, 1, P(0)) < 0
    assert lib.df_sparse_conv3x3(P(0x1000), P(0x1000), 2, bad, P(0x1000), P(0), ok64, 1, P(0)) < 0
    assert lib.df_sparse_wgrad3x3(P(0), P(0), 2, null_img, null_img, P(0), P(0), 1, P(0)) < 0
    assert lib.df_sparse_wgrad3x3(P(0x1000), P(0x1000), 2, bad, ok64, P(0x1000), P(0), 1, P(0)) < 0
    assert lib.df_pillar_input_grad(P(0), P(0), 2, 8, 8, 0, P(0), P(0), null_img, P(0), null_img, 1, 1, P(0)) < 0
    assert lib.df_pillar_input_grad(P(0x1000), P(0x1000), 2, 8, 8, 3, P(0x1000), P(0x1000), ok64, P(0x1000), ok64, 1, 1, P(0)) < 0  # cloud 3
    assert lib.df_sparse_in_wgrad(P(0), P(0), 2, 8, 8, 0, P(0), null_img, P(0), 1, P(0)) < 0
    assert lib.df_gru_wgrad(P(0), P(0), P(0), 2, 100, 4, P(0), 1, P(0)) < 0
    assert lib.df_colsum_stage(P(0x1000), 4, 16, 8, P(0x1000), P(0)) < 0                                     # groups > rows
    assert lib.df_conv2d_wgrad(bad, ok64, 3, 1, 1, P(0), 1, P(0), 0, P(0), P(0)) < 0


def test_counted_vmcnt_kernels_have_no_scratch_traffic_in_their_loops():
    """kernels that keep LDS-DMA in flight across barriers order it with COUNTED s_waitcnt vmcnt(n).  A register spill reloaded inside
    such a loop is a VMEM operation on the same counter, and the compiler follows it with s_waitcnt vmcnt(0): the prefetched ring is
    drained once per stage (round 6 found that in the dominant weight-gradient kernel -- correct, and its four-deep ring worth nothing).
    The compiled ISA of these kernels must have no scratch instruction inside any loop (spills in the prologue / epilogue are harmless);
    tools/scan_scratch_in_loops.py does the scan."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from scan_scratch_in_loops import scan
    src = os.path.join(ROOT, "deflow_amd", "csrc")
    for fname, kernels in (("conv_bf16.hip", {"conv64_roll_bf16_kernel": 0}), ("decoder_wgrad.hip", {"gru_wgrad_kernel": 0, "gru_wgrad4_kernel": 0}),
                           ("conv_wgrad.hip", {"wgrad3_tr_kernel": 0, "wgrad3_h2p_kernelILi4ELb1": 0})):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-Wno-unused-result", "-S",
                            os.path.join(src, fname), "-o", "-", "--cuda-device-only"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        found = scan(r.stdout)
        for k, allowed in kernels.items():
            hits = {name: v for name, v in found.items() if k in name}
            assert hits, (fname, k)
            for name, (n_in, tot) in hits.items():
                assert n_in <= allowed, (name, n_in, tot)


def test_no_bit_cast_of_a_vector_element_in_the_kernels():
    """ROCm 7.2's clang evaluates __builtin_bit_cast(T, v[i]) on an ext_vector_type value as T(v[0]) for every i
    (tools/repro/bitcast_vector_element.hip).  The kernels must copy the element to a scalar first (or cast an rvalue expression)."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deflow_amd", "csrc")
    # vector-typed locals the kernels index with [k]: names declared with one of the ext-vector typedefs
    vec_decl = re.compile(r"\b(?:const\s+)?(?:f32x4|f32x16|u32x2\w*|u32x4\w*|f16x\w+|bf16x\w+)\s+(\w+)\s*(?:=|;|\[)")
    cast = re.compile(r"__builtin_bit_cast\(\s*[\w ]+,\s*(\w+)\s*\[[^\]]+\]\s*\)")
    bad = []
    for f in sorted(os.listdir(root)):
        if not f.endswith((".hip", ".h")):
            continue
        src = open(os.path.join(root, f)).read()
        vecs = set(vec_decl.findall(src))
        for m in cast.finditer(src):
            # an ARRAY of vectors indexed once (af[s][i] is matched only up to the first index and yields a whole vector) is fine;
            # a vector-typed scalar local indexed once is the defect
            name = m.group(1)
            tail = src[m.end() - 1:m.end() + 1]
            if name in vecs and not re.search(r"\b" + name + r"\s*\[[^\]]+\]\s*\[", src[m.start():m.end() + 8]):
                decl = re.search(r"\b(?:const\s+)?(?:f32x4|f32x16|u32x2\w*|u32x4\w*|f16x\w+|bf16x\w+)\s+" + name + r"\s*\[", src)
                if decl is None:      # declared as a plain vector, not as an array of vectors
                    bad.append((f, m.group(0)))
    assert not bad, bad
