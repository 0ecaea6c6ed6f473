"""CPU tier: the C-ABI library loads, exports every symbol include/fa_mi355.h declares, rejects
bad arguments before touching the device, and the host logic (sharding, flop/byte model)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fa_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"\b(?:int|size_t|const char\*)\s+(\w+)\s*\(", text)


def test_header_symbols_exported(fa):
    names = _declared_symbols()
    assert set(names) == set(fa.capi.SYMBOLS), names
    raw = ctypes.CDLL(fa.capi.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), n


def test_version(fa):
    assert fa.version().startswith("fa_mi355 ") and fa.version().endswith("gfx950")


def test_invalid_arguments_rejected_without_device(fa):
    L = fa.lib()
    INVALID = 1  # hipErrorInvalidValue
    p = ctypes.c_void_p(16)
    null = ctypes.c_void_p(0)
    assert L.flashattn_forward_wmma(null, p, p, p, 1, 128, 64, 0.125, None) == INVALID
    assert L.flashattn_forward_wmma(p, p, p, p, 1, 128, 60, 0.125, None) == INVALID   # D % 16 != 0 (wmma.cu:63)
    assert L.flashattn_forward_wmma(p, p, p, p, 0, 128, 64, 0.125, None) == INVALID
    assert L.flashattn_forward_wmma(p, p, p, p, 1, 0, 64, 0.125, None) == INVALID
    assert L.flashattn_forward_wmma(p, p, p, p, 1, 128, 512, 0.125, None) == INVALID  # D > 256
    assert L.fa_forward(p, p, p, p, 0, 1, 128, 64, 0.125, 0, 0, None) == INVALID
    assert L.fa_forward(p, p, p, p, 1, 1, 128, 64, 0.125, 7, 0, None) == INVALID      # dtype
    assert L.fa_forward(p, p, p, p, 1, 1, 128, 64, 0.125, 0, 5, None) == INVALID      # out dtype
    assert L.fa_forward_ex(p, p, p, p, 1, 1, 128, 32, 0.125, 0, 0, 2, None) == INVALID  # tiled needs D in {64,128}
    assert L.fa_forward_ex(p, p, p, p, 1, 1, 128, 64, 0.125, 0, 0, 99, None) == INVALID
    assert L.fa_forward(p, p, p, p, 1, 1, 1 << 24, 128, 0.125, 0, 0, None) == INVALID  # per-head offsets must fit 32 bit
    assert L.fa_forward_causal(p, p, p, p, 1, 1, 128, 64, 0.125, 0, 0, 5, None) == INVALID   # only AUTO/GENERIC/TILED/2WG/W64
    assert L.fa_forward_causal(p, p, p, p, 1, 1, 128, 32, 0.125, 0, 0, 2, None) == INVALID   # tiled needs D in {64,128}
    assert L.fa_forward_causal(null, p, p, p, 1, 1, 128, 64, 0.125, 0, 0, 0, None) == INVALID
    assert L.fa_forward_splitkv(p, p, p, p, 1, 1, 1, 4096, 32, 0.125, 0, 0, None, 0, None) == INVALID    # d in {64,128}
    assert L.fa_forward_splitkv(p, p, p, p, 1, 1, 0, 4096, 64, 0.125, 0, 0, None, 0, None) == INVALID
    assert L.fa_forward_splitkv(p, p, p, p, 1, 1, 1, 8192, 64, 0.125, 0, 0, None, 0, None) == INVALID    # split needs a workspace
    assert L.fa_forward_splitkv_workspace_bytes(1, 1, 1, 8192, 64) > 0
    assert L.fa_forward_splitkv_workspace_bytes(8, 16, 4096, 4096, 64) == 0                             # enough workgroups already
    assert L.fa_forward_splitkv_workspace_bytes(0, 1, 1, 8192, 64) == 0
    assert L.fa_debug_stage(0, p, p, p, 1, 128, 64, 0.125, 0, None) == INVALID
    assert L.fa_debug_stage(1, p, p, p, 1, 128, 32, 0.125, 0, None) == INVALID
    assert L.fa_debug_stage(3, p, null, p, 1, 128, 64, 0.125, 0, None) == INVALID
    assert L.flashattn_streaming_16x16_mw(p, p, p, p, 0, 128, 0.25, None) == INVALID
    assert L.flashattn_streaming_16x16_mw(p, p, null, p, 4, 128, 0.25, None) == INVALID
    assert L.flashattn_streaming_16x16_mw_kt(p, p, p, p, 4, 0, 0.25, None) == INVALID


def test_ops_refuse_cpu_tensors(fa):
    torch = pytest.importorskip("torch")
    q = torch.zeros(1, 128, 64, dtype=torch.float16)
    with pytest.raises(ValueError):
        fa.fa_forward(q, q, q)
    o = torch.zeros(1, 128, 64, dtype=torch.float32)
    with pytest.raises(ValueError):
        fa.flashattn_forward_wmma(q, q, q, o, 1, 128, 64, 0.125)


def test_flop_and_byte_model(fa):
    # BASELINE.md section 2
    assert fa.attention_flops(1, 128, 64) == 4194304.0
    assert fa.attention_flops(128, 4096, 64) == pytest.approx(549.76e9, rel=1e-4)
    assert fa.attention_min_bytes(1, 128, 64) == 81920.0
    assert fa.attention_min_bytes(128, 4096, 64) == pytest.approx(335.5e6, rel=1e-3)
    assert fa.attention_min_bytes(128, 4096, 64, out_bytes=2) == pytest.approx(268.4e6, rel=1e-3)


def test_shard_range(fa):
    for total in (0, 1, 7, 128, 1024, 1031):
        for world in (1, 2, 3, 8):
            parts = [fa.shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            for (a0, a1), (b0, b1) in zip(parts, parts[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        fa.shard_range(8, 8, 8)


def test_torch_custom_op_registers(fa):
    """torch.ops.fa_mi355.forward exists after register(), traces on meta tensors, and has no CPU kernel."""
    import pytest
    import torch
    from flashattention_kernel_project_amd.torch_op import register
    register()
    register()
    q = torch.empty(2, 3, 50, 64, dtype=torch.float16, device="meta")
    o = torch.ops.fa_mi355.forward(q, q, q, 0.125, False, True)
    assert o.shape == q.shape and o.dtype == torch.float32
    assert torch.ops.fa_mi355.forward(q, q, q, 0.125, True, False).dtype == torch.float16
    with pytest.raises(Exception):   # no CPU implementation: the product path is the HIP library only
        c = torch.zeros(1, 1, 16, 64, dtype=torch.float16)
        torch.ops.fa_mi355.forward(c, c, c, 0.125, False, True)


def test_header_is_plain_c(tmp_path):
    """include/fa_mi355.h must compile as C99 on its own (the drop-in boundary is a C ABI)."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "fa_mi355.h"\nint main(void) { return (int)sizeof(size_t) == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "hdr.o")])


def test_spelled_out_matrix_instructions_keep_their_wait_states():
    """The one-wave-per-SIMD kernel (id 28) spells QK^T out as inline asm, which the compiler's hazard recogniser does not
    look into: tools/mfma_hazard_lint.py walks the gfx950 listing and fails on any vector instruction closer than the
    required wait states behind (or, for producers of the operands, in front of) one of those instructions."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    for tu in ("fa_fwd_rp16_d128w.hip", "fa_fwd_rp16_cw.hip"):   # the plain and the causal one-wave families
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_hazard_lint.py"),
                            os.path.join(root, "flashattention_kernel_project_amd", "csrc", tu)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, tu + "\n" + r.stdout[-3000:] + r.stderr[-1000:]
        assert re.search(r"total: \d+ spelled-out matrix instructions, 0 hazards", r.stdout), r.stdout[-500:]
