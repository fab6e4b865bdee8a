"""CPU tests of the C-ABI boundary: the shared library loads, exports every symbol that
include/pti_vae.h declares, and the ctypes binding table covers exactly that set.  No compute
calls (no GPU here); argument validation paths that return before any launch ARE exercised.
"""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "pti_vae.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pti_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_bound_and_exported():
    from pti_ldm_vae_amd import _lib
    declared = _declared()
    assert declared, "no declarations parsed"
    assert sorted(_lib.SIGNATURES) == declared
    handle = _lib.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} not exported by libpti_vae_hip.so"
    assert handle.pti_abi_version() == 5


def test_conv_desc_layout_matches_header(tmp_path):
    """ctypes mirror vs the C compiler's view of include/pti_vae.h (sizeof + every field offset)."""
    import subprocess
    from pti_ldm_vae_amd._lib import ConvDesc
    fields = [f[0] for f in ConvDesc._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pti_vae.h"\nint main(void){printf("%zu", sizeof(pti_conv_desc));'
                   + "".join(f'printf(" %zu", offsetof(pti_conv_desc, {f}));' for f in fields) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert C.sizeof(ConvDesc) == vals[0]
    assert [getattr(ConvDesc, f).offset for f in fields] == vals[1:]
    assert ConvDesc.in_stride.offset == 72 and ConvDesc.out_stride.offset == 104 and ConvDesc.in_f16.offset == 136


def test_validation_errors_before_launch():
    from pti_ldm_vae_amd import _lib
    h = _lib.lib()
    assert h.pti_conv_packed_bytes(32, 48, 3, 0) == 0          # cin not a multiple of 32
    assert h.pti_conv_packed_bytes(64, 32, 3, 0) == 2 * 64 * 32 * 9
    d = _lib.ConvDesc(n=1, h=8, w=8, cin=32, ho=8, wo=8, cout=32, ksize=3)
    rc = h.pti_conv2d_mfma(None, None, None, None, None, None, None, None, None, C.byref(d), None)
    assert rc == -1 and b"null" in h.pti_last_error_string()
    rc = h.pti_attention_fwd(C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), 1, 96, 96, None)
    assert rc == -2 and b"head dim" in h.pti_last_error_string()
    with pytest.raises(_lib.PtiError):
        _lib.check(rc, "attention")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pti_ldm_vae_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PtiError, match="no CPU/PyTorch fallback"):
        _lib.lib()


def test_workspace_size_constants_match_header():
    """ops sizes the scratch of the fixed-order reductions from the header's block caps."""
    from pti_ldm_vae_amd import ops
    text = open(os.path.join(ROOT, "include", "pti_vae.h")).read()
    caps = {k: int(v) for k, v in re.findall(r"#define (PTI_[A-Z_]+_MAX_BLOCKS) (\d+)", text)}
    assert caps == {"PTI_POST_QUANT_BWD_MAX_BLOCKS": ops.POST_QUANT_BWD_MAX_BLOCKS,
                    "PTI_LATENT_BWD_MAX_BLOCKS": ops.LATENT_BWD_MAX_BLOCKS,
                    "PTI_VAE_LOSS_MAX_BLOCKS": ops.VAE_LOSS_MAX_BLOCKS}


def test_tile_and_block_queries_need_no_gpu():
    """pti_conv_gnbwd_tiles / pti_gn_bwd_blocks are pure host arithmetic: callable on a CPU-only box."""
    from pti_ldm_vae_amd import _lib as L
    lib = L.lib()
    d = L.ConvDesc(n=2, h=64, w=48, cin=64, ho=64, wo=48, cout=32, ksize=3, mode=L.PTI_CONV_S1, groups=16, eps=1e-6)
    assert lib.pti_conv_gnbwd_tiles(C.byref(d)) == (64 // 16) * (48 // 16)      # 32-channel tile: 16 x 16 pixels
    d.cout = 128
    assert lib.pti_conv_gnbwd_tiles(C.byref(d)) == (64 // 8) * (48 // 16)       # 128-channel tile: 8 x 16 pixels
    d.cout = 48
    assert lib.pti_conv_gnbwd_tiles(C.byref(d)) == 0                            # not a multiple of 32: unsupported
    assert lib.pti_gn_bwd_blocks(2, 64 * 64, 64) >= 1
    assert lib.pti_gn_bwd_blocks(2, 64 * 64, 1024) == 0                         # above the kernel's channel cap


def test_entry_points_refuse_wrong_result_environment(monkeypatch):
    """no ``PTI_DIAG_*`` knob lives in the product any more; what is left (a native tuning aid that
    needs PTI_ALLOW_WRONG_RESULTS=1 as a second opt-in) makes bench.py and train_vae.py exit before touching the GPU."""
    import subprocess
    import sys
    hits = subprocess.run(["grep", "-rn", "PTI_DIAG", os.path.join(ROOT, "pti_ldm_vae_amd"), "--include=*.py",
                           "--include=*.hip", "--include=*.h", "--include=*.cpp"], capture_output=True, text=True).stdout
    assert hits == "", hits
    import bench
    from pti_ldm_vae_amd import train_vae
    for var in ("PTI_DIAG_SKIP_WGRAD", "PTI_WGRAD_V4_DIAG", "PTI_ALLOW_WRONG_RESULTS"):
        monkeypatch.setenv(var, "1")
        monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "1", "--warmup", "0", "--batch", "2", "--size", "64"])
        with pytest.raises(SystemExit) as e:
            bench.main()
        assert var in str(e.value)
        with pytest.raises(SystemExit) as e:
            train_vae.main(["--synthetic"])
        assert var in str(e.value)
        monkeypatch.delenv(var)
    from pti_ldm_vae_amd import _lib
    monkeypatch.setenv("PTI_WGRAD_STREAM", "0")
    assert _lib.env_overrides().get("PTI_WGRAD_STREAM") == "0"
