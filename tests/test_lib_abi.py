"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/hma_hip.h declares."""
import os
import re

from hma_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "hma_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(hma_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    names = _declared()
    assert len(names) >= 20
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names
    assert lib.hma_abi_version() == 0x484D4104


def test_invalid_arguments_are_rejected_without_a_gpu():
    # argument validation happens before any launch: NULL pointers / bad shapes -> HMA_EINVAL
    lib = _lib.load()
    assert lib.hma_ln_fwd(None, None, None, None, 4, 1e-5) == -10001
    g = _lib.GemmNT()
    assert lib.hma_gemm_nt(None, g) == -10001
    assert lib.hma_attn_temporal_fwd(None, 1, 1, 1, 17, 4, 0.25) == -10001
