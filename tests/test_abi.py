"""The C-ABI library loads and exports every symbol include/svt_hip.h declares
(no compute calls: runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "svt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svt_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_and_export_list_agree():
    from sparsearray_amd._hip import EXPORTS
    assert sorted(EXPORTS) == _declared_symbols()


def test_library_exports_every_declared_symbol():
    from sparsearray_amd._hip import load_library
    lib = load_library()
    for sym in _declared_symbols():
        assert hasattr(lib, sym), f"libsvt_hip.so does not export {sym}"


def test_thread_control_entry_points_need_no_gpu():
    """C_get_num_procs / C_get_max_threads / C_set_max_threads (src/thread_control.c:47-66):
    set returns the previous value, get returns what was set."""
    from sparsearray_amd._hip import load_library
    lib = load_library()
    assert lib.svt_get_num_procs() >= 1
    first = lib.svt_get_max_threads()
    assert first >= 1
    assert lib.svt_set_max_threads(3) == first
    assert lib.svt_get_max_threads() == 3
    assert lib.svt_set_max_threads(first) == 3


def test_product_fails_loudly_without_gpu():
    """On a box without an MI355X the product path must raise, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sparsearray_amd
    from sparsearray_amd._hip import HipBackendError
    with pytest.raises(HipBackendError, match="no HIP device|gfx950"):
        sparsearray_amd.hip_session()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "sparsearray_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "svt_oracle", "orc_"):
                    assert needle not in src, f"{f} references the oracle ({needle})"


def test_generated_asm_is_in_sync(tmp_path):
    """sparsearray_amd/csrc/pbc_dma_asm.inc is generated (tools/gen_pbc_asm.py) and committed:
    the committed text must be what the generator writes with its defaults."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "pbc_dma_asm.inc"
    env = {k: v for k, v in os.environ.items() if not k.startswith("PBC_")}
    env["PBC_ASM_OUT"] = str(out)
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_pbc_asm.py")], check=True, env=env,
                   stdout=subprocess.DEVNULL)
    committed = open(os.path.join(root, "sparsearray_amd", "csrc", "pbc_dma_asm.inc")).read()
    assert out.read_text() == committed


def test_generated_gatherx_asm_is_in_sync(tmp_path):
    """sparsearray_amd/csrc/pbgx_asm.inc (the pass loop of the XCD-paced gather kernel) is what
    tools/gen_pbgx_asm.py writes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "pbgx_asm.inc"
    env = dict(os.environ)
    env["PBGX_ASM_OUT"] = str(out)
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_pbgx_asm.py")], check=True, env=env,
                   stdout=subprocess.DEVNULL)
    committed = open(os.path.join(root, "sparsearray_amd", "csrc", "pbgx_asm.inc")).read()
    assert out.read_text() == committed
