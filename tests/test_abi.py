"""CPU tests of the C-ABI boundary: the library loads without a GPU, exports what include/safebo.h declares,
fails loudly instead of falling back, and the product package never touches the oracle."""
import ctypes as C
import os
import subprocess
import re

import pytest

import safebo_amd
from safebo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "safebo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbo_[a-z0-9_]+)\s*\(", text)) - {"sbo_relay_allreduce_fn", "sbo_relay_allgather_fn"})


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.library_path())
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/safebo.h but not exported"
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert bound == set(names), (sorted(bound - set(names)), sorted(set(names) - bound))


def test_version_and_struct_layout(tmp_path):
    lib = _lib.load()
    assert lib.sbo_version() == 3
    # the ctypes mirrors must have the sizes a C compiler gives the header's structs
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "safebo.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(sbo_sweep_opts), sizeof(sbo_profile), sizeof(sbo_safeopt_result), sizeof(sbo_goose_result), '
                   'sizeof(sbo_tr_result), sizeof(int64_t)); return 0; }\n')
    exe = tmp_path / "sizes"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-I", inc, str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes[:5] == [C.sizeof(_lib.SweepOpts), C.sizeof(_lib.Profile), C.sizeof(_lib.SafeOptResult),
                         C.sizeof(_lib.GooseResult), C.sizeof(_lib.TRResult)]
    assert C.sizeof(_lib.SweepOpts) == 24


def test_null_arguments_are_invalid_not_crashes():
    lib = _lib.load()
    assert lib.sbo_init(0, None) == _lib.SBO_E_INVALID
    assert lib.sbo_posterior_run(None) == _lib.SBO_E_INVALID
    assert lib.sbo_sweep_safeopt(None, None, None) == _lib.SBO_E_INVALID
    assert lib.sbo_model_set(None, 0, b"RBF", 1, 1, 1, None, None, None, None, None, None, None, None) == _lib.SBO_E_INVALID
    with pytest.raises(ValueError):
        _lib.check(lib.sbo_candidates_grid(None, 2, None, None, None, 0, 0))


def _has_gpu():
    n = C.c_int(0)
    _lib.load().sbo_device_count(C.byref(n))
    return n.value > 0


@pytest.mark.skipif(_has_gpu(), reason="this check is about the GPU-less build container")
def test_no_cpu_fallback_without_device():
    with pytest.raises(safebo_amd.SafeBOError) as e:
        safebo_amd.SweepEngine(0)
    assert e.value.code == _lib.SBO_E_HIP
    assert "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "safe-bayesian-optimization_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "gp_oracle" not in text, f
