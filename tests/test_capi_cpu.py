"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every
symbol include/lam_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "lam_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lam_hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lam):
    lam.build()
    L = C.CDLL(lam.lib_path())
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/lam_hip.h but not exported"


def test_binding_covers_header(lam):
    L = lam.lib()
    assert set(_declared_symbols()) == set(L._lam_symbols)
    assert L.lam_hip_abi_version() == 4


def test_build_id_covers_every_source_of_the_translation_unit(lam):
    """The library is ONE translation unit, csrc/lam_hip.hip, that includes csrc/lam_*.h: the build id the binding checks (and the
    Makefile's SRC_ID / stamp file) must hash every one of them, or an edited header would pass for the built library."""
    import glob
    from importlib import import_module
    capi = import_module("2024-eumaster4hpc-student-challenge_amd._capi")
    files = [os.path.relpath(f, ROOT) for f in capi._source_files()]
    pkg = "2024-eumaster4hpc-student-challenge_amd"
    assert files[0] == f"{pkg}/csrc/lam_hip.hip" and files[-1] == "include/lam_hip.h"
    headers = sorted(os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, pkg, "csrc", "*.h")))
    assert files[1:-1] == headers and {"lam_kernels.h", "lam_ctx.h", "lam_launch.h", "lam_exchange.h", "lam_iterate.h"} <= {os.path.basename(h) for h in headers}
    # every header is really included by the translation unit, and the Makefile hashes the same list
    # (directly, or -- lam_host_plan.h, the host-only arithmetic that tests/host_asan also builds -- through lam_kernels.h)
    tu = open(os.path.join(ROOT, pkg, "csrc", "lam_hip.hip")).read() + open(os.path.join(ROOT, pkg, "csrc", "lam_kernels.h")).read()
    assert all(f'#include "{os.path.basename(h)}"' in tu for h in headers)
    mk = open(os.path.join(ROOT, pkg, "Makefile")).read()
    assert "$(sort $(wildcard $(HERE)csrc/*.h))" in mk and "$(STAMP)" in mk
    assert lam.lib().lam_hip_build_id().decode() == capi.source_id()


def test_no_cpu_fallback(lam):
    """Without a GPU, creating a context must fail loudly with ENODEV."""
    import subprocess, sys
    code = (
        "import importlib,sys; sys.path.insert(0, %r);"
        "m = importlib.import_module('2024-eumaster4hpc-student-challenge_amd');\n"
        "try:\n    m.Solver(); print('CREATED')\n"
        "except m.LamHipError as e:\n    print('ERR', e.code)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout
    assert "ERR -2" in out, out


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may mention it."""
    pkg = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd")
    bad = []
    for base in (pkg, os.path.join(ROOT, "include")):
        for dp, _, fns in os.walk(base):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c", "Makefile")):
                    t = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"liboracle|pyoracle|cg_oracle|oracle/", t):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_header_is_plain_c(tmp_path):
    """include/lam_hip.h must be consumable by a C compiler (the boundary is a C ABI, not C++)."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "lam_hip.h"\n'
                   'int main(void) { lam_hip_stats st; lam_hip_ctx *c = 0; (void)st; (void)c;\n'
                   '  return lam_hip_abi_version() == LAM_HIP_ABI_VERSION ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_file_header_rule_is_the_same_in_cpp_and_python(lam, tmp_path):
    """One header rule for every loader: the reference's writers leave garbage in the upper
    32 bits of cols (ConjugateGradient_CPU_OMP.hpp:206-210) or of BOTH words (ConjugateGradient_MultiGPUS_
    CUDA_NCCL.cu:754-757).  LAM::parse_bin_header (C++ loaders) and _capi._read_bin (Python loaders) must
    accept and reject the same files."""
    import importlib, subprocess
    import numpy as np
    capi = importlib.import_module("2024-eumaster4hpc-student-challenge_amd._capi")
    G = 0x2E00B50000000000                      # the garbage pattern observed in reference-written files
    cases = {                                    # name: (hdr words, payload elements, expected (rows, cols) or None)
        "clean_matrix": ((8, 8), 64, (8, 8)),
        "clean_vector": ((8, 1), 8, (8, 1)),
        "garbage_cols": ((8, G | 1), 8, (8, 1)),
        "garbage_both": ((G | 8, G | 1), 8, (8, 1)),
        "truncated": ((8, 8), 60, None),
        "longer_than_needed": ((4, 1), 8, (4, 1)),
        "zero_rows": ((0, 1), 8, None),
    }
    src = tmp_path / "hdr.cpp"
    src.write_text('#include <cstdio>\n#include <cstdlib>\n#include "LAM.hpp"\n'
                   'int main(int c, char **v) { uint64_t h[2] = {strtoull(v[1], 0, 0), strtoull(v[2], 0, 0)}, r = 0, k = 0;\n'
                   '  bool ok = LAM::parse_bin_header(h, strtoull(v[3], 0, 0), 8, &r, &k);\n'
                   '  if (ok) printf("%llu %llu\\n", (unsigned long long)r, (unsigned long long)k); else printf("bad\\n"); return 0; }\n')
    exe = tmp_path / "hdr.out"
    pkg = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd")
    r = subprocess.run(["g++", "-std=c++17", "-DUSE_HIP", "-I", os.path.join(pkg, "LAM", "include"), "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for name, (hdr, nelem, want) in cases.items():
        f = tmp_path / (name + ".bin")
        f.write_bytes(np.array(hdr, dtype=np.uint64).tobytes() + np.zeros(nelem).tobytes())
        out = subprocess.run([str(exe), hex(hdr[0]), hex(hdr[1]), str(16 + 8 * nelem)], capture_output=True, text=True).stdout.strip()
        got_cpp = None if out == "bad" else tuple(int(x) for x in out.split())
        try:
            got_py = capi._read_bin(str(f), np.float64)
        except OSError:
            got_py = None
        assert got_cpp == want and got_py == want, (name, got_cpp, got_py, want)


def test_stale_library_is_refused(lam):
    """The library carries the identity of the sources it was built from (lam_hip_build_id, set by the Makefile); the
    binding compares it with the sources next to it, so a .so left over from an earlier edit can never pass for the
    current code (round 3: a silently failed rebuild had a whole GPU test run exercise the previous kernels)."""
    import importlib
    capi = importlib.import_module(lam.__name__ + "._capi")
    assert lam.lib().lam_hip_build_id().decode() == capi.source_id()
    code = ("import importlib, sys\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            f"capi = importlib.import_module({lam.__name__!r} + '._capi')\n"
            "capi.source_id = lambda: '0123456789abcdef'\n"
            "try:\n    capi.lib(); print('LOADED')\n"
            "except ImportError as e:\n    print('REFUSED', e)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "REFUSED" in r.stdout and "rebuild" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_symmetric_product_plan_covers_every_pair_once(lam, dtype):
    """Host logic of option "symmetric" without a GPU: lam_hip_debug_symv_plan builds the task lists of every shard exactly as the
    launcher does and walks each element through the kernel's own use rule (symv_use, compiled for the host as well).  Every
    directed product y_i += A_ij p_j must come out exactly once -- one shard: the upper triangle; several: cyclic half windows,
    antipodes of even N to the upper half's rows -- and a task flagged interior (processed without any test) may hold no element
    that the rule would have left out.  Odd and even N, N below / at / above a strip (512 fp64, 1024 fp32, 2048 bf16 columns), N not a
    multiple of the vector, shards that do not divide N (the plan itself does not need that; the exchange does)."""
    sizes = [1, 2, 3, 7, 64, 77, 511, 512, 513, 1000, 1023, 1024, 1025, 1536, 2047, 2048, 2050, 3000, 4097]
    for n in sizes:
        for shards in (1, 2, 3, 4, 5, 8):
            if shards > n:
                continue
            bad_pairs, bad_interior, tasks = lam.symv_plan_check(n, shards, dtype)
            assert (bad_pairs, bad_interior) == (0, 0) and tasks > 0, (n, shards, dtype, bad_pairs, bad_interior, tasks)
    # a larger one with many interior tasks in both forms
    for n, shards in ((6144, 1), (6144, 8), (5000, 4)):
        assert lam.symv_plan_check(n, shards, dtype)[:2] == (0, 0)


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("LAM_RUN_SLOW"), reason="set LAM_RUN_SLOW=1 (minutes of CPU, 4 GiB of memory); its output is committed as "
                                                                 "profiles/r05_symv_plan_full_size.txt")
def test_symmetric_product_plan_at_the_advertised_sizes(lam):
    """The plan the symmetric product runs on where it is advertised (README: N = 65536 fp64 on one GPU and on 8 row shards,
    N = 131072 fp32): from N = 65536 on the planner switches to 2048-row bulk tasks, a shape no smaller case builds.  The exhaustive
    check (lam_hip_debug_symv_plan, two bitmaps of n^2 / 8 bytes) must report every directed pair exactly once and nothing unused
    in an interior task.  Run once per change of the planner with LAM_RUN_SLOW=1; the table it prints is committed under profiles/."""
    import time
    rows = []
    for n, shards, dtype, name in ((65536, 1, 0, "fp64"), (65536, 8, 0, "fp64"), (65536, 2, 0, "fp64"), (65536, 4, 0, "fp64"), (131072, 1, 1, "fp32"),
                                   (131072, 8, 1, "fp32"), (131072, 1, 2, "bf16"), (50000, 6, 0, "fp64")):
        t0 = time.time()
        bad_pairs, bad_interior, tasks = lam.symv_plan_check(n, shards, dtype)
        rows.append(f"N={n:6d} {name} shards={shards}: tasks {tasks:7d}  pairs not made exactly once {bad_pairs}  bad interior elements {bad_interior}  "
                    f"({n * n:.3e} directed pairs walked in {time.time() - t0:.0f} s)")
        print(rows[-1], flush=True)
        assert (bad_pairs, bad_interior) == (0, 0) and tasks > 0, rows[-1]
    out = os.environ.get("LAM_SLOW_OUT")
    if out:
        with open(out, "w") as f:
            f.write("# tests/test_capi_cpu.py::test_symmetric_product_plan_at_the_advertised_sizes (LAM_RUN_SLOW=1): lam_hip_debug_symv_plan, host only\n")
            f.write("\n".join(rows) + "\n")
