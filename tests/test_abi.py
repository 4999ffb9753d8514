"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares;
the header is plain C (compiles with gcc -std=c99, no HIP/CUDA headers); the product never
touches oracle/."""
import ctypes
import glob
import os
import re
import subprocess

import fem_amd as fa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DIST_LIB = os.path.join(ROOT, "fem_amd", "lib", "libfem_dist.so")


def declared_symbols(header="fem_hip.h"):
    names = set()
    for h in [os.path.join(ROOT, "include", header)]:
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        text = re.sub(r"#.*", "", text)
        for m in re.finditer(r"\b(\w+)\s*\(", text):
            name = m.group(1)
            if name.startswith("fem_") or name == "step":
                names.add(name)
    names.discard("fem_uint")
    return names


def test_every_declared_symbol_is_exported():
    names = declared_symbols()
    assert "step" in names and "fem_step" in names and "fem_plan_run" in names and len(names) > 40
    lib = ctypes.CDLL(fa.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_dist_library_exports_its_header():
    """libfem_dist.so (RCCL halo exchange, include/fem_dist.h): symbol table only -- loading it here would bring a
    second RCCL into a process that may already hold torch's"""
    names = declared_symbols("fem_dist.h") - declared_symbols("fem_hip.h")
    assert "fem_dist_run" in names and "fem_dist_rccl_comm_create" in names and len(names) >= 12
    exported = set(re.findall(r" T (\w+)", subprocess.run(["nm", "-D", "--defined-only", DIST_LIB], capture_output=True, text=True).stdout))
    assert not (names - exported), sorted(names - exported)
    deps = subprocess.run(["ldd", DIST_LIB], capture_output=True, text=True).stdout
    assert "librccl" in deps and "libfem_hip" in deps and "oracle" not in deps
    assert "librccl" not in subprocess.run(["ldd", fa.LIB_PATH], capture_output=True, text=True).stdout


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "fem_hip.h"\n#include "fem_dist.h"\nint main(void){fem_model m; fem_default_model(&m); return sizeof(fem_double2)==16 && sizeof(fem_float4)==16 ? 0 : 1;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                           str(tmp_path / "t.o")])


def test_status_strings_and_default_model():
    m = fa.default_model()                       # reference src/FEM_common.cu:9-13
    assert (m.nMax, m.D, m.r, m.alpha, m.S) == (1.0, 0.25, 59.453790914308100, 269.0, 1.0)
    for code in range(0, -7, -1):
        assert fa.lib.fem_error_string(code) not in (None, b"", b"unknown status")


def test_product_does_not_use_the_oracle():
    for path in glob.glob(os.path.join(ROOT, "fem_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
            text = open(path, errors="ignore").read()
            assert "oracle_lib" not in text and "fem_oracle" not in text and "libfem_oracle" not in text, path
    deps = subprocess.run(["ldd", fa.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps
