"""The C++ host mirror (include/lz4.hpp: pierrec/lz4's Go API restated over the C ABI)
run as a compiled program: tests/cpp/host_test.cpp restates the reference's own tests
(block_test.go, decode_test.go, writer_test.go, reader_test.go) and checks compressed
bytes against the oracle.  The binary is built on the CPU by __graft_entry__.build()
(make -C tests/cpp); here it is only run, as a child process."""
import os
import subprocess

import pytest

from conftest import GOLDEN

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "host_test")


def test_lz4_hpp_compiles_standalone(tmp_path):
    """CPU: the header is self-contained C++17 (no HIP, no torch types in the API)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "lz4.hpp"\nint main() { return lz4::CompressBlockBound(0) == 16 ? 0 : 1; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror",
                        "-I", os.path.join(HERE, "..", "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_cpp_host_mirror_reference_tests():
    assert os.path.exists(BIN), "tests/cpp/host_test is not built: run __graft_entry__.build()"
    r = subprocess.run([BIN, os.path.join(GOLDEN, "testdata")], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout
