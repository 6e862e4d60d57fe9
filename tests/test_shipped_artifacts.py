"""CPU tier: the resource usage of every kernel in the shipped libfsea_hip.so, read from the code objects' own
metadata (the clang offload bundle inside the library -> gfx950 ELF -> NT_AMDGPU_METADATA note).  A register spill or
a scratch allocation in a hot kernel is a silent 5-10 % (it happened once during round 2: a change that moved the
twiddle build into the loop cost 11-25 spilled VGPRs); this pins the budget so that a build that regresses fails here,
without a GPU."""
import os
import struct

import msgpack
import pytest

from tests.conftest import ROOT

LIB = os.path.join(ROOT, "frequensea_amd", "libfsea_hip.so")
SIZES = (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384)
KINDS = ("u8_mag", "u8_db5", "u8_db10", "u8", "u8_rot", "f32")


def _kernels(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = {}
    pos = data.find(magic)
    while pos >= 0:
        count = struct.unpack_from("<Q", data, pos + 24)[0]
        off = pos + 32
        for _ in range(count):
            o, size, tlen = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24: off + 24 + tlen].decode()
            off += 24 + tlen
            if "gfx950" not in triple or size == 0:
                continue
            elf = data[pos + o: pos + o + size]
            shoff = struct.unpack_from("<Q", elf, 0x28)[0]
            shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
            for i in range(shnum):
                sh = shoff + i * shentsize
                if struct.unpack_from("<I", elf, sh + 4)[0] != 7:          # SHT_NOTE
                    continue
                s_off, s_size = struct.unpack_from("<QQ", elf, sh + 0x18)
                p = s_off
                while p < s_off + s_size:
                    namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
                    name = elf[p + 12: p + 12 + namesz].rstrip(b"\0")
                    d0 = p + 12 + ((namesz + 3) & ~3)
                    if name == b"AMDGPU" and ntype == 32:                   # NT_AMDGPU_METADATA
                        for k in msgpack.unpackb(elf[d0: d0 + descsz], raw=False)["amdhsa.kernels"]:
                            out[k[".name"]] = k
                    p = d0 + ((descsz + 3) & ~3)
        pos = data.find(magic, pos + 1)
    return out


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("libfsea_hip.so not built")
    ks = _kernels(LIB)
    assert ks, "no gfx950 code object found in libfsea_hip.so"
    return ks


def test_every_size_has_its_six_entry_points_and_nothing_experimental(kernels):
    fft = sorted(k for k in kernels if k.startswith("fsea_fft"))
    assert fft == sorted("fsea_fft%d_%s" % (n, kind) for n in SIZES for kind in KINDS)
    assert not [k for k in kernels if "abl" in k]


def test_hot_kernels_do_not_spill(kernels):
    for n in SIZES:
        for kind in KINDS:
            k = kernels["fsea_fft%d_%s" % (n, kind)]
            assert k[".vgpr_count"] <= 256 and k[".wavefront_size"] == 64
            if kind == "f32":
                # the f32-complex input branch (NUT_BUFFER_F64, one frame per call) prefetches 64 VGPRs of rows: a few
                # spilled registers at two sizes are accepted there, not more
                assert k[".vgpr_spill_count"] <= 16 and k[".private_segment_fixed_size"] <= 64, (n, kind)
            else:
                assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, (n, kind, k[".vgpr_spill_count"])


def test_occupancy_budget_of_the_multi_wave_sizes(kernels):
    """Two workgroups per CU at 4096 and 8192 points (LDS <= 80 KB each, <= 256 VGPRs at 2 waves per SIMD), one at 16384."""
    for n, lds_max in ((4096, 80 * 1024), (8192, 80 * 1024), (16384, 160 * 1024)):
        for kind in KINDS:
            k = kernels["fsea_fft%d_%s" % (n, kind)]
            assert k[".group_segment_fixed_size"] <= lds_max, (n, kind)
    assert kernels["fsea_fft8192_u8_mag"][".max_flat_workgroup_size"] == 256
    assert kernels["fsea_fft16384_u8_mag"][".max_flat_workgroup_size"] == 512


def test_libraries_link_no_fft_or_blas_library():
    """The transform is this repository's own kernels: nothing in the dynamic dependencies of the shipped libraries
    (or of the tools) that could compute an FFT or a GEMM for them -- no rocFFT / hipFFT / hipfftw / rocBLAS / hipBLASLt /
    MIOpen, and RCCL only in the multi-GPU gather library."""
    import subprocess
    pkg = os.path.join(ROOT, "frequensea_amd")
    files = [os.path.join(pkg, f) for f in ("libfsea_hip.so", "libfsea_nrf.so", "libfsea_nrf_fft.so", "libfsea_rccl.so")]
    files += [os.path.join(pkg, "bin", f) for f in ("fsea-fft-batch", "fsea-fft-stitch", "fsea-fft-sweep", "fsea-add-markers")]
    banned = ("fft", "blas", "miopen", "rocsolver", "rocsparse", "mkl", "torch")
    for path in files:
        if not os.path.exists(path):
            pytest.skip("not built: " + path)
        out = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, check=True).stdout
        needed = [ln.split("[")[1].split("]")[0].lower() for ln in out.splitlines() if "(NEEDED)" in ln]
        assert needed, path
        for lib in needed:
            assert not any(b in lib for b in banned if not lib.startswith("libfsea")), (path, lib)
            if lib.startswith("librccl"):                  # RCCL itself: only the gather library links it
                assert os.path.basename(path) == "libfsea_rccl.so", (path, lib)
            if lib == "libfsea_rccl.so":                   # and only the multi-GPU tool links that
                assert os.path.basename(path) == "fsea-fft-sweep", (path, lib)
    # and the product library does not reach the oracle
    for path in files:
        out = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, check=True).stdout
        assert "oracle" not in out.lower()


def test_the_oracle_is_reachable_from_the_checkers_only():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import,
    link or run it.  No source file of the product (frequensea_amd/, include/) names it in code, bench.py imports it in
    one function only, and that function feeds cpu_baseline and nothing else."""
    import ast
    import re
    pkg = os.path.join(ROOT, "frequensea_amd")
    for dirpath, _, names in os.walk(pkg):
        if "build" in dirpath or "__pycache__" in dirpath:
            continue
        for name in names:
            if not name.endswith((".py", ".c", ".h", ".hip", ".cpp")) and name != "Makefile":
                continue
            text = open(os.path.join(dirpath, name), errors="replace").read()
            code = re.sub(r'""".*?"""|/\\*.*?\\*/|//[^\\n]*|#[^\\n]*', "", text, flags=re.S)   # comments and docstrings may mention it
            assert "oracle" not in code.lower(), os.path.join(dirpath, name)
    for name in os.listdir(os.path.join(ROOT, "include")):
        text = open(os.path.join(ROOT, "include", name)).read()
        assert not re.search(r'#\\s*include\\s*[<"][^>"]*oracle', text), name
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    importers = []
    for fn in ast.walk(tree):
        if isinstance(fn, ast.FunctionDef):
            for node in ast.walk(fn):
                if isinstance(node, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(node):
                    importers.append(fn.name)
    assert importers == ["cpu_baseline"], importers
    for node in tree.body:                                   # and no module-level import of it
        assert not (isinstance(node, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(node))


def _code_objects(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = data.find(magic)
    while pos >= 0:
        count = struct.unpack_from("<Q", data, pos + 24)[0]
        off = pos + 32
        for _ in range(count):
            o, size, tlen = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24: off + 24 + tlen].decode()
            off += 24 + tlen
            if "gfx950" in triple and size:
                yield data[pos + o: pos + o + size]
        pos = data.find(magic, pos + 1)


def test_every_16_byte_store_is_followed_by_its_wait_states(tmp_path):
    """DESIGN.md section 3, the store-data hazard: a VALU write to a data register of a 16-byte store too soon behind it
    changes what the store writes.  Every such store in the shipped kernels must be followed by the fenced wait states of
    store_data_guard() (two `s_nop 7`) before the next vector instruction -- checked in the disassembly of the code
    objects inside libfsea_hip.so, so that a 16-byte store added past `bst` fails here without a GPU."""
    import re
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump) or not os.path.exists(LIB):
        pytest.skip("llvm-objdump or libfsea_hip.so not available")
    stores = 0
    for i, elf in enumerate(_code_objects(LIB)):
        path = tmp_path / ("co%d.elf" % i)
        path.write_bytes(elf)
        text = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(path)], capture_output=True, text=True, check=True).stdout
        lines = [ln.split("//")[0].strip() for ln in text.splitlines()]
        lines = [ln for ln in lines if ln and not ln.endswith(":") and not ln.startswith(("Disassembly", "/"))]
        for k, ln in enumerate(lines):
            if not re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]\b", ln):
                continue
            stores += 1
            waits = 0
            for nxt in lines[k + 1: k + 8]:
                m = re.match(r"s_nop (\d+)", nxt)
                if m:
                    waits += int(m.group(1)) + 1
                    continue
                if nxt.startswith(("v_", "ds_", "buffer_", "global_")):
                    break
            assert waits >= 8, "16-byte store without its wait states: %r followed by %r" % (ln, lines[k + 1: k + 4])
    assert stores > 50        # the 64-, 128- and 2048-point f32 rows and the complex rows do use them
