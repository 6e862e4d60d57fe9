"""CPU tier: the resource usage of every kernel in the shipped libfsea_hip.so, read from the code objects' own
metadata (the clang offload bundle inside the library -> gfx950 ELF -> NT_AMDGPU_METADATA note).  A register spill or
a scratch allocation in a hot kernel is a silent 5-10 % (it happened once during round 2: a change that moved the
twiddle build into the loop cost 11-25 spilled VGPRs); this pins the budget so that a build that regresses fails here,
without a GPU."""
import os
import re
import struct
import subprocess

import msgpack
import pytest

from tests.conftest import ROOT

LIB = os.path.join(ROOT, "frequensea_amd", "libfsea_hip.so")
SIZES = (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384)
KINDS = ("u8_mag", "u8_db5", "u8_db10", "u8", "u8_rot", "f32")
WIN_KINDS = ("u8_mag_win", "u8_db5_win", "u8_db10_win", "u8_win", "u8_rot_win", "f32_win")
ALT_CONFIGS = ("256rows", "512px", "1024rt")


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
    # + the half-overlap MAG kernels of the two sizes with one frame per workgroup (hop == N/2: every sample loaded once)
    # + the windowed kernels (fsea_plan_set_window): MAG, run-time mode, and the half-overlap MAG kernels again
    # + the per-(size, mode) configurations: 256 points for f32 rows, 512 for u8 pixels, 1024 for COMPLEX_F32 rows (full sets)
    assert fft == sorted(["fsea_fft%s_%s" % (n, kind) for n in SIZES + ALT_CONFIGS for kind in KINDS + WIN_KINDS] +
                         ["fsea_fft%d_u8_mag_half%s" % (n, w) for n in (8192, 16384) for w in ("", "_win")])
    assert not [k for k in kernels if "abl" in k]


def test_hot_kernels_do_not_spill(kernels):
    for name in ["fsea_fft%d_u8_mag_half%s" % (n, w) for n in (8192, 16384) for w in ("", "_win")]:
        k = kernels[name]
        assert k[".vgpr_count"] <= 256 and k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, name
    for n in SIZES + ALT_CONFIGS:
        for kind in KINDS + WIN_KINDS:
            k = kernels["fsea_fft%s_%s" % (n, kind)]
            assert k[".vgpr_count"] <= 256 and k[".wavefront_size"] == 64
            if (n, kind) == (1024, "u8_win"):
                # the run-time-mode windowed kernel at 1024 points (four bins per lane, 32 weights in flight): 4 registers
                assert k[".vgpr_spill_count"] <= 4 and k[".private_segment_fixed_size"] <= 32, (n, kind)
                continue
            if kind == "u8_rot_win":
                # the frequency-shifted kernel with the taper (lua/fft-shifted.lua's chain on a windowed plan, one frame per
                # call): phasors and 32 resident weights together spill 13 registers at 1024 points and 8 at 4096
                assert k[".vgpr_spill_count"] <= 16 and k[".private_segment_fixed_size"] <= 64, (n, kind)
                continue
            if kind == "f32":
                # the f32-complex input branch (NUT_BUFFER_F64, one frame per call) prefetches 64 VGPRs of rows: a few
                # spilled registers at two sizes are accepted there, not more
                assert k[".vgpr_spill_count"] <= 16 and k[".private_segment_fixed_size"] <= 64, (n, kind)
            else:
                assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, (n, kind, k[".vgpr_spill_count"])


def test_occupancy_budget_of_the_multi_wave_sizes(kernels):
    """Two workgroups per CU at 4096 and 8192 points (LDS <= 80 KB each, <= 256 VGPRs at 2 waves per SIMD), one at 16384."""
    for n, lds_max in ((4096, 80 * 1024), (8192, 80 * 1024), (16384, 160 * 1024)):
        for kind in KINDS + WIN_KINDS:
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


def _store_hazard_findings(lib, tmp_path, tag):
    """(number of 12/16-byte stores, list of violations) in the gfx950 code objects of `lib`.
    buffer_store_dwordx3/x4 -- only bst128() of fsea_fft_core.h emits them, with an SGPR soffset, the form LLVM's hazard
    recogniser exempts -- must be followed by the fenced wait states of store_data_guard() (two `s_nop 7`).
    global/flat/scratch_store_dwordx3/x4 (compiler-generated in the plain HIP kernels) are covered by LLVM's own rule; what
    is checked is the rule's effect: no VALU write to one of the store's data registers within two wait states."""
    import re
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    stores, bad = 0, []
    for i, elf in enumerate(_code_objects(lib)):
        path = tmp_path / ("%s%d.elf" % (tag, i))
        path.write_bytes(elf)
        text = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(path)], capture_output=True, text=True, check=True).stdout
        lines = [ln.split("//")[0].strip() for ln in text.splitlines()]
        lines = [ln for ln in lines if ln and not ln.endswith(":") and not ln.startswith(("Disassembly", "/"))]
        for k, ln in enumerate(lines):
            m = re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]\b(.*)", ln)
            if not m:
                continue
            stores += 1
            if m.group(1) == "buffer":
                waits = 0
                for nxt in lines[k + 1: k + 8]:
                    mm = re.match(r"s_nop (\d+)", nxt)
                    if mm:
                        waits += int(mm.group(1)) + 1
                        continue
                    if nxt.startswith(("v_", "ds_", "buffer_", "global_")):
                        break
                if waits < 8:
                    bad.append("16-byte buffer store without its wait states: %r followed by %r" % (ln, lines[k + 1: k + 4]))
                continue
            regs = re.findall(r"v\[(\d+):(\d+)\]", m.group(2))
            data = regs[-1] if m.group(1) == "global" and len(regs) > 1 else (regs[0] if regs else None)
            if m.group(1) != "global" and len(regs) > 1:
                data = regs[1]
            if data is None:
                continue
            lo, hi = int(data[0]), int(data[1])
            waited = 0
            for nxt in lines[k + 1: k + 4]:
                if waited >= 2:
                    break
                mm = re.match(r"s_nop (\d+)", nxt)
                if mm:
                    waited += int(mm.group(1)) + 1
                    continue
                if nxt.startswith("v_"):
                    d = re.match(r"v_\S+\s+(v\[(\d+):(\d+)\]|v(\d+))", nxt)
                    if d:
                        a, b = (int(d.group(2)), int(d.group(3))) if d.group(2) else (int(d.group(4)), int(d.group(4)))
                        if a <= hi and b >= lo:
                            bad.append("VALU write to store data %r right behind %r" % (nxt, ln))
                waited += 1
    return stores, bad


def test_every_16_byte_store_is_followed_by_its_wait_states(tmp_path):
    """DESIGN.md section 3, the store-data hazard: a VALU write to a data register of a 16-byte store too soon behind it
    changes what the store writes.  Checked in the disassembly of EVERY shipped code object (the product library incl. the
    plain HIP kernels of fsea_api.hip, the gather library, and the tuning library), so that a 12- or 16-byte store added
    past bst128() fails here without a GPU.  FSEA_ARTIFACT_LIB=<path> checks another build of libfsea_hip.so instead
    (a -DFSEA_STORE_GUARD=0 build must fail: scripts/store_guard_regression.sh)."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.environ.get("FSEA_ARTIFACT_LIB", LIB)
    if not os.path.exists(objdump) or not os.path.exists(lib):
        pytest.skip("llvm-objdump or libfsea_hip.so not available")
    stores, bad = _store_hazard_findings(lib, tmp_path, "hip")
    assert not bad, bad[:5]
    assert stores > 50        # the 64-, 128- and 2048-point f32 rows and the complex rows do use them
    pkg = os.path.join(ROOT, "frequensea_amd")
    for name in ("libfsea_hip_tune.so", "libfsea_rccl.so"):
        path = os.path.join(pkg, name)
        if os.path.exists(path) and "FSEA_ARTIFACT_LIB" not in os.environ:
            _, bad = _store_hazard_findings(path, tmp_path, name[:12])
            assert not bad, (name, bad[:5])


def test_a_16_byte_store_past_the_guarded_primitive_does_not_compile(tmp_path):
    """The raw 12- and 16-byte buffer-store builtins are poisoned behind bst128() in fsea_fft_core.h."""
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = tmp_path / "bypass.hip"
    src.write_text('#include "fsea_fft_core.h"\n'
                   '__global__ void k(float *p) {\n'
                   '    fsea::rsrc_t rs = fsea::buffer_window(p, 0, 64);\n'
                   '    __builtin_amdgcn_raw_buffer_store_b128(fsea::u32x4{1, 2, 3, 4}, rs, 0, 0, 0);\n'
                   '}\n')
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "frequensea_amd", "csrc"),
                        "-c", str(src), "-o", str(tmp_path / "bypass.o")], capture_output=True, text=True)
    assert r.returncode != 0 and "poison" in r.stderr, r.stderr[-800:]


def _frame_loop_waits(lib, tmp_path, kernel):
    """The s_waitcnt vmcnt(N) values in front of the frame loop's byte conversions (v_cvt_f32_i32_sdwa) of `kernel`, in the
    order they stand in the shipped code object: what the wave lets stay in flight while it takes the next frame's bytes."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    for i, elf in enumerate(_code_objects(lib)):
        path = tmp_path / ("loop%d.elf" % i)
        path.write_bytes(elf)
        text = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(path)], capture_output=True, text=True, check=True).stdout
        m = re.search(r"<%s>:\n(.*?)(?=\n\n[0-9a-f]+ <|\Z)" % kernel, text, re.S)
        if not m:
            continue
        lines = [ln.split("//")[0].strip() for ln in m.group(1).splitlines() if ln.strip()]
        waits = []
        for k, ln in enumerate(lines):
            w = re.match(r"s_waitcnt vmcnt\((\d+)\)$", ln)
            if w and any(nxt.startswith("v_cvt_f32_i32_sdwa") for nxt in lines[k + 1: k + 3]):
                waits.append(int(w.group(1)))
        return waits
    return None


def test_the_frame_loop_does_not_wait_for_the_previous_frames_row_stores(tmp_path):
    """DESIGN.md section 3 (round 5): gfx950 counts loads and stores in one vmcnt, and the compiler's merge at the loop header
    made every frame wait for the previous frame's row stores (vmcnt(15) ... vmcnt(0) in front of the byte conversions).  With
    the prologue's balancing stores (FftKernel::balance_vmcnt) the same compiler leaves the S stores of an iteration out of the
    wait: the smallest vmcnt in front of a conversion must be about S, not 0.  A compiler that merges differently, or a change
    to the loop that breaks the balance, shows up here without a GPU."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump) or not os.path.exists(LIB):
        pytest.skip("llvm-objdump or libfsea_hip.so not available")
    # kernel: stores per lane and iteration (rows x pieces) the wait has to leave out, with a margin of two for the DC-patch
    # store and the ticket request that may sit between them
    for kernel, stores in (("fsea_fft8192_u8_mag", 32), ("fsea_fft8192_u8_db5", 32), ("fsea_fft4096_u8_db5", 16),
                           ("fsea_fft4096_u8_mag", 16), ("fsea_fft2048_u8_mag", 8), ("fsea_fft1024_u8_mag", 8),
                           ("fsea_fft1024_u8_db10", 8), ("fsea_fft256_u8_db5", 16)):
        waits = _frame_loop_waits(LIB, tmp_path, kernel)
        assert waits, kernel
        assert min(waits) >= stores - 2, (kernel, waits)
    # the kernels that are left as they were (windowed: a longer prologue loses it again) still show the full wait -- if this
    # ever changes, profiles/r05_prologue_stores.txt's "left off" has to be measured again
    assert min(_frame_loop_waits(LIB, tmp_path, "fsea_fft8192_u8_mag_win")) == 0
