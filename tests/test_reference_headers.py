"""Drop-in boundary against the reference's own headers (this container only; skipped on the GPU box,
where /root/reference does not exist): every function of include/nut.h and include/nrf.h is
re-declared with the prototype text taken from /root/reference/src/{nut,nrf}.h at test time -- C
rejects a redeclaration whose type differs -- and the nut_buffer layout, enum values and NRF_*
constants are compared with the reference's through two small compiled probes.  Nothing of the
reference is copied into the repository: the prototypes are read, compiled and discarded."""
import os
import re
import subprocess

import pytest

from frequensea_amd import nrf
from tests.conftest import ROOT

REF_SRC = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF_SRC, "nrf.h")), reason="reference tree absent")


def _prototypes(header, names):
    text = open(os.path.join(REF_SRC, header)).read()
    found = {}
    for m in re.finditer(r"^([A-Za-z_][\w \*]*?\b(\w+)\s*\([^;{]*\))\s*;", text, flags=re.M):
        if m.group(2) in names:
            found[m.group(2)] = m.group(1)
    return found


def test_every_function_has_the_reference_prototype(tmp_path):
    nut = _prototypes("nut.h", set(nrf.NUT_EXPORTS))
    nrf_protos = _prototypes("nrf.h", set(nrf.NRF_EXPORTS))
    assert set(nut) == set(nrf.NUT_EXPORTS), set(nrf.NUT_EXPORTS) ^ set(nut)
    assert set(nrf_protos) == set(nrf.NRF_EXPORTS), set(nrf.NRF_EXPORTS) ^ set(nrf_protos)
    # the additions (nrf_fft_set_window*) are additions: the reference's header declares no function of those names, and
    # every function it does declare on this path is still here with its own prototype text (the redeclaration below)
    assert not _prototypes("nrf.h", set(nrf.NRF_ADDITIONS))
    src = tmp_path / "redeclare.c"
    src.write_text('#include "nut.h"\n#include "nrf.h"\n' +
                   "".join(p + ";\n" for p in list(nut.values()) + list(nrf_protos.values())))
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only",
                          "-I" + os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


PROBE = r"""
#include <stddef.h>
#include <stdio.h>
#include "nut.h"
int main(void) {
    printf("sizeof %zu type %zu length %zu channels %zu size_bytes %zu data %zu\n", sizeof(nut_buffer),
           offsetof(nut_buffer, type), offsetof(nut_buffer, length), offsetof(nut_buffer, channels),
           offsetof(nut_buffer, size_bytes), offsetof(nut_buffer, data));
    printf("u8 %d f64 %d\n", (int)NUT_BUFFER_U8, (int)NUT_BUFFER_F64);
    return 0;
}
"""


def test_nut_buffer_layout_and_enums_match_the_reference(tmp_path):
    (tmp_path / "probe.c").write_text(PROBE)
    outs = []
    for tag, inc in (("ref", REF_SRC), ("ours", os.path.join(ROOT, "include"))):
        exe = tmp_path / ("probe_" + tag)
        subprocess.run(["gcc", "-std=c99", "-I" + inc, "-o", str(exe), str(tmp_path / "probe.c")], check=True)
        outs.append(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout)
    assert outs[0] == outs[1], outs


def test_nrf_constants_and_block_types_match_the_reference():
    ref = open(os.path.join(REF_SRC, "nrf.h")).read()
    ours = open(os.path.join(ROOT, "include", "nrf.h")).read()

    def define(text, name):
        m = re.search(r"#define\s+%s\s+(.+)" % name, text)
        assert m, name
        return eval(re.sub(r"/\*.*?\*/", "", m.group(1)).strip())        # plain integer expressions

    for name in ("NRF_BUFFER_SIZE_BYTES", "NRF_SAMPLES_LENGTH", "NRF_IQ_RESOLUTION", "DEFAULT_FFT_SIZE",
                 "DEFAULT_FFT_HISTORY_SIZE"):
        assert define(ref, name) == define(ours, name), name
    # enum nrf_block_type: SOURCE = 1, GENERIC, SINK -- main.cpp:616-624 asserts type in {1,2,3}
    for text in (ref, ours):
        m = re.search(r"NRF_BLOCK_SOURCE\s*=\s*1\s*,\s*NRF_BLOCK_GENERIC\s*,\s*NRF_BLOCK_SINK", text)
        assert m
    # NRF_BLOCK first in every block struct we define
    for struct in ("nrf_fft", "nrf_freq_shifter"):
        m = re.search(r"typedef struct \{\s*NRF_BLOCK;[^}]*\}\s*%s;" % struct, ours)
        assert m, struct
    assert re.search(r"struct nrf_device \{\s*NRF_BLOCK;", ours)


LINK_MAIN = r"""
#include "nrf.h"
/* what src/main.cpp's wrappers do, in C: l_nrf_fft_new / _process / _get_buffer / _shift / _free */
int main(int argc, char **argv) {
    (void)argv;
    if (argc > 1000) { /* never: this program is linked, not run (no GPU in this container) */
        nrf_fft *fft = nrf_fft_new(1024, 1024);
        nut_buffer *in = nut_buffer_new_u8(NRF_SAMPLES_LENGTH, 2, NULL);
        nrf_fft_process(fft, in);
        nut_buffer *out = nrf_fft_get_buffer(fft);
        nrf_fft_shift(fft, 8.0);
        nrf_fft_set_window(fft, "hann");     /* the additions (include/nrf.h) link the same way */
        nrf_fft_set_window_weights(fft, NULL);
        nut_buffer_free(out);
        nut_buffer_free(in);
        nrf_fft_free(fft);
    }
    return 0;
}
"""

BLOCK_INIT = r"""
#include "nrf.h"
/* stands for the application's own src/nrf.c:24-31, which stays in the application */
void nrf_block_init(nrf_block *block, nrf_block_type type, nrf_block_process_fn process_fn,
                    nrf_block_result_fn result_fn) {
    block->type = type;
    block->process_fn = process_fn;
    block->result_fn = result_fn;
    block->n_outputs = 0;
}
"""


def test_fft_only_library_links_next_to_the_reference_nut(tmp_path):
    """libfsea_nrf_fft.so (the five nrf_fft_* functions only) + the reference's own src/nut.c compiled
    as is: every symbol resolves, none is defined twice -- the link line INTEGRATION.md describes."""
    pkg = os.path.join(ROOT, "frequensea_amd")
    if not os.path.exists(os.path.join(pkg, "libfsea_nrf_fft.so")):
        pytest.skip("libfsea_nrf_fft.so not built")
    (tmp_path / "main.c").write_text(LINK_MAIN)
    (tmp_path / "block.c").write_text(BLOCK_INIT)
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "--std=c99", "-g", "-Wall", "-Werror", "-pedantic", "-c", "-I" + REF_SRC,
                    os.path.join(REF_SRC, "nut.c"), "-o", str(tmp_path / "ref_nut.o")], check=True)
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + inc, str(tmp_path / "main.c"),
                          str(tmp_path / "block.c"), str(tmp_path / "ref_nut.o"), "-L" + pkg, "-lfsea_nrf_fft",
                          "-lfsea_hip", "-Wl,-rpath," + pkg, "-lm", "-lpthread", "-o", str(tmp_path / "app")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(pkg, "libfsea_nrf_fft.so")],
                          capture_output=True, text=True, check=True).stdout.split()
    defined = sorted(s for s in syms if s.startswith("nrf_") or s.startswith("nut_"))
    assert defined == ["nrf_fft_free", "nrf_fft_get_buffer", "nrf_fft_new", "nrf_fft_process", "nrf_fft_set_window",
                       "nrf_fft_set_window_weights", "nrf_fft_shift"]       # the reference's five + the two taper additions
