"""CPU tier: the TrueType label renderer (include/ntt_font.h, frequensea_amd/host/ntt_font.c).

Pinned three ways:
  * against the reference's own vendored rasteriser, stb_truetype v1.02, compiled as it lies under
    /root/reference/externals/stb into oracle/_ref/libstbtt_ref.so (oracle/ref_stbtt.c): glyph indices,
    font and glyph metrics, kerning and bitmap boxes must be IDENTICAL; coverage must agree closely (stb 1.02
    samples five scanlines per pixel row, this rasteriser integrates exactly);
  * against FreeType (PIL) for the shape of whole strings;
  * the helper semantics of c/fft-stitch.c:97-157 (centring, baseline, truncated advances, max-composite).
Fonts: the image's DejaVu files; the reference's Roboto files when /root/reference is present."""
import ctypes
import os

import numpy as np
import pytest

from frequensea_amd import nrf
from tests.conftest import ROOT

DEJAVU = "/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf"
FONTS = [p for p in (DEJAVU, "/usr/share/fonts/truetype/dejavu/DejaVuSansMono-Bold.ttf",
                     "/root/reference/fonts/RobotoCondensed-Regular.ttf", "/root/reference/fonts/RobotoCondensed-Bold.ttf")
         if os.path.exists(p)]
STB = os.path.join(ROOT, "oracle", "_ref", "libstbtt_ref.so")
LABEL_CHARS = "0123456789.-"

pytestmark = pytest.mark.skipif(not FONTS, reason="no TrueType font on this machine")


def _lib():
    L = nrf.nrf_lib()
    vp, ci, cf_ = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    ip = ctypes.POINTER(ci)
    L.ntt_font_load.restype = vp
    L.ntt_font_load.argtypes = [ctypes.c_char_p]
    L.ntt_font_free.argtypes = [vp]
    L.ntt_font_glyph_index.argtypes = [vp, ci]
    L.ntt_font_scale_for_pixel_height.restype = cf_
    L.ntt_font_scale_for_pixel_height.argtypes = [vp, cf_]
    L.ntt_font_vmetrics.argtypes = [vp, ip, ip, ip]
    L.ntt_font_hmetrics.argtypes = [vp, ci, ip, ip]
    L.ntt_font_kern_advance.argtypes = [vp, ci, ci]
    L.ntt_font_bitmap_box.argtypes = [vp, ci, cf_, ip, ip, ip, ip]
    L.ntt_font_glyph_bitmap.restype = ctypes.POINTER(ctypes.c_uint8)
    L.ntt_font_glyph_bitmap.argtypes = [vp, ci, cf_, ip, ip, ip, ip]
    L.ntt_font_measure.argtypes = [vp, ctypes.c_char_p, ci, ci, ci, ip, ip]
    L.ntt_font_draw.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ci, ci, ci]
    return L


def _glyph(L, font, g, scale):
    w, h, dx, dy = (ctypes.c_int() for _ in range(4))
    p = L.ntt_font_glyph_bitmap(font, g, scale, ctypes.byref(w), ctypes.byref(h), ctypes.byref(dx), ctypes.byref(dy))
    if not p:
        return np.zeros((0, 0), np.uint8), dx.value, dy.value
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value)).copy()
    ctypes.CDLL(None).free(p)
    return out, dx.value, dy.value


@pytest.mark.skipif(not os.path.exists(STB), reason="oracle/_ref/libstbtt_ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("path", FONTS)
@pytest.mark.parametrize("px", [48, 64, 17])
def test_against_the_references_stb_truetype(path, px):
    L = _lib()
    S = ctypes.CDLL(STB)
    data = open(path, "rb").read()
    buf = ctypes.create_string_buffer(data, len(data))
    info = ctypes.create_string_buffer(512)                         # stbtt_fontinfo, opaque
    assert S.stbtt_InitFont(info, buf, S.stbtt_GetFontOffsetForIndex(buf, 0))
    S.stbtt_ScaleForPixelHeight.restype = ctypes.c_float
    S.stbtt_ScaleForPixelHeight.argtypes = [ctypes.c_void_p, ctypes.c_float]
    S.stbtt_GetGlyphBitmapBox.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float] + [ctypes.POINTER(ctypes.c_int)] * 4
    S.stbtt_GetGlyphBitmap.restype = ctypes.POINTER(ctypes.c_uint8)
    S.stbtt_GetGlyphBitmap.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 4
    font = L.ntt_font_load(path.encode())
    assert font
    scale = L.ntt_font_scale_for_pixel_height(font, float(px))
    assert scale == S.stbtt_ScaleForPixelHeight(info, float(px))
    a, d, g = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    a2, d2, g2 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.ntt_font_vmetrics(font, ctypes.byref(a), ctypes.byref(d), ctypes.byref(g))
    S.stbtt_GetFontVMetrics(info, ctypes.byref(a2), ctypes.byref(d2), ctypes.byref(g2))
    assert (a.value, d.value, g.value) == (a2.value, d2.value, g2.value)
    for ch in LABEL_CHARS + "AgW":
        gi = L.ntt_font_glyph_index(font, ord(ch))
        assert gi == S.stbtt_FindGlyphIndex(info, ord(ch)) and gi > 0
        adv, lsb, adv2, lsb2 = (ctypes.c_int() for _ in range(4))
        L.ntt_font_hmetrics(font, gi, ctypes.byref(adv), ctypes.byref(lsb))
        S.stbtt_GetGlyphHMetrics(info, gi, ctypes.byref(adv2), ctypes.byref(lsb2))
        assert (adv.value, lsb.value) == (adv2.value, lsb2.value)
        for other in "17.":
            go = L.ntt_font_glyph_index(font, ord(other))
            assert L.ntt_font_kern_advance(font, gi, go) == S.stbtt_GetGlyphKernAdvance(info, gi, go)
        box = [ctypes.c_int() for _ in range(4)]
        box2 = [ctypes.c_int() for _ in range(4)]
        L.ntt_font_bitmap_box(font, gi, scale, *[ctypes.byref(b) for b in box])
        S.stbtt_GetGlyphBitmapBox(info, gi, scale, scale, *[ctypes.byref(b) for b in box2])
        assert [b.value for b in box] == [b.value for b in box2], ch
        ours, dx, dy = _glyph(L, font, gi, scale)
        w, h, xo, yo = (ctypes.c_int() for _ in range(4))
        p = S.stbtt_GetGlyphBitmap(info, scale, scale, gi, ctypes.byref(w), ctypes.byref(h), ctypes.byref(xo), ctypes.byref(yo))
        if ours.size == 0:                                          # composite glyph: blank here
            continue
        theirs = np.ctypeslib.as_array(p, shape=(h.value, w.value)).copy()
        assert ours.shape == theirs.shape and (dx, dy) == (xo.value, yo.value), ch
        diff = np.abs(ours.astype(int) - theirs.astype(int))
        # same ink: total coverage within 2 %, mean difference a few grey levels, no pixel flipped
        assert abs(int(ours.sum()) - int(theirs.sum())) <= 0.02 * theirs.sum() + 64, ch
        # (tiny glyphs are all edge pixels: a looser mean there)
        assert diff.mean() <= (6.0 if px >= 48 else 14.0) and diff.max() <= 110, (ch, diff.mean(), diff.max())
    L.ntt_font_free(font)


def _best_iou(a, b, reach=3):
    best = 0.0
    for dy in range(-reach, reach + 1):
        for dx in range(-reach, reach + 1):
            s = np.roll(np.roll(b, dy, axis=0), dx, axis=1)
            best = max(best, (a & s).sum() / float((a | s).sum()))
    return best


@pytest.mark.parametrize("text", ["1802.00", "650.00", "-3.50", "8", "4"])
def test_string_shape_against_freetype(text):
    """A label drawn by ntt_font_draw against FreeType's rendering of the same string (PIL) at the size whose
    ascent + descent equals the pixel height (stbtt_ScaleForPixelHeight's meaning): same advance width within
    3 px; the inked pixels coincide (intersection over union, best alignment within 3 px: FreeType hints and
    keeps fractional advances, the reference's helper truncates each advance, c/fft-stitch.c:112-116)."""
    from PIL import Image, ImageDraw, ImageFont
    L = _lib()
    font = L.ntt_font_load(DEJAVU.encode())
    px = 64
    w, h = ctypes.c_int(), ctypes.c_int()
    L.ntt_font_measure(font, text.encode(), 0, 0, px, ctypes.byref(w), ctypes.byref(h))
    img = np.zeros((120, 400), np.uint8)
    L.ntt_font_draw(font, img.ctypes.data, 400, 120, text.encode(), 200, 20, px)
    scale = L.ntt_font_scale_for_pixel_height(font, float(px))
    asc = ctypes.c_int()
    L.ntt_font_vmetrics(font, ctypes.byref(asc), None, None)
    ft = ImageFont.truetype(DEJAVU, 64)
    a, d = ft.getmetrics()
    ft = ImageFont.truetype(DEJAVU, int(round(64 * px / float(a + d))))
    assert abs(ft.getlength(text) - w.value) <= 3
    ref = Image.new("L", (400, 120), 0)
    ImageDraw.Draw(ref).text((200 - w.value // 2, 20 + int(asc.value * scale)), text, fill=255, font=ft, anchor="ls")
    ref = np.array(ref)
    iou = _best_iou(img > 96, ref > 96)
    assert iou > (0.85 if len(text) == 1 else 0.7), iou
    L.ntt_font_free(font)


def test_draw_semantics_of_the_reference_helpers():
    L = _lib()
    font = L.ntt_font_load(FONTS[0].encode())
    w, h = ctypes.c_int(), ctypes.c_int()
    L.ntt_font_measure(font, b"100.00", 7, 9, 48, ctypes.byref(w), ctypes.byref(h))
    w0, h0 = ctypes.c_int(), ctypes.c_int()
    L.ntt_font_measure(font, b"100.00", 0, 0, 48, ctypes.byref(w0), ctypes.byref(h0))
    assert (w.value, h.value) == (w0.value + 7, h0.value + 9)                # c/fft-stitch.c:105-109
    scale = L.ntt_font_scale_for_pixel_height(font, 48.0)
    a, d = ctypes.c_int(), ctypes.c_int()
    L.ntt_font_vmetrics(font, ctypes.byref(a), ctypes.byref(d), None)
    assert h0.value == int(a.value * scale) - int(d.value * scale)
    img = np.zeros((80, 300), np.uint8)
    L.ntt_font_draw(font, img.ctypes.data, 300, 80, b"100.00", 150, 10, 48)
    ys, xs = np.nonzero(img)
    assert abs((xs.min() + xs.max()) / 2.0 - 150) <= 6                      # centred on x (:129)
    assert ys.min() >= 10 and ys.max() <= 10 + h0.value
    before = img.copy()
    L.ntt_font_draw(font, img.ctypes.data, 300, 80, b"100.00", 150, 10, 48)
    assert np.array_equal(img, before)                                      # max-composite is idempotent (:46-54)
    clipped = np.zeros((80, 300), np.uint8)
    L.ntt_font_draw(font, clipped.ctypes.data, 300, 80, b"100.00", 20, 10, 48)
    assert not clipped.any()                                                # start_x < 0: nothing drawn (:130)
    edge = np.zeros((30, 300), np.uint8)
    L.ntt_font_draw(font, edge.ctypes.data, 300, 30, b"100.00", 250, 10, 48)   # runs off the right and bottom edges
    assert edge.any()
    assert not L.ntt_font_load(b"/nonexistent.ttf")
    assert not L.ntt_font_load(os.path.join(ROOT, "README.md").encode())
    L.ntt_font_free(font)


def test_rulers_with_truetype_labels(tmp_path):
    """img_draw_frequency_axis / fsea-add-markers with a font: lines and ticks unchanged, labels centred on their ticks."""
    import subprocess
    from oracle import oracle as O
    from tests.test_host_api import _AxisCfg
    L = _lib()
    font = L.ntt_font_load(DEJAVU.encode())
    width, height, rows_ = 1024 + 4 * 512, 700, 100
    L.img_draw_frequency_axis.restype = ctypes.c_int
    L.img_draw_frequency_axis.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_AxisCfg)]
    plain = np.zeros((height, width), np.uint8)
    cfg = _AxisCfg(1024, rows_, 5000000, 2000000, 1802000000, 1810000000, 100000, 1000000, 0, 255, None)
    L.img_draw_frequency_axis(plain.ctypes.data, width, height, ctypes.byref(cfg))
    text = np.zeros((height, width), np.uint8)
    cfg = _AxisCfg(1024, rows_, 5000000, 2000000, 1802000000, 1810000000, 100000, 1000000, 48, 255, font)
    n = L.img_draw_frequency_axis(text.ctypes.data, width, height, ctypes.byref(cfg))
    _, labels = O.frequency_axis(width, height, rows_, 1024, 5000000, 2000000, 1802000000, 1810000000)
    assert n == len(labels)
    extra = text != plain
    ys, xs = np.nonzero(extra)
    markers_y = rows_ + (600 // 2 - 48 // 2)
    assert extra.any() and ys.min() >= markers_y and ys.max() <= markers_y + 48
    # the second label ("1801.00" at its tick) is centred on the tick column
    x1 = labels[1][0]
    cols = np.nonzero(extra[:, x1 - 100: x1 + 100].any(axis=0))[0] + x1 - 100
    assert abs((cols.min() + cols.max()) / 2.0 - x1) <= 6
    L.ntt_font_free(font)
    # the tool
    src = np.zeros((50, 256 * 11), np.uint8)
    L.write_gray_png.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    assert L.write_gray_png(str(tmp_path / "broad-stitched-630-680.png").encode(), src.shape[1], src.shape[0], src.ctypes.data) == 0
    tool = os.path.join(ROOT, "frequensea_amd", "bin", "fsea-add-markers")
    subprocess.run([tool, "--start", "630", "--end", "680", "--dir", str(tmp_path), "--font", DEJAVU], check=True,
                   capture_output=True)
    from PIL import Image
    got = np.array(Image.open(tmp_path / "broad-stitched-630-680-markers.png"))
    want, labels = O.broad_markers(src, 300, 300, 630000000, 680000000)
    (x, y, _), = labels
    band = np.zeros(got.shape, bool)
    band[y:y + 64, x - 120:x + 120] = True
    assert np.array_equal(got[~band], want[~band]) and got[band].max() == 255
    cols = np.nonzero((got != want)[y:y + 64].any(axis=0))[0]
    assert abs((cols.min() + cols.max()) / 2.0 - x) <= 6                     # centred on the 650 MHz tick
    assert subprocess.run([tool, "--start", "630", "--end", "680", "--dir", str(tmp_path), "--font", "/nonexistent.ttf"],
                          capture_output=True).returncode != 0


def test_damaged_font_files_are_refused_or_render_without_faults(tmp_path):
    """Truncated and bit-flipped copies of a font: ntt_font_load either refuses them or every call on the result
    stays inside the file image (scripts/asan_cpu_tests.sh runs this under AddressSanitizer; the long form --
    tens of thousands of cases, also aimed at single tables -- is scripts/fuzz_host_readers.sh, which found the
    out-of-box outline and the unchecked table reads this reader now guards against)."""
    L = _lib()
    data = bytearray(open(DEJAVU, "rb").read())
    rng = np.random.default_rng(5)
    cases = [bytes(data[:n]) for n in (0, 3, 12, 100, 1000, 5000, len(data) // 2, len(data) - 1)]
    for _ in range(40):
        d = bytearray(data)
        # damage the table directory, or a stretch anywhere
        lo = 0 if rng.integers(2) else int(rng.integers(0, len(d) - 64))
        for k in rng.integers(lo, lo + (300 if lo == 0 else 64), 6):
            d[int(k)] = int(rng.integers(0, 256))
        cases.append(bytes(d))
    img = np.zeros((80, 300), np.uint8)
    loaded = 0
    for i, blob in enumerate(cases):
        p = tmp_path / ("f%d.ttf" % i)
        p.write_bytes(blob)
        font = L.ntt_font_load(str(p).encode())
        if not font:
            continue
        loaded += 1
        w, h = ctypes.c_int(), ctypes.c_int()
        L.ntt_font_measure(font, b"0123456789.-AgW", 0, 0, 48, ctypes.byref(w), ctypes.byref(h))
        L.ntt_font_draw(font, img.ctypes.data, 300, 80, b"0123456789.-", 150, 10, 48)
        L.ntt_font_free(font)
    assert loaded >= 1                        # damage outside the tables the labels use still loads


@pytest.mark.parametrize("path", FONTS)
@pytest.mark.parametrize("px", [48, 31])
def test_measure_accumulates_kerning_like_the_reference_helper(path, px):
    """c/fft-stitch.c:104-123: `int glyph_width = advance * font_scale; *width += glyph_width;` truncates the addend, but
    `*width += font_scale * kern` on an int truncates the SUM.  The two differ for negative fractional kerns;
    the expected widths are computed here with exactly that arithmetic from this library's own metrics (which are pinned
    to the reference's stb_truetype in test_against_the_references_stb_truetype), on strings with kerned pairs."""
    L = _lib()
    font = L.ntt_font_load(path.encode())
    assert font
    scale = np.float32(L.ntt_font_scale_for_pixel_height(font, float(px)))
    kerned = 0
    for text in ("AV", "AVAVATo", "To.Wa,LT", "1802.5", "WAVE-11"):
        width = 0
        for i, ch in enumerate(text):
            g = L.ntt_font_glyph_index(font, ord(ch))
            adv, lsb = ctypes.c_int(), ctypes.c_int()
            L.ntt_font_hmetrics(font, g, ctypes.byref(adv), ctypes.byref(lsb))
            width += int(np.float32(adv.value) * scale)                       # truncated addend
            if i + 1 < len(text):
                k = L.ntt_font_kern_advance(font, g, L.ntt_font_glyph_index(font, ord(text[i + 1])))
                kerned += k != 0
                width = int(np.float32(width) + scale * np.float32(k))        # truncated sum
        w, h = ctypes.c_int(), ctypes.c_int()
        L.ntt_font_measure(font, text.encode(), 0, 0, px, ctypes.byref(w), ctypes.byref(h))
        assert w.value == width, (text, w.value, width)
    L.ntt_font_free(font)
    if "DejaVuSans" in path and "Mono" not in path:
        assert kerned > 0                                                     # DejaVu Sans kerns A-V, T-o, W-a
