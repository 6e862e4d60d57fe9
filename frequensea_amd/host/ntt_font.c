/*
 * ntt_font.c -- the label renderer of the stitched-image rulers (include/ntt_font.h).
 *
 * The reference draws its tick labels with three static helpers around the vendored stb_truetype
 * (paths under /root/reference):
 *   c/fft-stitch.c:81-94     ntt_font_load     read the .ttf, stbtt_InitFont
 *   c/fft-stitch.c:97-124    ntt_font_measure  advance widths (+ kerning) and the font's line height
 *   c/fft-stitch.c:126-157   ntt_font_draw     centre the string on x, one anti-aliased bitmap per glyph at
 *                                              (pen + left bearing, y + baseline + top), max-composited
 * (the same three in c/add-markers.c:45-133).  stb_truetype is a third-party dependency and is not
 * re-shipped; this file is a from-scratch TrueType reader and rasteriser that provides what those helpers
 * take from it, with stb's conventions where they decide pixel positions:
 *   scale            = pixel height / (hhea.ascent - hhea.descent)
 *   glyph bitmap box = floor(xMin s), floor(-yMax s) ... ceil(xMax s), ceil(-yMin s)
 *   baseline         = (int)(ascent * scale); advances and kerns are truncated to int per glyph
 * Coverage is the exact signed area of the outline inside every pixel (curves flattened to 0.35 px).
 * Supported: TrueType outlines ('glyf'), simple glyphs (the digits, '.', '-' of the labels are simple in
 * every font tried), cmap formats 4 and 12, 'kern' format 0.  Composite glyphs render as blanks.
 * The font FILE is the user's: nothing is embedded (fsea-fft-stitch --font FILE.ttf).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ntt_font.h"

struct ntt_font {
    uint8_t *data;
    size_t size;
    uint32_t cmap, loca, glyf, head, hhea, hmtx, kern; /* table offsets, 0 = absent */
    uint32_t cmap_sub;                                 /* chosen cmap subtable */
    int cmap_format;
    int loc_format, num_glyphs, num_hmetrics, units_per_em;
    int ascent, descent, line_gap;
};

/* Every read of the file image goes through these: an offset past the end reads as zero, so a damaged or
 * truncated file can make a label wrong but never makes the reader leave its buffer (tests/fuzz, scripts/fuzz_host_readers.sh). */
static uint32_t rd8(const ntt_font *f, size_t off) { return off < f->size ? f->data[off] : 0u; }
static uint32_t rd16(const ntt_font *f, size_t off) {
    return (off + 2 <= f->size && off + 2 > off) ? (uint32_t)((f->data[off] << 8) | f->data[off + 1]) : 0u;
}
static int rds16(const ntt_font *f, size_t off) { return (int16_t)rd16(f, off); }
static uint32_t rd32(const ntt_font *f, size_t off) {
    if (off + 4 > f->size || off + 4 < off) return 0u;
    const uint8_t *p = f->data + off;
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

static uint32_t find_table(const ntt_font *f, const char *tag) {
    const uint32_t n = rd16(f, 4);
    for (uint32_t i = 0; i < n; i++) {
        const size_t rec = 12 + 16 * (size_t)i;
        if (rec + 16 > f->size) return 0;
        if (memcmp(f->data + rec, tag, 4) == 0) {
            const uint32_t off = rd32(f, rec + 8), len = rd32(f, rec + 12);
            return ((size_t)off + len <= f->size) ? off : 0;
        }
    }
    return 0;
}

ntt_font *ntt_font_load(const char *font_file) {
    FILE *fp = fopen(font_file, "rb");
    if (!fp) {
        fprintf(stderr, "ERROR ntt_font_load: cannot open %s\n", font_file);
        return NULL;
    }
    fseek(fp, 0L, SEEK_END);
    const long size = ftell(fp);
    rewind(fp);
    ntt_font *f = (ntt_font *)calloc(1, sizeof(ntt_font));
    if (!f || size < 12) {
        fclose(fp);
        free(f);
        return NULL;
    }
    f->data = (uint8_t *)malloc((size_t)size);
    f->size = (size_t)size;
    if (!f->data || fread(f->data, 1, (size_t)size, fp) != (size_t)size) {
        fclose(fp);
        ntt_font_free(f);
        return NULL;
    }
    fclose(fp);
    f->cmap = find_table(f, "cmap");
    f->loca = find_table(f, "loca");
    f->glyf = find_table(f, "glyf");
    f->head = find_table(f, "head");
    f->hhea = find_table(f, "hhea");
    f->hmtx = find_table(f, "hmtx");
    f->kern = find_table(f, "kern");
    const uint32_t maxp = find_table(f, "maxp");
    if (!f->cmap || !f->loca || !f->glyf || !f->head || !f->hhea || !f->hmtx || !maxp) {
        fprintf(stderr, "ERROR ntt_font_load: %s is not a TrueType-outline font\n", font_file);
        ntt_font_free(f);
        return NULL;
    }
    f->num_glyphs = (int)rd16(f, (size_t)maxp + 4);
    f->units_per_em = (int)rd16(f, (size_t)f->head + 18);
    f->loc_format = rds16(f, (size_t)f->head + 50);
    f->ascent = rds16(f, (size_t)f->hhea + 4);
    f->descent = rds16(f, (size_t)f->hhea + 6);
    f->line_gap = rds16(f, (size_t)f->hhea + 8);
    f->num_hmetrics = (int)rd16(f, (size_t)f->hhea + 34);
    if (f->ascent - f->descent <= 0 || f->num_hmetrics < 1) {
        fprintf(stderr, "ERROR ntt_font_load: %s has unusable vertical / horizontal metrics\n", font_file);
        ntt_font_free(f);
        return NULL;
    }
    /* a Unicode cmap subtable: (3,10) / (0,4+) format 12 first, else (3,1) / (0,x) format 4 */
    const uint32_t n_sub = rd16(f, (size_t)f->cmap + 2);
    for (int pass = 0; pass < 2 && !f->cmap_sub; pass++) {
        for (uint32_t i = 0; i < n_sub; i++) {
            const size_t rec = (size_t)f->cmap + 4 + 8 * (size_t)i;
            if (rec + 8 > f->size) break;
            const uint32_t platform = rd16(f, rec), encoding = rd16(f, rec + 2);
            const size_t off = (size_t)f->cmap + rd32(f, rec + 4);
            if (off + 16 > f->size) continue;
            const uint32_t format = rd16(f, off);
            const int unicode = (platform == 0) || (platform == 3 && (encoding == 1 || encoding == 10));
            if (unicode && ((pass == 0 && format == 12) || (pass == 1 && format == 4))) {
                f->cmap_sub = (uint32_t)off;
                f->cmap_format = (int)format;
                break;
            }
        }
    }
    if (!f->cmap_sub) {
        fprintf(stderr, "ERROR ntt_font_load: %s has no Unicode cmap (format 4 or 12)\n", font_file);
        ntt_font_free(f);
        return NULL;
    }
    return f;
}

void ntt_font_free(ntt_font *font) {
    if (!font) return;
    free(font->data);
    free(font);
}

int ntt_font_glyph_index(const ntt_font *f, int codepoint) {
    const size_t t = f->cmap_sub;
    if (codepoint < 0) return 0;
    if (f->cmap_format == 4) {
        if (codepoint > 0xffff) return 0;
        const size_t segx2 = rd16(f, t + 6) & ~1u;
        const size_t end_code = t + 14, start_code = end_code + segx2 + 2;
        const size_t id_delta = start_code + segx2, id_range = id_delta + segx2;
        for (size_t i = 0; i < segx2; i += 2) {
            if (end_code + i + 2 > f->size) return 0;
            if ((uint32_t)codepoint <= rd16(f, end_code + i)) {
                const uint32_t start = rd16(f, start_code + i);
                if ((uint32_t)codepoint < start) return 0;
                const uint32_t range = rd16(f, id_range + i);
                if (range == 0) return (int)(((uint32_t)codepoint + rd16(f, id_delta + i)) & 0xffffu);
                const uint32_t glyph = rd16(f, id_range + i + range + 2 * (size_t)((uint32_t)codepoint - start));
                return glyph ? (int)((glyph + rd16(f, id_delta + i)) & 0xffffu) : 0;
            }
        }
        return 0;
    }
    const uint32_t n_groups = rd32(f, t + 12);
    for (uint32_t i = 0; i < n_groups; i++) {
        const size_t g = t + 16 + 12 * (size_t)i;
        if (g + 12 > f->size) return 0;
        const uint32_t start = rd32(f, g), end = rd32(f, g + 4);
        if ((uint32_t)codepoint >= start && (uint32_t)codepoint <= end) {
            return (int)((rd32(f, g + 8) + ((uint32_t)codepoint - start)) & 0xffffu);
        }
    }
    return 0;
}

float ntt_font_scale_for_pixel_height(const ntt_font *f, float pixels) {
    return pixels / (float)(f->ascent - f->descent);
}

void ntt_font_vmetrics(const ntt_font *f, int *ascent, int *descent, int *line_gap) {
    if (ascent) *ascent = f->ascent;
    if (descent) *descent = f->descent;
    if (line_gap) *line_gap = f->line_gap;
}

void ntt_font_hmetrics(const ntt_font *f, int glyph, int *advance, int *lsb) {
    const size_t h = f->hmtx;
    if (glyph < 0) glyph = 0;
    if (glyph < f->num_hmetrics) {
        if (advance) *advance = (int)rd16(f, h + 4 * (size_t)glyph);
        if (lsb) *lsb = rds16(f, h + 4 * (size_t)glyph + 2);
    } else {
        if (advance) *advance = (int)rd16(f, h + 4 * (size_t)(f->num_hmetrics - 1));
        if (lsb) *lsb = rds16(f, h + 4 * (size_t)f->num_hmetrics + 2 * (size_t)(glyph - f->num_hmetrics));
    }
}

int ntt_font_kern_advance(const ntt_font *f, int glyph1, int glyph2) {
    if (!f->kern) return 0;
    const size_t k = f->kern;
    if (rd16(f, k + 2) < 1 || rd16(f, k + 8) != 1) return 0; /* first subtable: horizontal, format 0 */
    int lo = 0, hi = (int)rd16(f, k + 10) - 1;
    const uint32_t needle = ((uint32_t)glyph1 << 16) | ((uint32_t)glyph2 & 0xffffu);
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t key = rd32(f, k + 18 + 6 * (size_t)mid);
        if (needle < key) hi = mid - 1;
        else if (needle > key) lo = mid + 1;
        else return rds16(f, k + 22 + 6 * (size_t)mid);
    }
    return 0;
}

/* file offset of the glyph's data and one past its end; 0 for an empty glyph (space) or a bad index */
static size_t glyph_data(const ntt_font *f, int glyph, size_t *end) {
    if (glyph < 0 || glyph >= f->num_glyphs) return 0;
    uint32_t g0, g1;
    if (f->loc_format == 0) {
        g0 = 2u * rd16(f, (size_t)f->loca + 2 * (size_t)glyph);
        g1 = 2u * rd16(f, (size_t)f->loca + 2 * (size_t)glyph + 2);
    } else {
        g0 = rd32(f, (size_t)f->loca + 4 * (size_t)glyph);
        g1 = rd32(f, (size_t)f->loca + 4 * (size_t)glyph + 4);
    }
    if (g1 <= g0 || g1 - g0 < 10 || (size_t)f->glyf + g1 > f->size) return 0;
    if (end) *end = (size_t)f->glyf + g1;
    return (size_t)f->glyf + g0;
}

int ntt_font_glyph_box(const ntt_font *f, int glyph, int *x0, int *y0, int *x1, int *y1) {
    const size_t g = glyph_data(f, glyph, NULL);
    if (!g) return 0;
    *x0 = rds16(f, g + 2);
    *y0 = rds16(f, g + 4);
    *x1 = rds16(f, g + 6);
    *y1 = rds16(f, g + 8);
    return 1;
}

void ntt_font_bitmap_box(const ntt_font *f, int glyph, float scale, int *ix0, int *iy0, int *ix1, int *iy1) {
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (!ntt_font_glyph_box(f, glyph, &x0, &y0, &x1, &y1)) {
        *ix0 = *iy0 = *ix1 = *iy1 = 0;
        return;
    }
    *ix0 = (int)floorf((float)x0 * scale);
    *iy0 = (int)floorf((float)-y1 * scale);
    *ix1 = (int)ceilf((float)x1 * scale);
    *iy1 = (int)ceilf((float)-y0 * scale);
}

/* ---- rasteriser: signed area of the outline inside every pixel, accumulated per row ---- */
typedef struct {
    float *acc; /* h rows of (w + 2) */
    int w, h;
} raster;

static void raster_line(raster *r, float x0, float y0, float x1, float y1) {
    if (y0 == y1) return;
    float dir = 1.0f;
    if (y0 > y1) {
        const float tx = x0, ty = y0;
        x0 = x1; y0 = y1; x1 = tx; y1 = ty;
        dir = -1.0f;
    }
    if (!(y1 > 0.0f) || !(y0 < (float)r->h)) return; /* above or below the bitmap (or not a number) */
    const float dxdy = (x1 - x0) / (y1 - y0);
    float x = x0;
    if (y0 < 0.0f) x -= y0 * dxdy;
    const int y_begin = y0 < 0.0f ? 0 : (int)floorf(y0);
    const int y_end = y1 > (float)r->h ? r->h : (int)ceilf(y1);
    for (int y = y_begin; y < y_end; y++) {
        float *row = r->acc + (size_t)y * (size_t)(r->w + 2);
        const float top = y0 > (float)y ? y0 : (float)y, bot = y1 < (float)(y + 1) ? y1 : (float)(y + 1);
        const float dy = bot - top, xnext = x + dxdy * dy, d = dy * dir;
        float xa = x < xnext ? x : xnext, xb = x < xnext ? xnext : x;
        /* outlines of a damaged file may leave the glyph's declared box: left of it a segment still covers every
         * pixel of its rows, right of it none (column w and w + 1 are scratch) */
        if (xa < 0.0f) xa = 0.0f;
        if (xa > (float)r->w) xa = (float)r->w;
        if (xb > (float)r->w) xb = (float)r->w;
        if (xb < xa) xb = xa;
        const float xa_floor = floorf(xa), xb_ceil = ceilf(xb);
        const int ia = (int)xa_floor, ib = (int)xb_ceil;
        if (ib <= ia + 1) {
            const float xmf = 0.5f * (xa + xb) - xa_floor; /* mean x inside the pixel */
            row[ia] += d - d * xmf;
            row[ia + 1] += d * xmf;
        } else {
            const float s = 1.0f / (xb - xa);
            const float fa = xa - xa_floor, a0 = 0.5f * s * (1.0f - fa) * (1.0f - fa);
            const float fb = xb - xb_ceil + 1.0f, am = 0.5f * s * fb * fb;
            row[ia] += d * a0;
            if (ib == ia + 2) {
                row[ia + 1] += d * (1.0f - a0 - am);
            } else {
                const float a1 = s * (1.5f - fa);
                row[ia + 1] += d * (a1 - a0);
                for (int xi = ia + 2; xi < ib - 1; xi++) row[xi] += d * s;
                const float a2 = a1 + (float)(ib - ia - 3) * s;
                row[ib - 1] += d * (1.0f - a2 - am);
            }
            row[ib] += d * am;
        }
        x = xnext;
    }
}

static void raster_quad(raster *r, float x0, float y0, float cx, float cy, float x1, float y1) {
    /* flatten to within 0.35 px (stb_truetype's flatness): the deviation of a quadratic from its chord
     * is at most |p0 - 2c + p1| / 4, and it shrinks by 4 per halving */
    const float ddx = x0 - 2.0f * cx + x1, ddy = y0 - 2.0f * cy + y1;
    const float dev = 0.25f * sqrtf(ddx * ddx + ddy * ddy);
    int n = 1;
    while ((dev / (float)(n * n)) > 0.35f && n < 64) n *= 2;
    float px = x0, py = y0;
    for (int i = 1; i <= n; i++) {
        const float t = (float)i / (float)n, mt = 1.0f - t;
        const float qx = mt * mt * x0 + 2.0f * mt * t * cx + t * t * x1;
        const float qy = mt * mt * y0 + 2.0f * mt * t * cy + t * t * y1;
        raster_line(r, px, py, qx, qy);
        px = qx;
        py = qy;
    }
}

uint8_t *ntt_font_glyph_bitmap(const ntt_font *f, int glyph, float scale, int *width, int *height, int *xoff, int *yoff) {
    int ix0, iy0, ix1, iy1;
    ntt_font_bitmap_box(f, glyph, scale, &ix0, &iy0, &ix1, &iy1);
    *width = ix1 - ix0;
    *height = iy1 - iy0;
    *xoff = ix0;
    *yoff = iy0;
    size_t g_end = 0;
    const size_t g = glyph_data(f, glyph, &g_end);
    /* a bitmap of more than 16 Mpixel is not a tick label (damaged box or an absurd size) */
    if (!g || *width <= 0 || *height <= 0 || (int64_t)*width * (int64_t)*height > (int64_t)(1 << 24)) {
        *width = *height = 0;
        return NULL;
    }
    const int n_contours = rds16(f, g);
    if (n_contours <= 0) { /* composite glyph: not needed for the labels */
        *width = *height = 0;
        return NULL;
    }
    const size_t end_pts = g + 10;
    const int n_points = (int)rd16(f, end_pts + 2 * (size_t)(n_contours - 1)) + 1;
    const size_t n_instr = rd16(f, end_pts + 2 * (size_t)n_contours);
    size_t p = end_pts + 2 * (size_t)n_contours + 2 + n_instr;
    if (p > g_end) { /* the header runs past the glyph's own data */
        *width = *height = 0;
        return NULL;
    }
    uint8_t *flags = (uint8_t *)malloc((size_t)n_points);
    float *px = (float *)malloc(sizeof(float) * (size_t)n_points), *py = (float *)malloc(sizeof(float) * (size_t)n_points);
    raster r;
    r.w = *width;
    r.h = *height;
    r.acc = (float *)calloc((size_t)(r.w + 2) * (size_t)r.h, sizeof(float));
    uint8_t *out = (uint8_t *)malloc((size_t)r.w * (size_t)r.h);
    if (!flags || !px || !py || !r.acc || !out) {
        free(flags); free(px); free(py); free(r.acc); free(out);
        *width = *height = 0;
        return NULL;
    }
    for (int i = 0; i < n_points;) { /* flags, run-length coded */
        const uint8_t fl = (uint8_t)rd8(f, p++);
        flags[i++] = fl;
        if (fl & 8) {
            int rep = (int)rd8(f, p++);
            while (rep-- > 0 && i < n_points) flags[i++] = fl;
        }
    }
    long long v = 0; /* 65536 points of +-32767 do not fit an int */
    for (int i = 0; i < n_points; i++) { /* x deltas */
        if (flags[i] & 2) {
            const int dx = (int)rd8(f, p++);
            v += (flags[i] & 16) ? dx : -dx;
        } else if (!(flags[i] & 16)) {
            v += rds16(f, p);
            p += 2;
        }
        px[i] = (float)v * scale - (float)ix0;
    }
    v = 0;
    for (int i = 0; i < n_points; i++) { /* y deltas; bitmap y grows downwards */
        if (flags[i] & 4) {
            const int dy = (int)rd8(f, p++);
            v += (flags[i] & 32) ? dy : -dy;
        } else if (!(flags[i] & 32)) {
            v += rds16(f, p);
            p += 2;
        }
        py[i] = (float)-v * scale - (float)iy0;
    }
    /* per contour: insert the implied on-curve midpoint between consecutive control points, start at an
     * on-curve point, then every step is a line (on -> on) or a quadratic (on -> control -> on) */
    float *ex = (float *)malloc(sizeof(float) * 2 * (size_t)n_points + 2), *ey = (float *)malloc(sizeof(float) * 2 * (size_t)n_points + 2);
    uint8_t *eon = (uint8_t *)malloc(2 * (size_t)n_points + 2);
    if (!ex || !ey || !eon) {
        free(flags); free(px); free(py); free(r.acc); free(out); free(ex); free(ey); free(eon);
        *width = *height = 0;
        return NULL;
    }
    int first = 0;
    for (int c = 0; c < n_contours; c++) {
        const int last = (int)rd16(f, end_pts + 2 * (size_t)c), n = last - first + 1;
        if (n >= 2 && first >= 0 && last < n_points) {
            int m = 0, s = -1;
            for (int i = 0; i < n; i++) {
                const int a = first + i, b = first + (i + 1) % n;
                ex[m] = px[a]; ey[m] = py[a]; eon[m] = flags[a] & 1;
                if (eon[m] && s < 0) s = m;
                m++;
                if (!(flags[a] & 1) && !(flags[b] & 1)) {
                    ex[m] = 0.5f * (px[a] + px[b]); ey[m] = 0.5f * (py[a] + py[b]); eon[m] = 1;
                    if (s < 0) s = m;
                    m++;
                }
            }
            if (s >= 0) {
                float curx = ex[s], cury = ey[s], cx = 0.0f, cy = 0.0f;
                int have_ctrl = 0;
                for (int k = 1; k <= m; k++) {
                    const int idx = (s + k) % m;
                    if (eon[idx]) {
                        if (have_ctrl) raster_quad(&r, curx, cury, cx, cy, ex[idx], ey[idx]);
                        else raster_line(&r, curx, cury, ex[idx], ey[idx]);
                        curx = ex[idx]; cury = ey[idx];
                        have_ctrl = 0;
                    } else {
                        cx = ex[idx]; cy = ey[idx];
                        have_ctrl = 1;
                    }
                }
            }
        }
        if (last + 1 > first) first = last + 1; /* end points must ascend; a contour that does not is skipped */
    }
    free(ex); free(ey); free(eon);
    for (int y = 0; y < r.h; y++) {
        float acc = 0.0f;
        const float *row = r.acc + (size_t)y * (size_t)(r.w + 2);
        for (int x = 0; x < r.w; x++) {
            acc += row[x];
            float a = fabsf(acc);
            if (a > 1.0f) a = 1.0f;
            out[(size_t)y * (size_t)r.w + (size_t)x] = (uint8_t)(a * 255.0f + 0.5f);
        }
    }
    free(flags); free(px); free(py); free(r.acc);
    return out;
}

/* ---- the reference's three helpers ---- */
void ntt_font_measure(const ntt_font *font, const char *text, const int x, const int y, const int font_size, int *width,
                      int *height) {
    const float font_scale = ntt_font_scale_for_pixel_height(font, (float)font_size);
    const int baseline = (int)((float)font->ascent * font_scale);
    const int descent_scaled = (int)((float)font->descent * font_scale);
    *width = x;
    *height = y + (baseline - descent_scaled); /* constant for the font; descent is negative */
    for (int ch = 0; text[ch]; ch++) {
        const int g = ntt_font_glyph_index(font, (unsigned char)text[ch]);
        int advance = 0, lsb = 0;
        ntt_font_hmetrics(font, g, &advance, &lsb);
        *width += (int)((float)advance * font_scale);
        if (text[ch + 1]) {
            /* `*width += font_scale * kern` on an int (c/fft-stitch.c:118): the SUM is truncated, not the addend */
            *width = (int)((float)*width + font_scale * (float)ntt_font_kern_advance(font, g, ntt_font_glyph_index(font, (unsigned char)text[ch + 1])));
        }
    }
}

void ntt_font_draw(const ntt_font *font, uint8_t *img, const uint32_t img_stride, const uint32_t img_height, const char *text,
                   const int x, const int y, const int font_size) {
    int text_width, text_height;
    ntt_font_measure(font, text, 0, 0, font_size, &text_width, &text_height);
    const int start_x = x - text_width / 2;
    if (start_x < 0) return; /* c/fft-stitch.c:130 */
    const float font_scale = ntt_font_scale_for_pixel_height(font, (float)font_size);
    const int baseline = (int)((float)font->ascent * font_scale);
    int pen = 0;
    for (int ch = 0; text[ch]; ch++) {
        const int g = ntt_font_glyph_index(font, (unsigned char)text[ch]);
        int advance = 0, lsb = 0, w = 0, h = 0, dx = 0, dy = 0;
        ntt_font_hmetrics(font, g, &advance, &lsb);
        uint8_t *bitmap = ntt_font_glyph_bitmap(font, g, font_scale, &w, &h, &dx, &dy);
        if (bitmap) { /* img_gray_copy: max-composite (c/fft-stitch.c:46-54), clipped to the image here */
            for (int j = 0; j < h; j++) {
                const long py = (long)y + baseline + dy + j;
                if (py < 0 || py >= (long)img_height) continue;
                for (int i = 0; i < w; i++) {
                    const long px = (long)start_x + pen + dx + i;
                    if (px < 0 || px >= (long)img_stride) continue;
                    uint8_t *d = img + (size_t)py * img_stride + (size_t)px;
                    const uint8_t s = bitmap[(size_t)j * (size_t)w + (size_t)i];
                    if (s > *d) *d = s;
                }
            }
            free(bitmap);
        }
        pen += (int)((float)advance * font_scale);
        if (text[ch + 1]) {
            /* `_x += font_scale * kern` on an int (c/fft-stitch.c:152): the sum is truncated */
            pen = (int)((float)pen + font_scale * (float)ntt_font_kern_advance(font, g, ntt_font_glyph_index(font, (unsigned char)text[ch + 1])));
        }
    }
}
