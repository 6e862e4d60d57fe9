/*
 * easypng.c -- 8-bit gray PNG writer / reader on zlib (include/easypng.h).
 * Writer: IHDR (8-bit, colour type 0, no interlace), one IDAT of the zlib-compressed
 * scanlines (filter type 0), IEND; what c/easypng.h:6-53 asks libpng for.  Large images are
 * deflated in up to sixteen slabs of rows on as many threads (one zlib stream, see deflate_slab).
 * Reader: all five scanline filters, colour types 0/2/4/6 at 8 bits, converted to gray with
 * stb_image's integer luma (77 r + 150 g + 29 b) >> 8, as c/fft-stitch.c:172 requests.
 */
#include "easypng.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static void put_u32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

static uint32_t get_u32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

static int write_chunk(FILE *fp, const char *type, const uint8_t *data, uint32_t len) {
    uint8_t head[8], tail[4];
    put_u32(head, len);
    memcpy(head + 4, type, 4);
    uLong crc = crc32(0L, head + 4, 4);
    if (len) crc = crc32(crc, data, len);
    put_u32(tail, (uint32_t)crc);
    if (fwrite(head, 1, 8, fp) != 8) return -1;
    if (len && fwrite(data, 1, len, fp) != len) return -1;
    if (fwrite(tail, 1, 4, fp) != 4) return -1;
    return 0;
}

/* One slab of scanlines, compressed on its own thread as a raw deflate stream that ends on a byte boundary
 * (Z_FULL_FLUSH; the last slab finishes the stream), so that the slabs' outputs concatenate into ONE valid deflate stream
 * -- the way pigz writes.  A 1024 x 16384 tile (c/fft-batch.c's geometry) took 0.36 s to deflate on one core, a hundred
 * times the GPU's share of that capture; sixteen slabs take it to a few tens of milliseconds (eight until round 6: at the
 * reference's own sweep geometry the encode was the stage the other two waited for, profiles/r06_reference_geometry_narrow.json). */
typedef struct {
    const uint8_t *pixels; /* first row of the slab */
    int width, rows, last;
    uint8_t *out;
    size_t out_len, out_cap;
    uLong adler; /* of the slab's raw bytes (filter bytes included) */
    uLong out_crc; /* crc32 of the slab's compressed bytes, summed on the slab's own thread: the IDAT chunk's CRC is combined
                      from these instead of one serial pass over the whole stream (a third of a 16 MiB tile's encode time) */
    size_t raw_len;
    int ok;
} png_slab;

static void *deflate_slab(void *p) {
    png_slab *sl = (png_slab *)p;
    const size_t stride = (size_t)sl->width + 1;
    sl->raw_len = stride * (size_t)sl->rows;
    sl->ok = 0;
    if (sl->raw_len > 0xf0000000u) return NULL; /* zlib's avail_in / avail_out are 32 bits: write_gray_png sizes the slabs below that */
    uint8_t *raw = (uint8_t *)malloc(sl->raw_len);
    sl->out_cap = compressBound((uLong)sl->raw_len) + 64;
    sl->out = (uint8_t *)malloc(sl->out_cap);
    if (!raw || !sl->out) {
        free(raw);
        return NULL;
    }
    for (int y = 0; y < sl->rows; y++) {
        raw[(size_t)y * stride] = 0; /* filter type: none */
        memcpy(raw + (size_t)y * stride + 1, sl->pixels + (size_t)y * (size_t)sl->width, (size_t)sl->width);
    }
    sl->adler = adler32(adler32(0L, Z_NULL, 0), raw, (uInt)sl->raw_len);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    /* Z_RLE: spectrogram pixels are either noise-like (a synthetic or a busy capture: no LZ77 match worth coding) or long runs
     * of one value (a recorded capture's empty band, the stitched image's footer); on both the run-length strategy gives a
     * SMALLER stream than the default one at four times its speed (a 1024 x 4096 noise tile: 0.891 of the raw bytes at
     * 90 MB/s against 0.897 at 24 MB/s; a recorded capture's tile: 0.312 against 0.337) -- and at the reference's sweep
     * geometry the encode is the stage the other two wait for */
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_RLE) == Z_OK) {
        zs.next_in = raw;
        zs.avail_in = (uInt)sl->raw_len;
        zs.next_out = sl->out;
        zs.avail_out = (uInt)sl->out_cap;
        const int rc = deflate(&zs, sl->last ? Z_FINISH : Z_FULL_FLUSH);
        if ((sl->last ? rc == Z_STREAM_END : rc == Z_OK) && zs.avail_in == 0) {
            sl->out_len = sl->out_cap - zs.avail_out;
            sl->out_crc = crc32(0L, Z_NULL, 0);
            for (size_t off = 0; off < sl->out_len;) {
                const size_t piece = sl->out_len - off < ((size_t)1 << 30) ? sl->out_len - off : ((size_t)1 << 30);
                sl->out_crc = crc32(sl->out_crc, sl->out + off, (uInt)piece);
                off += piece;
            }
            sl->ok = 1;
        }
        deflateEnd(&zs);
    }
    free(raw);
    return NULL;
}

#include <pthread.h>
#include <unistd.h>

/* IDAT chunks written as a stream: `left` bytes of zlib data are still to come in all; a chunk is opened with
 * min(left, PNG_IDAT_MAX) bytes, filled by idat_put from however many pieces, and closed with its CRC when full. */
#define PNG_IDAT_MAX ((size_t)1 << 30)
typedef struct {
    FILE *fp;
    size_t chunk_max; /* bytes per IDAT chunk at most */
    size_t left;     /* zlib bytes not yet handed to idat_put */
    size_t in_chunk; /* bytes the open chunk still takes; 0 = no chunk open */
    uLong crc;
} idat_stream;

/* `known_crc`: crc32 of data[0..len) computed elsewhere (a slab's own thread), or NULL.  Used when the whole piece falls into
 * the open chunk (crc32_combine is O(log len)); a piece that a chunk boundary cuts is summed here byte by byte. */
static int idat_put_crc(idat_stream *st, const uint8_t *data, size_t len, const uLong *known_crc) {
    while (len > 0) {
        if (st->in_chunk == 0) {
            if (st->left == 0) return -1;
            const size_t take = st->left < st->chunk_max ? st->left : st->chunk_max;
            uint8_t head[8];
            put_u32(head, (uint32_t)take);
            memcpy(head + 4, "IDAT", 4);
            if (fwrite(head, 1, 8, st->fp) != 8) return -1;
            st->crc = crc32(0L, head + 4, 4);
            st->in_chunk = take;
        }
        size_t n = len < st->in_chunk ? len : st->in_chunk;
        if (n > st->left) return -1;
        if (known_crc != NULL && n == len && len <= 0x7fffffffu) {
            st->crc = crc32_combine(st->crc, *known_crc, (z_off_t)len);
            known_crc = NULL;
        } else {
            known_crc = NULL; /* the piece is cut by a chunk boundary: its own sum no longer applies */
            for (size_t off = 0; off < n;) { /* crc32 takes a uInt length */
                const size_t piece = n - off < ((size_t)1 << 30) ? n - off : ((size_t)1 << 30);
                st->crc = crc32(st->crc, data + off, (uInt)piece);
                off += piece;
            }
        }
        if (fwrite(data, 1, n, st->fp) != n) return -1;
        data += n;
        len -= n;
        st->in_chunk -= n;
        st->left -= n;
        if (st->in_chunk == 0) {
            uint8_t tail[4];
            put_u32(tail, (uint32_t)st->crc);
            if (fwrite(tail, 1, 4, st->fp) != 4) return -1;
        }
    }
    return 0;
}

static int idat_put(idat_stream *st, const uint8_t *data, size_t len) { return idat_put_crc(st, data, len, NULL); }

#define PNG_MAX_SLABS 16
#define PNG_SLAB_MIN_BYTES ((size_t)1 << 16) /* bytes per slab at least: smaller images are one slab, compressed on the calling thread */

int write_gray_png(const char *fname, int width, int height, const uint8_t *buffer) {
    return write_gray_png_chunked(fname, width, height, buffer, PNG_IDAT_MAX);
}

int write_gray_png_chunked(const char *fname, int width, int height, const uint8_t *buffer, size_t idat_max) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (width <= 0 || height <= 0 || buffer == NULL) {
        printf("ERROR: invalid image %d x %d.\n", width, height);
        return -1;
    }
    FILE *fp = fopen(fname, "wb");
    if (!fp) {
        printf("ERROR: Could not write open file %s for writing.\n", fname);
        return -1;
    }
    const size_t stride = (size_t)width + 1, total = stride * (size_t)height;
    long cpus = sysconf(_SC_NPROCESSORS_ONLN);
    int n_slabs = (int)(total / PNG_SLAB_MIN_BYTES);
    if (n_slabs > PNG_MAX_SLABS) n_slabs = PNG_MAX_SLABS;
    if (cpus > 0 && n_slabs > cpus) n_slabs = (int)cpus;
    if (n_slabs > height) n_slabs = height;
    if (n_slabs < 1) n_slabs = 1;
    png_slab slabs[PNG_MAX_SLABS];
    pthread_t threads[PNG_MAX_SLABS];
    int started[PNG_MAX_SLABS];
    const int per = (height + n_slabs - 1) / n_slabs;
    n_slabs = (height + per - 1) / per;
    for (int k = 0; k < n_slabs; k++) {
        memset(&slabs[k], 0, sizeof(slabs[k]));
        slabs[k].pixels = buffer + (size_t)k * (size_t)per * (size_t)width;
        slabs[k].width = width;
        slabs[k].rows = (k + 1) * per <= height ? per : height - k * per;
        slabs[k].last = (k == n_slabs - 1);
        started[k] = (k > 0) && pthread_create(&threads[k], NULL, deflate_slab, &slabs[k]) == 0;
    }
    deflate_slab(&slabs[0]);
    for (int k = 1; k < n_slabs; k++) {
        if (started[k]) pthread_join(threads[k], NULL);
        else deflate_slab(&slabs[k]); /* no thread to be had: compress it here */
    }
    int rc = -1;
    int all_ok = 1;
    for (int k = 0; k < n_slabs; k++) all_ok = all_ok && slabs[k].ok;
    if (all_ok) {
        /* the zlib stream = 2-byte header, the slabs' deflate output, Adler-32 of the raw bytes; written as IDAT chunks of at
         * most PNG_IDAT_MAX bytes each (a chunk length is 31 bits; libpng, which the reference links, splits likewise).  The
         * reference's own stitched image is 154112 x 11811 (c/fft-stitch.c:16-27): 1.8 GB of scanlines in one IDAT would sit
         * just under that limit, the same sweep at full tile height above it. */
        uint8_t zhead[2] = {0x78, 0x9c}; /* deflate, 32 KiB window; default compression, no dictionary (0x789c % 31 == 0) */
        uint8_t ztail[4];
        uLong adler = adler32(0L, Z_NULL, 0);
        for (int k = 0; k < n_slabs; k++) {
            adler = (k == 0) ? slabs[k].adler : adler32_combine(adler, slabs[k].adler, (z_off_t)slabs[k].raw_len);
        }
        put_u32(ztail, (uint32_t)adler);
        uint8_t ihdr[13];
        put_u32(ihdr, (uint32_t)width);
        put_u32(ihdr + 4, (uint32_t)height);
        ihdr[8] = 8;  /* bit depth */
        ihdr[9] = 0;  /* gray */
        ihdr[10] = 0; /* deflate */
        ihdr[11] = 0; /* adaptive filtering */
        ihdr[12] = 0; /* no interlace */
        idat_stream st;
        memset(&st, 0, sizeof(st));
        st.fp = fp;
        st.chunk_max = idat_max < 1 ? 1 : idat_max > PNG_IDAT_MAX ? PNG_IDAT_MAX : idat_max;
        st.left = 2 + 4;
        for (int k = 0; k < n_slabs; k++) st.left += slabs[k].out_len;
        int ok = fwrite(sig, 1, 8, fp) == 8 && write_chunk(fp, "IHDR", ihdr, 13) == 0 && idat_put(&st, zhead, 2) == 0;
        for (int k = 0; ok && k < n_slabs; k++) ok = idat_put_crc(&st, slabs[k].out, slabs[k].out_len, &slabs[k].out_crc) == 0;
        ok = ok && idat_put(&st, ztail, 4) == 0 && st.left == 0 && st.in_chunk == 0;
        if (ok && write_chunk(fp, "IEND", NULL, 0) == 0) rc = 0;
    }
    for (int k = 0; k < n_slabs; k++) free(slabs[k].out);
    if (fclose(fp) != 0) rc = -1;
    if (rc == 0) {
        printf("Written %s.\n", fname);
    } else {
        printf("ERROR: writing %s failed.\n", fname);
    }
    return rc;
}

static int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

uint8_t *read_gray_png(const char *fname, int *width, int *height) {
    FILE *fp = fopen(fname, "rb");
    if (!fp) return NULL;
    fseek(fp, 0L, SEEK_END);
    long size = ftell(fp);
    rewind(fp);
    uint8_t *file = size > 8 ? (uint8_t *)malloc((size_t)size) : NULL;
    if (!file || fread(file, 1, (size_t)size, fp) != (size_t)size) {
        free(file);
        fclose(fp);
        return NULL;
    }
    fclose(fp);
    uint8_t *idat = (uint8_t *)malloc((size_t)size);
    size_t idat_len = 0;
    uint32_t w = 0, h = 0;
    int channels = 0, ok = idat != NULL && memcmp(file, "\x89PNG\r\n\x1a\n", 8) == 0;
    for (size_t pos = 8; ok && pos + 12 <= (size_t)size;) {
        uint32_t len = get_u32(file + pos);
        const uint8_t *type = file + pos + 4, *data = file + pos + 8;
        if (pos + 12 + (size_t)len > (size_t)size) { ok = 0; break; }
        if (memcmp(type, "IHDR", 4) == 0 && len == 13) {
            w = get_u32(data);
            h = get_u32(data + 4);
            int ct = data[9];
            channels = ct == 0 ? 1 : ct == 4 ? 2 : ct == 2 ? 3 : ct == 6 ? 4 : 0;
            if (data[8] != 8 || channels == 0 || data[12] != 0 || w == 0 || h == 0) ok = 0;
            /* sizes the sweep never produces (and whose byte counts would not fit the arithmetic below) are refused:
             * the widest stitched image is 2 097 152 x 256, the tallest tile 1024 x 16384 */
            if (w > (1u << 24) || h > (1u << 24) || (uint64_t)w * (uint64_t)h > ((uint64_t)1 << 33)) ok = 0;
        } else if (memcmp(type, "IDAT", 4) == 0) {
            memcpy(idat + idat_len, data, len);
            idat_len += len;
        } else if (memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    uint8_t *out = NULL;
    if (ok && channels && idat_len) {
        const size_t bpp = (size_t)channels, stride = (size_t)w * bpp;
        uLongf raw_len = (uLongf)((stride + 1) * (size_t)h);
        uint8_t *raw = (uint8_t *)malloc(raw_len);
        out = (uint8_t *)malloc((size_t)w * (size_t)h);
        if (raw && out && uncompress(raw, &raw_len, idat, (uLong)idat_len) == Z_OK &&
            raw_len == (stride + 1) * (size_t)h) {
            uint8_t *prev = (uint8_t *)calloc(stride, 1);
            for (uint32_t y = 0; y < h && prev; y++) {
                uint8_t *line = raw + (size_t)y * (stride + 1);
                const int ft = line[0];
                uint8_t *cur = line + 1;
                for (size_t i = 0; i < stride; i++) {
                    int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                    int add = ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) / 2 : ft == 4 ? paeth(a, b, c) : 0;
                    cur[i] = (uint8_t)(cur[i] + add);
                }
                for (uint32_t x = 0; x < w; x++) {
                    const uint8_t *px = cur + (size_t)x * bpp;
                    out[(size_t)y * w + x] =
                        channels <= 2 ? px[0] : (uint8_t)((px[0] * 77 + px[1] * 150 + px[2] * 29) >> 8);
                }
                memcpy(prev, cur, stride);
            }
            free(prev);
            *width = (int)w;
            *height = (int)h;
        } else {
            free(out);
            out = NULL;
        }
        free(raw);
    }
    free(idat);
    free(file);
    return out;
}
