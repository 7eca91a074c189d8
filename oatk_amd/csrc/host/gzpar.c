/*
 * oatk_amd/csrc/host/gzpar.c -- ONE gzip member inflated on many threads (round 5), for host/gzsrc.c.
 *
 * BASELINE.json's configs[0] is a plain `.fa.gz`: one member, which zlib inflates on one thread at ~0.4 GB/s of text -- 3.5 s of a 5.8 s run of the drop-in CLI on the
 * config-1 surrogate's 100 k reads, on a host with 128 cores (the reference waits for the same thread: sstream.c:39-54 reads through gzread).  A deflate stream has no
 * index, but it can be entered at any BLOCK boundary by a decoder that does not know the 32 KiB of text before it (the construction of pugz / rapidgzip):
 *   1. the compressed bytes are cut into chunks; for every chunk but the first a thread looks for a block header at or after the chunk's first bit -- a dynamic-Huffman
 *      header whose code lengths form complete codes, whose block decodes to its end-of-block symbol with nothing but text in its literals and no distance beyond the
 *      window, and which is followed by another valid header;
 *   2. every chunk is decoded from its boundary into 16-bit SYMBOLS: a byte, or "whatever stood at position w of the window before this chunk" -- a back-reference into
 *      the unknown copies such symbols like any other.  A chunk is decoded up to the first block boundary at or after the next chunk's cut, which is where the next chunk
 *      looks for ITS boundary.  The chain starts from a known state (the member's first bit) and moves block by block, so a claim it arrives at EXACTLY is a true
 *      boundary and the symbols behind it are the true text; where it stands before a claim (a flush marker or another block the search does not accept lies between, or
 *      a chunk found nothing) its last chunk goes on, in order, until it stands there; a claim it passes was false;
 *   3. in order, each chunk's last 32 KiB are turned into bytes with the window handed down from the chunk before (that is all the serial work there is), and then all
 *      chunks are turned into bytes at their places in the caller's buffer, in parallel, with a CRC each (combined: crc32_combine).
 * The member's CRC-32 and length are checked at its end as gzread checks them.  Text that does not look like text (no boundary found in the first chunks) is left to zlib.
 * Nothing of this changes a byte of what is delivered: tests/test_host_gzsrc.py compares with zlib on files of every kind.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

#include "oatk_hip.h"
#include "host_internal.h"

#define GP_WIN 32768
#define GP_FAST 11
#define GP_MAX_THREADS 64
#define GP_INF (~0ULL)

typedef struct { const uint8_t *base, *p, *end; uint64_t bits; int nb; int over; } br_t;

static inline void br_init(br_t *b, const uint8_t *base, uint64_t n, uint64_t bitpos)
{
    b->base = base, b->end = base + n, b->p = base + (bitpos >> 3), b->bits = 0, b->nb = 0, b->over = 0;
    const int skip = (int) (bitpos & 7);
    if (skip) {
        if (b->p < b->end) { b->bits = (uint64_t) *b->p++ >> skip, b->nb = 8 - skip; }
        else b->over = 1;
    }
}
static inline void br_refill(br_t *b)
{
    if (b->p + 8 <= b->end) {                    /* eight bytes at once: what does not fit stays where it is and is read again */
        uint64_t w;
        memcpy(&w, b->p, 8);
        b->bits |= w << b->nb;
        const int adv = (63 - b->nb) >> 3;
        b->p += adv, b->nb += adv << 3;
        return;
    }
    while (b->nb <= 56) {
        if (b->p < b->end) b->bits |= (uint64_t) *b->p++ << b->nb;
        else b->over++;                          /* (zeros; taking them is an error the caller sees through br_pos) */
        b->nb += 8;
    }
}
static inline uint32_t br_peek(const br_t *b, int n) { return (uint32_t) (b->bits & ((1ULL << n) - 1)); }
static inline void br_drop(br_t *b, int n) { b->bits >>= n, b->nb -= n; }
static inline uint32_t br_get(br_t *b, int n) { if (b->nb < n) br_refill(b); const uint32_t v = br_peek(b, n); br_drop(b, n); return v; }
/* the position of the next unread bit; past the end of the input when zeros were taken */
static inline uint64_t br_pos(const br_t *b) { return (uint64_t) (b->p - b->base) * 8 + (uint64_t) b->over * 8 - (uint64_t) b->nb; }

/* A code's table: 2^11 entries indexed by the next bits as they come.  An entry says everything the decoder needs without a second look-up:
 *   bits 0-3 the code's length (0: longer than eleven bits, or no such code -- the canonical walk decides), bits 4-7 the number of extra bits that follow,
 *   bit 8 literal, bit 9 end of block, bit 10 a symbol that must not occur (286, 287; distances 30, 31), bits 16-31 the literal / the length's or distance's base */
#define E_LIT (1u << 8)
#define E_EOB (1u << 9)
#define E_BAD (1u << 10)
enum { K_PLAIN = 0, K_LITLEN = 1, K_DIST = 2 };
typedef struct {
    uint32_t fast[1 << GP_FAST];
    uint16_t count[16], sym[288];                /* canonical decoding for the long codes (puff.c's way) */
    int max_len, kind;
} huff_t;

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

static inline uint32_t entry_of(int kind, int sym, int l)
{
    if (kind == K_PLAIN) return (uint32_t) sym << 16 | (uint32_t) l;
    if (kind == K_LITLEN) {
        if (sym < 256) return (uint32_t) sym << 16 | E_LIT | (uint32_t) l;
        if (sym == 256) return E_EOB | (uint32_t) l;
        if (sym >= 286) return E_BAD | (uint32_t) l;
        return (uint32_t) LBASE[sym - 257] << 16 | (uint32_t) LEXT[sym - 257] << 4 | (uint32_t) l;
    }
    if (sym >= 30) return E_BAD | (uint32_t) l;
    return (uint32_t) DBASE[sym] << 16 | (uint32_t) DEXT[sym] << 4 | (uint32_t) l;
}

/* 0: a complete code; 1: incomplete (allowed for a single distance code); -1: over-subscribed or empty where that is not allowed */
/* (need_complete: an incomplete code is of no use to the caller -- said before any table is filled: the boundary search asks this of a hundred thousand chance headers
 *  per chunk, and nearly all of them end here) */
static int huff_build(huff_t *h, const uint8_t *len, int n, int need_complete, int kind)
{
    int i, l, left = 1;
    uint16_t offs[16];
    memset(h->count, 0, sizeof(h->count));
    for (i = 0; i < n; ++i) h->count[len[i]]++;
    if (h->count[0] == n) return -1;
    for (l = 1; l <= 15; ++l) { left <<= 1; left -= h->count[l]; if (left < 0) return -1; }
    if (left > 0 && need_complete) return 1;
    offs[1] = 0;
    for (l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h->count[l];
    for (i = 0; i < n; ++i) if (len[i]) h->sym[offs[len[i]]++] = (uint16_t) i;
    memset(h->fast, 0, sizeof(h->fast));
    h->max_len = 0, h->kind = kind;
    {   /* codes in canonical order; the table is indexed by the code's bits as they come (least significant first) */
        uint32_t code = 0;
        int idx = 0;
        for (l = 1; l <= 15; ++l) {
            int c;
            for (c = 0; c < h->count[l]; ++c, ++idx, ++code) {
                h->max_len = l;
                if (l <= GP_FAST) {
                    uint32_t rev = 0, t = code;
                    int k;
                    for (k = 0; k < l; ++k) rev = rev << 1 | (t & 1), t >>= 1;
                    const uint32_t e = entry_of(kind, h->sym[idx], l);
                    for (t = rev; t < (1u << GP_FAST); t += 1u << l) h->fast[t] = e;
                }
            }
            code <<= 1;
        }
    }
    return left > 0;
}
/* the canonical walk for a code the table does not hold (at least 15 bits in `bits`): its entry, 0: no such code */
static inline uint32_t huff_long(const huff_t *h, uint64_t bits)
{
    int code = 0, first = 0, index = 0, l;
    for (l = 1; l <= 15; ++l) {
        code |= (int) (bits & 1), bits >>= 1;
        const int c = h->count[l];
        if (code - c < first) return entry_of(h->kind, h->sym[index + (code - first)], l);
        index += c, first += c, first <<= 1, code <<= 1;
    }
    return 0;
}
/* (at least 15 bits in the buffer) a plain code's next symbol; -1: no such code */
static inline int huff_decode(const huff_t *h, br_t *b)
{
    uint32_t e = h->fast[br_peek(b, GP_FAST)];
    if (!(e & 15)) { e = huff_long(h, b->bits); if (!e) return -1; }
    br_drop(b, (int) (e & 15));
    return (int) (e >> 16);
}

typedef struct { uint16_t *s; uint64_t n, m, lim; int refused; } sym_t;      /* refused: room was denied because of lim (not for want of memory) */          /* lim: more symbols than this are refused (0: no limit) -- a decoder that entered the stream at a wrong place must not eat the machine's memory */            /* a chunk's text as symbols: < 256 a byte, else 256 + position in the window before the chunk */

static int sym_room(sym_t *o, uint64_t more)
{
    if (o->n + more <= o->m) return 0;
    if (o->lim && o->n + more > o->lim) { o->refused = 1; return -1; }
    uint64_t m = o->m? o->m : 1 << 20;
    while (m < o->n + more) m += m >> 1;
    uint16_t *s = (uint16_t *) realloc(o->s, m * 2);
    if (!s) return -1;
    o->s = s, o->m = m;
    return 0;
}

/* the dynamic block's two codes (RFC 1951 3.2.7); strict: what zlib refuses is refused.  0 ok */
static int read_dynamic(br_t *b, huff_t *lit, huff_t *dst)
{
    uint8_t len[320], cl[19];
    huff_t clh;
    int i;
    br_refill(b);
    const int hlit = (int) br_get(b, 5) + 257, hdist = (int) br_get(b, 5) + 1, hclen = (int) br_get(b, 4) + 4;
    if (hlit > 286 || hdist > 30) return -1;
    memset(cl, 0, sizeof(cl));
    for (i = 0; i < hclen; ++i) cl[CLORD[i]] = (uint8_t) br_get(b, 3);
    if (huff_build(&clh, cl, 19, 1, K_PLAIN) != 0) return -1;
    for (i = 0; i < hlit + hdist; ) {
        br_refill(b);
        const int s = huff_decode(&clh, b);
        if (s < 0) return -1;
        if (s < 16) len[i++] = (uint8_t) s;
        else {
            int rep, v = 0;
            if (s == 16) { if (i == 0) return -1; v = len[i - 1]; rep = 3 + (int) br_get(b, 2); }
            else if (s == 17) rep = 3 + (int) br_get(b, 3);
            else rep = 11 + (int) br_get(b, 7);
            if (i + rep > hlit + hdist) return -1;
            while (rep--) len[i++] = (uint8_t) v;
        }
    }
    if (len[256] == 0) return -1;
    if (huff_build(lit, len, hlit, 1, K_LITLEN) != 0) return -1;              /* (zlib: an incomplete literal/length code is an error) */
    {
        const int r = huff_build(dst, len + hlit, hdist, 0, K_DIST);
        if (r < 0) { int nz = 0; for (i = 0; i < hdist; ++i) nz += len[hlit + i] != 0; if (nz) return -1; memset(dst, 0, sizeof(*dst)); dst->kind = K_DIST; }      /* (no distance codes at all: a block of literals) */
        else if (r > 0 && dst->count[1] + dst->count[2] + dst->count[3] + dst->count[4] + dst->count[5] + dst->count[6] + dst->count[7] + dst->count[8] + dst->count[9] + dst->count[10] + dst->count[11] + dst->count[12] + dst->count[13] + dst->count[14] + dst->count[15] != 1) return -1;
    }
    return 0;
}
static huff_t g_fix_lit, g_fix_dst;
static uint8_t g_not_text[256];                  /* 1: a byte no FASTA / FASTQ file holds (the boundary search refuses a block with such a literal) */
static uint16_t g_kraft3[512];                  /* three 3-bit code lengths -> their Kraft weights in 128ths */
static pthread_once_t g_fix_once = PTHREAD_ONCE_INIT;
static void crc_init(void);
static const char *gp_crc_kind(void);
static void fixed_init(void)
{
    uint8_t len[288];
    int i;
    for (i = 0; i < 144; ++i) len[i] = 8;
    for (; i < 256; ++i) len[i] = 9;
    for (; i < 280; ++i) len[i] = 7;
    for (; i < 288; ++i) len[i] = 8;
    (void) huff_build(&g_fix_lit, len, 288, 0, K_LITLEN);
    for (i = 0; i < 30; ++i) len[i] = 5;
    (void) huff_build(&g_fix_dst, len, 30, 0, K_DIST);
    crc_init();
    for (i = 0; i < 256; ++i) g_not_text[i] = !(i == '\n' || i == '\r' || i == '\t' || (i >= 32 && i < 127));
    for (i = 0; i < 512; ++i) { int q; g_kraft3[i] = 0; for (q = 0; q < 9; q += 3) { const int l = i >> q & 7; g_kraft3[i] += (uint16_t) (l? 128 >> l : 0); } }
}


/* The symbols of one Huffman-coded block.  One refill (56 bits and more) serves three literals (45 bits at most) or a whole match: length code, its extra bits, distance
 * code, its extra bits (15 + 5 + 15 + 13); the table's entry says what a code is and what follows it, so nothing else is looked up on the way.  0 ok, -1 not a block / corrupt */
static inline __attribute__((always_inline)) int decode_huff(br_t *bp, const huff_t *lit, const huff_t *dst, sym_t *o, const int text_only, uint64_t total_bits, uint64_t have)
{
    br_t b = *bp;
    uint16_t *os = o->s;
    uint64_t n = o->n, room = o->m;
    int rc = -1;
    for (;;) {
        if (__builtin_expect(n + 288 > room, 0)) { o->n = n; if (sym_room(o, 288)) goto out; os = o->s, room = o->m; if (br_pos(&b) > total_bits) goto out; }
        br_refill(&b);
        uint32_t e = lit->fast[b.bits & ((1u << GP_FAST) - 1)];
        if (e & E_LIT) {
            if (text_only && g_not_text[e >> 16]) goto out;
            b.bits >>= e & 15, b.nb -= (int) (e & 15);
            os[n++] = (uint16_t) (e >> 16);
            e = lit->fast[b.bits & ((1u << GP_FAST) - 1)];
            if (e & E_LIT) {
                if (text_only && g_not_text[e >> 16]) goto out;
                b.bits >>= e & 15, b.nb -= (int) (e & 15);
                os[n++] = (uint16_t) (e >> 16);
                e = lit->fast[b.bits & ((1u << GP_FAST) - 1)];
                if (e & E_LIT) {
                    if (text_only && g_not_text[e >> 16]) goto out;
                    b.bits >>= e & 15, b.nb -= (int) (e & 15);
                    os[n++] = (uint16_t) (e >> 16);
                    continue;
                }
            }
            br_refill(&b);                       /* (what was looked up is not a literal, or a long code: it is still in the buffer, and now so is all that can follow it) */
        }
        if (__builtin_expect(!(e & 15), 0)) {   /* a code of twelve bits and more */
            e = huff_long(lit, b.bits);
            if (!e) goto out;
            if (e & E_LIT) {
                if (text_only && g_not_text[e >> 16]) goto out;
                b.bits >>= e & 15, b.nb -= (int) (e & 15);
                os[n++] = (uint16_t) (e >> 16);
                continue;
            }
        }
        if (__builtin_expect(e & (E_EOB | E_BAD), 0)) {
            if (e & E_BAD) goto out;
            b.bits >>= e & 15, b.nb -= (int) (e & 15);
            break;
        }
        b.bits >>= e & 15, b.nb -= (int) (e & 15);
        const uint32_t lx = e >> 4 & 15, len = (e >> 16) + (uint32_t) (b.bits & ((1u << lx) - 1));
        b.bits >>= lx, b.nb -= (int) lx;
        uint32_t d = dst->fast[b.bits & ((1u << GP_FAST) - 1)];
        if (__builtin_expect(!(d & 15), 0)) { d = huff_long(dst, b.bits); if (!d) goto out; }
        if (__builtin_expect(d & E_BAD, 0)) goto out;
        b.bits >>= d & 15, b.nb -= (int) (d & 15);
        const uint32_t dx = d >> 4 & 15;
        const uint64_t dist = (d >> 16) + (b.bits & ((1u << dx) - 1));
        b.bits >>= dx, b.nb -= (int) dx;
        if (__builtin_expect(dist > n + have, 0)) goto out;        /* (zlib: "invalid distance too far back" -- `have` is the text before the chunk, at most a window's worth) */
        {
            uint32_t i = 0;
            if (__builtin_expect(dist > n, 0)) {                  /* it begins before the chunk: positions in the window */
                const uint32_t pre = (uint32_t) (dist - n < len? dist - n : len);
                for (; i < pre; ++i) os[n + i] = (uint16_t) (256 + (GP_WIN + n + i - dist));
            }
            if (dist >= 16 && i == 0) {         /* sixteen symbols at a time (the source lies at least sixteen behind: no overlap within a move; up to fifteen symbols beyond the match
                                                   are scribbled on and overwritten by what follows -- there is room: 288 were asked for) */
                const uint16_t *src = os + n - dist;
                uint16_t *dst2 = os + n;
                for (; i < len; i += 16) memcpy(dst2 + i, src + i, 32);
            } else if (dist >= 4 && i == 0) {   /* four at a time */
                const uint16_t *src = os + n - dist;
                uint16_t *dst2 = os + n;
                for (; i < len; i += 4) memcpy(dst2 + i, src + i, 8);
            } else {
                for (; i < len; ++i) os[n + i] = os[n + i - dist];
            }
            n += len;
        }
        if (__builtin_expect(b.over > 16, 0)) goto out;           /* (far beyond the input's end: zeros decode for ever) */
    }
    o->n = n;
    rc = br_pos(&b) > total_bits? -1 : 0;
out:
    *bp = b;
    return rc;
}

/* One block's data (the header's three bits are read) into o.  text_only: literals must be text (a candidate boundary is being tried).  0 ok, -1 not a block / corrupt */
static int decode_block(br_t *b, int btype, sym_t *o, int text_only, uint64_t total_bits, uint64_t have)
{
    if (btype == 0) {
        br_drop(b, b->nb & 7);
        br_refill(b);
        const uint32_t len = br_get(b, 16), nlen = br_get(b, 16);
        if ((len ^ nlen) != 0xFFFF) return -1;
        if (br_pos(b) + (uint64_t) len * 8 > total_bits) return -1;
        if (sym_room(o, len)) return -1;
        /* (the buffer holds whole bytes now) */
        uint32_t i;
        for (i = 0; i < len; ++i) {
            const uint32_t c = br_get(b, 8);
            if (text_only && g_not_text[c]) return -1;
            o->s[o->n++] = (uint16_t) c;
        }
        return 0;
    }
    huff_t lit_d, dst_d;
    const huff_t *lit = &g_fix_lit, *dst = &g_fix_dst;
    if (btype == 2) { if (read_dynamic(b, &lit_d, &dst_d)) return -1; lit = &lit_d, dst = &dst_d; }
    else if (btype != 1) return -1;
    return text_only? decode_huff(b, lit, dst, o, 1, total_bits, have) : decode_huff(b, lit, dst, o, 0, total_bits, have);
}

/* a block header that can be trusted at or after bit `from` (below `to`): dynamic, not the last, decodes as text and is followed by another header.  GP_INF: none */
static uint64_t find_boundary(const uint8_t *in, uint64_t n_in, uint64_t from, uint64_t to)
{
    const uint64_t total_bits = n_in * 8;
    sym_t tmp = {0, 0, 0, 16u << 20};             /* (a block of sixteen million symbols is no block zlib, pigz or bgzip writes) */
    uint64_t at, found = GP_INF;
    uint64_t x = 0;
    for (at = from; at < to && at + 192 < total_bits; ++at) {
        /* cheap tests first: BFINAL = 0, BTYPE = 2 (bits 0, 0, 1), HLIT <= 29, HDIST <= 29 */
        const uint64_t byte = at >> 3;
        if ((at & 7) == 0 || at == from) memcpy(&x, in + byte, 8);        /* (eight positions from one load) */
        const uint32_t w = (uint32_t) (x >> (at & 7));
        if ((w & 7) != 4) continue;
        if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
        {   /* ... and the code-length code must be complete (Kraft's sum over its 4 .. 19 three-bit lengths, the header's bits 17 ..): one chance header in hundreds is */
            uint64_t hi;
            memcpy(&hi, in + byte + 8, 8);                               /* (the loop's bound leaves these sixteen bytes inside the input) */
            const int sh = (int) (at & 7) + 17, ncl = (int) ((w >> 13) & 15) + 4;
            uint64_t c = (x >> sh | hi << (64 - sh)) & ((1ULL << (3 * ncl)) - 1);
            uint32_t kraft = 0;
            for (; c; c >>= 9) kraft += g_kraft3[c & 511];
            if (kraft != 128) continue;
        }
        br_t b;
        br_init(&b, in, n_in, at);
        br_refill(&b);
        br_drop(&b, 3);
        tmp.n = 0;
        if (decode_block(&b, 2, &tmp, 1, total_bits, GP_WIN)) continue;
        if (tmp.n < 1024) continue;                                  /* (a real block of a text file is not this small; chance finds are) */
        {   /* what follows must be a header too */
            br_refill(&b);
            const uint32_t h = br_peek(&b, 3);
            const int bt = (int) (h >> 1);
            if (bt == 3) continue;
            if (bt == 2) {
                huff_t l2, d2;
                br_t c = b;
                br_drop(&c, 3);
                if (read_dynamic(&c, &l2, &d2)) continue;
            } else if (bt == 0) {
                br_t c = b;
                br_drop(&c, 3);
                br_drop(&c, c.nb & 7);
                br_refill(&c);
                const uint32_t len = br_get(&c, 16), nlen = br_get(&c, 16);
                if ((len ^ nlen) != 0xFFFF) continue;
            }
        }
        found = at;
        break;
    }
    free(tmp.s);
    return found;
}

typedef struct {
    uint64_t have;                               /* the text known to lie before it, at most GP_WIN (a chunk entered on a guess: GP_WIN; checked when it is chained) */
    uint64_t nominal, start, end;                /* bits: where the chunk was cut, the boundary it claims (GP_INF none), where its decoding stopped */
    sym_t o;
    int ok, last;                                /* decoded without error; reached the member's last block */
    int capped;                                  /* given up because it grew beyond anything a chunk of this size inflates to (it may still be the text: then it is decoded again, unbounded) */
    int chained;                                 /* a chunk before it arrived exactly at its boundary: its symbols are the text */
    uint64_t out_off, take, given;               /* where its bytes go in this call, how many of its symbols go there, how many went before */
    uint8_t lut[256 + GP_WIN];                   /* symbol -> byte: 256 bytes as they are, then the window before the chunk (one load a symbol, no branch: in DNA text most
                                                    symbols of a chunk stay window references -- short matches into short matches -- to its end) */
    uint32_t crc;
} chunk_t;

struct oatk_gzpar {
    const uint8_t *in; uint64_t n_in;
    int n_threads;
    uint64_t chunk_bits;
    uint64_t pos;                                /* the boundary the next batch begins at (bits) */
    uint8_t win[GP_WIN];                         /* the text before it */
    uint64_t total_out; uint32_t crc;
    int done, failed, first;
    chunk_t *ch; int n_ch, next_out;             /* the batch at hand, and the first of its chunks not delivered yet */
    int n_slots;                                 /* chunks a batch may hold: more than there are threads, so that a thread that is done early takes another */
    /* work sharing: the threads live as long as the member (created at open; a phase is a generation they all take part in) */
    int phase; volatile int next;
    uint8_t *dst;
    pthread_mutex_t mu; pthread_cond_t cv_go, cv_done;
    pthread_t th[GP_MAX_THREADS];
    int n_started, gen, busy, quit;
    /* OATK_GZPAR_LOG=1: where the time goes, on stderr when the member is closed */
    int log, n_batches, n_again, n_gaps;
    uint64_t gap_syms;
    double t_phase[4];                           /* boundaries, symbols, chain + windows, bytes */
    uint64_t n_chunks, n_chained;
};
static double gp_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec; }

/* chunk j from the boundary it claims to the first block boundary at or after the next chunk's cut (where the next chunk looks for ITS boundary: if both are right they
 * meet; nothing here depends on what another thread finds) */
static void decode_chunk(oatk_gzpar_t *p, int j, int bounded)
{
    chunk_t *c = &p->ch[j];
    const uint64_t total_bits = p->n_in * 8, stop = c->nominal + p->chunk_bits;
    br_t b;
    c->o.n = 0, c->ok = 0, c->last = 0, c->capped = 0, c->o.refused = 0;
    /* a chunk entered on a guess (every one but the batch's first) gets room for what two chunks of text compressed fortyfold would need: garbage that decodes as one long
     * run of matches stops there, instead of at the end of a gigabyte of input.  And the batch as a whole holds at most 2^30 symbols, 2 GB (192 slots of
     * 144 MB were 27 GB on text that deflates seventyfold): a chunk that is the text and wants more is decoded again without the bound, one at a time, as before. */
    {
        const uint64_t per_chunk = (p->chunk_bits >> 3) * 40 + (32u << 20), share = ((uint64_t) 1 << 30) / (uint64_t) (p->n_slots > 0? p->n_slots : 1);
        c->o.lim = bounded? (per_chunk < share? per_chunk : (share > (4u << 20)? share : (4u << 20))) : 0;
    }
    if (c->start == GP_INF) return;
    br_init(&b, p->in, p->n_in, c->start);
    for (;;) {
        br_refill(&b);
        const uint32_t h = br_get(&b, 3);
        if (decode_block(&b, (int) (h >> 1), &c->o, 0, total_bits, c->have)) { c->capped = c->o.refused; return; }      /* (a stored block is refused room for up to 65535 symbols, not 288: the flag says so, the arithmetic did not) */
        const uint64_t at = br_pos(&b);
        if (h & 1) { c->last = 1, c->end = at, c->ok = 1; return; }
        if (at >= stop) { c->end = at, c->ok = 1; return; }
    }
}
/* the chained chunk j goes on from the boundary `*at` (where it stopped), block by block, until it stands at or beyond `target`: the blocks between its end and a later
 * chunk's claim -- a flush marker the search does not take for a boundary, a chunk whose claim was false or that found none.  Its symbols are the text: no bound.  0 ok */
static int extend_chunk(oatk_gzpar_t *p, int j, uint64_t *at, uint64_t target)
{
    chunk_t *c = &p->ch[j];
    const uint64_t total_bits = p->n_in * 8;
    br_t b;
    c->o.lim = 0;
    br_init(&b, p->in, p->n_in, *at);
    for (;;) {
        br_refill(&b);
        const uint32_t h = br_get(&b, 3);
        if (decode_block(&b, (int) (h >> 1), &c->o, 0, total_bits, c->have)) return -1;
        const uint64_t pos = br_pos(&b);
        c->end = *at = pos;
        if (h & 1) { c->last = 1; return 0; }
        if (pos >= target) return 0;
    }
}
/* ---- CRC-32 of the text, as gzip defines it ----
 * zlib 1.2.11's crc32 does ~1.1 GB/s on a core, a third of what turning symbols into bytes costs.  Where the processor multiplies without carries (PCLMULQDQ) the CRC is
 * folded sixty-four bytes at a time instead (Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ", Intel 2009; the constants are x^n mod P for the
 * reflected polynomial 0xEDB88320: 4 x 128 + 64 / 4 x 128 bits for the four-lane fold, 128 + 64 / 128 for the one-lane fold, 64 -> 32, and P' with mu for Barrett's
 * reduction).  It is checked against zlib's on the first use -- all lengths 0 .. 300 and a few long ones -- and zlib's is used if they differ or the instruction is absent. */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("pclmul,sse4.1")))
static uint32_t crc_fold(uint32_t crc, const uint8_t *buf, uint64_t len)          /* len a multiple of 16, at least 64; crc and the result WITHOUT zlib's inversions */
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596LL, 0x0154442bd4LL), k3k4 = _mm_set_epi64x(0x00ccaa009eLL, 0x01751997d0LL);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124LL), pm = _mm_set_epi64x(0x01f7011641LL, 0x01db710641LL), lo32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i x1 = _mm_loadu_si128((const __m128i *) buf), x2 = _mm_loadu_si128((const __m128i *) (buf + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *) (buf + 32)), x4 = _mm_loadu_si128((const __m128i *) (buf + 48)), t;
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int) crc));
    buf += 64, len -= 64;
    for (; len >= 64; buf += 64, len -= 64) {
        __m128i a = _mm_clmulepi64_si128(x1, k1k2, 0x00), b = _mm_clmulepi64_si128(x2, k1k2, 0x00), c = _mm_clmulepi64_si128(x3, k1k2, 0x00), d = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11), x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11), x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11), x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a), _mm_loadu_si128((const __m128i *) buf));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, b), _mm_loadu_si128((const __m128i *) (buf + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, c), _mm_loadu_si128((const __m128i *) (buf + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, d), _mm_loadu_si128((const __m128i *) (buf + 48)));
    }
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00), x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11), x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), t);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00), x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11), x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), t);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00), x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11), x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), t);
    for (; len >= 16; buf += 16, len -= 16) {
        t = _mm_clmulepi64_si128(x1, k3k4, 0x00), x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, _mm_loadu_si128((const __m128i *) buf)), t);
    }
    /* 128 -> 64 -> 32 bits, then Barrett */
    x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, lo32), k5, 0x00), x2);
    x2 = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, lo32), pm, 0x10), lo32);
    x2 = _mm_clmulepi64_si128(x2, pm, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t) _mm_extract_epi32(x1, 1);
}
static int g_crc_fast;                           /* decided once (pthread_once with the fixed codes) */
static uint32_t gp_crc32(uint32_t crc, const uint8_t *d, uint64_t n)
{
    if (g_crc_fast && n >= 64) {
        const uint64_t m = n & ~15ULL;
        crc = ~crc_fold(~crc, d, m);
        d += m, n -= m;
    }
    for (; n; ) { const uint64_t m = n < (1u << 30)? n : (1u << 30); crc = (uint32_t) crc32(crc, d, (uInt) m); d += m, n -= m; }
    return crc;
}
static void crc_init(void)
{
    g_crc_fast = 0;
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return;
    {
        enum { N = 70000 };
        uint8_t *t = (uint8_t *) malloc(N);
        uint32_t x = 12345, ok = 1;
        int n;
        if (!t) return;
        for (n = 0; n < N; ++n) x = x * 1664525u + 1013904223u, t[n] = (uint8_t) (x >> 24);
        g_crc_fast = 1;
        for (n = 0; n <= 300 && ok; ++n) ok = gp_crc32(0x1234u + (uint32_t) n, t + (n & 7), (uint64_t) n) == (uint32_t) crc32(0x1234u + (uint32_t) n, t + (n & 7), (uInt) n);
        for (n = 4093; n < N && ok; n = n * 2 + 3) ok = gp_crc32(0, t + 1, (uint64_t) n) == (uint32_t) crc32(0L, t + 1, (uInt) n);
        g_crc_fast = (int) ok;
        free(t);
    }
}
/* sixteen symbols that are all bytes become sixteen bytes in one move */
static inline uint64_t resolve_run(uint8_t *d, const uint16_t *s, uint64_t n, const uint8_t *lut)
{
    uint64_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128((const __m128i *) (s + i)), b = _mm_loadu_si128((const __m128i *) (s + i + 8));
        if (_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_srli_epi16(_mm_or_si128(a, b), 8), _mm_setzero_si128())) == 0xFFFF) _mm_storeu_si128((__m128i *) (d + i), _mm_packus_epi16(a, b));
        else { int q; for (q = 0; q < 16; ++q) d[i + q] = lut[s[i + q]]; }
    }
    for (; i < n; ++i) d[i] = lut[s[i]];
    return n;
}
#else
static uint32_t gp_crc32(uint32_t crc, const uint8_t *d, uint64_t n)
{
    for (; n; ) { const uint64_t m = n < (1u << 30)? n : (1u << 30); crc = (uint32_t) crc32(crc, d, (uInt) m); d += m, n -= m; }
    return crc;
}
static void crc_init(void) {}
static inline uint64_t resolve_run(uint8_t *d, const uint16_t *s, uint64_t n, const uint8_t *lut)
{
    uint64_t i;
    for (i = 0; i < n; ++i) d[i] = lut[s[i]];
    return n;
}
#endif

#if defined(__x86_64__)
static const char *gp_crc_kind(void) { return g_crc_fast? "folded" : "zlib's"; }
#else
static const char *gp_crc_kind(void) { return "zlib's"; }
#endif
/* a chunk's symbols -> bytes at their place, with their CRC: a piece at a time, so that the CRC reads what the cache still holds */
static void resolve_chunk(oatk_gzpar_t *p, int j)
{
    chunk_t *c = &p->ch[j];
    uint8_t *d = p->dst + c->out_off;
    const uint16_t *s = c->o.s + c->given;
    uint64_t i;
    c->crc = (uint32_t) crc32(0L, Z_NULL, 0);
    for (i = 0; i < c->take; i += 32768) {
        const uint64_t m = c->take - i < 32768? c->take - i : 32768;
        resolve_run(d + i, s + i, m, c->lut);
        c->crc = gp_crc32(c->crc, d + i, m);
    }
}
static void gp_work(oatk_gzpar_t *p)
{
    for (;;) {
        const int j = __atomic_fetch_add(&p->next, 1, __ATOMIC_RELAXED);
        if (j >= p->n_ch) break;
        chunk_t *c = &p->ch[j];
        if (p->phase == 0) {
            if (j > 0) c->start = find_boundary(p->in, p->n_in, c->nominal, c->nominal + p->chunk_bits);
            decode_chunk(p, j, j > 0);
        } else if (c->chained && j >= p->next_out && c->out_off != GP_INF && c->take) resolve_chunk(p, j);
    }
}
static void *gp_worker(void *arg)
{
    oatk_gzpar_t *p = (oatk_gzpar_t *) arg;
    int seen = 0;                                /* (every worker is created in oatk_gzpar_open, before the first generation) */
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (p->gen == seen && !p->quit) pthread_cond_wait(&p->cv_go, &p->mu);
        if (p->quit) { pthread_mutex_unlock(&p->mu); return 0; }
        seen = p->gen;
        pthread_mutex_unlock(&p->mu);
        gp_work(p);
        pthread_mutex_lock(&p->mu);
        if (--p->busy == 0) pthread_cond_signal(&p->cv_done);
        pthread_mutex_unlock(&p->mu);
    }
}
static int run_phase(oatk_gzpar_t *p, int phase)
{
    const double t0 = gp_now();
    pthread_mutex_lock(&p->mu);
    p->phase = phase, p->next = 0, p->busy = p->n_started, ++p->gen;
    pthread_cond_broadcast(&p->cv_go);
    pthread_mutex_unlock(&p->mu);
    gp_work(p);
    pthread_mutex_lock(&p->mu);
    while (p->busy) pthread_cond_wait(&p->cv_done, &p->mu);
    pthread_mutex_unlock(&p->mu);
    p->t_phase[phase == 2? 3 : phase] += gp_now() - t0;
    return 0;
}

oatk_gzpar_t *oatk_gzpar_open(const uint8_t *deflate, uint64_t n_in, int n_threads)
{
    oatk_gzpar_t *p = (oatk_gzpar_t *) calloc(1, sizeof(*p));
    int i;
    if (!p) return 0;
    pthread_once(&g_fix_once, fixed_init);
    p->in = deflate, p->n_in = n_in;
    p->n_threads = n_threads < 1? 1 : (n_threads > GP_MAX_THREADS? GP_MAX_THREADS : n_threads);
    {
        const char *e = getenv("OATK_HOST_GZ_CHUNK_KB");
        uint64_t kb = e && atoi(e) > 0? (uint64_t) atoi(e) : 1024;
        p->chunk_bits = kb * 1024 * 8;
        e = getenv("OATK_HOST_GZ_SLOTS");         /* chunks per batch and thread (3: a batch is done when its slowest chunk is, and chunks differ by a third) */
        p->n_slots = p->n_threads * (e && atoi(e) > 0 && atoi(e) <= 16? atoi(e) : 3);
    }
    p->first = 1;
    { const char *e = getenv("OATK_GZPAR_LOG"); p->log = e && e[0] == '1'; }
    p->crc = (uint32_t) crc32(0L, Z_NULL, 0);
    p->ch = (chunk_t *) calloc((size_t) p->n_slots, sizeof(chunk_t));
    if (!p->ch) { free(p); return 0; }
    { int q; for (i = 0; i < p->n_slots; ++i) for (q = 0; q < 256; ++q) p->ch[i].lut[q] = (uint8_t) q; }
    pthread_mutex_init(&p->mu, 0), pthread_cond_init(&p->cv_go, 0), pthread_cond_init(&p->cv_done, 0);
    for (i = 0; i < p->n_threads - 1; ++i) { if (pthread_create(&p->th[p->n_started], 0, gp_worker, p) != 0) break; ++p->n_started; }      /* (fewer than asked for: the others do their share) */
    return p;
}
void oatk_gzpar_close(oatk_gzpar_t *p)
{
    int j;
    if (!p) return;
    pthread_mutex_lock(&p->mu);
    p->quit = 1;
    pthread_cond_broadcast(&p->cv_go);
    pthread_mutex_unlock(&p->mu);
    for (j = 0; j < p->n_started; ++j) pthread_join(p->th[j], 0);
    pthread_mutex_destroy(&p->mu), pthread_cond_destroy(&p->cv_go), pthread_cond_destroy(&p->cv_done);
    if (p->log)
        fprintf(stderr, "[M::gzpar] %d threads, %d batches, %lu chunks of %lu KB (%lu the text, %d decoded again, %d gaps of %lu symbols closed in order), crc %s: boundaries and symbols %.3f s, chain and windows %.3f s, bytes %.3f s\n",
                p->n_started + 1, p->n_batches, (unsigned long) p->n_chunks, (unsigned long) (p->chunk_bits >> 13), (unsigned long) p->n_chained, p->n_again, p->n_gaps, (unsigned long) p->gap_syms, gp_crc_kind(),
                p->t_phase[0], p->t_phase[2], p->t_phase[3]);
    for (j = 0; j < p->n_slots; ++j) free(p->ch[j].o.s);
    free(p->ch);
    free(p);
}
uint64_t oatk_gzpar_in_used(const oatk_gzpar_t *p) { return (p->pos + 7) >> 3; }
uint32_t oatk_gzpar_crc(const oatk_gzpar_t *p) { return p->crc; }
uint64_t oatk_gzpar_total(const oatk_gzpar_t *p) { return p->total_out; }

/* the next batch of chunks, decoded to symbols and chained.  0 ok; -1 corrupt; -2 (first batch only) this is not text one can enter in the middle: zlib's job */
static int next_batch(oatk_gzpar_t *p)
{
    int j;
    const uint64_t total_bits = p->n_in * 8;
    p->n_ch = 0;
    for (j = 0; j < p->n_slots; ++j) {
        const uint64_t nominal = p->pos + (uint64_t) j * p->chunk_bits;
        if (j > 0 && nominal + 1024 >= total_bits) break;
        p->ch[j].nominal = nominal, p->ch[j].start = j == 0? p->pos : GP_INF, p->ch[j].chained = 0, p->ch[j].out_off = GP_INF, p->ch[j].take = 0, p->ch[j].given = 0;
        p->ch[j].have = j == 0 && p->total_out < GP_WIN? p->total_out : GP_WIN;
        ++p->n_ch;
    }
    p->next_out = 0;
    run_phase(p, 0);
    if (p->first && p->n_ch >= 4) {
        int found = 0;
        for (j = 1; j < p->n_ch; ++j) found += p->ch[j].start != GP_INF;
        if (2 * found < p->n_ch - 1) return -2;
    }
    p->first = 0;
    const double t_chain = gp_now();
    p->n_batches++, p->n_chunks += (uint64_t) p->n_ch;
    /* the chain: chunk 0 begins at a known boundary and stops at a boundary; a later chunk is the text if the chain stands exactly where it begins.  Where the chain stands
     * BEFORE a claim (the boundary it stopped at was one the search does not accept -- a flush marker, a stored or a short block -- or a chunk between found none) its last
     * chunk goes on, in order, until it stands there or has passed it; a claim the chain has passed was false */
    {
        uint64_t at = p->pos, before = p->total_out;         /* (bits; bytes of text before the chunk at hand) */
        int prev = -1;
        for (j = 0; j < p->n_ch; ++j) {
            chunk_t *c = &p->ch[j];
            if (c->start == GP_INF || c->start < at) continue;
            if (c->start > at) {
                if (prev < 0) return -1;
                const uint64_t n0 = p->ch[prev].o.n;
                if (extend_chunk(p, prev, &at, c->start)) return -1;
                p->n_gaps++, p->gap_syms += p->ch[prev].o.n - n0;
                before += p->ch[prev].o.n - n0;
                if (p->ch[prev].last) break;
                if (at != c->start) continue;
            }
            if (!c->ok && c->capped) { decode_chunk(p, j, 0); p->n_again++; }      /* (it IS the text, and longer than the guess allowed: once more, without the bound) */
            if (!c->ok) return -1;               /* (decoded from a true boundary and failed: the stream is damaged) */
            if (before < GP_WIN && j > 0) {      /* entered on a guess within the member's first 32 KiB: a reference to text before the member's first byte is zlib's "invalid distance too far back" */
                uint64_t q;
                for (q = 0; q < c->o.n; ++q) if (c->o.s[q] >= 256 && (uint64_t) (c->o.s[q] - 256) < GP_WIN - before) return -1;
            }
            c->chained = 1, prev = j, at = c->end, p->n_chained++;
            before += c->o.n;
            if (c->last) break;
        }
        if (prev < 0) return -1;
    }
    /* windows, in order: a chunk's is the 32 KiB of text before it */
    {
        uint8_t win[GP_WIN];
        memcpy(win, p->win, GP_WIN);
        for (j = 0; j < p->n_ch; ++j) {
            chunk_t *c = &p->ch[j];
            if (!c->chained) continue;
            memcpy(c->lut + 256, win, GP_WIN);
            const uint64_t n = c->o.n, tail = n < GP_WIN? n : GP_WIN;
            uint64_t i;
            if (tail < GP_WIN) memmove(win, win + tail, GP_WIN - tail);
            for (i = 0; i < tail; ++i) { const uint16_t s = c->o.s[n - tail + i]; win[GP_WIN - tail + i] = c->lut[s]; }
        }
    }
    p->t_phase[2] += gp_now() - t_chain;
    return 0;
}

int oatk_gzpar_done(const oatk_gzpar_t *p) { return p->done && p->next_out >= p->n_ch; }

/* text of the member into dst, at most cap bytes; fewer than cap (even 0) while oatk_gzpar_done() is false: call again; -1 corrupt; -2 see next_batch.  When it is done
 * oatk_gzpar_in_used / _crc / _total say where the deflate data ended and what came out. */
int64_t oatk_gzpar_read(oatk_gzpar_t *p, uint8_t *dst, uint64_t cap)
{
    uint64_t out = 0;
    if (!p || p->failed) return -1;
    while (out < cap) {
        int j, stop;
        if (p->next_out >= p->n_ch) {
            if (p->done) break;
            const int rc = next_batch(p);
            if (rc) { if (rc == -1) p->failed = 1; return out? (int64_t) out : rc; }      /* (-2 only before anything was delivered) */
        }
        /* the chunks that fit (the last of them perhaps in part), at their places */
        {
            uint64_t at = out;
            for (j = p->next_out; j < p->n_ch && at < cap; ++j) {
                chunk_t *c = &p->ch[j];
                c->out_off = GP_INF, c->take = 0;
                if (!c->chained) continue;
                c->take = c->o.n - c->given < cap - at? c->o.n - c->given : cap - at;
                c->out_off = at, at += c->take;
            }
            stop = j;
            p->dst = dst;
            run_phase(p, 2);
            for (j = p->next_out; j < stop; ++j) {
                chunk_t *c = &p->ch[j];
                if (!c->chained) { p->next_out = j + 1; continue; }
                p->crc = (uint32_t) crc32_combine(p->crc, c->crc, (z_off_t) c->take);
                p->total_out += c->take;
                if (c->take) {                   /* the window for the batch after this one */
                    const uint64_t tail = c->take < GP_WIN? c->take : GP_WIN;
                    if (tail < GP_WIN) memmove(p->win, p->win + tail, GP_WIN - tail);
                    memcpy(p->win + GP_WIN - tail, dst + c->out_off + c->take - tail, tail);
                }
                c->given += c->take, c->out_off = GP_INF;
                if (c->given < c->o.n) { p->next_out = j; break; }          /* the caller's buffer is full in the middle of this chunk */
                p->pos = c->end, p->next_out = j + 1;
                if (c->last) { p->done = 1, p->next_out = p->n_ch; break; }
            }
            out = at;
        }
    }
    return (int64_t) out;
}
